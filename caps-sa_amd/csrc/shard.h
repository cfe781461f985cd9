// caps-sa_amd/csrc/shard.h
//
// One rank of the multi-GPU construction (one process per GPU).  The path
// shards exactly where the reference's samplesort has its exchange step:
//
//   phase1   every rank packs the (replicated) text and sorts ITS slice of the p subarrays
//            (src/Suffix_Array.cpp:161-184) and samples them                       [local]
//   pivots   samples are all-gathered by the caller; every rank sorts all of them and picks
//            the same p-1 pivots (cpp:197-222), locates them in its subarrays (cpp:225-249)
//            and counts its contribution to every partition (cpp:305-316)          [local]
//   collate  partition sizes are all-gathered by the caller; partitions are assigned to
//            ranks as contiguous ranges balanced by SIZE; the rank's sub-subarrays are laid
//            out partition-major = destination-major (cpp:335-364)                 [local]
//   -------- all-to-all-v of (key, sa) by the caller: RCCL over xGMI --------      [exchange]
//   phase2   received runs are regrouped per partition, every owned partition is sorted
//            (cpp:371-409) and gets its boundary LCPs (cpp:431-447); the result is a
//            contiguous slice [slice_off, slice_off + slice_len) of the global SA / LCP
//   fix      LCP of the slice's first suffix against the previous rank's last one    [1 idx]
//
// That is the samplesort path.  The DIRECT path (pipeline.h, Builder::run_direct) shards more simply, and is what a
// build takes unless the text forbids it (long repeats: keys alone cannot balance the groups):
//
//   scatter  every rank packs the text, draws the SAME samples from it and derives the SAME pivots (nothing to
//            exchange), then runs level A on every world-th tile of the text: its suffixes go, group by group, into
//            stream regions of the send buffers; groups are owned by ranks in contiguous ranges (equal counts: groups
//            are sample quantiles, hence of equal size), so a destination's streams are one contiguous block    [local]
//   -------- all_gather of the stream sizes (+ flags); all-to-all of the blocks of (key, sa) --------          [exchange]
//   plan     from all ranks' reports: agree on going on (or falling back), lay out the received streams           [host]
//   sort_owned  level B + tile sort over the owned groups -> the rank's contiguous slice of SA / LCP            [local]
//
// The collectives themselves are issued by the host driver (caps_sa_dist.py) with
// torch.distributed; this class only runs kernels on the rank's stream.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <numeric>

#include "pipeline.h"

namespace caps {

struct DevAllocs;   // capi_impl.h

struct ShardBase {
    virtual ~ShardBase() {}
    virtual void info(caps_sa_shard_info* out) const = 0;
    virtual void phase1(void* d_sample_keys, void* d_sample_sa) = 0;
    virtual void pivots(const void* d_all_keys, const void* d_all_sa, void* d_local_sizes) = 0;
    virtual void collate(const uint64_t* all_sizes, void* d_send_keys, void* d_send_sa, uint64_t* send_counts,
                         uint64_t* recv_counts) = 0;
    virtual void phase2(const void* d_recv_keys, const void* d_recv_sa, void* dSA, void* dLCP) = 0;
    virtual uint64_t last_sa() = 0;
    virtual void fix_first_lcp(uint64_t prev_sa, void* dLCP) = 0;
    // direct path (see Shard)
    virtual void scatter(void* d_send_keys, void* d_send_sa, void* d_report) = 0;
    virtual int plan(const uint64_t* all_reports, uint64_t* send_counts, uint64_t* recv_counts) = 0;
    virtual int sort_owned(const void* d_recv_keys, const void* d_recv_sa, void* dSA, void* dLCP) = 0;
    virtual void set_key_bits(int bits) = 0;
    virtual void phase1_arrays(void* d_keys_out, void* d_sa_out, uint64_t* count, uint64_t* subarray_len) = 0;
};

template <typename idx_t> class Shard : public ShardBase {
public:
    Shard(const void* dT, uint64_t n, uint64_t p_arg, int rank, int world, void* stream)
        : be_(static_cast<decltype(Backend::stream)>(stream)), dT_(static_cast<const uint8_t*>(dT)), n_(n), rank_(rank),
          world_(world)
    {
        effective_params(n, p_arg, &p_, &ppp_);
        if (p_ < 2) throw std::invalid_argument("sharded build needs an effective subproblem count >= 2 (n >= 32)");
        s_ = n / p_;
        g0_ = (uint32_t)((uint64_t)rank * p_ / world);
        g1_ = (uint32_t)((uint64_t)(rank + 1) * p_ / world);
        G_ = g1_ - g0_;
        text_base_ = (uint64_t)g0_ * s_;
        local_n_ = G_ ? (g1_ == p_ ? n : (uint64_t)g1_ * s_) - text_base_ : 0;
        m_local_ = (uint64_t)G_ * ppp_;
        m_total_ = (uint64_t)p_ * ppp_;
        // direct path: groups, sub-streams, capacity of a stream region (sample quantiles: +-1.3 % at C3, Poisson on top)
        direct_fb_ = direct_shape(n, p_, m_total_, &PG_, &K1_);
        if (const char* force = std::getenv("CAPS_SA_PATH")) if (std::string(force) == "classic") direct_fb_ = CAPS_SA_FB_FORCED;
        // Direct path, two ways to shard it.  LOCAL (the default): the text is replicated anyway, so every rank
        // scatters ALL of it and keeps only the groups it owns -- level A costs every rank a pass over the whole (packed, 0.75 GB
        // at C3) text but no element ever crosses a link: no data-path collective at all.  EXCHANGE (CAPS_SA_SHARD_EXCHANGE=1):
        // every rank scatters every world-th tile of the text and the streams are exchanged by one all-to-all -- 8 B per suffix
        // over xGMI, which costs more than the redundant classification at every world size a node offers.
        local_ = std::getenv("CAPS_SA_SHARD_EXCHANGE") == nullptr;
        const int tw = local_ ? 1 : world;                        // ranks that share the tiles of the text
        const uint64_t ga_tiles = (n + GA_E - 1) / GA_E;
        my_tiles_ = local_ ? (uint32_t)ga_tiles : (uint32_t)(ga_tiles > (uint64_t)rank ? (ga_tiles - rank + world - 1) / world : 0);
        my_elems_ = (uint64_t)my_tiles_ * GA_E;
        if (my_tiles_ && (local_ || (ga_tiles - 1) % world == (uint64_t)rank)) my_elems_ -= ga_tiles * GA_E - n;   // the text's last tile is short
        if (direct_fb_ == CAPS_SA_FB_NONE) {
            SUB_ = ga_tiles / tw >= 32ull * DIRECT_SUB ? DIRECT_SUB : 1u;
            if (const char* e = std::getenv("CAPS_SA_DIRECT_SUB")) if (std::atoi(e) >= 1 && (uint32_t)std::atoi(e) <= DIRECT_SUB) SUB_ = (uint32_t)std::atoi(e);
            n_streams_ = K1_ * SUB_;
            own_lo_ = (uint32_t)((uint64_t)rank * K1_ / world);           // rank d owns groups [d K1 / world, (d + 1) K1 / world)
            own_hi_ = (uint32_t)((uint64_t)(rank + 1) * K1_ / world);
            const double mean = (double)n / ((double)n_streams_ * tw);
            capA_ = (uint64_t)(mean * 1.10 + 6.0 * std::sqrt(mean) + 2.0 * GA_E / K1_ + 64.0);
            capA_ += capA_ & 1;
            capA_ = test_stream_cap(capA_, n, (uint64_t)n_streams_ * tw);
            if (n_streams_ * capA_ / (local_ ? world : 1) + TILE_E >= (uint64_t)std::numeric_limits<idx_t>::max()) direct_fb_ = CAPS_SA_FB_SHAPE;
            // quantile mode (pipeline.h Builder::run_direct): NB buckets of BUCKET_Q suffixes, KPG per group, QUANTILE_SPB samples each
            const uint64_t NBt = (n + BUCKET_Q - 1) / BUCKET_Q;
            KPG_ = (uint32_t)((NBt + K1_ - 1) / K1_);
            NB_ = (uint64_t)K1_ * KPG_;
            m2_ = NB_ * QUANTILE_SPB;
            if (m2_ > n / 4) m2_ = n / 4;
        }
        // segments of level B: (owned group, sub-stream), in exchange mode (owned group, source rank, sub-stream) -- a rank owns
        // ceil(K1 / world) groups at most, i.e. up to K1 * SUB + world * SUB segments when the groups do not divide evenly (found
        // by tools/stress_gpu.py STRESS_MULTI: p = 3 on 8 ranks overran the tables by 40 entries; HIP refused the copy)
        const uint32_t gseg = std::max<uint32_t>(p_, DIRECT_SUB * std::min<uint32_t>(p_, BUCKET_LDS)) + (uint32_t)world * DIRECT_SUB;
        try {
            // Everything is allocated ONCE here (hipMalloc / hipFree of tens of GB cost seconds): the
            // element arrays serve phase 1 (this rank's subarrays) and phase 2 (its partitions), sized
            // for the larger of the two with 25 % slack for the imbalance of the partition ownership.
            const uint64_t share = n / (uint64_t)world + 1;
            cap_ = std::max<uint64_t>(local_n_, share + share / 4 + 16 * TILE_E);
            if (world == 1) cap_ = local_n_;               // a single rank owns everything: no ownership imbalance to provide for
            if (direct_fb_ == CAPS_SA_FB_NONE) {           // received streams of the owned groups, gaps included
                const uint64_t own_max = ((uint64_t)K1_ + world - 1) / world * SUB_ * tw * capA_;
                cap_ = std::max<uint64_t>(cap_, own_max + TILE_E);
            }
            P_ = get<uint32_t>(text_alloc_words(n));
            present_ = get<uint32_t>(16);
            lut_ = get<uint8_t>(256);
            A_ = elems(cap_);
            B_ = elems(cap_);
            seg1_ = segs(G_ ? G_ : 1, cap_ / TILE_E + p_ + 2);
            seg1_.G = G_;
            seg2_ = segs(gseg, cap_ / TILE_E + gseg + 2);
            seg_cap_ = gseg;
            seg2_.G = p_;
            seg2_.seg_end = get<uint64_t>((size_t)gseg + 1);
            gkey_ = get<uint64_t>(p_);
            glut_ = get<uint16_t>(SPLIT_LUT_CELLS + 2);
            dcur_ = get<idx_t>(gseg);
            gshift_ = get<uint8_t>(gseg);
            rcap_ = get<uint64_t>((size_t)gseg + 1);
            rstart_ = get<uint64_t>((size_t)gseg + 2);
            quantile_ok_ = direct_fb_ == CAPS_SA_FB_NONE && local_ && KPG_ >= 2 && KPG_ <= BUCKET_LDS && m2_ >= 8 * NB_ && m2_ <= (1ull << 31) &&
                           m2_ <= cap_ && n_streams_ <= gseg && (uint64_t)(own_hi_ - own_lo_) * KPG_ <= BucketBufs::bucket_bound(cap_, gseg);
            knots_ = get<uint64_t>(quantile_ok_ ? NB_ : 1);
            dstat_ = get<uint64_t>(4);
            SA_ = elems(m_total_);
            SB_ = elems(m_total_);
            segS_ = segs(1, m_total_ / TILE_E + 3);
            pkey_ = get<uint64_t>(p_);
            psa_ = get<idx_t>(p_);
            Pm_ = get<idx_t>((size_t)(G_ ? G_ : 1) * (p_ + 1));
            ruler_ = get<idx_t>((size_t)(G_ ? G_ : 1) * p_);
            sizes_ = get<uint64_t>(p_);
            partial_ = get<uint64_t>((size_t)PART_CHUNKS * p_);
            lstart_ = get<uint64_t>((size_t)p_ + 1);
            bk_ = buckets(cap_, gseg);
            hot_rec_ = quantile_ok_ ? get<TileInfo>(bk_.tile_cap + 1) : nullptr;      // (segmented_sort: speculative split by knots)
            const uint64_t a = bk_.tile_cap + 3, b = m_total_ / TILE_E + 3;
            tdesc_ = get<TileDesc>(a > b ? a : b);
            desc_ = get<uint64_t>((size_t)3 * world * p_ + 3);
        } catch (...) {
            release();
            throw;
        }
    }
    ~Shard() override { release(); }

    void info(caps_sa_shard_info* o) const override
    {
        o->n = n_; o->p = p_; o->ppp = ppp_; o->rank = (uint32_t)rank_; o->world = (uint32_t)world_;
        o->g0 = g0_; o->g1 = g1_; o->bits_per_char = (uint32_t)bits_; o->idx_bytes = sizeof(idx_t);
        o->local_elems = local_n_; o->m_local = m_local_; o->m_total = m_total_;
        o->recv_total = recv_total_; o->slice_off = slice_off_; o->capacity = cap_;
        o->part_lo = jlo_; o->part_hi = jhi_;
        o->ms_phase1 = ms_phase1_; o->ms_pivots = ms_pivots_; o->ms_collate = ms_collate_; o->ms_phase2 = ms_phase2_;
        o->direct_fallback = (uint32_t)direct_fb_; o->direct_groups = K1_; o->direct_sub = SUB_;
        o->n_streams = n_streams_; o->stream_cap = capA_;
        o->send_capacity = std::max<uint64_t>(local_n_, direct_fb_ != CAPS_SA_FB_NONE ? 0 :
                                              local_ ? ((uint64_t)K1_ + world_ - 1) / world_ * SUB_ * capA_ : (uint64_t)n_streams_ * capA_);
        o->exchange = direct_fb_ == CAPS_SA_FB_NONE && local_ ? 0u : 1u;
        o->direct_quantile = quantile_ ? 1u : 0u;
        o->run_buckets = run_buckets_;
        o->tie_groups_deferred = tie_groups_;
        o->tie_levels = tie_levels_;
        o->reserved_ = 0;
        o->ms_scatter = ms_scatter_; o->ms_sort = ms_sort_;
        o->key_bytes = key_bits_ / 8;
        o->ms_level_a = ms_level_a_; o->ms_level_b = ms_level_b_ + ms_count_; o->ms_tile_sort = ms_tile_sort_; o->ms_merge_passes = ms_merge_;
        o->level_a_elems = my_elems_;
        o->slot_splits = slot_stats_[0]; o->slot_splits_redone = slot_stats_[1];
    }

    void phase1(void* d_sample_keys, void* d_sample_sa) override
    {
        slot_stats_[0] = slot_stats_[1] = 0;
        BackendEvent e0 = be_.record();
        bits_ = prepare_text(be_, dT_, n_, P_, present_, lut_);
        if (G_) {
            CAPS_LAUNCH(uniform_segments_kernel, (G_ + 256) / 256, 256, be_, seg1_.seg_start, G_, s_, local_n_);
            const uint64_t last = local_n_ - (uint64_t)(G_ - 1) * s_;
            n_tiles1_ = (G_ - 1) * tiles_of(s_) + tiles_of(last);
            prepare_segments(be_, seg1_, n_tiles1_);
            cur_ = (bits_ == 2 ? sort<2>(seg1_, n_tiles1_, last, true, A_, B_, local_n_, text_base_, false, false, &bk_, true)
                               : sort<8>(seg1_, n_tiles1_, last, true, A_, B_, local_n_, text_base_, false, false, &bk_, true)).uniform();
            oth_ = cur_.key == A_.key ? B_ : A_;
            CAPS_LAUNCH((sample_kernel<idx_t>), (uint32_t)((m_local_ + 255) / 256), 256, be_, (const uint64_t*)seg1_.seg_start, G_,
                        ppp_, (const uint64_t*)cur_.key, (const idx_t*)cur_.sa, static_cast<uint64_t*>(d_sample_keys),
                        static_cast<idx_t*>(d_sample_sa));
        }
        BackendEvent e1 = be_.record();
        be_.sync();
        ms_phase1_ = be_.elapsed_ms(e0, e1);
        be_.release_events();
    }

    // the sorted subarrays of this rank after phase1 (differential tests): local element i of subarray g is at g * s + i
    void phase1_arrays(void* d_keys_out, void* d_sa_out, uint64_t* count, uint64_t* subarray_len) override
    {
        *count = local_n_;
        *subarray_len = s_;
        if (d_keys_out && cur_.key) be_.d2d(d_keys_out, cur_.key, local_n_ * sizeof(uint64_t));
        if (d_sa_out && cur_.sa) be_.d2d(d_sa_out, cur_.sa, local_n_ * sizeof(idx_t));
        be_.sync();
    }

    void pivots(const void* d_all_keys, const void* d_all_sa, void* d_local_sizes) override
    {
        BackendEvent e0 = be_.record();
        be_.d2d(SA_.key, d_all_keys, m_total_ * sizeof(uint64_t));
        be_.d2d(SA_.sa, d_all_sa, m_total_ * sizeof(idx_t));
        CAPS_LAUNCH(uniform_segments_kernel, 1, 256, be_, segS_.seg_start, 1u, m_total_, m_total_);
        prepare_segments(be_, segS_, tiles_of(m_total_));
        ElemBuf<idx_t> smp = (bits_ == 2 ? sort<2>(segS_, tiles_of(m_total_), m_total_, false, SA_, SB_, m_total_, 0, false, false)
                                         : sort<8>(segS_, tiles_of(m_total_), m_total_, false, SA_, SB_, m_total_, 0, false, false)).uniform();
        CAPS_LAUNCH((pick_pivots_kernel<idx_t>), (p_ + 255) / 256, 256, be_, (const uint64_t*)smp.key, (const idx_t*)smp.sa,
                    m_total_, p_, pkey_, psa_);
        const uint32_t np = p_ - 1, bpr = (np + 255) / 256;
        if (G_) {
            if (bits_ == 2)
                CAPS_LAUNCH((locate_kernel<idx_t, 2>), capped_grid((uint64_t)G_ * bpr, 256), 256, be_, (const uint32_t*)P_, n_, (const uint64_t*)seg1_.seg_start, G_,
                            (const uint64_t*)cur_.key, (const idx_t*)cur_.sa, (const uint64_t*)pkey_, (const idx_t*)psa_, np, Pm_);
            else
                CAPS_LAUNCH((locate_kernel<idx_t, 8>), capped_grid((uint64_t)G_ * bpr, 256), 256, be_, (const uint32_t*)P_, n_, (const uint64_t*)seg1_.seg_start, G_,
                            (const uint64_t*)cur_.key, (const idx_t*)cur_.sa, (const uint64_t*)pkey_, (const idx_t*)psa_, np, Pm_);
        }
        CAPS_LAUNCH((partition_partial_kernel<idx_t>), ((p_ + 255) / 256) * PART_CHUNKS, 256, be_, (const idx_t*)Pm_, G_, p_, partial_);
        CAPS_LAUNCH((partition_sizes_kernel<idx_t>), ((p_ + 255) / 256) * PART_CHUNKS, 256, be_, (const idx_t*)Pm_, G_, p_,
                    (const uint64_t*)partial_, ruler_, sizes_);
        be_.d2d(d_local_sizes, sizes_, (size_t)p_ * sizeof(uint64_t));
        BackendEvent e1 = be_.record();
        be_.sync();
        ms_pivots_ = be_.elapsed_ms(e0, e1);
        be_.release_events();
    }

    void collate(const uint64_t* all_sizes, void* d_send_keys, void* d_send_sa, uint64_t* send_counts,
                 uint64_t* recv_counts) override
    {
        BackendEvent e0 = be_.record();
        // global partition sizes and size-balanced contiguous ownership
        std::vector<uint64_t> gs(p_, 0);
        for (int r = 0; r < world_; ++r)
            for (uint32_t j = 0; j < p_; ++j) gs[j] += all_sizes[(size_t)r * p_ + j];
        std::vector<uint32_t> lo(world_ + 1, p_);
        {
            uint64_t cum = 0;
            uint32_t j = 0;
            for (int r = 0; r < world_; ++r) {
                lo[r] = j;
                // partition j belongs to rank min(world-1, floor(midpoint * world / n))
                while (j < p_) {
                    const long double mid = (long double)cum + (long double)gs[j] / 2;
                    uint64_t own = (uint64_t)(mid * world_ / (long double)n_);
                    if (own >= (uint64_t)world_) own = world_ - 1;
                    if ((int)own > r) break;
                    cum += gs[j];
                    ++j;
                }
            }
            lo[world_] = p_;
        }
        jlo_ = lo[rank_];
        jhi_ = lo[rank_ + 1];
        G2_ = jhi_ - jlo_;
        slice_off_ = 0;
        for (uint32_t j = 0; j < jlo_; ++j) slice_off_ += gs[j];
        recv_total_ = 0;
        for (uint32_t j = jlo_; j < jhi_; ++j) recv_total_ += gs[j];
        for (int r = 0; r < world_; ++r) {
            uint64_t sc = 0, rc = 0;
            for (uint32_t j = lo[r]; j < lo[r + 1]; ++j) sc += all_sizes[(size_t)rank_ * p_ + j];
            for (uint32_t j = jlo_; j < jhi_; ++j) rc += all_sizes[(size_t)r * p_ + j];
            send_counts[r] = sc;
            recv_counts[r] = rc;
        }
        // local collate: partition-major == destination-major
        CAPS_LAUNCH(scan_sizes_kernel, 1, 1024, be_, (const uint64_t*)sizes_, p_, lstart_);
        if (G_)
        {
            uint32_t* first_part = reinterpret_cast<uint32_t*>(tdesc_);        // [n_tiles1_], idle between the sorts
            CAPS_LAUNCH((collate_plan_kernel<idx_t>), (n_tiles1_ + 255) / 256, 256, be_, seg1_.desc(), p_, (const idx_t*)Pm_, first_part);
            CAPS_LAUNCH((collate_kernel<idx_t>), n_tiles1_, TILE_NT, be_, seg1_.desc(), p_, (const idx_t*)Pm_, (const idx_t*)ruler_,
                        (const uint64_t*)lstart_, (const uint32_t*)first_part, (const uint64_t*)cur_.key, (const idx_t*)cur_.sa,
                        static_cast<uint64_t*>(d_send_keys), static_cast<idx_t*>(d_send_sa));
        }

        // phase 2 re-uses the phase-1 arrays (dead once the send buffers are filled).  Ownership is by partition midpoint,
        // so a rank can own up to n / world plus one whole partition: with few partitions per rank that can exceed the
        // buffers.  Every rank sees the same sizes: all of them fail together (a rank failing alone would leave the
        // others waiting in the exchange).
        {
            const uint64_t share = n_ / (uint64_t)world_ + 1, common_cap = share + share / 4 + 16 * TILE_E;
            for (int r = 0; r < world_; ++r) {
                uint64_t own = 0;
                for (uint32_t j = lo[r]; j < lo[r + 1]; ++j) own += gs[j];
                if (own > common_cap)
                    throw std::runtime_error("partition ownership is more imbalanced than the shards' capacity allows (rank " +
                                             std::to_string(r) + " would own " + std::to_string(own) + " of " + std::to_string(n_) +
                                             " suffixes): use more subproblems per GPU");
            }
        }
        seg2_.G = G2_;
        std::vector<uint64_t> st((size_t)G2_ + 1, 0);
        max_len2_ = 0;
        n_tiles2_ = 0;
        for (uint32_t k = 0; k < G2_; ++k) {
            const uint64_t z = gs[jlo_ + k];
            st[k + 1] = st[k] + z;
            max_len2_ = z > max_len2_ ? z : max_len2_;
            n_tiles2_ += tiles_of(z);
        }
        if (G2_) be_.h2d(seg2_.seg_start, st.data(), st.size() * sizeof(uint64_t));
        // regroup descriptors: run (source r, partition k)
        std::vector<uint64_t> desc;
        desc.reserve((size_t)3 * world_ * G2_);
        uint64_t src = 0;
        std::vector<uint64_t> fill(G2_, 0);
        for (int r = 0; r < world_; ++r)
            for (uint32_t k = 0; k < G2_; ++k) {
                const uint64_t z = all_sizes[(size_t)r * p_ + jlo_ + k];
                desc.push_back(src);
                desc.push_back(st[k] + fill[k]);
                desc.push_back(z);
                src += z;
                fill[k] += z;
            }
        n_desc_ = (uint32_t)(desc.size() / 3);
        if (!desc.empty()) be_.h2d(desc_, desc.data(), desc.size() * sizeof(uint64_t));
        BackendEvent e1 = be_.record();
        be_.sync();                      // st/desc live on this frame
        ms_collate_ = be_.elapsed_ms(e0, e1);
        be_.release_events();
    }

    void phase2(const void* d_recv_keys, const void* d_recv_sa, void* dSA, void* dLCP) override
    {
        const bool dbg = std::getenv("CAPS_SA_DEBUG") != nullptr;
        auto mark = [&](const char* what) { if (dbg) { be_.sync(); std::fprintf(stderr, "[shard %d] phase2: %s\n", rank_, what); } };
        BackendEvent e0 = be_.record();
        mark("start");
        if (recv_total_) {
            if (n_desc_)
                CAPS_LAUNCH((regroup_kernel<idx_t>), n_desc_, 256, be_, (const uint64_t*)desc_, static_cast<const uint64_t*>(d_recv_keys),
                            static_cast<const idx_t*>(d_recv_sa), A_.key, A_.sa);
            mark("regrouped");
            prepare_segments(be_, seg2_, n_tiles2_);
            mark("segments prepared");
            if (bits_ == 2) {
                SortResult<idx_t> r = sort<2>(seg2_, n_tiles2_, max_len2_, false, A_, B_, recv_total_, 0, true, true, &bk_, false, true, dSA, dLCP);
                finalize<idx_t, 2>(be_, P_, n_, r, static_cast<idx_t*>(dSA), static_cast<idx_t*>(dLCP));
            } else {
                SortResult<idx_t> r = sort<8>(seg2_, n_tiles2_, max_len2_, false, A_, B_, recv_total_, 0, true, true, &bk_, false, true, dSA, dLCP);
                finalize<idx_t, 8>(be_, P_, n_, r, static_cast<idx_t*>(dSA), static_cast<idx_t*>(dLCP));
            }
            mark("sorted + finalized");
            dSA_ = static_cast<idx_t*>(dSA);
        }
        BackendEvent e1 = be_.record();
        be_.sync();
        ms_phase2_ = be_.elapsed_ms(e0, e1);
        be_.release_events();
    }

    // ---- direct path ---------------------------------------------------------------------------------------
    // d_send_keys: u64[send_capacity], d_send_sa: idx[send_capacity]; d_report: u64[n_streams + 2] (stream_report_kernel)
    void scatter(void* d_send_keys, void* d_send_sa, void* d_report) override
    {
        last_send_keys_ = d_send_keys;
        last_send_sa_ = d_send_sa;
        if (direct_fb_ != CAPS_SA_FB_NONE) throw std::invalid_argument("this shard's shape does not allow the direct path");
        slot_stats_[0] = slot_stats_[1] = 0;
        level_a_ran_ = false;
        BackendEvent e0 = be_.record();
        bits_ = prepare_text(be_, dT_, n_, P_, present_, lut_);
        uint64_t* report = static_cast<uint64_t*>(d_report);
        be_.memset(report, 0, ((size_t)n_streams_ + 2) * sizeof(uint64_t));
        be_.memset(dstat_, 0, 4 * sizeof(uint64_t));
        uint32_t* dflag = reinterpret_cast<uint32_t*>(dstat_ + 2);
        quantile_ = false;
        if (be_.long_runs && !quantile_ok_) {
            const uint32_t code = CAPS_SA_FB_LONG_RUNS;        // every rank sees the same text: all report the same
            be_.h2d(dflag, &code, sizeof code);
        } else {
            // 32-bit keys (text.h key32_of) whenever the elements are going to cross xGMI (exchange mode): a third fewer bytes on the wire
            // (8 + w -> 4 + w per suffix).  On one GPU they cost more than they save, so a world of one keeps 64
            // bits unless CAPS_SA_KEYS=32 asks (tests); CAPS_SA_KEYS=64 and set_key_bits(64) (the retry after a slot overflow
            // in level B) force 64.
            const char* ke = std::getenv("CAPS_SA_KEYS");
            const bool want32 = ke ? std::string(ke) == "32" : world_ > 1 && !local_;
            key_bits_ = want32 && !force64_ && bits_ == 2 ? 32u : 64u;
            if (bits_ == 2) scatter_bits<2>(d_send_keys, d_send_sa, dflag);
            else scatter_bits<8>(d_send_keys, d_send_sa, dflag);
        }
        CAPS_LAUNCH((stream_report_kernel<idx_t>), (n_streams_ + 256) / 256, 256, be_, (const idx_t*)dcur_, n_streams_, capA_,
                    (const uint32_t*)dflag, report, quantile_ ? (const uint64_t*)(rcap_ + (size_t)own_lo_ * SUB_) : (const uint64_t*)nullptr,
                    K1_, SUB_, local_ ? own_lo_ : 0u, local_ ? own_hi_ : K1_);
        BackendEvent e1 = be_.record();
        be_.sync();
        ms_scatter_ = be_.elapsed_ms(e0, e1);
        ms_level_a_ = level_a_ran_ ? be_.elapsed_ms(a0_, a1_) : 0.0;
        be_.release_events();
    }

    // all_reports: HOST u64[world][n_streams + 2] (every rank's report).  Returns CAPS_SA_FB_NONE and the exchange's counts
    // (elements per peer, gaps included), or the reason why every rank must take the samplesort path instead.
    int plan(const uint64_t* all_reports, uint64_t* send_counts, uint64_t* recv_counts) override
    {
        if (direct_fb_ != CAPS_SA_FB_NONE) return direct_fb_;
        const size_t W = (size_t)n_streams_ + 2;
        for (int r = 0; r < world_; ++r) {
            const uint64_t flag = all_reports[r * W + n_streams_], over = all_reports[r * W + n_streams_ + 1];
            if (flag == CAPS_SA_FB_LONG_RUNS) return CAPS_SA_FB_LONG_RUNS;
            if (flag != 0) return CAPS_SA_FB_PIVOT_TIES;
            if (over != 0) return CAPS_SA_FB_GROUP_OVERFLOW;
        }
        auto bound = [&](int d) { return (uint32_t)((uint64_t)d * K1_ / world_); };          // rank d owns groups [bound(d), bound(d+1))
        jlo_ = bound(rank_);
        jhi_ = bound(rank_ + 1);
        G2_ = jhi_ - jlo_;
        const int srcs = local_ ? 1 : world_;                        // ranks whose streams of an owned group this rank sorts
        for (int d = 0; d < world_; ++d) {
            send_counts[d] = local_ ? 0 : (uint64_t)(bound(d + 1) - bound(d)) * SUB_ * capA_;
            recv_counts[d] = local_ ? 0 : (uint64_t)G2_ * SUB_ * capA_;
        }
        if ((uint64_t)G2_ * SUB_ * capA_ * srcs > cap_) throw std::runtime_error("receive buffer too small for the owned streams");
        // size of stream x of group g at rank r: the cursors are stream-major (group_scatter_kernel)
        auto sz = [&](int r, uint32_t g, uint32_t x) { return all_reports[r * W + (size_t)x * K1_ + g]; };
        slice_off_ = 0;
        for (uint32_t g = 0; g < jlo_; ++g)
            for (int r = 0; r < world_; ++r)
                for (uint32_t x = 0; x < SUB_; ++x) slice_off_ += sz(r, g, x);
        // level B's segments: (owned group, source rank, sub-stream), all sub-streams of a group consecutive
        const uint32_t per_group = (uint32_t)srcs * SUB_;
        const size_t nseg = (size_t)G2_ * per_group;
        if (nseg > seg_cap_) throw std::runtime_error("more level-B segments than the shard's tables hold");
        std::vector<uint64_t> st(nseg + 1, 0), en(nseg + 1, 0);
        recv_total_ = 0;
        max_len2_ = 0;
        n_tiles2_ = 0;
        const uint64_t block = (uint64_t)G2_ * SUB_ * capA_;                                // elements received from one rank
        for (uint32_t k = 0; k < G2_; ++k)
            for (int r = 0; r < srcs; ++r)
                for (uint32_t x = 0; x < SUB_; ++x) {
                    const size_t s = ((size_t)k * srcs + r) * SUB_ + x;
                    const uint64_t z = sz(local_ ? rank_ : r, jlo_ + k, x);
                    st[s] = quantile_ ? h_rstart_[(size_t)k * SUB_ + x] - h_rstart_[0]      // (local mode only: r = 0)
                                      : (uint64_t)r * block + ((uint64_t)k * SUB_ + x) * capA_;
                    en[s] = st[s] + z;
                    recv_total_ += z;
                    max_len2_ = z > max_len2_ ? z : max_len2_;
                    n_tiles2_ += tiles_of(z);
                }
        seg2_.G = (uint32_t)nseg;
        if (nseg) {
            be_.h2d(seg2_.seg_start, st.data(), (nseg + 1) * sizeof(uint64_t));
            be_.h2d(seg2_.seg_end, en.data(), (nseg + 1) * sizeof(uint64_t));
            be_.sync();                                  // st / en live on this frame
        }
        direct_planned_ = true;
        return CAPS_SA_FB_NONE;
    }

    // d_recv_*: the blocks received from ranks 0 .. world-1, in rank order (no exchange -- shard_info.exchange = 0 --: the
    // send buffers themselves)
    // Returns 0, or CAPS_SA_FB_KEY32 when a slot of level B overflowed under 32-bit keys (nothing sorted): the caller makes
    // ALL ranks agree (max over ranks), calls set_key_bits(64) and repeats scatter / exchange / sort.
    int sort_owned(const void* d_recv_keys, const void* d_recv_sa, void* dSA, void* dLCP) override
    {
        if (!direct_planned_) throw std::invalid_argument("shard_plan has not accepted the direct path");
        BackendEvent e0 = be_.record();
        int code = CAPS_SA_FB_NONE;
        if (recv_total_) {
            ::caps::prepare_segments(be_, seg2_, n_tiles2_ + 1, nullptr, nullptr, true);
            // large groups of equal keys are re-keyed, not compared (kernels.h "Deferred ties"): when every rank sorts the streams it
            // scattered itself (no exchange) under 64-bit keys.  Groups that do not fit the work memory, or need more levels than
            // msd_refine allows: level A again -- the sort has used its buffers -- and the sort with every tie compared
            bool retry = false;
            // ties are deferred only when the streams sorted here are the ones THIS shard scattered last (d_recv_* = the send buffers of
            // scatter()): a failed refinement is repaired by scattering into them again -- buffers of the caller's own are inputs, never
            // rewritten, and their ties are compared
            const bool own_streams = d_recv_keys == last_send_keys_ && d_recv_sa == last_send_sa_;
            const bool defer = local_ && own_streams && key_bits_ == 64 && !std::getenv("CAPS_SA_NO_DEFER");
            bool ok = bits_ == 2 ? sort_owned_bits<2>(d_recv_keys, d_recv_sa, dSA, dLCP, defer, &retry)
                                 : sort_owned_bits<8>(d_recv_keys, d_recv_sa, dSA, dLCP, defer, &retry);
            if (ok && retry) {
                uint32_t* dflag = reinterpret_cast<uint32_t*>(dstat_ + 2);
                be_.memset(dstat_, 0, 4 * sizeof(uint64_t));
                if (bits_ == 2) scatter_bits<2>(const_cast<void*>(d_recv_keys), const_cast<void*>(d_recv_sa), dflag);
                else scatter_bits<8>(const_cast<void*>(d_recv_keys), const_cast<void*>(d_recv_sa), dflag);
                ::caps::prepare_segments(be_, seg2_, n_tiles2_ + 1, nullptr, nullptr, true);
                ok = bits_ == 2 ? sort_owned_bits<2>(d_recv_keys, d_recv_sa, dSA, dLCP, false, &retry)
                                : sort_owned_bits<8>(d_recv_keys, d_recv_sa, dSA, dLCP, false, &retry);
            }
            if (!ok) code = CAPS_SA_FB_KEY32;
            dSA_ = static_cast<idx_t*>(dSA);
        }
        BackendEvent e1 = be_.record();
        be_.sync();
        ms_sort_ = be_.elapsed_ms(e0, e1);
        double* out[4] = {&ms_tile_sort_, &ms_level_b_, &ms_count_, &ms_merge_};
        for (size_t c = 0; c < clocks_.size() && c < 4; ++c) {
            *out[c] = 0;
            for (auto& sp : clocks_[c].spans) *out[c] += be_.elapsed_ms(sp.first, sp.second);
        }
        clocks_.clear();
        be_.release_events();
        return code;
    }

    void set_key_bits(int bits) override { force64_ = bits == 64; }

    uint64_t last_sa() override
    {
        if (!recv_total_) return ~0ull;
        idx_t v = 0;
        be_.d2h(&v, dSA_ + (recv_total_ - 1), sizeof(idx_t));
        be_.sync();
        return (uint64_t)v;
    }

    void fix_first_lcp(uint64_t prev_sa, void* dLCP) override
    {
        if (!recv_total_ || prev_sa == ~0ull) return;
        if (bits_ == 2)
            CAPS_LAUNCH((first_lcp_kernel<idx_t, 2>), 1, 64, be_, (const uint32_t*)P_, n_, prev_sa, (const idx_t*)dSA_,
                        static_cast<idx_t*>(dLCP));
        else
            CAPS_LAUNCH((first_lcp_kernel<idx_t, 8>), 1, 64, be_, (const uint32_t*)P_, n_, prev_sa, (const idx_t*)dSA_,
                        static_cast<idx_t*>(dLCP));
        be_.sync();
    }

private:
    Backend be_;
    std::vector<void*> owned_;
    const uint8_t* dT_;
    uint64_t n_;
    int rank_, world_;
    uint32_t p_ = 0, ppp_ = 0, g0_ = 0, g1_ = 0, G_ = 0, G2_ = 0, jlo_ = 0, jhi_ = 0, n_tiles1_ = 0, n_tiles2_ = 0, n_desc_ = 0;
    uint64_t s_ = 0, text_base_ = 0, local_n_ = 0, m_local_ = 0, m_total_ = 0, recv_total_ = 0, slice_off_ = 0, max_len2_ = 0;
    int bits_ = 0;
    uint32_t slot_stats_[2] = {0, 0};    // bucket splits of the current build: kept with slots / redone
    uint32_t run_buckets_ = 0;           // letter-run buckets of the last sort_owned (text.h "letter runs")
    uint64_t tie_groups_ = 0;            // groups of equal keys the last sort_owned deferred (kernels.h "Deferred ties")
    uint32_t tie_levels_ = 0;
    uint32_t h_flag_ = 0;                // scatter_bits: the fallback word on its way to the device
    uint32_t* P_ = nullptr;
    uint32_t* present_ = nullptr;
    uint8_t* lut_ = nullptr;
    ElemBuf<idx_t> A_, B_, SA_, SB_, cur_, oth_;
    uint64_t cap_ = 0;
    idx_t* dSA_ = nullptr;
    BucketBufs bk_;
    TileDesc* tdesc_ = nullptr;
    SegBufs seg1_, segS_, seg2_;
    uint64_t *pkey_ = nullptr, *sizes_ = nullptr, *lstart_ = nullptr, *desc_ = nullptr, *partial_ = nullptr;
    idx_t *psa_ = nullptr, *Pm_ = nullptr, *ruler_ = nullptr;
    double ms_phase1_ = 0, ms_pivots_ = 0, ms_collate_ = 0, ms_phase2_ = 0, ms_scatter_ = 0, ms_sort_ = 0;
    // direct path
    double ms_level_a_ = 0, ms_level_b_ = 0, ms_tile_sort_ = 0, ms_count_ = 0, ms_merge_ = 0;
    BackendEvent a0_, a1_;
    std::vector<KernelClock> clocks_;
    int direct_fb_ = CAPS_SA_FB_SHAPE;
    bool direct_planned_ = false;
    uint32_t seg_cap_ = 0;               // entries of seg2_ (level B's segments)
    bool local_ = true;                  // no exchange: every rank scatters the whole text and keeps its groups
    bool quantile_ok_ = false, quantile_ = false, skewed_ = false, level_a_ran_ = false;   // quantile level B (local mode only)
    uint32_t KPG_ = 0;
    uint64_t NB_ = 0, m2_ = 0;
    uint64_t *knots_ = nullptr, *rcap_ = nullptr, *rstart_ = nullptr;
    TileInfo* hot_rec_ = nullptr;
    uint64_t spill_stats_[3] = {0, 0, 0};
    std::vector<uint64_t> h_rstart_, h_rcap_;       // regions of the owned streams (host copy for shard_plan)
    uint32_t PG_ = 0, K1_ = 0, SUB_ = 1, n_streams_ = 0, my_tiles_ = 0, own_lo_ = 0, own_hi_ = 0;
    uint64_t capA_ = 0, my_elems_ = 0;
    uint32_t key_bits_ = 64;             // width of the keys the last scatter() wrote (32: text.h key32_of)
    const void* last_send_keys_ = nullptr;      // the buffers scatter() filled last (sort_owned may scatter into them again)
    const void* last_send_sa_ = nullptr;
    bool force64_ = false;               // set_key_bits(64): after a slot overflow under 32-bit keys
    uint8_t* gshift_ = nullptr;
    uint64_t *gkey_ = nullptr, *dstat_ = nullptr;
    uint16_t* glut_ = nullptr;
    idx_t* dcur_ = nullptr;

    // pivots (identical on every rank: same text, same samples) + level A over this rank's tiles
    template <int BITS> void scatter_bits(void* d_send_keys, void* d_send_sa, uint32_t* dflag)
    {
        const uint64_t m = m_total_;
        CAPS_LAUNCH((sample_text_kernel<idx_t, BITS>), (uint32_t)((m + 255) / 256), 256, be_, (const uint32_t*)P_, (uint64_t)0, n_, m,
                    SA_.key, SA_.sa);
        CAPS_LAUNCH(uniform_segments_kernel, 1, 256, be_, segS_.seg_start, 1u, m, m);
        prepare_segments(be_, segS_, tiles_of(m));
        ElemBuf<idx_t> smp = sort<BITS>(segS_, tiles_of(m), m, false, SA_, SB_, m, 0, false, false).uniform();
        CAPS_LAUNCH((pick_pivots_kernel<idx_t>), (p_ + 255) / 256, 256, be_, (const uint64_t*)smp.key, (const idx_t*)smp.sa, m, p_,
                    pkey_, psa_);
        CAPS_LAUNCH(group_keys_kernel, (p_ + 255) / 256, 256, be_, (const uint64_t*)pkey_, p_, PG_, K1_, gkey_, dflag);
        // ---- linear or quantile level B (pipeline.h Builder::run_direct; every rank decides the same from the same pivots)
        const uint64_t* rs = nullptr;
        const uint64_t* rc = nullptr;
        if (quantile_ok_) {
            CAPS_LAUNCH(skew_probe_kernel, (p_ + 255) / 256, 256, be_, (const uint64_t*)pkey_, p_, PG_, K1_, dflag + 2);
            uint32_t probe[4];
            be_.d2h(probe, dflag, sizeof probe);
            be_.sync();                               // {pivot-key ties, -, skewed, longest run of equal pivot keys}
            quantile_ = probe[0] != 0 || probe[2] != 0 || be_.long_runs;
            if (const char* mode = std::getenv("CAPS_SA_DIRECT_MODE")) {
                if (std::string(mode) == "linear" && !be_.long_runs) quantile_ = false;
                if (std::string(mode) == "quantile") quantile_ = true;
            }
            skewed_ = probe[2] != 0;
            if (quantile_) key_bits_ = 64;            // (knot buckets carry 64-bit keys)
            const uint64_t token = 2 * GA_E / K1_ + 16;
            uint32_t& code = h_flag_;                 // (a member: the asynchronous copy below must not read a dead stack slot)
            code = 0;
            if (quantile_ && probe[3] > p_ / 4) code = CAPS_SA_FB_PIVOT_TIES;        // one key over a quarter of the text: the samplesort path's
            else if (quantile_ && capA_ <= 2 * token) code = CAPS_SA_FB_SHAPE;
            if (quantile_ && code == 0) {
                // more samples, sorted in the (still idle) element arrays; their quantiles are the bucket boundaries and, every
                // KPG-th, the group keys; groups that share a frequent key share the room of all of them (group_caps_kernel)
                CAPS_LAUNCH((sample_text_kernel<idx_t, BITS>), (uint32_t)((m2_ + 255) / 256), 256, be_, (const uint32_t*)P_, (uint64_t)0, n_, m2_,
                            A_.key, A_.sa);
                SegBufs sseg = seg1_;
                sseg.G = 1;
                CAPS_LAUNCH(uniform_segments_kernel, 1, 256, be_, sseg.seg_start, 1u, m2_, m2_);
                prepare_segments(be_, sseg, tiles_of(m2_));
                ElemBuf<idx_t> smp2 = sort<BITS>(sseg, tiles_of(m2_), m2_, false, A_, B_, m2_, 0, false, false, &bk_, true, false, nullptr, nullptr,
                                                 gkey_, K1_, skewed_).uniform();
                CAPS_LAUNCH(knots_kernel, (uint32_t)((NB_ + 255) / 256), 256, be_, (const uint64_t*)smp2.key, m2_, NB_, KPG_, K1_, knots_, gkey_);
                CAPS_LAUNCH(group_caps_kernel, (K1_ + 255) / 256, 256, be_, (const uint64_t*)knots_, NB_, KPG_, K1_, SUB_, capA_ - token, token, rcap_);
                CAPS_LAUNCH(scan_sizes_kernel, 1, 1024, be_, (const uint64_t*)rcap_, n_streams_, rstart_);
                const size_t own_streams = (size_t)(own_hi_ - own_lo_) * SUB_;
                h_rstart_.assign(own_streams + 1, 0);
                h_rcap_.assign(own_streams + 1, 0);
                be_.d2h(h_rstart_.data(), rstart_ + (size_t)own_lo_ * SUB_, (own_streams + 1) * sizeof(uint64_t));
                be_.d2h(h_rcap_.data(), rcap_ + (size_t)own_lo_ * SUB_, own_streams * sizeof(uint64_t));
                be_.sync();
                // the regions of the owned streams must fit the send buffers (a rank that owns a frequent key's groups gets more room)
                const uint64_t room = ((uint64_t)K1_ + world_ - 1) / world_ * SUB_ * capA_;
                if (h_rstart_[own_streams] - h_rstart_[0] > room) code = CAPS_SA_FB_GROUP_OVERFLOW;
                rs = rstart_ + (size_t)own_lo_ * SUB_;
                rc = rcap_ + (size_t)own_lo_ * SUB_;
            }
            // the word the other ranks see: pivot ties are quantile mode's business, not a reason to fall back
            if (quantile_ || code) be_.h2d(dflag, &code, sizeof code);
            if (code) { quantile_ = false; return; }                                  // (every rank: the same text, the same code)
        }
        CAPS_LAUNCH(split_lut_kernel, (SPLIT_LUT_CELLS + 256) / 256, 256, be_, (const uint64_t*)gkey_, K1_ - 1, glut_, dflag + 1);
        be_.memset(dcur_, 0, (size_t)n_streams_ * sizeof(idx_t));
        if (key_bits_ == 32) CAPS_LAUNCH(group_shift_kernel, (K1_ + 255) / 256, 256, be_, (const uint64_t*)gkey_, K1_, gshift_);
        // local: every tile of the text, the owned groups kept; exchange: every world-th tile, every group kept
        const uint32_t tile_first = local_ ? 0u : (uint32_t)rank_, tile_stride = local_ ? 1u : (uint32_t)world_;
        const uint32_t keep_lo = local_ ? own_lo_ : 0u, keep_hi = local_ ? own_hi_ : K1_;
        a0_ = be_.record();
        if (my_tiles_ && key_bits_ == 32)
            CAPS_LAUNCH((group_scatter_kernel<idx_t, BITS, uint32_t>), my_tiles_, TILE_NT, be_, (const uint32_t*)P_, packed_words(n_, BITS),
                        (uint64_t)0, n_, (const uint64_t*)gkey_, K1_, (const uint16_t*)glut_, (const uint32_t*)(dflag + 1), SUB_, capA_, dcur_,
                        static_cast<uint32_t*>(d_send_keys), static_cast<idx_t*>(d_send_sa), tile_first, tile_stride,
                        rs, rc, (const uint8_t*)gshift_, keep_lo, keep_hi);
        else if (my_tiles_)
            CAPS_LAUNCH((group_scatter_kernel<idx_t, BITS>), my_tiles_, TILE_NT, be_, (const uint32_t*)P_, packed_words(n_, BITS), (uint64_t)0,
                        n_, (const uint64_t*)gkey_, K1_, (const uint16_t*)glut_, (const uint32_t*)(dflag + 1), SUB_, capA_, dcur_,
                        static_cast<uint64_t*>(d_send_keys), static_cast<idx_t*>(d_send_sa), tile_first, tile_stride,
                        rs, rc, (const uint8_t*)nullptr, keep_lo, keep_hi);
        a1_ = be_.record();
        level_a_ran_ = true;
    }

    template <int BITS> bool sort_owned_bits(const void* d_recv_keys, const void* d_recv_sa, void* dSA, void* dLCP, bool defer, bool* retry)
    {
        *retry = false;
        tie_groups_ = 0;
        tie_levels_ = 0;
        SortOpts o;
        o.need_lcp = true;
        o.skip_finished = true;
        o.bk = &bk_;
        o.range_mode = 1;                         // group g holds the keys in (gkey[g-1], gkey[g]]
        o.pkey = gkey_;
        o.part_off = jlo_;
        o.part_total = K1_;
        o.sub = (local_ ? 1u : (uint32_t)world_) * SUB_;
        o.seg_ends = true;
        o.in_key = static_cast<const uint64_t*>(d_recv_keys);
        o.in_sa = d_recv_sa;
        o.final_sa = dSA;
        o.final_lcp = dLCP;
        o.bnd.first_key = bk_.first_key;
        o.bnd.last_key = bk_.last_key;
        o.bnd.first_sa = bk_.first_sa;
        o.bnd.last_sa = bk_.last_sa;
        o.slot_stats = slot_stats_;
        o.speculate = std::getenv("CAPS_SA_NO_SLOTS") == nullptr;
        if (key_bits_ == 32) { o.k32 = true; o.range_mode = 2; o.gshift = gshift_; o.speculate = true; }
        if (defer && bk_.fcount) {
            o.defer_flags = bk_.fcount;           // (the fine-count table: idle outside the equalised split)
            o.run_tables = &seg1_;                // (idle since level A; the bucket tables stay intact for msd_refine)
        }
        if (quantile_) {                          // the owned groups' buckets are (knots[k - 1], knots[k]], KPG per group
            o.knots = knots_ + (size_t)jlo_ * KPG_;
            o.knots_per_parent = KPG_;
            o.knots_have_prev = jlo_ > 0;                 // bucket 0 of the slice starts above the previous group's last knot
            o.in_extent = ((uint64_t)K1_ + world_ - 1) / world_ * SUB_ * capA_;   // the owned streams' regions fit this (scatter_bits)
            o.skewed_keys = skewed_ && !std::getenv("CAPS_SA_TRY_LINEAR_TILES");
            o.spill_slots = hot_rec_ != nullptr;      // no count pass (as Builder::run_direct)
            o.hot_tile_rec = hot_rec_;
            o.spill_stats = spill_stats_;
        }
        KernelClock tile_clock, scatter_clock, count_clock, merge_clock;
        o.tile_clock = &tile_clock;
        o.scatter_clock = &scatter_clock;
        o.count_clock = &count_clock;
        o.merge_clock = &merge_clock;
        SortResult<idx_t> r = segmented_sort<idx_t, BITS>(be_, P_, n_, tdesc_, seg2_, n_tiles2_, max_len2_, A_, B_, recv_total_, o);
        clocks_ = {tile_clock, scatter_clock, count_clock, merge_clock};
        run_buckets_ = (uint32_t)r.run_buckets.size();
        if (r.failed) return false;
        finalize<idx_t, BITS>(be_, P_, n_, r, static_cast<idx_t*>(dSA), static_cast<idx_t*>(dLCP));
        if (r.msd_failed) { *retry = true; return true; }
        tie_groups_ = r.msd_groups;
        tie_levels_ = r.msd_levels;
        return true;
    }

    template <typename T> T* get(size_t count)
    {
        T* p = static_cast<T*>(be_.alloc((count ? count : 1) * sizeof(T)));
        owned_.push_back(p);
        return p;
    }
    ElemBuf<idx_t> elems(uint64_t cnt)
    {
        // one allocation (key | sa | lcp): a speculative bucket split views it as its slot buffer
        ElemBuf<idx_t> b;
        const size_t c = cnt ? cnt : 1;
        const size_t key_bytes = (c * sizeof(uint64_t) + 255) & ~size_t(255), idx_bytes = (c * sizeof(idx_t) + 255) & ~size_t(255);
        char* base = get<char>(key_bytes + 2 * idx_bytes);
        b.key = reinterpret_cast<uint64_t*>(base);
        b.sa = reinterpret_cast<idx_t*>(base + key_bytes);
        b.lcp = reinterpret_cast<idx_t*>(base + key_bytes + idx_bytes);
        b.region_bytes = key_bytes + 2 * idx_bytes;
        return b;
    }
    SegBufs segs(uint32_t G, uint64_t cap)
    {
        SegBufs s;
        s.G = G;
        s.seg_start = get<uint64_t>((size_t)G + 1);
        s.tile_off = get<uint32_t>((size_t)G + 1);
        s.tile_rec = get<TileInfo>(cap);
        s.out2 = get<uint64_t>(2);
        return s;
    }
    void drop(void* p)
    {
        if (!p) return;
        for (size_t i = 0; i < owned_.size(); ++i)
            if (owned_[i] == p) { be_.free(p); owned_[i] = owned_.back(); owned_.pop_back(); return; }
    }
    void release()
    {
        for (void* p : owned_) be_.free(p);
        owned_.clear();
    }
    template <int BITS>
    SortResult<idx_t> sort(const SegBufs& s, uint32_t n_tiles, uint64_t max_len, bool from_text, ElemBuf<idx_t> a, ElemBuf<idx_t> b,
                           uint64_t n_elems, uint64_t text_base, bool need_lcp, bool skip_finished, const BucketBufs* bk = nullptr,
                           bool unify = false, bool between_pivots = false, void* final_sa = nullptr, void* final_lcp = nullptr,
                           const uint64_t* knots = nullptr, uint32_t knots_per_parent = 0, bool skewed = false)
    {
        SortOpts o;
        o.keys_only = knots != nullptr;           // (the second sample: its keys become the knots, nothing else is used)
        o.knots = knots;                          // (the second sample: split at the first sample's group keys, as Builder::run_direct)
        o.knots_per_parent = knots_per_parent;
        o.skewed_keys = skewed;
        if (final_sa && bk) {                     // completed segments go straight to the caller's arrays
            o.final_sa = final_sa;
            o.final_lcp = final_lcp;
            o.bnd.first_key = bk->first_key;
            o.bnd.last_key = bk->last_key;
            o.bnd.first_sa = bk->first_sa;
            o.bnd.last_sa = bk->last_sa;
        }
        if (between_pivots) {                     // owned partitions jlo_ .. jhi_-1 of p_
            o.range_mode = 1;
            o.pkey = pkey_;
            o.part_off = jlo_;
            o.part_total = p_;
        }
        o.from_text = from_text;
        o.text_base = text_base;
        o.need_lcp = need_lcp;
        o.skip_finished = skip_finished;
        o.bk = bk;
        o.unify = unify;
        o.slot_stats = slot_stats_;
        o.speculate = bk != nullptr && std::getenv("CAPS_SA_NO_SLOTS") == nullptr && slot_stats_[1] == 0;   // as in Builder::seg_sort
        return segmented_sort<idx_t, BITS>(be_, P_, n_, tdesc_, s, n_tiles, max_len, a, b, n_elems, o);
    }
    BucketBufs buckets(uint64_t n_elems, uint32_t G)
    {
        BucketBufs k;
        k.nb_cap = BucketBufs::bucket_bound(n_elems ? n_elems : 1, G);
        k.tile_cap = (n_elems ? n_elems : 1) / TILE_E + k.nb_cap + 2;
        k.params = get<BucketParams>(G);
        k.segB = get<uint64_t>(G);
        k.bstart = get<uint64_t>((size_t)G + 1);
        k.count = get<uint64_t>(k.nb_cap);
        k.cursor = get<idx_t>(k.nb_cap);
        k.scan_tmp = get<uint64_t>(2 * ((size_t)k.nb_cap / SCAN_CHUNK + 2));
        k.tile_map = get<BucketParams>(k.nb_cap);
        k.first_key = get<uint64_t>(k.nb_cap);
        k.last_key = get<uint64_t>(k.nb_cap);
        k.first_sa = get<idx_t>(k.nb_cap);
        k.last_sa = get<idx_t>(k.nb_cap);
        k.sub = segs(k.nb_cap, k.tile_cap);
        k.fparams = get<BucketParams>(G);
        k.fsegB = get<uint64_t>(G);
        k.fstart = get<uint64_t>((size_t)G + 1);
        k.fcount = get<uint64_t>((size_t)EQ_FINE * k.nb_cap);
        k.gfirst = get<uint32_t>(k.nb_cap);
        k.kshift = get<uint8_t>(k.nb_cap);
        k.skip = get<uint8_t>(k.nb_cap);
        k.run_list = get<uint64_t>(1 + 4 * RUN_BUCKET_MAX);
        return k;
    }
};

}  // namespace caps
