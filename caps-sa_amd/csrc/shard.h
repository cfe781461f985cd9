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
        try {
            // Everything is allocated ONCE here (hipMalloc / hipFree of tens of GB cost seconds): the
            // element arrays serve phase 1 (this rank's subarrays) and phase 2 (its partitions), sized
            // for the larger of the two with 25 % slack for the imbalance of the partition ownership.
            const uint64_t share = n / (uint64_t)world + 1;
            cap_ = std::max<uint64_t>(local_n_, share + share / 4 + 16 * TILE_E);
            P_ = get<uint32_t>(text_alloc_words(n));
            present_ = get<uint32_t>(8);
            lut_ = get<uint8_t>(256);
            A_ = elems(cap_);
            B_ = elems(cap_);
            seg1_ = segs(G_ ? G_ : 1, cap_ / TILE_E + p_ + 2);
            seg1_.G = G_;
            seg2_ = segs(p_, cap_ / TILE_E + p_ + 2);
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
            bk_ = buckets(cap_, p_);
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

        // phase 2 re-uses the phase-1 arrays (dead once the send buffers are filled)
        if (recv_total_ > cap_)
            throw std::runtime_error("partition ownership is more imbalanced than the shard's capacity allows");
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
    double ms_phase1_ = 0, ms_pivots_ = 0, ms_collate_ = 0, ms_phase2_ = 0;

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
                           bool unify = false, bool between_pivots = false, void* final_sa = nullptr, void* final_lcp = nullptr)
    {
        SortOpts o;
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
        return k;
    }
};

}  // namespace caps
