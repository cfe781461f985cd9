// caps-sa_amd/csrc/caps_sa_cli.cpp -- command-line driver with the reference CLI's contract
// (reference src/main.cpp:43-93):
//
//     caps_sa <input_path> <output_path> [subproblem-count] [bounded-context] [--pretty-print] [--gpus N]
//
// * --gpus N (not in the reference): shard the build over HIP devices 0 .. N-1;
// * bounded-context (main.cpp:57): 0 or >= the text length = the suffix array; a smaller bound takes the reference's own
//   sequence of merges (csrc/bounded.h: its output depends on them; a compatibility mode, seconds instead of milliseconds);
// * every input byte is remapped to {A,C,G,T} with lookup[(toupper(c) & 0x6) >> 1],
//   lookup = {A,C,T,G} (main.cpp:61-70) -- FASTA headers and newlines are kept and remapped;
// * n = file size; n <= UINT32_MAX selects 32-bit indices, else 64-bit (main.cpp:76-87);
// * output = Suffix_Array::dump format (u64 n, SA, LCP; src/Suffix_Array.cpp:497-509), or,
//   with --pretty-print (advertised at main.cpp:49 but never wired up there), the text
//   form of main.cpp:32-40: one line of SA values, one line of LCP values.
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "Suffix_Array.hpp"

static bool read_input(const std::string& path, std::string& text)
{
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) { std::cerr << path << " : cannot open\n"; return false; }
    const std::streamsize size = in.tellg();
    in.seekg(0);
    text.resize(static_cast<size_t>(size));
    in.read(&text[0], size);
    return static_cast<bool>(in) || size == 0;
}

// positional numeric argument: digits only, fits size_t
static bool parse_count(const std::string& s, size_t& out)
{
    if (s.empty() || s.size() > 19) return false;
    size_t v = 0;
    for (char c : s) {
        if (c < '0' || c > '9') return false;
        v = v * 10 + static_cast<size_t>(c - '0');
    }
    out = v;
    return true;
}

template <typename idx_t>
static int run(const std::string& text, const std::string& out_path, size_t p, size_t ctx, bool pretty, const std::vector<int>& devices)
{
    if (p > std::numeric_limits<idx_t>::max()) { std::cerr << "subproblem-count does not fit the index type\n"; return EXIT_FAILURE; }
    // the bounded-context mode replays the reference's merge history, one GPU thread per merge node -- the top levels
    // of its trees are sequential merges of O(n) elements (9.5 s at 64 Mi chars; the reference itself: 67 s on one thread there).
    // Say so before a genome-sized run looks like a hang.
    if (ctx != 0 && ctx < text.size() && text.size() >= (size_t(64) << 20))
        std::cerr << "bounded-context " << ctx << ": compatibility mode (csrc/bounded.h), about " << (text.size() >> 20) * 0.15
                  << " seconds for " << (text.size() >> 20) << " Mi chars; the unbounded construction takes milliseconds.\n";
    CaPS_SA::Suffix_Array<idx_t> suf_arr(text.c_str(), static_cast<idx_t>(text.size()), static_cast<idx_t>(p),
                                         static_cast<idx_t>(ctx >= text.size() ? 0 : ctx), devices);
    suf_arr.construct();
    const caps_sa_stats& st = suf_arr.stats();
    std::cerr << "Constructed the suffix array. Device time: " << st.ms_total / 1e3 << " seconds (h2d "
              << st.ms_h2d / 1e3 << ", d2h " << st.ms_d2h / 1e3 << "); subproblems " << st.p_eff << ", "
              << st.bits_per_char << " bits/char, " << devices.size() << " GPU(s).\n";
    std::ofstream output(out_path, std::ios::binary);           // only now: a failed build leaves no empty file behind
    if (!output) { std::cerr << out_path << " : cannot open for writing\n"; return EXIT_FAILURE; }
    if (pretty) {
        const size_t n = suf_arr.n();
        for (size_t i = 0; i < n; ++i) output << suf_arr.SA()[i] << " \n"[i == n - 1];
        for (size_t i = 0; i < n; ++i) output << suf_arr.LCP()[i] << " \n"[i == n - 1];
    } else
        suf_arr.dump(output);
    return output ? 0 : EXIT_FAILURE;
}

int main(int argc, char* argv[])
{
    std::vector<std::string> pos;
    bool pretty = false;
    size_t gpus = 1;
    for (int i = 1; i < argc; ++i) {
        if (std::strcmp(argv[i], "--pretty-print") == 0) pretty = true;
        else if (std::strcmp(argv[i], "--gpus") == 0 && i + 1 < argc) {
            if (!parse_count(argv[++i], gpus) || gpus < 1 || gpus > 64) { std::cerr << "--gpus: a device count, please\n"; return EXIT_FAILURE; }
        } else pos.push_back(argv[i]);
    }
    if (pos.size() < 2) {
        std::cerr << "Usage: caps_sa <input_path> <output_path> <(optional)-subproblem-count> "
                     "<(optional)-bounded-context> <(optional)--pretty-print> <(optional)--gpus N>\n";
        return EXIT_FAILURE;
    }
    size_t p = 0, ctx = 0;
    if (pos.size() >= 3 && !parse_count(pos[2], p)) { std::cerr << "subproblem-count: not a number: " << pos[2] << "\n"; return EXIT_FAILURE; }
    if (pos.size() >= 4 && !parse_count(pos[3], ctx)) { std::cerr << "bounded-context: not a number: " << pos[3] << "\n"; return EXIT_FAILURE; }

    std::string text;
    if (!read_input(pos[0], text)) return EXIT_FAILURE;
    // the byte remap of main.cpp:61-70, on all host threads like the reference's parallel_for
    {
        static const char lookup[4] = {'A', 'C', 'T', 'G'};
        const size_t n = text.size();
        const size_t nt = std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), n / (1u << 20) + 1));
        std::vector<std::thread> th;
        char* d = &text[0];
        for (size_t t = 0; t < nt; ++t)
            th.emplace_back([=]() {
                for (size_t j = n * t / nt, e = n * (t + 1) / nt; j < e; ++j)
                    d[j] = lookup[(std::toupper(static_cast<unsigned char>(d[j])) & 0x6) >> 1];
            });
        for (auto& t : th) t.join();
    }
    std::vector<int> devices;
    for (size_t i = 0; i < gpus; ++i) devices.push_back(static_cast<int>(i));

    std::cerr << "Text length: " << text.size() << ".\n";
    try {
        if (text.size() <= std::numeric_limits<uint32_t>::max()) return run<uint32_t>(text, pos[1], p, ctx, pretty, devices);
        return run<uint64_t>(text, pos[1], p, ctx, pretty, devices);
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return EXIT_FAILURE;
    }
}
