// caps-sa_amd/csrc/caps_sa_cli.cpp -- command-line driver with the reference CLI's contract
// (reference src/main.cpp:43-93):
//
//     caps_sa <input_path> <output_path> [subproblem-count] [bounded-context] [--pretty-print]
//
// * every input byte is remapped to {A,C,G,T} with lookup[(toupper(c) & 0x6) >> 1],
//   lookup = {A,C,T,G} (main.cpp:61-70) -- FASTA headers and newlines are kept and remapped;
// * n = file size; n <= UINT32_MAX selects 32-bit indices, else 64-bit (main.cpp:76-87);
// * output = Suffix_Array::dump format (u64 n, SA, LCP; src/Suffix_Array.cpp:497-509), or,
//   with --pretty-print (advertised at main.cpp:49 but never wired up there), the text
//   form of main.cpp:32-40: one line of SA values, one line of LCP values.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <string>
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

template <typename idx_t>
static int run(const std::string& text, std::ofstream& output, size_t p, size_t ctx, bool pretty)
{
    CaPS_SA::Suffix_Array<idx_t> suf_arr(text.c_str(), static_cast<idx_t>(text.size()), static_cast<idx_t>(p),
                                         static_cast<idx_t>(ctx));
    suf_arr.construct();
    const caps_sa_stats& st = suf_arr.stats();
    std::cerr << "Constructed the suffix array. Device time: " << st.ms_total / 1e3 << " seconds (h2d "
              << st.ms_h2d / 1e3 << ", d2h " << st.ms_d2h / 1e3 << "); subproblems " << st.p_eff << ", "
              << st.bits_per_char << " bits/char.\n";
    if (pretty) {
        const size_t n = suf_arr.n();
        for (size_t i = 0; i < n; ++i) output << suf_arr.SA()[i] << " \n"[i == n - 1];
        for (size_t i = 0; i < n; ++i) output << suf_arr.LCP()[i] << " \n"[i == n - 1];
    } else
        suf_arr.dump(output);
    return 0;
}

int main(int argc, char* argv[])
{
    std::vector<std::string> pos;
    bool pretty = false;
    for (int i = 1; i < argc; ++i) {
        if (std::strcmp(argv[i], "--pretty-print") == 0) pretty = true;
        else pos.push_back(argv[i]);
    }
    if (pos.size() < 2) {
        std::cerr << "Usage: caps_sa <input_path> <output_path> <(optional)-subproblem-count> "
                     "<(optional)-bounded-context> <(optional)--pretty-print>\n";
        return EXIT_FAILURE;
    }
    const size_t p = pos.size() >= 3 ? static_cast<size_t>(std::atoll(pos[2].c_str())) : 0;
    const size_t ctx = pos.size() >= 4 ? static_cast<size_t>(std::atoll(pos[3].c_str())) : 0;

    std::string text;
    if (!read_input(pos[0], text)) return EXIT_FAILURE;
    static const char lookup[4] = {'A', 'C', 'T', 'G'};
    for (size_t j = 0; j < text.size(); ++j)
        text[j] = lookup[(std::toupper(static_cast<unsigned char>(text[j])) & 0x6) >> 1];

    std::ofstream output(pos[1], std::ios::binary);
    std::cerr << "Text length: " << text.size() << ".\n";
    try {
        if (text.size() <= std::numeric_limits<uint32_t>::max()) return run<uint32_t>(text, output, p, ctx, pretty);
        return run<uint64_t>(text, output, p, ctx, pretty);
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return EXIT_FAILURE;
    }
}
