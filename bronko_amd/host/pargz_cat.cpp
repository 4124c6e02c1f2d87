// pargz_cat -- `zcat` through pargz.hpp: pargz_cat FILE [threads [chunk bytes]] writes the text to stdout; exit code 1 and a
// message on stderr for a damaged file (what was read before the damage is written, as with gzread).  tests/test_pargz.py.
// pargz_cat --lines FILE [threads]: the same through fastx.hpp's GzLineReader, the way `bronko call` opens its inputs (plain or
// gzip, a file or a stream), every line written back with "\n".
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fastx.hpp"
#include "pargz.hpp"

int main(int argc, char** argv) {
    if (argc >= 3 && !strcmp(argv[1], "--lines")) {
        try {
            bronko::GzLineReader in(argv[2], argc > 3 ? (unsigned)atoi(argv[3]) : 8u);
            std::string line;
            while (in.next(line)) { fwrite(line.data(), 1, line.size(), stdout); fputc('\n', stdout); }
        } catch (const std::exception& e) {
            fflush(stdout);
            fprintf(stderr, "pargz_cat: %s\n", e.what());
            return 1;
        }
        return 0;
    }
    if (argc < 2) { fprintf(stderr, "usage: pargz_cat FILE [threads [chunk bytes]] | pargz_cat --lines FILE [threads]\n"); return 2; }
    const unsigned threads = argc > 2 ? (unsigned)atoi(argv[2]) : 8u;
    const size_t chunk = argc > 3 ? (size_t)strtoull(argv[3], nullptr, 10) : 0;
    try {
        bronko::ParallelGunzip in(argv[1], threads, chunk);
        std::vector<char> buf(1u << 20);
        for (size_t n; (n = in.read(buf.data(), buf.size())) > 0;) fwrite(buf.data(), 1, n, stdout);
    } catch (const std::exception& e) {
        fflush(stdout);
        fprintf(stderr, "pargz_cat: %s\n", e.what());
        return 1;
    }
    return 0;
}
