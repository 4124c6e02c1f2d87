// pack_cat -- what fastq_pack.hpp makes of a FASTQ(.gz) file, as text: pack_cat FILE K THREADS [quiet] writes every packed record's bases,
// one per line, in the order they come, then "reads N records M"; exit code 1 and a message on stderr for a damaged file.
// tests/test_fastq_pack.py compares the parallel reader (THREADS > 1) with the line loop (THREADS = 1).
#include <cstdio>
#include <cstdlib>

#include "fastq_pack.hpp"

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: pack_cat FILE K THREADS\n"); return 2; }
    const int k = atoi(argv[2]);
    const unsigned threads = (unsigned)atoi(argv[3]);
    const bool quiet = argc > 4;   // (a fourth argument: only the counts -- timing the reader, not the printing)
    try {
        bronko::FastqPacker in(argv[1], k, threads);
        bronko::PackedBatch b;
        uint64_t reads = 0, records = 0;
        std::string line;
        while (in.next(b)) {
            reads += b.n_reads; records += b.n_records;
            for (uint64_t r = 0; r < b.n_records && !quiet; r++) {
                line.clear();
                const uint32_t* w = b.words.data() + r * b.stride;
                for (uint32_t i = 0; i < b.lens[r]; i++) line.push_back("ACGT"[(w[i >> 4] >> (2 * (i & 15))) & 3u]);
                line.push_back('\n');
                fwrite(line.data(), 1, line.size(), stdout);
            }
        }
        printf("reads %llu records %llu\n", (unsigned long long)reads, (unsigned long long)records);
    } catch (const std::exception& e) {
        fflush(stdout);
        fprintf(stderr, "pack_cat: %s\n", e.what());
        return 1;
    }
    return 0;
}
