// bk_kernels.h -- launch interface between the engine (host) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "bk_device.h"

namespace bk {

struct ScanArgs {
    IndexView ix;
    const uint32_t* words;          // [n_records][stride_words] 2-bit packed, 16 bases per word, LSB first
    const uint16_t* lens;           // [n_records] valid bases
    uint64_t n_records;
    uint32_t stride_words;
    unsigned long long* counters;   // [n_slots][4 bases][2 orientations]
    unsigned long long* kmer_total; // optional: += k-mer occurrences scanned
};

struct FinalizeArgs {
    IndexView ix;
    const unsigned long long* counters;
    unsigned long long ci, cs, cx;
    unsigned long long* pileup;     // 4 planes of `plane` u64: fwd depth, rev depth, fwd #kmers, rev #kmers
    size_t plane;                   // total_cells * 4
    unsigned long long* stats;      // [n_files][3]
    unsigned char* present;         // [n_files]
    unsigned long long* kept_total; // optional: += distinct k-mers that passed the thresholds
};

void launch_scan_count(const ScanArgs& a, hipStream_t stream);
void launch_finalize(const FinalizeArgs& a, hipStream_t stream);
size_t finalize_lds_bytes(int n_files);

}  // namespace bk
