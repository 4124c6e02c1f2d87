// bk_finalize_common.h -- device helpers shared by the finalize kernels (bk_kernels.hip: K2a / K2e / K2b; bk_finalize_lean.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "bk_device.h"
#include "bk_kernels.h"
#include "bk_scan_common.h"

namespace bk {

// The vote of call.rs:1327-1384 for one BucketInfo.
__device__ __forceinline__ void vote(const FinalizeArgs& a, const DevEntry& e, uint64_t c, uint32_t isrc, int k, unsigned long long v) {
    if (a.mode == 1 || (a.mode == 2 && (int)e.file != a.sel_file)) return;   // statistics pass / votes for the selected genome only
    uint32_t bit_idx;
    bool forward;
    if (e.canonical) {
        bit_idx = ((uint32_t)(c >> (2 * e.idx)) & 3u) ^ 3u;
        forward = isrc != 0;
    } else {
        bit_idx = (uint32_t)(c >> (2 * (k - 1 - e.idx))) & 3u;
        forward = isrc == 0;
    }
    const size_t cell = (size_t)e.cell * 4 + bit_idx;
    atomicAdd(a.pileup + (forward ? 2 : 3) * a.plane + cell, 1ull);   // #kmers  += 1
    atomicMax(a.pileup + (forward ? 0 : 1) * a.plane + cell, v);      // depth = max(depth, n)
}


// lstats[idx] += 1 from every active lane: the lanes of a wave walk the entry lists of their buckets in step, and with many
// genomes that share a k-mer they name the same genome at the same time -- one LDS atomic for all lanes that agree with the
// first active one instead of up to 64 on one address
__device__ __forceinline__ void tally(uint32_t* lstats, uint32_t idx) {
    const unsigned long long active = __ballot(true);
    const int leader = __builtin_ctzll(active);
    const uint32_t lidx = (uint32_t)__shfl((int)idx, leader);
    const bool same = idx == lidx;
    const unsigned long long sm = __ballot(same);
    if (!same) atomicAdd(&lstats[idx], 1u);
    else if ((int)(threadIdx.x & 63u) == leader) atomicAdd(&lstats[idx], (uint32_t)__popcll(sm));
}

// End of a finalize workgroup: per-genome tallies (LDS) and the kept / distinct k-mer tallies either go to this
// workgroup's row of `partials` (no atomics; finalize_reduce adds the rows up) or, without a partials buffer, straight
// to the global words.  Thousands of workgroups doing same-address atomics would serialise at ~12 ns each.
__device__ __forceinline__ void finalize_epilogue(const FinalizeArgs& a, const uint32_t* lstats, unsigned int kept, unsigned int distinct,
                                                  uint32_t* scratch2 /* LDS, 2 words, zeroed */, int row) {
    const int n3 = a.ix.n_files * 3;
#pragma unroll
    for (int off = 32; off; off >>= 1) { kept += (unsigned int)__shfl_xor((int)kept, off); distinct += (unsigned int)__shfl_xor((int)distinct, off); }
    if ((threadIdx.x & 63) == 0) { if (kept) atomicAdd(&scratch2[0], kept); if (distinct) atomicAdd(&scratch2[1], distinct); }
    __syncthreads();
    if (a.partials) {
        uint32_t* out = a.partials + (size_t)row * (n3 + 2);
        for (int g = threadIdx.x; g < n3; g += blockDim.x) out[g] = lstats[g];
        if (threadIdx.x == 0) { out[n3] = scratch2[0]; out[n3 + 1] = scratch2[1]; }
    } else {
        for (int g = threadIdx.x; g < a.ix.n_files; g += blockDim.x) {
            const uint32_t pf = lstats[g * 3], vr = lstats[g * 3 + 1], un = lstats[g * 3 + 2];
            if (pf) atomicAdd(a.stats + (size_t)g * 3 + 0, (unsigned long long)pf);
            if (vr) atomicAdd(a.stats + (size_t)g * 3 + 1, (unsigned long long)vr);
            if (un) atomicAdd(a.stats + (size_t)g * 3 + 2, (unsigned long long)un);
            if (pf | vr) a.present[g] = 1;
        }
        if (threadIdx.x == 0) {
            if (scratch2[0] && a.kept_total) atomicAdd(a.kept_total, (unsigned long long)scratch2[0]);
            if (scratch2[1] && a.distinct_total) atomicAdd(a.distinct_total, (unsigned long long)scratch2[1]);
        }
    }
}


// Reverse complement of a k-mer (first base on top, like every canonical k-mer here).
__device__ __forceinline__ uint64_t revcomp_kmer(uint64_t c, int k) {
    const uint64_t t = ~c;
    const uint64_t r = ((uint64_t)rev2_32((uint32_t)t) << 32) | rev2_32((uint32_t)(t >> 32));
    return r >> (64 - 2 * k);
}


}  // namespace bk
