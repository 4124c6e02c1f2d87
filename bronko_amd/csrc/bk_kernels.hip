// bk_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the k-mer -> pileup engine.
//
// K1 scan_count : packed 2-bit read records -> rolling forward / reverse-complement k-mer -> canonical
//                 (lcb.rs:87-95) -> probe the window sub-tables in ascending wildcard position -> one u64
//                 atomicAdd on the occurrence counter of that distinct k-mer.  Replaces the external KMC3
//                 run of call.rs:1166-1211 for every k-mer that can touch the index.
// K2 finalize   : counters -> KMC thresholds (-ci/-cs/-cx) -> the literal map_kmers vote of
//                 call.rs:1286-1418 (max into depth, +1 into #kmers, per-genome perfect/variant/unique).
//
// Why one counter identifies one distinct k-mer: a window bucket (wildcard position j, the other k-1 bases)
// plus the base at j is the whole canonical k-mer, and the read-orientation flag tells which strand-specific
// k-mer it was (k odd => a k-mer never equals its reverse complement).  scan_count always credits the
// *lowest* hitting wildcard position, so every occurrence of a k-mer lands on the same counter and no k-mer
// owns two counters; finalize re-derives the k-mer from the counter's coordinates and replays map_kmers on it.
#include <hip/hip_runtime.h>

#include "bk_device.h"
#include "bk_kernels.h"

namespace bk {

__device__ __forceinline__ int probe_table(const TableSlot* __restrict__ sub, uint32_t log2s, uint64_t key) {
    const uint32_t smask = (1u << log2s) - 1u;
    uint32_t h = hash_key(key, log2s);
    for (;;) {
        const uint4 e = *reinterpret_cast<const uint4*>(sub + h);  // one dwordx4 load per probe
        const uint64_t kk = (uint64_t)e.x | ((uint64_t)e.y << 32);
        if (kk == key) return (int)e.z;
        if (kk == kEmptyKey) return -1;
        h = (h + 1) & smask;
    }
}

// ------------------------------------------------------------------------------------------------ K1
// One thread per record, 64 records per wave.  The k-mer loop is wave-uniform (trip count = longest record of
// the wave) so that ballots and the LDS miss queue always see the whole wave.
//
// Fast path: probe sub-table 0 (wildcard at the first window position).  A read k-mer that equals a
// reference k-mer -- the overwhelmingly common case -- hits here and costs one probe + one atomic.
// Slow path: k-mers that miss sub-table 0 are compacted (ballot + prefix popcount) into a per-wave LDS queue;
// whenever 64 are pending the wave drains them together, each lane walking the remaining W-1 sub-tables of one
// queued k-mer.  This keeps the 16-probe worst case off the common path instead of making every wave step pay
// for its slowest lane.
constexpr int kQueueCap = 128;

__device__ __forceinline__ void drain_queue(const unsigned long long* q, uint32_t n, int lane, const IndexView& ix,
                                            unsigned long long* __restrict__ counters) {
    if ((uint32_t)lane < n) {
        const unsigned long long e = q[lane];
        const uint64_t c = e & ~(1ull << 63);
        const uint32_t isrc = (uint32_t)(e >> 63);
        const size_t S = (size_t)1 << ix.log2s;
        for (int t = 1; t < ix.W; ++t) {
            const int sh = 2 * (ix.k - 1 - (ix.wstart + t));
            const int s = probe_table(ix.table + (size_t)t * S, ix.log2s, c & ~(3ull << sh));
            if (s >= 0) {
                const uint32_t b = (uint32_t)(c >> sh) & 3u;
                atomicAdd(counters + (size_t)s * kCountersPerSlot + b * 2 + isrc, 1ull);
                break;
            }
        }
    }
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void scan_count_kernel(ScanArgs a) {
    __shared__ unsigned long long queue[BLOCK / 64][kQueueCap];
    const int lane = threadIdx.x & 63;
    unsigned long long* q = queue[threadIdx.x >> 6];
    const IndexView& ix = a.ix;

    const uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool live = r < a.n_records;
    const uint32_t len = live ? (uint32_t)a.lens[r] : 0u;
    uint32_t maxlen = len;
#pragma unroll
    for (int off = 32; off; off >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, off));

    const uint32_t* __restrict__ w = a.words + (live ? r : 0) * a.stride_words;
    const int k = ix.k;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;  // k <= 31
    const int rcshift = 2 * (k - 1);
    const int sh0 = 2 * (k - 1 - ix.wstart);
    const uint64_t m0 = ~(3ull << sh0);

    uint64_t fwd = 0, rc = 0;
    uint32_t nkm = 0;  // k-mer occurrences of this record
    uint32_t qn = 0;   // wave-uniform queue fill

    for (uint32_t i0 = 0; i0 < maxlen; i0 += 16) {
        uint32_t x = (i0 < len) ? w[i0 >> 4] : 0u;
        const uint32_t nb = min(16u, maxlen - i0);
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t i = i0 + b;
            const uint32_t base = x & 3u;
            x >>= 2;
            fwd = ((fwd << 2) | base) & kmask;
            rc = (rc >> 2) | ((uint64_t)(3u - base) << rcshift);
            bool miss = false;
            uint64_t c = 0;
            uint32_t isrc = 0;
            if (i < len && i + 1 >= (uint32_t)k) {
                ++nkm;
                isrc = fwd < rc ? 0u : 1u;  // lcb.rs:90-94
                c = isrc ? rc : fwd;
                const int s = probe_table(ix.table, ix.log2s, c & m0);
                if (s >= 0) {
                    const uint32_t bb = (uint32_t)(c >> sh0) & 3u;
                    atomicAdd(a.counters + (size_t)s * kCountersPerSlot + bb * 2 + isrc, 1ull);
                } else {
                    miss = ix.W > 1;
                }
            }
            const unsigned long long mm = __ballot(miss);
            if (mm) {
                const uint32_t pos = qn + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                if (miss) q[pos] = c | ((unsigned long long)isrc << 63);
                qn += (uint32_t)__popcll(mm);
                __builtin_amdgcn_wave_barrier();
                if (qn >= 64) {
                    drain_queue(q, 64, lane, ix, a.counters);
                    const uint32_t rest = qn - 64;
                    const unsigned long long tmp = ((uint32_t)lane < rest) ? q[64 + lane] : 0ull;
                    __builtin_amdgcn_wave_barrier();
                    if ((uint32_t)lane < rest) q[lane] = tmp;
                    __builtin_amdgcn_wave_barrier();
                    qn = rest;
                }
            }
        }
    }
    if (qn) drain_queue(q, qn, lane, ix, a.counters);

    if (a.kmer_total) {
        uint32_t tot = nkm;
#pragma unroll
        for (int off = 32; off; off >>= 1) tot += (uint32_t)__shfl_xor((int)tot, off);
        if (lane == 0 && tot) atomicAdd(a.kmer_total, (unsigned long long)tot);
    }
}

void launch_scan_count(const ScanArgs& a, hipStream_t stream) {
    if (a.n_records == 0 || a.ix.W <= 0) return;
    constexpr int BLOCK = 256;
    const uint64_t blocks = (a.n_records + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(scan_count_kernel<BLOCK>, dim3((unsigned)blocks), dim3(BLOCK), 0, stream, a);
}

// ------------------------------------------------------------------------------------------------ K2
// One wave per workgroup.  The wave sweeps 64 counters at a time; every counter that survives the KMC
// thresholds is one distinct k-mer, which the whole wave then maps like call.rs:1286-1418 does: lane t probes
// the k-mer's t-th window bucket and votes once per BucketInfo found there.  Per-genome hit totals live in
// LDS (hits[n_files]); genomes touched by the current k-mer are listed so that only they are classified and
// re-zeroed.
__global__ __launch_bounds__(64) void finalize_kernel(FinalizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const IndexView& ix = a.ix;
    uint32_t* hits = reinterpret_cast<uint32_t*>(smem);                       // [n_files]
    uint32_t* touched = hits + ix.n_files;                                      // [n_files]
    uint32_t* ntouched = touched + ix.n_files;                                  // [1]
    const int lane = threadIdx.x;
    for (int g = lane; g < ix.n_files; g += 64) hits[g] = 0;
    if (lane == 0) *ntouched = 0;
    __syncthreads();

    const int k = ix.k;
    const size_t S = (size_t)1 << ix.log2s;
    const uint64_t n_counters = ix.n_slots * kCountersPerSlot;
    unsigned long long kept = 0;

    for (uint64_t base = (uint64_t)blockIdx.x * 64; base < n_counters; base += (uint64_t)gridDim.x * 64) {
        const uint64_t idx = base + lane;
        const unsigned long long n = idx < n_counters ? a.counters[idx] : 0ull;
        const bool pass = n >= a.ci && n <= a.cx && n != 0;     // kmc -ci / -cx act on the true count
        unsigned long long todo = __ballot(pass);
        kept += __popcll(todo);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const uint64_t cidx = base + src;
            const uint64_t slot = cidx >> 3;
            const uint32_t bb = (uint32_t)(cidx >> 1) & 3u;
            const uint32_t isrc = (uint32_t)cidx & 1u;
            unsigned long long v = __shfl(n, src);
            v = v > a.cs ? a.cs : v;                              // kmc -cs: reported count saturates
            const int t0 = ix.slot_t[slot];
            const uint64_t c = ix.slot_key[slot] | ((uint64_t)bb << (2 * (k - 1 - (ix.wstart + t0))));

            if (lane < ix.W) {
                const int sh = 2 * (k - 1 - (ix.wstart + lane));
                const int s = probe_table(ix.table + (size_t)lane * S, ix.log2s, c & ~(3ull << sh));
                if (s >= 0) {
                    const uint32_t off = ix.ent_off[s], cnt = ix.ent_len[s];
                    for (uint32_t q = 0; q < cnt; ++q) {
                        const DevEntry e = ix.entries[off + q];
                        // call.rs:1316-1318 per_genome_bucket_hits
                        if (atomicAdd(&hits[e.file], 1u) == 0u) touched[atomicAdd(ntouched, 1u)] = e.file;
                        // call.rs:1327-1384
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
                        unsigned long long* depth = a.pileup + (forward ? 0 : 1) * a.plane + cell;
                        unsigned long long* nk = a.pileup + (forward ? 2 : 3) * a.plane + cell;
                        atomicAdd(nk, 1ull);
                        atomicMax(depth, v);
                    }
                }
            }
            __syncthreads();
            // call.rs:1390-1418: perfect iff hits == number of window buckets; unique iff exactly one perfect
            const uint32_t nt = *ntouched;
            uint32_t n_perfect = 0;
            int my_perfect = -1;
            for (uint32_t q = lane; q < ((nt + 63u) & ~63u); q += 64) {
                bool perfect = false;
                if (q < nt) {
                    const uint32_t g = touched[q];
                    const uint32_t h = hits[g];
                    hits[g] = 0;
                    perfect = h == (uint32_t)ix.W;
                    atomicAdd(a.stats + (size_t)g * 3 + (perfect ? 0 : 1), 1ull);
                    a.present[g] = 1;
                    if (perfect) my_perfect = (int)g;
                }
                n_perfect += (uint32_t)__popcll(__ballot(perfect));
            }
            if (n_perfect == 1 && my_perfect >= 0) atomicAdd(a.stats + (size_t)my_perfect * 3 + 2, 1ull);
            __syncthreads();
            if (lane == 0) *ntouched = 0;
            __syncthreads();
        }
    }
    if (lane == 0 && kept && a.kept_total) atomicAdd(a.kept_total, kept);
}

size_t finalize_lds_bytes(int n_files) { return ((size_t)n_files * 2 + 4) * sizeof(uint32_t); }

void launch_finalize(const FinalizeArgs& a, hipStream_t stream) {
    if (a.ix.n_slots == 0 || a.ix.W <= 0) return;
    const uint64_t n_counters = a.ix.n_slots * kCountersPerSlot;
    uint64_t blocks = (n_counters + 63) / 64;
    if (blocks > 256 * 32) blocks = 256 * 32;  // 32 single-wave workgroups per CU, grid-stride beyond that
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)blocks), dim3(64), finalize_lds_bytes(a.ix.n_files), stream, a);
}

}  // namespace bk
