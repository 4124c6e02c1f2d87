// bk_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the k-mer -> pileup engine.
//
// K1  scan_count       : packed 2-bit read records -> rolling forward / reverse-complement k-mer -> canonical
//                        (lcb.rs:87-95) -> which distinct index-touching k-mer is it? -> +1 on that k-mer's
//                        occurrence counter.  Replaces the external KMC3 run of call.rs:1166-1211 for every
//                        k-mer that can touch the index.
// K1b fold             : adds the workgroup histogram slabs (and the per-XCD overflow planes) into the u64 plane.
// K2a finalize_variant : V counters -> KMC thresholds -> map_kmers vote, one thread per non-reference k-mer.
// K2b finalize_general : E counters (+ the k-mers K2a defers) -> thresholds -> map_kmers vote, one wave per k-mer.
//                        K2a/K2b together are call.rs:1286-1418 applied to KMC's kept k-mers (-ci/-cs/-cx).
//
// Counter naming (bk_device.h): a read k-mer equal to a reference k-mer u owns E[2*pos(u) + rc]; a read k-mer
// at Hamming distance 1 from reference k-mers, differing at a window position, owns the V counter of the
// smallest (position, pos(u)).  Both are functions of the k-mer alone, so every occurrence of a k-mer lands on
// the same counter and no k-mer owns two; a k-mer that touches no window bucket is not counted at all
// (map_kmers would ignore it: call.rs:1307).  finalize re-derives the k-mer from the counter's coordinates and
// replays map_kmers on it with its exact count.
#include <hip/hip_runtime.h>

#include <algorithm>

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

// If a and b differ in exactly one base, return its position counted from the left (0..k-1), else -1.
__device__ __forceinline__ int single_diff_pos(uint64_t a, uint64_t b, int k) {
    const uint64_t x = a ^ b;
    const uint64_t y = (x | (x >> 1)) & 0x5555555555555555ull;
    if (y == 0 || (y & (y - 1)) != 0) return -1;
    return k - 1 - (__builtin_ctzll(y) >> 1);
}

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & (kXcdPlanes - 1);
}

__device__ __forceinline__ HalfDir half_lookup(const HalfView& hv, uint64_t half) {
    const uint32_t pilot = hv.pilots[phf_bucket(half, hv.log2nb)];
    const uint4 e = *reinterpret_cast<const uint4*>(hv.dir + phf_pos(half, pilot, hv.m));
    HalfDir d;
    d.key = e.x; d.off = e.y; d.cnt = (e.x == (uint32_t)half) ? e.z : 0u; d.pad = 0u;
    return d;
}

// Reference k-mers at Hamming distance exactly 1 from c whose differing position lies in the window.
// Calls f(j, p) for each (pigeonhole: such a k-mer shares c's low half or c's high half).
template <typename F>
__device__ __forceinline__ void for_each_neighbour(const IndexView& ix, uint64_t c, F&& f) {
    const int k = ix.k;
    const int lo_bits = 2 * ix.lo_bases;
    const uint64_t lo = c & ((1ull << lo_bits) - 1ull), hi = c >> lo_bits;
    const int wlo = ix.wstart, whi = ix.wstart + ix.W;
    const HalfDir dl = half_lookup(ix.lo, lo);   // the two lookups are independent: their loads overlap
    const HalfDir dh = half_lookup(ix.hi, hi);
    for (uint32_t i = 0; i < dl.cnt; ++i) {
        const uint4 e = *reinterpret_cast<const uint4*>(ix.lo.cand + dl.off + i);
        const int j = single_diff_pos((uint64_t)e.x | ((uint64_t)e.y << 32), c, k);
        if (j >= wlo && j < whi) f(j, e.z);
    }
    for (uint32_t i = 0; i < dh.cnt; ++i) {
        const uint4 e = *reinterpret_cast<const uint4*>(ix.hi.cand + dh.off + i);
        const int j = single_diff_pos((uint64_t)e.x | ((uint64_t)e.y << 32), c, k);
        if (j >= wlo && j < whi) f(j, e.z);
    }
}

// ------------------------------------------------------------------------------------------------ K1
// Persistent workgroups of 16 waves (one per CU: the LDS histogram takes most of the CU's 160 KB); each wave
// takes tiles of 64 records, one record per lane.  The k-mer loop is wave-uniform (trip count = longest record
// of the tile) so that ballots and the LDS miss queue always see the whole wave.
//
// Fast path -- is the read k-mer a reference k-mer?  Perfect hash: pilot load (LDS when it fits, else L1/L2) +
// one 8-byte key load from an L2-resident table, no probe chain, so no lane waits for another lane's
// collisions.  A hit (the overwhelmingly common case) is counted in the workgroup's private LDS histogram:
// one 32-bit word per reference k-mer, low half = read as-is, high half = read as reverse complement.  A half
// that reaches 0x8000 is spilled (by the one lane that saw 0x7fff -> 0x8000) as 0x8000 into the u64 plane, so
// degenerate inputs (millions of identical k-mers) cannot overflow 16 bits.  At the end the histogram is
// written as one coalesced slab per workgroup; fold adds the slabs into the u64 plane.  This replaces one
// global atomic per k-mer occurrence by an LDS atomic plus |U| stores per workgroup.
//
// Slow path -- k-mers that are not reference k-mers are compacted (ballot + prefix popcount) into a per-wave
// LDS queue; whenever 64 are pending the wave drains them together, each lane looking up the neighbours of
// one queued k-mer and adding to the V counter of the smallest (position, reference k-mer).  Without the queue
// every wave step would pay for its slowest lane.
//
// Reference k-mers beyond the LDS histogram's capacity (large multi-genome indexes) are counted with
// workgroup-scope (non-sc1) atomics in a u32 plane private to the XCD the workgroup runs on (HW_REG_XCC_ID,
// read at run time, so nothing depends on how workgroups are placed); fold adds the planes up afterwards.
constexpr int kScanBlock = 1024;
constexpr int kScanWaves = kScanBlock / 64;
constexpr int kQueueCap = 128;
constexpr size_t kQueueBytes = (size_t)kScanWaves * kQueueCap * (sizeof(unsigned long long) + 1);
constexpr size_t kScanLdsFixed = kQueueBytes + 16;

struct QueueView { unsigned long long* c; unsigned char* meta; };

template <bool COUNT = true>
__device__ __forceinline__ void drain_queue(const QueueView& q, uint32_t n, int lane, const IndexView& ix,
                                            unsigned long long* __restrict__ v_counters) {
    if ((uint32_t)lane < n) {
        const uint64_t c = q.c[lane];
        const uint32_t isrc = q.meta[lane];
        uint64_t best = ~0ull;   // (j << 32) | p, smallest wins
        for_each_neighbour(ix, c, [&](int j, uint32_t p) {
            const uint64_t key = ((uint64_t)j << 32) | p;
            if (key < best) best = key;
        });
        if (best != ~0ull) {
            const int j = (int)(best >> 32);
            const uint32_t p = (uint32_t)best;
            const uint32_t b = (uint32_t)(c >> (2 * (ix.k - 1 - j))) & 3u;
            unsigned long long* ctr = v_counters + (((uint64_t)p * ix.W + (uint32_t)(j - ix.wstart)) * 4 + b) * 2 + isrc;
            if (COUNT) atomicAdd(ctr, 1ull);
            else if (best == 0x123456789ull) *ctr = 1;   // measurement aid: keep the lookup alive without the atomic
        }
    }
}

// MODE is a measurement aid (BK_SCAN_ABLATE): 0 = product kernel; 1 = exact-match counting replaced by a
// register sink; 2 = lookups and counting replaced by a register sink; 3 = product fast path, slow path
// dropped; 4 = slow path without its final atomic.  Modes 1-4 produce incomplete counts.
template <int MODE>
__global__ __launch_bounds__(kScanBlock) void scan_count_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* queue_c = reinterpret_cast<unsigned long long*>(smem);
    unsigned char* queue_m = smem + (size_t)kScanWaves * kQueueCap * sizeof(unsigned long long);
    unsigned int* block_kmers = reinterpret_cast<unsigned int*>(smem + kQueueBytes);   // 16 B reserved
    unsigned int* bins = block_kmers + 4;
    unsigned short* lds_pilots = reinterpret_cast<unsigned short*>(bins + a.n_lds_bins);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const QueueView q{queue_c + wave * kQueueCap, queue_m + wave * kQueueCap};
    const IndexView& ix = a.ix;

    for (uint32_t i = threadIdx.x; i < a.n_lds_bins; i += kScanBlock) bins[i] = 0u;
    if (a.pilots_in_lds)
        for (uint32_t i = threadIdx.x; i < (1u << ix.log2nb); i += kScanBlock) lds_pilots[i] = ix.pilots[i];
    __syncthreads();

    const int k = ix.k;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;  // k <= 31
    const int rcshift = 2 * (k - 1);
    const uint64_t n_e = e_plane_len(ix.m);
    unsigned int* const e_local = a.e_planes ? a.e_planes + (size_t)xcc_id() * n_e : nullptr;
    unsigned long long* const v_counters = a.counters + n_e;

    uint32_t nkm = 0;  // k-mer occurrences seen by this lane
    uint32_t qn = 0;   // wave-uniform queue fill
    uint64_t sink = 0; // MODE 1/2 only

    const uint64_t n_tiles = (a.n_records + 63) / 64;
    for (uint64_t tile = (uint64_t)blockIdx.x * kScanWaves + wave; tile < n_tiles; tile += (uint64_t)gridDim.x * kScanWaves) {
        const uint64_t r = tile * 64 + lane;
        const bool live = r < a.n_records;
        const uint32_t len = live ? (uint32_t)a.lens[r] : 0u;
        uint32_t maxlen = len;
#pragma unroll
        for (int off = 32; off; off >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, off));
        const uint32_t* __restrict__ w = a.words + (live ? r : 0) * a.stride_words;

        uint64_t fwd = 0, rc = 0;
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
                    if (MODE == 2) {
                        sink += c;
                    } else {
                        const uint32_t bkt = phf_bucket(c, ix.log2nb);
                        const uint32_t pilot = a.pilots_in_lds ? lds_pilots[bkt] : ix.pilots[bkt];
                        const uint32_t pos = phf_pos(c, pilot, ix.m);
                        if (ix.kmer_pos[pos] == c) {
                            if (MODE == 1) {
                                sink += pos;
                            } else if (pos < a.n_lds_bins) {
                                const unsigned int old = atomicAdd(&bins[pos], isrc ? 0x10000u : 1u);
                                if (((isrc ? old >> 16 : old) & 0xffffu) == 0x7fffu) {   // this add made the half 0x8000: spill it
                                    atomicSub(&bins[pos], isrc ? 0x80000000u : 0x8000u);
                                    atomicAdd(a.counters + 2 * (size_t)pos + isrc, 0x8000ull);
                                }
                            } else if (e_local) {
                                __hip_atomic_fetch_add(e_local + 2 * (size_t)pos + isrc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            } else {
                                atomicAdd(a.counters + 2 * (size_t)pos + isrc, 1ull);
                            }
                        } else {
                            miss = MODE != 3;
                        }
                    }
                }
                const unsigned long long mm = __ballot(miss);
                if (mm) {
                    const uint32_t pos = qn + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                    if (miss) { q.c[pos] = c; q.meta[pos] = (unsigned char)isrc; }
                    qn += (uint32_t)__popcll(mm);
                    __builtin_amdgcn_wave_barrier();
                    if (qn >= 64) {
                        drain_queue<MODE != 4>(q, 64, lane, ix, v_counters);
                        const uint32_t rest = qn - 64;
                        const unsigned long long tc = ((uint32_t)lane < rest) ? q.c[64 + lane] : 0ull;
                        const unsigned char tm = ((uint32_t)lane < rest) ? q.meta[64 + lane] : (unsigned char)0;
                        __builtin_amdgcn_wave_barrier();
                        if ((uint32_t)lane < rest) { q.c[lane] = tc; q.meta[lane] = tm; }
                        __builtin_amdgcn_wave_barrier();
                        qn = rest;
                    }
                }
            }
        }
    }
    if (qn) drain_queue<MODE != 4>(q, qn, lane, ix, v_counters);
    if ((MODE == 1 || MODE == 2) && sink == 0x1234567) a.counters[0] = sink;   // keeps the sink alive, never true in practice

    // histogram -> this workgroup's slab (coalesced); k-mer tally -> one atomic per workgroup
    if (threadIdx.x == 0) *block_kmers = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < a.n_lds_bins; i += kScanBlock) a.slabs[(size_t)blockIdx.x * a.n_lds_bins + i] = bins[i];
    uint32_t tot = nkm;
#pragma unroll
    for (int off = 32; off; off >>= 1) tot += (uint32_t)__shfl_xor((int)tot, off);
    if (lane == 0 && tot) atomicAdd(block_kmers, tot);
    __syncthreads();
    if (threadIdx.x == 0 && *block_kmers && a.kmer_total) atomicAdd(a.kmer_total, (unsigned long long)*block_kmers);
}

size_t scan_lds_budget() { return 160u * 1024u - 64u - kScanLdsFixed; }
size_t scan_lds_bytes(uint32_t n_lds_bins, bool pilots_in_lds, uint32_t log2nb) {
    return kScanLdsFixed + (size_t)n_lds_bins * sizeof(unsigned int) + (pilots_in_lds ? ((size_t)2 << log2nb) : 0);
}

uint32_t scan_grid(uint64_t n_records, int n_cus) {
    const uint64_t want = (n_records + kScanBlock - 1) / kScanBlock;
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)n_cus));
}

hipError_t launch_scan_count(const ScanArgs& a, uint32_t grid, hipStream_t stream) {
    if (a.n_records == 0 || a.ix.W <= 0) return hipSuccess;
    const size_t lds = scan_lds_bytes(a.n_lds_bins, a.pilots_in_lds != 0, a.ix.log2nb);
    void (*kern)(ScanArgs) = a.ablate == 1 ? scan_count_kernel<1> : a.ablate == 2 ? scan_count_kernel<2>
                           : a.ablate == 3 ? scan_count_kernel<3> : a.ablate == 4 ? scan_count_kernel<4> : scan_count_kernel<0>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kScanBlock), lds, stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K1b
// E[2i + h] += sum over workgroup slabs of half h of slab[b][i]   (i < n_lds_bins), and
// E[i]      += sum over the 8 XCD planes of e_planes[x][i]        (planes re-zeroed for the next batch).
__global__ __launch_bounds__(256) void fold_kernel(FoldArgs f) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * 256;
    for (uint64_t i = tid; i < f.n_lds_bins; i += nthreads) {
        unsigned long long s0 = 0, s1 = 0;
        for (uint32_t b = 0; b < f.n_slabs; ++b) {
            const unsigned int v = f.slabs[(size_t)b * f.n_lds_bins + i];
            s0 += v & 0xffffu;
            s1 += v >> 16;
        }
        if (s0) f.counters[2 * i] += s0;
        if (s1) f.counters[2 * i + 1] += s1;
    }
    if (f.e_planes) {
        for (uint64_t i = 2ull * f.n_lds_bins + tid; i < f.n_e; i += nthreads) {
            unsigned long long s = 0;
#pragma unroll
            for (int x = 0; x < kXcdPlanes; ++x) {
                const unsigned int v = f.e_planes[(size_t)x * f.n_e + i];
                if (v) { s += v; f.e_planes[(size_t)x * f.n_e + i] = 0u; }
            }
            if (s) f.counters[i] += s;
        }
    }
}

void launch_fold(const FoldArgs& f, hipStream_t stream) {
    const uint64_t work = std::max<uint64_t>(f.n_lds_bins, f.e_planes ? f.n_e : 0);
    if (work == 0) return;
    uint64_t blocks = (work + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fold_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, f);
}

// ------------------------------------------------------------------------------------------------ K2
// The vote of call.rs:1327-1384 for one BucketInfo.
__device__ __forceinline__ void vote(const FinalizeArgs& a, const DevEntry& e, uint64_t c, uint32_t isrc, int k, unsigned long long v) {
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

// K2a: one thread per V counter.  A kept non-reference k-mer almost always touches exactly one window bucket
// (the one its name says); then the whole of map_kmers for it is: vote once per BucketInfo of that bucket,
// and per genome file "variant" (or "perfect" if the file has exactly W entries there, which needs W == 1 or
// repeats).  K-mers that touch several buckets need cross-bucket per-genome totals; they are appended (by
// counter index) to `deferred` and mapped by K2b.
__global__ __launch_bounds__(256) void finalize_variant_kernel(FinalizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const IndexView& ix = a.ix;
    uint32_t* lstats = reinterpret_cast<uint32_t*>(smem);   // [n_files][3] block-local tallies
    for (int g = threadIdx.x; g < ix.n_files * 3; g += 256) lstats[g] = 0;
    __syncthreads();

    const int k = ix.k;
    const uint64_t n_e = e_plane_len(ix.m);
    const uint64_t n_v = v_plane_len(ix.m, ix.W);
    const unsigned long long* __restrict__ vc = a.counters + n_e;
    const size_t S = (size_t)1 << ix.log2s;
    unsigned int kept = 0;

    for (uint64_t vi = (uint64_t)blockIdx.x * 256 + threadIdx.x; vi < n_v; vi += (uint64_t)gridDim.x * 256) {
        const unsigned long long n = vc[vi];
        if (n == 0 || n < a.ci || n > a.cx) continue;           // kmc -ci / -cx act on the true count
        ++kept;
        const unsigned long long v = n > a.cs ? a.cs : n;       // kmc -cs: reported count saturates
        const uint32_t isrc = (uint32_t)vi & 1u;
        const uint32_t bb = (uint32_t)(vi >> 1) & 3u;
        const uint64_t pt = vi >> 3;
        const uint32_t t = (uint32_t)(pt % (uint32_t)ix.W);
        const uint32_t p = (uint32_t)(pt / (uint32_t)ix.W);
        const int j = ix.wstart + (int)t;
        const int sh = 2 * (k - 1 - j);
        const uint64_t c = (ix.kmer_pos[p] & ~(3ull << sh)) | ((uint64_t)bb << sh);

        uint32_t jmask = 0;   // window positions at which c has a neighbouring reference k-mer
        for_each_neighbour(ix, c, [&](int jj, uint32_t) { jmask |= 1u << (jj - ix.wstart); });
        if (jmask != (1u << t)) {   // several buckets (or, defensively, an unexpected set): general path
            const unsigned int at = atomicAdd(a.n_deferred, 1u);
            a.deferred[at] = (uint32_t)vi;
            continue;
        }
        const int s = probe_table(ix.table + (size_t)t * S, ix.log2s, c & ~(3ull << sh));
        if (s < 0) continue;   // cannot happen: the neighbour owns this bucket
        const uint32_t off = ix.ent_off[s], cnt = ix.ent_len[s];
        // entries of one bucket are grouped by file (index build appends file by file): run lengths = hits per file
        uint32_t n_perfect = 0, perfect_file = 0;
        for (uint32_t q = 0; q < cnt;) {
            const uint32_t file = ix.entries[off + q].file;
            uint32_t run = 0;
            while (q < cnt && ix.entries[off + q].file == file) { vote(a, ix.entries[off + q], c, isrc, k, v); ++run; ++q; }
            if (run == (uint32_t)ix.W) { atomicAdd(&lstats[file * 3 + 0], 1u); ++n_perfect; perfect_file = file; }
            else atomicAdd(&lstats[file * 3 + 1], 1u);
        }
        if (n_perfect == 1) atomicAdd(&lstats[perfect_file * 3 + 2], 1u);
    }
    __syncthreads();
    for (int g = threadIdx.x; g < ix.n_files; g += 256) {
        const uint32_t pf = lstats[g * 3], vr = lstats[g * 3 + 1], un = lstats[g * 3 + 2];
        if (pf) atomicAdd(a.stats + (size_t)g * 3 + 0, (unsigned long long)pf);
        if (vr) atomicAdd(a.stats + (size_t)g * 3 + 1, (unsigned long long)vr);
        if (un) atomicAdd(a.stats + (size_t)g * 3 + 2, (unsigned long long)un);
        if (pf | vr) a.present[g] = 1;
    }
    // kept tally: wave reduce, one atomic per wave
    unsigned int tot = kept;
#pragma unroll
    for (int off = 32; off; off >>= 1) tot += (unsigned int)__shfl_xor((int)tot, off);
    if ((threadIdx.x & 63) == 0 && tot && a.kept_total) atomicAdd(a.kept_total, (unsigned long long)tot);
}

// K2b: one wave per workgroup and per k-mer.  Items are the E counters (reference k-mers: all W buckets are
// non-empty) followed by the deferred V counters.  Lane t probes the k-mer's t-th window bucket and votes once
// per BucketInfo found there.  Per-genome hit totals live in LDS (hits[n_files]); genomes touched by the
// current k-mer are listed so that only they are classified and re-zeroed.  Per-genome statistics are tallied
// in LDS and flushed once per workgroup (millions of k-mers voting for the same genome would otherwise
// serialise on one global atomic word).
__global__ __launch_bounds__(64) void finalize_general_kernel(FinalizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const IndexView& ix = a.ix;
    uint32_t* hits = reinterpret_cast<uint32_t*>(smem);                       // [n_files]
    uint32_t* touched = hits + ix.n_files;                                      // [n_files]
    uint32_t* lstats = touched + ix.n_files;                                    // [n_files][3]
    uint32_t* ntouched = lstats + (size_t)ix.n_files * 3;                       // [1]
    const int lane = threadIdx.x;
    for (int g = lane; g < ix.n_files; g += 64) hits[g] = 0;
    for (int g = lane; g < ix.n_files * 3; g += 64) lstats[g] = 0;
    if (lane == 0) *ntouched = 0;
    __syncthreads();

    const int k = ix.k;
    const size_t S = (size_t)1 << ix.log2s;
    const uint64_t n_e = e_plane_len(ix.m);
    const uint64_t n_def = *a.n_deferred;
    const uint64_t n_items = n_e + n_def;
    unsigned long long kept = 0;

    for (uint64_t base = (uint64_t)blockIdx.x * 64; base < n_items; base += (uint64_t)gridDim.x * 64) {
        const uint64_t item = base + lane;
        uint64_t cidx = ~0ull;                                   // index into the counter plane
        if (item < n_e) cidx = item;
        else if (item < n_items) cidx = n_e + a.deferred[item - n_e];
        const unsigned long long n = cidx != ~0ull ? a.counters[cidx] : 0ull;
        const bool pass = n >= a.ci && n <= a.cx && n != 0;     // kmc -ci / -cx act on the true count
        unsigned long long todo = __ballot(pass);
        kept += __popcll(__ballot(pass && item < n_e));          // deferred items were already tallied by K2a
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const uint64_t ci = __shfl(cidx, src);
            unsigned long long v = __shfl(n, src);
            v = v > a.cs ? a.cs : v;                              // kmc -cs: reported count saturates
            uint64_t c;
            uint32_t isrc;
            if (ci < n_e) {                                       // E: a reference k-mer itself
                c = ix.kmer_pos[ci >> 1];
                isrc = (uint32_t)ci & 1u;
            } else {                                              // V: reference k-mer p with base bb at window position t
                const uint64_t vi = ci - n_e;
                isrc = (uint32_t)vi & 1u;
                const uint32_t bb = (uint32_t)(vi >> 1) & 3u;
                const uint64_t pt = vi >> 3;
                const int sh = 2 * (k - 1 - (ix.wstart + (int)(pt % (uint32_t)ix.W)));
                c = (ix.kmer_pos[pt / (uint32_t)ix.W] & ~(3ull << sh)) | ((uint64_t)bb << sh);
            }

            if (lane < ix.W) {
                const int sh = 2 * (k - 1 - (ix.wstart + lane));
                const int s = probe_table(ix.table + (size_t)lane * S, ix.log2s, c & ~(3ull << sh));
                if (s >= 0) {
                    const uint32_t off = ix.ent_off[s], cnt = ix.ent_len[s];
                    for (uint32_t q = 0; q < cnt; ++q) {
                        const DevEntry e = ix.entries[off + q];
                        // call.rs:1316-1318 per_genome_bucket_hits
                        if (atomicAdd(&hits[e.file], 1u) == 0u) touched[atomicAdd(ntouched, 1u)] = e.file;
                        vote(a, e, c, isrc, k, v);
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
                    lstats[g * 3 + (perfect ? 0 : 1)] += 1;   // g is distinct per lane within one k-mer
                    if (perfect) my_perfect = (int)g;
                }
                n_perfect += (uint32_t)__popcll(__ballot(perfect));
            }
            if (n_perfect == 1 && my_perfect >= 0) lstats[my_perfect * 3 + 2] += 1;
            __syncthreads();
            if (lane == 0) *ntouched = 0;
            __syncthreads();
        }
    }
    __syncthreads();
    for (int g = lane; g < ix.n_files; g += 64) {
        const uint32_t p = lstats[g * 3], v = lstats[g * 3 + 1], u = lstats[g * 3 + 2];
        if (p) atomicAdd(a.stats + (size_t)g * 3 + 0, (unsigned long long)p);
        if (v) atomicAdd(a.stats + (size_t)g * 3 + 1, (unsigned long long)v);
        if (u) atomicAdd(a.stats + (size_t)g * 3 + 2, (unsigned long long)u);
        if (p | v) a.present[g] = 1;
    }
    if (lane == 0 && kept && a.kept_total) atomicAdd(a.kept_total, kept);
}

size_t finalize_lds_bytes(int n_files) { return ((size_t)n_files * 5 + 4) * sizeof(uint32_t); }

void launch_finalize(const FinalizeArgs& a, hipStream_t stream) {
    if (a.ix.W <= 0) return;
    // K2a
    {
        const uint64_t n_v = v_plane_len(a.ix.m, a.ix.W);
        uint64_t blocks = (n_v + 255) / 256;
        if (blocks > 256 * 8) blocks = 256 * 8;
        if (blocks < 1) blocks = 1;
        const size_t lds = std::max<size_t>((size_t)a.ix.n_files * 3 * sizeof(uint32_t), 16);
        hipLaunchKernelGGL(finalize_variant_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, a);
    }
    // K2b
    {
        const size_t lds = finalize_lds_bytes(a.ix.n_files);
        uint64_t per_cu = (160u * 1024u) / lds;      // resident single-wave workgroups per CU: LDS-limited ...
        if (per_cu > 16) per_cu = 16;                // ... and capped so the end-of-block flush stays small
        if (per_cu < 1) per_cu = 1;
        hipLaunchKernelGGL(finalize_general_kernel, dim3((unsigned)(256 * per_cu)), dim3(64), lds, stream, a);
    }
}

}  // namespace bk
