// bk_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the k-mer -> pileup engine.
//
// K1 scan_count   : packed 2-bit read records -> rolling forward / reverse-complement k-mer -> canonical
//                   (lcb.rs:87-95) -> which distinct table-touching k-mer is it? -> one atomic on that k-mer's
//                   occurrence counter.  Replaces the external KMC3 run of call.rs:1166-1211 for every k-mer
//                   that can touch the index.
// K1b fold        : adds the workgroup histogram slabs (and the per-XCD overflow planes) into the u64 plane.
// K2 finalize     : counters -> KMC thresholds (-ci/-cs/-cx) -> the literal map_kmers vote of
//                   call.rs:1286-1418 (max into depth, +1 into #kmers, per-genome perfect/variant/unique).
//
// Counter naming (bk_device.h): a read k-mer equal to a reference k-mer u owns E[2*pos(u) + rc]; a read k-mer
// at Hamming distance 1 from a reference k-mer, differing at a window position, owns
// V[slot(j, masked)][base][rc] for the LOWEST such position j.  Both are functions of the k-mer alone, so
// every occurrence of a k-mer lands on the same counter and no k-mer owns two; a k-mer that touches no window
// bucket is not counted at all (map_kmers would ignore it: call.rs:1307).  finalize re-derives the k-mer
// from the counter's coordinates and replays map_kmers on it with its exact count.
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

// ------------------------------------------------------------------------------------------------ K1
// Persistent workgroups of 16 waves (one per CU while the LDS histogram is in use); each wave takes tiles of 64
// records, one record per lane.  The k-mer loop is wave-uniform (trip count = longest record of the tile) so
// that ballots and the LDS miss queue always see the whole wave.
//
// Fast path -- is the read k-mer a reference k-mer?  Perfect hash: pilot load (small, L1-resident) + one
// 8-byte key load from an L2-resident table, no probe chain, so no lane waits for another lane's collisions.
// A hit (the overwhelmingly common case) is counted in the workgroup's private LDS histogram: one 32-bit word
// per reference k-mer, low half = read as-is, high half = read as reverse complement.  A half that reaches
// 0x8000 is spilled (by the one lane that saw 0x7fff -> 0x8000) as 0x8000 into the u64 plane, so degenerate
// inputs (millions of identical k-mers) cannot overflow 16 bits.  At the end the histogram is written as one
// coalesced slab per workgroup; fold_slabs adds the slabs into the u64 plane.  This replaces ~one global atomic
// per k-mer occurrence by LDS atomics plus |U| stores per workgroup.
//
// Slow path -- k-mers that are not reference k-mers are compacted (ballot + prefix popcount) into a per-wave
// LDS queue; whenever 64 are pending the wave drains them together: walk the low-half and high-half chains of
// U (pigeonhole: a reference k-mer at Hamming distance 1 agrees with the read k-mer on one half), keep the
// lowest differing position that lies in the window, and name the counter through that position's window
// sub-table.  Without the queue every wave step would pay for its slowest lane.
//
// Reference k-mers beyond the LDS histogram's capacity (large multi-genome indexes) are counted with
// workgroup-scope (non-sc1) atomics in a u32 plane private to the XCD the workgroup runs on (HW_REG_XCC_ID,
// read at run time, so nothing depends on how workgroups are placed); fold adds the planes up afterwards.
constexpr int kScanBlock = 1024;
constexpr int kScanWaves = kScanBlock / 64;
constexpr int kQueueCap = 128;
constexpr size_t kQueueBytes = (size_t)kScanWaves * kQueueCap * (sizeof(unsigned long long) + 1);

struct QueueView { unsigned long long* c; unsigned char* meta; };

__device__ __forceinline__ int chain_best(const uint64_t* __restrict__ tab, uint32_t log2u, uint64_t half, int shift, uint64_t half_mask,
                                          uint64_t c, int k, int wlo, int whi, int best) {
    const uint32_t umask = (1u << log2u) - 1u;
    uint32_t h = hash_key(half, log2u);
    for (;;) {
        const uint64_t u = tab[h];
        if (u == kEmptyKey) break;
        if (((u >> shift) & half_mask) == half) {
            const int j = single_diff_pos(u, c, k);
            if (j >= wlo && j < whi && j < best) best = j;
        }
        h = (h + 1) & umask;
    }
    return best;
}

__device__ __forceinline__ void drain_queue(const QueueView& q, uint32_t n, int lane, const IndexView& ix,
                                            unsigned long long* __restrict__ v_counters) {
    if ((uint32_t)lane < n) {
        const uint64_t c = q.c[lane];
        const uint32_t isrc = q.meta[lane];
        const int k = ix.k;
        const int lo_bits = 2 * ix.lo_bases;
        const uint64_t lo_mask = (1ull << lo_bits) - 1ull;
        const int wlo = ix.wstart, whi = ix.wstart + ix.W;
        int best = 127;
        best = chain_best(ix.kmer_lo, ix.log2u, c & lo_mask, 0, lo_mask, c, k, wlo, whi, best);
        best = chain_best(ix.kmer_hi, ix.log2u, c >> lo_bits, lo_bits, ~0ull >> lo_bits, c, k, wlo, whi, best);
        if (best != 127) {
            const int sh = 2 * (k - 1 - best);
            const size_t S = (size_t)1 << ix.log2s;
            const int s = probe_table(ix.table + (size_t)(best - ix.wstart) * S, ix.log2s, c & ~(3ull << sh));
            if (s >= 0) {   // always true: the candidate reference k-mer owns this bucket
                const uint32_t b = (uint32_t)(c >> sh) & 3u;
                atomicAdd(v_counters + (size_t)s * kCountersPerSlot + b * 2 + isrc, 1ull);
            }
        }
    }
}

// MODE is a measurement aid (BK_SCAN_ABLATE): 0 = product kernel; 1 = exact-match counting replaced by a
// register sink; 2 = lookups and counting replaced by a register sink.  Modes 1/2 produce no counts.
template <int MODE>
__global__ __launch_bounds__(kScanBlock) void scan_count_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* queue_c = reinterpret_cast<unsigned long long*>(smem);
    unsigned char* queue_m = smem + (size_t)kScanWaves * kQueueCap * sizeof(unsigned long long);
    unsigned int* block_kmers = reinterpret_cast<unsigned int*>(smem + kQueueBytes);   // 16 B reserved
    unsigned int* bins = block_kmers + 4;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const QueueView q{queue_c + wave * kQueueCap, queue_m + wave * kQueueCap};
    const IndexView& ix = a.ix;

    for (uint32_t i = threadIdx.x; i < a.n_lds_bins; i += kScanBlock) bins[i] = 0u;
    __syncthreads();

    const int k = ix.k;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;  // k <= 31
    const int rcshift = 2 * (k - 1);
    const uint64_t n_e = e_plane_len(ix.m);
    unsigned int* const e_local = a.e_planes ? a.e_planes + (size_t)xcc_id() * n_e : nullptr;
    unsigned long long* const v_counters = a.counters + n_e;

    uint32_t nkm = 0;  // k-mer occurrences seen by this lane
    uint32_t qn = 0;   // wave-uniform queue fill
    uint64_t sink = 0; // MODE != 0 only

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
                        const uint32_t pilot = ix.pilots[phf_bucket(c, ix.log2nb)];
                        const uint32_t pos = phf_pos(c, pilot, ix.m);
                        if (ix.kmer_pos[pos] == c) {
                            if (MODE != 0) {
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
                            miss = true;
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
                        drain_queue(q, 64, lane, ix, v_counters);
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
    if (qn) drain_queue(q, qn, lane, ix, v_counters);
    if (MODE != 0 && sink == 0x1234567) a.counters[0] = sink;   // keeps the sink alive, never true in practice

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

size_t scan_lds_bytes(uint32_t n_lds_bins) { return kQueueBytes + 16 + (size_t)n_lds_bins * sizeof(unsigned int); }
uint32_t scan_max_lds_bins() { return (uint32_t)((160u * 1024u - 64u - kQueueBytes) / sizeof(unsigned int)); }

uint32_t scan_grid(uint64_t n_records, int n_cus) {
    const uint64_t want = (n_records + kScanBlock - 1) / kScanBlock;
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)n_cus));
}

hipError_t launch_scan_count(const ScanArgs& a, uint32_t grid, hipStream_t stream) {
    if (a.n_records == 0 || a.ix.W <= 0) return hipSuccess;
    const size_t lds = scan_lds_bytes(a.n_lds_bins);
    void (*kern)(ScanArgs) = a.ablate == 1 ? scan_count_kernel<1> : a.ablate == 2 ? scan_count_kernel<2> : scan_count_kernel<0>;
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
// One wave per workgroup.  The wave sweeps 64 counters at a time; every counter that survives the KMC
// thresholds is one distinct k-mer, which the whole wave then maps like call.rs:1286-1418 does: lane t probes
// the k-mer's t-th window bucket and votes once per BucketInfo found there.  Per-genome hit totals live in
// LDS (hits[n_files]); genomes touched by the current k-mer are listed so that only they are classified and
// re-zeroed.  Per-genome statistics are tallied in LDS and flushed once per workgroup.
__global__ __launch_bounds__(64) void finalize_kernel(FinalizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const IndexView& ix = a.ix;
    uint32_t* hits = reinterpret_cast<uint32_t*>(smem);                       // [n_files]
    uint32_t* touched = hits + ix.n_files;                                      // [n_files]
    uint32_t* lstats = touched + ix.n_files;                                    // [n_files][3] block-local tallies
    uint32_t* ntouched = lstats + (size_t)ix.n_files * 3;                       // [1]
    const int lane = threadIdx.x;
    for (int g = lane; g < ix.n_files; g += 64) hits[g] = 0;
    for (int g = lane; g < ix.n_files * 3; g += 64) lstats[g] = 0;
    if (lane == 0) *ntouched = 0;
    __syncthreads();

    const int k = ix.k;
    const size_t S = (size_t)1 << ix.log2s;
    const uint64_t n_e = e_plane_len(ix.m);
    const uint64_t n_counters = n_e + ix.n_slots * kCountersPerSlot;
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
            unsigned long long v = __shfl(n, src);
            v = v > a.cs ? a.cs : v;                              // kmc -cs: reported count saturates
            uint64_t c;
            uint32_t isrc;
            if (cidx < n_e) {                                     // E: a reference k-mer itself
                c = ix.kmer_pos[cidx >> 1];
                isrc = (uint32_t)cidx & 1u;
            } else {                                              // V: slot's masked k-mer + base at the wildcard
                const uint64_t vi = cidx - n_e;
                const uint64_t slot = vi >> 3;
                const uint32_t bb = (uint32_t)(vi >> 1) & 3u;
                isrc = (uint32_t)vi & 1u;
                c = ix.slot_key[slot] | ((uint64_t)bb << (2 * (k - 1 - (ix.wstart + ix.slot_t[slot]))));
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
                    // tallied in LDS and flushed once per workgroup: millions of k-mers voting for the same
                    // genome would otherwise serialise on one global atomic word
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
    const uint64_t n_counters = e_plane_len(a.ix.m) + a.ix.n_slots * kCountersPerSlot;
    const size_t lds = finalize_lds_bytes(a.ix.n_files);
    uint64_t per_cu = (160u * 1024u) / lds;      // resident single-wave workgroups per CU: LDS-limited ...
    if (per_cu > 16) per_cu = 16;                // ... and capped so the end-of-block flush stays small
    if (per_cu < 1) per_cu = 1;
    uint64_t blocks = (n_counters + 63) / 64;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;   // grid-stride beyond one resident wave set
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)blocks), dim3(64), lds, stream, a);
}

}  // namespace bk
