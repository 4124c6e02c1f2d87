// bk_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the k-mer -> pileup engine.
//
// K1  scan_count       : packed 2-bit read records -> for every k-mer, canonical form (lcb.rs:87-95) -> which distinct
//                        index-touching k-mer is it? -> +1 on that k-mer's occurrence counter.  Replaces the external
//                        KMC3 run of call.rs:1166-1211 for every k-mer that can touch the index.  Two levels: a
//                        word-parallel comparison of the read with the reference along its diagonal proves most
//                        k-mers exact; the rest go through the rolling k-mer / neighbour-search machinery.
// K1b fold             : adds the workgroup histogram slabs (and the per-XCD overflow planes) into the u64 plane.
// K2a finalize_variant : V counters -> KMC thresholds -> map_kmers vote, one thread per non-reference k-mer.
// K2e finalize_exact   : E counters -> thresholds -> map_kmers vote, one thread per (reference k-mer, bucket).
// K2b finalize_general : the k-mers K2a defers -> map_kmers vote, one wave per k-mer.
//                        K2a/K2e/K2b together are call.rs:1286-1418 applied to KMC's kept k-mers (-ci/-cs/-cx).
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
        if (j >= wlo && j < whi && ((e.w >> (j - wlo)) & 1u)) f(j, e.z);
    }
    for (uint32_t i = 0; i < dh.cnt; ++i) {
        const uint4 e = *reinterpret_cast<const uint4*>(ix.hi.cand + dh.off + i);
        const int j = single_diff_pos((uint64_t)e.x | ((uint64_t)e.y << 32), c, k);
        if (j >= wlo && j < whi && ((e.w >> (j - wlo)) & 1u)) f(j, e.z);
    }
}

// ------------------------------------------------------------------------------------------------ K0
// ASCII reads -> fixed-stride 2-bit records, one thread per read (the host-side twin is bk_pack_reads): split at
// every non-ACGT/acgt symbol (KMC contract), drop runs shorter than k, cut runs longer than the stride into chunks
// overlapping by k-1 bases.  Records are appended through one device counter (their order is irrelevant).
__device__ __forceinline__ int acgt_code(unsigned char c) {
    switch (c | 0x20) {
        case 'a': return 0;
        case 'c': return 1;
        case 'g': return 2;
        case 't': return 3;
        default: return -1;
    }
}

__global__ __launch_bounds__(256) void pack_reads_kernel(PackArgs a) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n_reads) return;
    const uint8_t* s = a.bases + a.offsets[r];
    const uint64_t len = a.offsets[r + 1] - a.offsets[r];
    const uint64_t maxb = min((uint64_t)a.stride_words * 16, (uint64_t)65535);
    uint64_t start = 0;
    for (uint64_t i = 0; i <= len; ++i) {
        if (i < len && acgt_code(s[i]) >= 0) continue;
        const uint64_t run = i - start;            // maximal ACGT run [start, i)
        if (run >= (uint64_t)a.k) {
            uint64_t pos = 0;
            for (;;) {
                const uint64_t take = min(maxb, run - pos);
                const unsigned long long rec = atomicAdd(a.n_records, 1ull);
                if (rec < a.cap) {
                    uint32_t* w = a.words + rec * a.stride_words;
                    uint32_t acc = 0;
                    for (uint64_t j = 0; j < take; ++j) {
                        acc |= (uint32_t)acgt_code(s[start + pos + j]) << (2 * (j & 15));
                        if ((j & 15) == 15) { w[j >> 4] = acc; acc = 0; }
                    }
                    if (take & 15) w[take >> 4] = acc;
                    for (uint64_t j = (take + 15) >> 4; j < a.stride_words; ++j) w[j] = 0;
                    a.lens[rec] = (uint16_t)take;
                }
                if (pos + take >= run) break;
                pos += take - (uint64_t)(a.k - 1);
            }
        }
        start = i + 1;
    }
}

__global__ void add_u64_kernel(unsigned long long* dst, const unsigned long long* src) { *dst += *src; }
void launch_add_u64(unsigned long long* dst, const unsigned long long* src, hipStream_t stream) {
    hipLaunchKernelGGL(add_u64_kernel, dim3(1), dim3(1), 0, stream, dst, src);
}

void launch_pack_reads(const PackArgs& a, hipStream_t stream) {
    if (a.n_reads == 0) return;
    hipLaunchKernelGGL(pack_reads_kernel, dim3((unsigned)((a.n_reads + 255) / 256)), dim3(256), 0, stream, a);
}

// full_kmer_stats: +1 on a k-mer that does not touch the index, in an open-addressing table keyed by
// (canonical k-mer, read orientation, mate file) -- i.e. by the strand-specific k-mer KMC -b counts.
struct KmerTable {
    unsigned long long* keys;
    unsigned int* cnt;
    uint32_t log2n;
    unsigned long long* overflow;
    uint32_t mate;
};

__device__ __forceinline__ void ktab_insert(const KmerTable& t, uint64_t c, uint32_t isrc) {
    const unsigned long long key = c | ((unsigned long long)isrc << 63) | ((unsigned long long)t.mate << 62);
    const uint32_t mask = (1u << t.log2n) - 1u;   // log2n <= 32 handled by the host (<= 34 uses 64-bit below)
    uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> (64 - t.log2n);
    for (uint32_t probes = 0; probes < 4096; ++probes) {
        const unsigned long long old = atomicCAS(t.keys + h, ~0ull, key);
        if (old == ~0ull || old == key) {
            if (t.cnt[h] < 0xfffffff0u) atomicAdd(t.cnt + h, 1u);   // saturates far above any -cx
            return;
        }
        h = (h + 1) & (uint64_t)mask;
    }
    *t.overflow = 1ull;   // table (nearly) full: statistics are reported as unavailable
}

__global__ __launch_bounds__(256) void ktab_stats_kernel(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ cnt,
                                                         uint64_t n, unsigned long long ci, unsigned long long cx, unsigned long long* out) {
    unsigned int d0 = 0, d1 = 0, k0 = 0, k1 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const unsigned long long key = keys[i];
        if (key == ~0ull) continue;
        const unsigned int c = cnt[i];
        const bool kept = c >= ci && c <= cx;
        if ((key >> 62) & 1ull) { ++d1; k1 += kept; } else { ++d0; k0 += kept; }
    }
    auto wave_sum = [](unsigned int v) {
#pragma unroll
        for (int off = 32; off; off >>= 1) v += (unsigned int)__shfl_xor((int)v, off);
        return v;
    };
    d0 = wave_sum(d0); d1 = wave_sum(d1); k0 = wave_sum(k0); k1 = wave_sum(k1);
    if ((threadIdx.x & 63) == 0) {
        if (d0) atomicAdd(out + 0, (unsigned long long)d0);
        if (k0) atomicAdd(out + 1, (unsigned long long)k0);
        if (d1) atomicAdd(out + 2, (unsigned long long)d1);
        if (k1) atomicAdd(out + 3, (unsigned long long)k1);
    }
}

void launch_ktab_stats(const unsigned long long* keys, const unsigned int* cnt, uint32_t log2n, unsigned long long ci,
                       unsigned long long cx, unsigned long long* out, hipStream_t stream) {
    const uint64_t n = 1ull << log2n;
    uint64_t blocks = std::min<uint64_t>((n + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(ktab_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, keys, cnt, n, ci, cx, out);
}

// ------------------------------------------------------------------------------------------------ K1
// Persistent workgroups of 16 waves (one per CU: the LDS histogram takes most of the CU's 160 KB); each wave
// takes tiles of 64 records, one record per lane, and works in two levels.
//
// Level 1 -- word-parallel verification along a diagonal.  A read follows the reference, so a few seed k-mers
// of the read (evenly spaced, looked up in the perfect hash of U) give its diagonal: read k-mer s <-> reference
// cell a + s (same strand) or a - s (opposite strand).  Then, 16 bases at a time, the read word is XORed with the
// reference word aligned to it (packed reference in LDS, one funnel shift; reversed and complemented for the
// opposite strand) and folded to one mismatch flag per base.  The per-base loop only shifts that flag into a
// k-bit window `dwin`:
//   dwin == 0 and a reference k-mer starts at the cell (2-bit per-cell code, LDS)   the read k-mer IS that
//        reference k-mer: one non-returning LDS add on the cell's bin (low half = read as canonical, high half =
//        as reverse complement; the code says which).  Bins are per CELL, not per k-mer id -- the fast path needs
//        no ids; fold adds bin c into E[id_at[c]].
//   anything else                                                                   the k-mer belongs to a "range"
//        of consecutive non-exact k-mers of this read; when the range ends, (lane, first k-mer, length) goes to
//        the wave's range queue in LDS.
// That is ~a dozen VALU instructions per base.  Reads without a usable diagonal (no seed hit, diagonal leaving
// the reference, cells beyond the LDS bins) become one big range.
//
// Level 2 -- whenever 64 ranges are pending (and at the end), they are processed one per lane with the exact
// per-k-mer logic: rolling canonical k-mer + 2-bit mismatch mask against the diagonal,
//   one base differs, cell "clean"  clean = no reference k-mer of either strand within Hamming distance 2 (host
//                                   precomputed).  Then the read k-mer is provably not a reference k-mer and the
//                                   reference k-mer here is its only possible neighbour: if both have the same
//                                   canonical orientation and the differing base lies in the window, its V
//                                   counter is known on the spot (one fire-and-forget global atomic); otherwise
//                                   it touches nothing.
//   anything else                   (no diagonal, several errors within one k-mer, strain-specific or repetitive
//                                   neighbourhoods, sequence ends): perfect-hash membership test when the lane has
//                                   no diagonal (a hit re-seeds it), else the asynchronous slow pipeline.
// Level 2 only ever sees the k-mers Level 1 could not prove exact (~one in eight at 0.5 % error), densely packed:
// every lane of the wave has work, which the per-base divergent branches of a one-level design cannot offer.
//
// Slow path -- k-mers not resolved on the diagonal are compacted (ballot + prefix popcount) into a per-wave LDS
// queue; whenever 64 are pending they become a batch of the SlowPipe (one k-mer per lane), which looks up the
// neighbours of each and adds to the V counter of the smallest (position, V row) while Level 2 keeps going.
//
// LDS bins: one 32-bit word per cell, two 16-bit halves.  Level 1 adds without looking at the result; a launch
// gives a workgroup at most kMaxRecordsPerGroup records and a record hits a (cell, half) at most once on its
// diagonal, so Level 1 adds at most 0x4000 per launch to a half.  Every other add (Level 2, slow path) is a
// returning add that moves 0x2000 to the u64 plane (compare-and-swap) whenever it sees the half at or above
// 0x2000.  Hence a half stays below 0x2000 + 0x4000 + (adds in flight) < 0x8000 and can never carry into its
// neighbour, whatever the input (millions of identical k-mers included).  At the end the bins are written as one
// coalesced slab per workgroup; fold adds the slabs into the u64 plane.
//
// Cells beyond the LDS bins (large multi-genome indexes) are counted by Level 2 with workgroup-scope (non-sc1)
// atomics in a u32 plane private to the XCD the workgroup runs on (HW_REG_XCC_ID, read at run time, so nothing
// depends on how workgroups are placed); fold adds the planes up afterwards.  A reference too large for LDS is
// read from global memory instead (REF_LDS = false).
constexpr int kScanBlock = 1024;
constexpr int kScanWaves = kScanBlock / 64;
constexpr int kQueueCap = 128;
constexpr int kRangeCap = 128;
constexpr int kMaxRangeLen = 1023;          // 10 bits of a range entry
constexpr uint32_t kMaxRecordsPerGroup = 16384;
constexpr uint32_t kSpillAt = 0x2000u;
constexpr int kSeeds = 4;
constexpr int kRefPadWords = 3;             // words of padding in front of ref_words / cell_codes (48 cells)
constexpr size_t kQueueBytes = (size_t)kScanWaves * kQueueCap * (sizeof(unsigned long long) + 1);
constexpr size_t kRangeBytes = (size_t)kScanWaves * kRangeCap * sizeof(unsigned int);
constexpr size_t kScanLdsFixed = kQueueBytes + kRangeBytes + 16;

struct QueueView { unsigned long long* c; unsigned char* meta; };

// number of set bits of a wave mask below this lane
__device__ __forceinline__ uint32_t lane_prefix(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// reverse the order of the sixteen 2-bit groups of a word
__device__ __forceinline__ uint32_t rev2_32(uint32_t x) {
    const uint32_t t = __builtin_bitreverse32(x);
    return ((t >> 1) & 0x55555555u) | ((t & 0x55555555u) << 1);
}
__device__ __forceinline__ uint64_t rev2_64(uint64_t x) {
    return ((uint64_t)rev2_32((uint32_t)x) << 32) | rev2_32((uint32_t)(x >> 32));
}
// 32 consecutive 2-bit symbols starting at symbol `pos` of a packed array (16 per word, LSB first); the caller
// guarantees words [pos/16, pos/16 + 2] exist
__device__ __forceinline__ uint64_t symbols_at(const uint32_t* __restrict__ w, int32_t pos) {
    const int32_t wi = pos >> 4;
    const uint32_t sh = 2u * ((uint32_t)pos & 15u);
    const uint32_t w0 = w[wi], w1 = w[wi + 1], w2 = w[wi + 2];
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    return ((uint64_t)hi << 32) | lo;
}
// same for a read record: word indices are clamped to the record (symbols beyond its length are never used)
__device__ __forceinline__ uint64_t read_symbols_at(const uint32_t* __restrict__ w, uint32_t pos, uint32_t last_word) {
    const uint32_t wi = pos >> 4;
    const uint32_t sh = 2u * (pos & 15u);
    const uint32_t w0 = w[min(wi, last_word)], w1 = w[min(wi + 1, last_word)], w2 = w[min(wi + 2, last_word)];
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    return ((uint64_t)hi << 32) | lo;
}

// The slow path as a software pipeline.  A batch of up to 64 queued k-mers (one per lane) advances one stage per
// k-mer step of Level 2: stage 1 has the three pilots in flight (reference k-mer set, low half, high half),
// stage 2 the perfect-hash entry and the two directory entries, stage 3 the first candidate of each list (unless
// the k-mer turned out to be a reference k-mer: then it is counted and done); stage 4 resolves the V counter
// and issues the atomic.  Each stage only *issues* its loads; they are consumed one step later, so the
// slow path's memory latency hides behind Level 2's work instead of stalling the wave.
struct SlowPipe {
    int stage = 0;          // wave-uniform: 0 = empty
    bool have = false;      // this lane holds a k-mer of the batch
    bool stat_only = false; // known not to touch the index: only the k-mer statistics table wants it
    uint64_t c = 0;
    uint32_t isrc = 0;
    // what is in flight, by stage (one set of registers, reused):
    //   after stage 1: b0 = {pilot of U, pilot of the low half, pilot of the high half}
    //   after stage 2: b0 = entry of U, b1 = low directory entry, b2 = high directory entry
    //   after stage 3: b0 = first low candidate, b1 = first high candidate, b2 = {low off, low cnt, high off, high cnt}
    uint4 b0{}, b1{}, b2{};

    __device__ __forceinline__ void start(const QueueView& q, uint32_t n, int lane, const IndexView& ix) {
        have = (uint32_t)lane < n;
        c = have ? q.c[lane] : 0ull;
        const uint32_t meta = have ? q.meta[lane] : 0u;
        isrc = meta & 1u;
        stat_only = (meta & 2u) != 0;
        const int lo_bits = 2 * ix.lo_bases;
        const uint64_t lo = c & ((1ull << lo_bits) - 1ull), hi = c >> lo_bits;
        b0.x = ix.pilots[phf_bucket(c, ix.log2nb)];
        b0.y = ix.lo.pilots[phf_bucket(lo, ix.lo.log2nb)];
        b0.z = ix.hi.pilots[phf_bucket(hi, ix.hi.log2nb)];
        stage = 1;
    }

    template <typename CountExact>
    __device__ __forceinline__ void advance(const IndexView& ix, unsigned long long* __restrict__ v_counters, CountExact&& count_exact,
                                            const KmerTable& kt) {
        const int lo_bits = 2 * ix.lo_bases;
        const uint64_t lo = c & ((1ull << lo_bits) - 1ull), hi = c >> lo_bits;
        if (stage == 1) {
            const uint32_t pu = b0.x, pl = b0.y, ph = b0.z;
            b0 = *reinterpret_cast<const uint4*>(ix.kmer_pos + phf_pos(c, pu, ix.m));
            b1 = *reinterpret_cast<const uint4*>(ix.lo.dir + phf_pos(lo, pl, ix.lo.m));
            b2 = *reinterpret_cast<const uint4*>(ix.hi.dir + phf_pos(hi, ph, ix.hi.m));
            stage = 2;
        } else if (stage == 2) {
            if (have && !stat_only && ((uint64_t)b0.x | ((uint64_t)b0.y << 32)) == c) {   // a reference k-mer after all
                count_exact(b0.z, b0.w & kIdMask, isrc);
                have = false;
            }
            const uint32_t cnt_lo = (have && !stat_only && b1.x == (uint32_t)lo) ? b1.z : 0u;
            const uint32_t cnt_hi = (have && !stat_only && b2.x == (uint32_t)hi) ? b2.z : 0u;
            const uint32_t off_lo = b1.y, off_hi = b2.y;
            if (cnt_lo) b0 = *reinterpret_cast<const uint4*>(ix.lo.cand + off_lo);
            if (cnt_hi) b1 = *reinterpret_cast<const uint4*>(ix.hi.cand + off_hi);
            b2 = make_uint4(off_lo, cnt_lo, off_hi, cnt_hi);
            stage = 3;
        } else if (stage == 3) {
            const int k = ix.k, wlo = ix.wstart, whi = ix.wstart + ix.W;
            const uint32_t off_lo = b2.x, cnt_lo = b2.y, off_hi = b2.z, cnt_hi = b2.w;
            uint64_t best = ~0ull;   // (j << 32) | V row, smallest wins
            auto consider = [&](const uint4& e) {
                const int j = single_diff_pos((uint64_t)e.x | ((uint64_t)e.y << 32), c, k);
                if (j >= wlo && j < whi && ((e.w >> (j - wlo)) & 1u)) {   // the neighbour owns a bucket at j
                    const uint64_t key = ((uint64_t)j << 32) | (e.z + (uint32_t)__popc(e.w & ((1u << (j - wlo)) - 1u)));
                    if (key < best) best = key;
                }
            };
            if (cnt_lo) consider(b0);
            for (uint32_t i = 1; i < cnt_lo; ++i) consider(*reinterpret_cast<const uint4*>(ix.lo.cand + off_lo + i));   // rare
            if (cnt_hi) consider(b1);
            for (uint32_t i = 1; i < cnt_hi; ++i) consider(*reinterpret_cast<const uint4*>(ix.hi.cand + off_hi + i));   // rare
            if (best != ~0ull) {
                const int j = (int)(best >> 32);
                const uint32_t row = (uint32_t)best;
                const uint32_t b = (uint32_t)(c >> (2 * (k - 1 - j))) & 3u;
                atomicAdd(v_counters + ((uint64_t)row * 4 + b) * 2 + isrc, 1ull);
            } else if (have && kt.keys) {
                ktab_insert(kt, c, isrc);   // touches no window bucket: only KMC's distinct / counted totals see it
            }
            stage = 0;
        }
    }

    template <typename CountExact>
    __device__ __forceinline__ void finish(const IndexView& ix, unsigned long long* __restrict__ v_counters, CountExact&& count_exact,
                                           const KmerTable& kt) {
        while (stage) advance(ix, v_counters, count_exact, kt);
    }
};

template <bool REF_LDS, bool STATS>
__global__ __launch_bounds__(kScanBlock) void scan_count_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* queue_c = reinterpret_cast<unsigned long long*>(smem);
    unsigned char* queue_m = smem + (size_t)kScanWaves * kQueueCap * sizeof(unsigned long long);
    unsigned int* range_q = reinterpret_cast<unsigned int*>(smem + kQueueBytes);
    unsigned int* block_kmers = reinterpret_cast<unsigned int*>(smem + kQueueBytes + kRangeBytes);   // 16 B reserved
    unsigned int* bins = block_kmers + 4;
    unsigned int* lds_ref = bins + a.n_lds_bins;   // REF_LDS: padded ref words, then the padded per-cell codes

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform by construction: tell the compiler
    const QueueView q{queue_c + wave * kQueueCap, queue_m + wave * kQueueCap};
    unsigned int* const rq = range_q + wave * kRangeCap;

    const uint32_t total = a.total_cells;
    const uint32_t n_refw = kRefPadWords + (total + 15) / 16 + 4;   // both arrays: front pad, cells, >= k cells of back pad
    for (uint32_t i = threadIdx.x; i < a.n_lds_bins; i += kScanBlock) bins[i] = 0u;
    if (REF_LDS) {
        for (uint32_t i = threadIdx.x; i < n_refw; i += kScanBlock) { lds_ref[i] = a.ref_words[i]; lds_ref[n_refw + i] = a.cell_codes[i]; }
    }
    __syncthreads();
    // symbol 0 of both arrays is cell 0; negative symbol positions down to -48 are readable padding
    const unsigned int* refw = (REF_LDS ? lds_ref : a.ref_words) + kRefPadWords;
    const unsigned int* codew = (REF_LDS ? lds_ref + n_refw : a.cell_codes) + kRefPadWords;
    const unsigned int* cflags = a.cell_flags;   // Level 2 only: 4 flag bits per cell, global memory (L1 / L2 cached)

    const int k = a.k;
    const uint32_t W = (uint32_t)a.W;
    const int wlo = a.wstart, whi = a.wstart + a.W;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;  // k <= 31
    // 2k-bit values are kept as explicit 32-bit halves: 64-bit shifts and compares are slow-rate VALU ops
    const uint32_t kmask_lo = (uint32_t)kmask, kmask_hi = (uint32_t)(kmask >> 32);
    const uint32_t kmask1 = (uint32_t)((1ull << k) - 1ull);   // one flag per base of a k-mer
    const int rcshift = 2 * (k - 1);
    const uint32_t rc_sh = (uint32_t)rcshift & 31u;
    const uint32_t rc_in_hi = rcshift >= 32 ? 0xffffffffu : 0u;   // which half receives the new complemented base
    const uint32_t km1 = (uint32_t)k - 1u;
    const uint64_t n_e = e_plane_len(a.n_u);
    unsigned int* const e_local = a.e_planes ? a.e_planes + (size_t)xcc_id() * n_e : nullptr;
    unsigned long long* const v_counters = a.counters + n_e;
    const uint32_t last_word = a.stride_words - 1u;

    // +1 on the E counter of reference k-mer `id` (first occurrence at `cell`) read in orientation `isrc`: Level 2 and
    // the slow path (returning add + spill, see the header comment)
    auto count_exact = [&](uint32_t cell, uint32_t id, uint32_t isrc) {
        if (cell < a.n_lds_bins) {
            const unsigned int old = atomicAdd(&bins[cell], isrc ? 0x10000u : 1u);
            if (((isrc ? old >> 16 : old) & 0xffffu) >= kSpillAt) {
                unsigned int cur = __hip_atomic_load(&bins[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (;;) {
                    if (((isrc ? cur >> 16 : cur) & 0xffffu) < kSpillAt) break;   // somebody else moved it
                    const unsigned int prev = atomicCAS(&bins[cell], cur, cur - (isrc ? (kSpillAt << 16) : kSpillAt));
                    if (prev == cur) { atomicAdd(a.counters + 2 * (size_t)id + isrc, (unsigned long long)kSpillAt); break; }
                    cur = prev;
                }
            }
        } else if (e_local) {
            __hip_atomic_fetch_add(e_local + 2 * (size_t)id + isrc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            atomicAdd(a.counters + 2 * (size_t)id + isrc, 1ull);
        }
    };
    // +1 on the V counter of "reference k-mer id with base bb at canonical position j" (reference k-mers: V row = id*W + t)
    auto count_variant = [&](uint32_t id, int j, uint32_t bb, uint32_t isrc) {
        atomicAdd(v_counters + (((uint64_t)id * W + (uint32_t)(j - wlo)) * 4 + bb) * 2 + isrc, 1ull);   // fire and forget
    };

    const KmerTable kt{STATS ? a.ktab_keys : nullptr, a.ktab_cnt, a.ktab_log2, a.ktab_overflow, a.mate};   // STATS = full_kmer_stats
    uint32_t nkm = 0;  // k-mer occurrences of this lane's records
    uint32_t qn = 0;   // wave-uniform fill of the slow-path queue
    uint32_t qr = 0;   // wave-uniform fill of the range queue
    SlowPipe pipe;
    const IndexView& ix = *a.ixp;

    uint64_t n_records = a.n_records;
    if (a.n_records_dev) {
        const uint64_t nd = *a.n_records_dev;
        n_records = nd > a.rec_base ? min(nd - a.rec_base, a.n_records) : 0ull;
    }
    const uint32_t* const words0 = a.words + a.rec_base * a.stride_words;
    const uint16_t* const lens0 = a.lens + a.rec_base;
    const uint64_t n_tiles = (n_records + 63) / 64;
    for (uint64_t tile = (uint64_t)blockIdx.x * kScanWaves + wave; tile < n_tiles; tile += (uint64_t)gridDim.x * kScanWaves) {
        const uint64_t r = tile * 64 + lane;
        const bool live = r < n_records;
        uint32_t len = live ? (uint32_t)lens0[r] : 0u;
        if (len < (uint32_t)k) len = 0u;   // no k-mer
        uint32_t maxlen = len;
#pragma unroll
        for (int off = 32; off; off >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, off));
        maxlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxlen);
        const uint32_t* __restrict__ w = words0 + (live ? r : 0) * a.stride_words;
        nkm += len ? len - km1 : 0u;

        // ---- seeds -> diagonal ----------------------------------------------------------------------------------
        // dg: cell of the reference k-mer aligned with read k-mer 0; read k-mer s <-> cell dg + s (fwd) / dg - s
        int32_t dg = 0;
        bool fwd = true, seeded = false, l1ok = false;   // seeded: diagonal known; l1ok: ... and all its cells have LDS bins
        if (maxlen) {
            uint64_t sc[kSeeds];
            uint32_t sisrc[kSeeds], spil[kSeeds];
#pragma unroll
            for (int sq = 0; sq < kSeeds; ++sq) {
                const uint32_t s = len ? ((len - (uint32_t)k) * (uint32_t)sq) / (uint32_t)(kSeeds - 1) : 0u;
                const uint64_t g = read_symbols_at(w, s, last_word) & kmask;       // base t of the k-mer at bits 2t
                const uint64_t rr = ~g & kmask;                                      // its reverse complement, first base on top
                const uint64_t ff = rev2_64(g) >> (64 - 2 * k);                      // the k-mer, first base on top
                const bool lt = ff < rr;                                             // lcb.rs:90-94
                sc[sq] = lt ? ff : rr;
                sisrc[sq] = lt ? 0u : 1u;
                spil[sq] = ix.pilots[phf_bucket(sc[sq], ix.log2nb)];
            }
            uint32_t best_cell = 0xffffffffu;
#pragma unroll
            for (int sq = 0; sq < kSeeds; ++sq) {
                const uint32_t s = len ? ((len - (uint32_t)k) * (uint32_t)sq) / (uint32_t)(kSeeds - 1) : 0u;
                const uint4 e = *reinterpret_cast<const uint4*>(ix.kmer_pos + phf_pos(sc[sq], spil[sq], ix.m));
                if (len && ((uint64_t)e.x | ((uint64_t)e.y << 32)) == sc[sq] && e.z < best_cell) {
                    // several seeds may hit (usually all, on one diagonal); prefer the lowest cell: in a multi-genome
                    // index that is the first genome, whose cells have the LDS bins
                    const bool f = sisrc[sq] == (e.w >> 31);   // same strand as the reference?
                    const int64_t d0 = f ? (int64_t)e.z - (int64_t)s : (int64_t)e.z + (int64_t)s;
                    const int64_t lo_cell = f ? d0 : d0 - (int64_t)(len - (uint32_t)k);
                    const int64_t hi_cell = f ? d0 + (int64_t)(len - (uint32_t)k) : d0;
                    // the whole read must lie on the reference (hi_cell + k <= total); Level 1 also needs its cells in the LDS bins
                    if (lo_cell >= 0 && hi_cell + k <= (int64_t)total) {
                        best_cell = e.z; dg = (int32_t)d0; fwd = f; seeded = true;
                        l1ok = hi_cell < (int64_t)a.n_lds_bins;
                    }
                }
            }
        }

        // ---- Level 1 / Level 2 state machine (all control flow wave-uniform) --------------------------------------
        uint32_t i = 0;                 // base index of the next Level-1 step
        uint32_t x = 0, Mw = 0, Zc = 0; // current read word (unused bases), its mismatch flags, the aligned cell codes
        uint32_t dwin = 0xffffffffu;    // mismatch flags of the last k bases
        uint32_t run_start = 0;         // first k-mer of the lane's open range
        bool inrun = false;             // the lane has an open range
        // LDS byte offset (within bins) of the cell of the k-mer that ends at base i: starts k-1 steps "before" cell dg
        uint32_t bin_off = (uint32_t)((fwd ? dg - (int32_t)km1 : dg + (int32_t)km1) * 4);
        const uint32_t bin_step = fwd ? 4u : (uint32_t)-4;
        int phase = maxlen ? 0 : 2;     // 0 = stepping, 1 = close the open ranges, 2 = Level 1 done for this tile
        for (;;) {
            if (qr >= 64u || (phase == 2 && qr)) {
                // ================= Level 2: one queued range per lane ==============================================
                const uint32_t nb2 = min(qr, 64u);
                const uint32_t ent = (uint32_t)lane < nb2 ? rq[lane] : 0u;
                {   // move the rest of the queue down
                    const uint32_t rest = qr - nb2;
                    const uint32_t t = (uint32_t)lane < rest ? rq[64 + lane] : 0u;
                    __builtin_amdgcn_wave_barrier();
                    if ((uint32_t)lane < rest) rq[lane] = t;
                    __builtin_amdgcn_wave_barrier();
                    qr = rest;
                }
                if (a.ablate == 1) continue;   // measurement aid: Level 1 alone (incomplete counts)
                const int src = (int)(ent & 63u);
                const uint32_t s_first = (ent >> 6) & 0xffffu;
                const uint32_t n2 = (uint32_t)lane < nb2 ? ent >> 22 : 0u;
                const int32_t dg2 = __shfl(dg, src);
                const bool fwd2 = __shfl((int)fwd, src) != 0;
                const bool seeded2 = __shfl((int)seeded, src) != 0 && n2;
                const uint32_t r_lo = (uint32_t)__shfl((int)(uint32_t)r, src), r_hi = (uint32_t)__shfl((int)(uint32_t)(r >> 32), src);
                const uint32_t* __restrict__ w2 = words0 + (n2 ? (((uint64_t)r_hi << 32) | r_lo) : 0ull) * a.stride_words;
                uint32_t nmax = n2;
#pragma unroll
                for (int off = 32; off; off >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, off));
                nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);

                // prime the rolling k-mers and the mismatch mask with the k-1 bases [s_first, s_first + k - 1)
                const int kp = k - 1;
                const uint64_t pmask = (1ull << (2 * kp)) - 1ull;
                const uint64_t g = read_symbols_at(w2, s_first, last_word) & pmask;
                const uint64_t f0 = kp ? rev2_64(g) >> (64 - 2 * kp) : 0ull;
                const uint64_t r0 = (~g & pmask) << 2;
                uint32_t f_lo = (uint32_t)f0, f_hi = (uint32_t)(f0 >> 32), r_lo2 = (uint32_t)r0, r_hi2 = (uint32_t)(r0 >> 32);
                uint32_t d_lo = 0, d_hi = 0;
                const uint32_t ddir = fwd2 ? 1u : 0xffffffffu;
                // cellc: cell of the reference k-mer the current read k-mer is aligned with (>= total: no diagonal)
                uint32_t cellc = 0xffffffffu;
                if (seeded2) {
                    cellc = (uint32_t)(fwd2 ? dg2 + (int32_t)s_first : dg2 - (int32_t)s_first);
                    // reference bases as the read sees them, aligned with read bases s_first .. s_first + k - 2
                    uint64_t gref;
                    if (fwd2) gref = symbols_at(refw, (int32_t)cellc);
                    else gref = kp ? (~rev2_64(symbols_at(refw, (int32_t)cellc + 1))) >> (64 - 2 * kp) : 0ull;
                    const uint64_t gx = (g ^ gref) & pmask;
                    const uint64_t d0 = kp ? rev2_64(gx) >> (64 - 2 * kp) : 0ull;
                    d_lo = (uint32_t)d0; d_hi = (uint32_t)(d0 >> 32);
                }
                uint32_t ddir2 = ddir, id = 0, bad = 0;
                bool id_ok = false;
                // base stream: xw = unused bases of the current word, xn = the next word (loaded a word ahead)
                uint32_t bi = s_first + km1;
                uint32_t xw = w2[min(bi >> 4, last_word)] >> (2u * (bi & 15u));
                uint32_t xn = w2[min((bi >> 4) + 1u, last_word)];
                for (uint32_t t = 0; t < nmax; ++t) {
                    const uint32_t base = xw & 3u;
                    xw >>= 2;
                    ++bi;
                    if ((bi & 15u) == 0u) { xw = xn; xn = w2[min((bi >> 4) + 1u, last_word)]; }
                    f_hi = ((f_hi << 2) | (f_lo >> 30)) & kmask_hi;
                    f_lo = ((f_lo << 2) | base) & kmask_lo;
                    const uint32_t cb = (3u - base) << rc_sh;
                    r_lo2 = ((r_lo2 >> 2) | (r_hi2 << 30)) | (cb & ~rc_in_hi);
                    r_hi2 = (r_hi2 >> 2) | (cb & rc_in_hi);
                    const bool valid = t < n2;

                    const bool fwd_dir = ddir2 == 1u;
                    const bool ok = valid && cellc < total;
                    const uint32_t nc = ok ? cellc : 0u;                         // clamped: the loads below are unconditional
                    const uint32_t bpos = fwd_dir ? nc + km1 : nc;               // reference base aligned with the new read base
                    const uint32_t rb = ((refw[bpos >> 4] >> (2 * (bpos & 15))) & 3u) ^ (fwd_dir ? 0u : 3u);
                    const uint32_t nd_hi = ((d_hi << 2) | (d_lo >> 30)) & kmask_hi;
                    const uint32_t nd_lo = ((d_lo << 2) | (base ^ rb)) & kmask_lo;
                    d_hi = ok ? nd_hi : d_hi;
                    d_lo = ok ? nd_lo : d_lo;
                    const uint32_t fl = (cflags[nc >> 3] >> (4 * (nc & 7))) & 15u;   // this cell's clean / rc / follow bits
                    const bool clean = ok && (fl & 1u);
                    const bool follow = fl & (fwd_dir ? 4u : 8u);                   // id continues from the cell we came from
                    const uint32_t ref_isrc = ((fl >> 1) & 1u) ^ (fwd_dir ? 0u : 1u); // orientation of the reference k-mer as the read sees it
                    const bool id_known = ok && id_ok && follow;                 // previous id +-1 along an unbroken stretch
                    id = id_known ? id + ddir2 : id;
                    id_ok = id_known;
                    const uint32_t dbits = (d_lo | (d_lo >> 1)) & 0x55555555u, dbits_hi = (d_hi | (d_hi >> 1)) & 0x55555555u;
                    const uint32_t n_diff = (uint32_t)__popc(dbits) + (uint32_t)__popc(dbits_hi);
                    const bool exact = id_known && n_diff == 0;                  // follow => a k-mer starts here; diff == 0 => it is this one
                    if (exact) { count_exact(nc, id, ref_isrc); bad = 0; }      // read orientation == the reference k-mer's
                    // one base differs from a clean reference k-mer whose id is known (the sequencing-error case).
                    // Provably not a reference k-mer, and that k-mer is its only possible neighbour (bk_device.h, amb):
                    // name its V counter on the spot, or it touches nothing.
                    const bool simple = id_known && clean && n_diff == 1;
                    bool stat_only = false;   // full_kmer_stats: a k-mer known to touch nothing still has to be counted somewhere
                    if (simple) {
                        const bool lt = f_hi < r_hi2 || (f_hi == r_hi2 && f_lo < r_lo2);   // lcb.rs:90-94
                        const uint32_t isrc = lt ? 0u : 1u;
                        const int from_right = dbits ? (__builtin_ctz(dbits) >> 1) : 16 + (__builtin_ctz(dbits_hi) >> 1);
                        const int j = isrc ? from_right : k - 1 - from_right;    // differing position in canonical orientation
                        if (ref_isrc == isrc && j >= wlo && j < whi) {
                            const int sh = 2 * (k - 1 - j);                     // base of the canonical k-mer at j
                            const uint32_t c_lo = lt ? f_lo : r_lo2, c_hi = lt ? f_hi : r_hi2;
                            const uint32_t bb = (sh >= 32 ? c_hi >> (sh - 32) : c_lo >> sh) & 3u;
                            count_variant(id, j, bb, isrc);
                        } else {
                            stat_only = STATS;
                        }
                    }
                    // everything else -- unknown id, several differences, dirty neighbourhoods, lanes without a
                    // diagonal, the miss queue and the slow pipeline
                    const bool slow = valid && !exact && !simple;
                    if (__ballot(slow | stat_only) || qn >= 64 || pipe.stage) {
                        bool lookup = false, miss = false;
                        uint64_t c = 0;
                        uint32_t isrc = 0;
                        if (slow | stat_only) {
                            const bool lt = f_hi < r_hi2 || (f_hi == r_hi2 && f_lo < r_lo2);
                            isrc = lt ? 0u : 1u;
                            c = lt ? (((uint64_t)f_hi << 32) | f_lo) : (((uint64_t)r_hi2 << 32) | r_lo2);
                            miss = stat_only;
                        }
                        if (slow) {
                            if (!ok) {
                                lookup = true;                                   // no diagonal: perfect-hash lookup, may re-seed
                            } else {
                                bool done = false;
                                if (n_diff == 0 || (n_diff == 1 && clean)) {
                                    id = a.id_at[nc];                            // repeat, or first step on this stretch
                                    id_ok = id != 0xffffffffu;
                                    if (id_ok && n_diff == 0) {
                                        count_exact(nc, id, ref_isrc);
                                        bad = 0;
                                        done = true;
                                    } else if (id_ok) {
                                        done = true;
                                        const int from_right = dbits ? (__builtin_ctz(dbits) >> 1) : 16 + (__builtin_ctz(dbits_hi) >> 1);
                                        const int j = isrc ? from_right : k - 1 - from_right;
                                        if (ref_isrc == isrc && j >= wlo && j < whi)
                                            count_variant(id, j, (uint32_t)(c >> (2 * (k - 1 - j))) & 3u, isrc);
                                        else if (STATS) { miss = true; stat_only = true; }
                                    }
                                }
                                if (!done) {
                                    // several differences, a dirty neighbourhood or no k-mer at this cell: full search,
                                    // asynchronously; the lane keeps walking its diagonal and gives it up only after more
                                    // than a k-mer of such steps
                                    miss = true;
                                    bad += 1;
                                    if (bad > (uint32_t)k + 4u) cellc = 0xffffffffu;
                                }
                            }
                        }
                        if (lookup) {
                            // perfect-hash membership test; a hit (re-)seeds the diagonal
                            const uint32_t pilot = ix.pilots[phf_bucket(c, ix.log2nb)];
                            const uint4 e = *reinterpret_cast<const uint4*>(ix.kmer_pos + phf_pos(c, pilot, ix.m));
                            if (((uint64_t)e.x | ((uint64_t)e.y << 32)) == c) {
                                id = e.w & kIdMask;
                                id_ok = true;
                                count_exact(e.z, id, isrc);
                                ddir2 = (isrc == (e.w >> 31)) ? 1u : 0xffffffffu;   // same strand as the reference?
                                cellc = e.z;
                                d_lo = d_hi = 0;
                                bad = 0;
                            } else {
                                miss = true;
                            }
                        }
                        const unsigned long long mm = __ballot(miss);
                        if (mm) {
                            const uint32_t pos = qn + lane_prefix(mm);
                            if (miss) { q.c[pos] = c; q.meta[pos] = (unsigned char)(isrc | (stat_only ? 2u : 0u)); }
                            qn += (uint32_t)__popcll(mm);
                            __builtin_amdgcn_wave_barrier();
                        }
                        if (qn >= 64) {
                            // a full batch is waiting: retire the batch in flight (its loads were issued steps ago), then
                            // take 64 k-mers off the queue and issue the first loads of the new batch
                            pipe.finish(ix, v_counters, count_exact, kt);
                            pipe.start(q, 64, lane, ix);
                            const uint32_t rest = qn - 64;
                            const unsigned long long tc = ((uint32_t)lane < rest) ? q.c[64 + lane] : 0ull;
                            const unsigned char tm = ((uint32_t)lane < rest) ? q.meta[64 + lane] : (unsigned char)0;
                            __builtin_amdgcn_wave_barrier();
                            if ((uint32_t)lane < rest) { q.c[lane] = tc; q.meta[lane] = tm; }
                            __builtin_amdgcn_wave_barrier();
                            qn = rest;
                        } else if (pipe.stage) {
                            pipe.advance(ix, v_counters, count_exact, kt);
                        }
                    }
                    cellc = cellc < total ? cellc + ddir2 : cellc;               // without a diagonal the lane stays without
                }
                continue;
            }
            if (phase == 2) break;
            if (phase == 1) {
                // ---- close the ranges still open at the end of the longest read (qr < 64 here) ----
                const unsigned long long om = __ballot(inrun);
                if (om) {
                    if (inrun) rq[qr + lane_prefix(om)] = (uint32_t)lane | (run_start << 6) | ((maxlen - km1 - run_start) << 22);
                    qr += (uint32_t)__popcll(om);
                    __builtin_amdgcn_wave_barrier();
                    inrun = false;
                }
                phase = 2;
                continue;
            }
            // ================= Level 1: one base =============================================================
            const uint32_t b = i & 15u;
            if (b == 0u) {
                // next read word, the reference word aligned with it, the cell codes of the 16 k-mers that end in it
                x = (i < len) ? w[i >> 4] : 0u;
                const bool act = l1ok && i < len;
                const int32_t p0 = act ? (fwd ? dg + (int32_t)i : dg + (int32_t)km1 - (int32_t)i - 15) : 0;
                const uint32_t sh = 2u * ((uint32_t)p0 & 15u);
                const uint32_t yl = __builtin_amdgcn_alignbit(refw[(p0 >> 4) + 1], refw[p0 >> 4], sh);
                const uint32_t y = fwd ? yl : ~rev2_32(yl);
                const uint32_t dd = x ^ y;
                Mw = act ? (dd | (dd >> 1)) & 0x55555555u : 0x55555555u;   // no usable diagonal: every base "differs"
                const int32_t c0 = act ? (fwd ? dg + (int32_t)i - (int32_t)km1 : dg - (int32_t)i + (int32_t)km1 - 15) : 0;
                const uint32_t zsh = 2u * ((uint32_t)c0 & 15u);
                const uint32_t zl = __builtin_amdgcn_alignbit(codew[(c0 >> 4) + 1], codew[c0 >> 4], zsh);
                Zc = fwd ? zl : __builtin_bitreverse32(zl);   // reversing the bits also swaps codes 1 <-> 2: the strand flips
                // a range that has grown long is cut here, so that its length always fits the queue entry
                if (i >= (uint32_t)kMaxRangeLen - 64u) {
                    const uint32_t s_now = i - km1;   // an open range implies i >= k
                    const bool cut = inrun && s_now - run_start >= (uint32_t)(kMaxRangeLen - 32);
                    const unsigned long long cm = __ballot(cut);
                    if (cm) {
                        if (cut) { rq[qr + lane_prefix(cm)] = (uint32_t)lane | (run_start << 6) | ((s_now - run_start) << 22); run_start = s_now; }
                        qr += (uint32_t)__popcll(cm);
                        __builtin_amdgcn_wave_barrier();
                        if (qr >= 64u) continue;   // (the word set-up is idempotent: i has not moved)
                    }
                }
            }
            {
                const uint32_t mbit = (Mw >> (2u * b)) & 1u;
                dwin = ((dwin << 1) | mbit) & kmask1;
                const uint32_t z = (Zc >> (2u * b)) & 3u;
                const bool valid = i < len && i >= km1;
                const bool ex = valid && dwin == 0u && z != 0u;
                if (ex) {
                    // z = 1: the reference k-mer here is canonical as the read sees it (low half), 2: as its reverse complement
                    __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(bins) + bin_off),
                                           ((z << 15) | z) & 0x10001u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                bin_off += bin_step;
                const bool ne = valid && !ex;
                const bool end = inrun && !ne;
                const uint32_t s = i - km1;   // the k-mer that ends at base i (meaningful when i >= k - 1)
                run_start = (ne && !inrun) ? s : run_start;
                inrun = ne;
                const unsigned long long ends = __ballot(end);
                if (ends) {
                    if (end) rq[qr + lane_prefix(ends)] = (uint32_t)lane | (run_start << 6) | ((s - run_start) << 22);
                    qr += (uint32_t)__popcll(ends);
                    __builtin_amdgcn_wave_barrier();
                }
                ++i;
                if (i == maxlen) phase = 1;
            }
        }
    }
    pipe.finish(ix, v_counters, count_exact, kt);
    if (qn) {
        pipe.start(q, qn, lane, ix);
        pipe.finish(ix, v_counters, count_exact, kt);
    }

    // bins -> this workgroup's slab (coalesced); k-mer tally -> one atomic per workgroup
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
size_t scan_ref_lds_bytes(uint32_t total_cells) {
    return 2 * (size_t)(kRefPadWords + (total_cells + 15) / 16 + 4) * sizeof(unsigned int);
}
size_t scan_lds_bytes(uint32_t n_lds_bins, bool ref_in_lds, uint32_t total_cells) {
    return kScanLdsFixed + (size_t)n_lds_bins * sizeof(unsigned int) + (ref_in_lds ? scan_ref_lds_bytes(total_cells) : 0);
}
int scan_ref_pad_words() { return kRefPadWords; }

uint32_t scan_grid(uint64_t n_records, int n_cus) {
    const uint64_t want = (n_records + kScanBlock - 1) / kScanBlock;
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)n_cus));
}
// records one launch may take so that no workgroup sees more than kMaxRecordsPerGroup of them
uint64_t scan_max_records(uint32_t grid) { return (uint64_t)grid * (kMaxRecordsPerGroup - kScanBlock - 64); }

hipError_t launch_scan_count(const ScanArgs& a, uint32_t grid, hipStream_t stream) {
    if (a.n_records == 0 || a.W <= 0) return hipSuccess;
    if (a.n_records > scan_max_records(grid)) return hipErrorInvalidValue;
    const size_t lds = scan_lds_bytes(a.n_lds_bins, a.ref_in_lds != 0, a.total_cells);
    const bool stats = a.ktab_keys != nullptr;
    void (*kern)(ScanArgs) = a.ref_in_lds ? (stats ? scan_count_kernel<true, true> : scan_count_kernel<true, false>)
                                          : (stats ? scan_count_kernel<false, true> : scan_count_kernel<false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kScanBlock), lds, stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K1b
// E[2 id_at[c] + h] += sum over workgroup slabs of half h of slab[b][c]   (c < n_lds_bins: bins are per cell), and
// E[i]              += sum over the 8 XCD planes of e_planes[x][i]         (planes re-zeroed for the next batch).
__global__ __launch_bounds__(256) void fold_kernel(FoldArgs f) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * 256;
    // blockIdx.y splits the slabs into groups so that the 31 MB of slabs are streamed by the whole chip; each
    // group adds its partial sums with one u64 atomic per non-zero bin half
    const uint32_t per = (f.n_slabs + gridDim.y - 1) / gridDim.y;
    const uint32_t b0 = blockIdx.y * per, b1 = min(f.n_slabs, b0 + per);
    for (uint64_t i = tid; i < f.n_lds_bins && b0 < b1; i += nthreads) {
        unsigned long long s0 = 0, s1 = 0;
        for (uint32_t b = b0; b < b1; ++b) {
            const unsigned int v = f.slabs[(size_t)b * f.n_lds_bins + i];
            s0 += v & 0xffffu;
            s1 += v >> 16;
        }
        if (s0 | s1) {
            const uint32_t id = f.id_at[i];   // a counted cell always has a reference k-mer
            if (s0) atomicAdd(f.counters + 2 * (size_t)id, s0);
            if (s1) atomicAdd(f.counters + 2 * (size_t)id + 1, s1);
        }
    }
    if (f.e_planes && blockIdx.y == 0) {
        for (uint64_t i = tid; i < f.n_e; i += nthreads) {
            unsigned long long s = 0;
#pragma unroll
            for (int x = 0; x < kXcdPlanes; ++x) {
                const unsigned int v = f.e_planes[(size_t)x * f.n_e + i];
                if (v) { s += v; f.e_planes[(size_t)x * f.n_e + i] = 0u; }
            }
            if (s) atomicAdd(f.counters + i, s);
        }
    }
}

void launch_fold(const FoldArgs& f, hipStream_t stream) {
    const uint64_t work = std::max<uint64_t>(f.n_lds_bins, f.e_planes ? f.n_e : 0);
    if (work == 0) return;
    uint64_t blocks = (work + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    const unsigned groups = std::max(1u, std::min(16u, f.n_slabs / 8));
    hipLaunchKernelGGL(fold_kernel, dim3((unsigned)blocks, groups), dim3(256), 0, stream, f);
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

// stats / present / kept / distinct += column sums of the partials rows written by the finalize workgroups
__global__ __launch_bounds__(256) void finalize_reduce_kernel(FinalizeArgs a, int n_rows) {
    __shared__ unsigned long long wave_sums[4];
    const int n3 = a.ix.n_files * 3, cols = n3 + 2;
    const int col = blockIdx.x;   // one workgroup per column, rows strided over its threads
    unsigned long long s = 0;
    for (int r = threadIdx.x; r < n_rows; r += 256) s += a.partials[(size_t)r * cols + col];
#pragma unroll
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x != 0) return;
    s = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
    if (!s) return;
    if (col < n3) { a.stats[col] += s; if (col % 3 != 2) a.present[col / 3] = 1; }
    else if (col == n3) { if (a.kept_total) *a.kept_total += s; }
    else if (a.distinct_total) *a.distinct_total += s;
}

// K2a: one thread per V counter.  A kept non-reference k-mer almost always touches exactly one window bucket
// (the one its name says); then the whole of map_kmers for it is: vote once per BucketInfo of that bucket,
// and per genome file "variant" (or "perfect" if the file has exactly W entries there, which needs W == 1 or
// repeats).  K-mers that touch several buckets need cross-bucket per-genome totals; they are appended (by
// counter index) to `deferred` and mapped by K2b.
__global__ __launch_bounds__(256) void finalize_variant_kernel(FinalizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const IndexView& ix = a.ix;
    uint32_t* lstats = reinterpret_cast<uint32_t*>(smem);   // [n_files][3] block-local tallies + 2 scratch words
    for (int g = threadIdx.x; g < ix.n_files * 3 + 2; g += 256) lstats[g] = 0;
    __syncthreads();

    const int k = ix.k;
    const uint64_t n_e = e_plane_len(ix.n_u);
    const uint64_t n_v = v_plane_len(ix.n_rows);
    const unsigned long long* __restrict__ vc = a.counters + n_e;
    unsigned int kept = 0, distinct = 0;

    for (uint64_t vi = (uint64_t)blockIdx.x * 256 + threadIdx.x; vi < n_v; vi += (uint64_t)gridDim.x * 256) {
        const unsigned long long n = vc[vi];
        distinct += n != 0;
        if (n == 0 || n < a.ci || n > a.cx) continue;           // kmc -ci / -cx act on the true count
        ++kept;
        const unsigned long long v = n > a.cs ? a.cs : n;       // kmc -cs: reported count saturates
        const uint32_t isrc = (uint32_t)vi & 1u;
        const uint32_t bb = (uint32_t)(vi >> 1) & 3u;
        uint32_t t, p;
        row_owner(ix, vi >> 3, p, t);
        const int j = ix.wstart + (int)t;
        const int sh = 2 * (k - 1 - j);
        const uint64_t c = (ix.kmer_of[p] & ~(3ull << sh)) | ((uint64_t)bb << sh);

        // c = u with one base changed at window position j.  It can touch a second window bucket only if another
        // reference k-mer lies at Hamming distance 2 from u (amb[p], precomputed); otherwise its one bucket is
        // u's own bucket at j.  Ambiguous u: enumerate the neighbours; several buckets -> general path (K2b).
        if (ix.amb[p]) {
            uint32_t jmask = 0;   // window positions at which c has a neighbouring reference k-mer
            for_each_neighbour(ix, c, [&](int jj, uint32_t) { jmask |= 1u << (jj - ix.wstart); });
            if (jmask != (1u << t)) {
                const unsigned int at = atomicAdd(a.n_deferred, 1u);
                a.deferred[at] = (uint32_t)vi;
                continue;
            }
        }
        const uint32_t s = ix.slot_of[(size_t)p * ix.W + t];
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
    finalize_epilogue(a, lstats, kept, distinct, lstats + ix.n_files * 3, (int)blockIdx.x);
}

// K2e: the E counters (reference k-mers).  A reference k-mer owns all W of its window buckets (slot_of), and its
// per-genome hit totals -- hence perfect / variant / unique -- depend on the index alone, so the host precomputed
// them (estat).  That makes the map embarrassingly parallel: one thread per (E counter, window bucket) votes for
// the BucketInfos of that bucket; the bucket-0 thread also tallies the statistics.
__global__ __launch_bounds__(256) void finalize_exact_kernel(FinalizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const IndexView& ix = a.ix;
    uint32_t* lstats = reinterpret_cast<uint32_t*>(smem);   // [n_files][3] block-local tallies + 2 scratch words
    for (int g = threadIdx.x; g < ix.n_files * 3 + 2; g += 256) lstats[g] = 0;
    __syncthreads();
    const int k = ix.k;
    const uint32_t W = (uint32_t)ix.W;
    const uint64_t n_work = e_plane_len(ix.n_u) * W;
    unsigned int kept = 0, distinct = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < n_work; g += (uint64_t)gridDim.x * 256) {
        const uint64_t cidx = g / W;
        const uint32_t t = (uint32_t)(g % W);
        const unsigned long long n = a.counters[cidx];
        distinct += (n != 0 && t == 0);
        if (n == 0 || n < a.ci || n > a.cx) continue;           // kmc -ci / -cx act on the true count
        const unsigned long long v = n > a.cs ? a.cs : n;       // kmc -cs: reported count saturates
        const uint32_t id = (uint32_t)(cidx >> 1), isrc = (uint32_t)cidx & 1u;
        const uint64_t c = ix.kmer_of[id];
        const uint32_t s = ix.slot_of[(size_t)id * W + t];
        const uint32_t off = ix.ent_off[s], cnt = ix.ent_len[s];
        for (uint32_t q = 0; q < cnt; ++q) vote(a, ix.entries[off + q], c, isrc, k, v);
        if (t == 0) {
            ++kept;
            uint32_t n_perfect = 0, perfect_file = 0;
            for (uint32_t q = ix.estat_off[id]; q < ix.estat_off[id + 1]; ++q) {   // (file << 1) | perfect
                const uint32_t e = ix.estat[q];
                atomicAdd(&lstats[(e >> 1) * 3 + ((e & 1u) ? 0 : 1)], 1u);
                if (e & 1u) { ++n_perfect; perfect_file = e >> 1; }
            }
            if (n_perfect == 1) atomicAdd(&lstats[perfect_file * 3 + 2], 1u);
        }
    }
    __syncthreads();
    finalize_epilogue(a, lstats, kept, distinct, lstats + ix.n_files * 3, a.row_exact + (int)blockIdx.x);
}

// K2b: one wave per workgroup and per k-mer, for the V counters K2a deferred (k-mers that touch several window
// buckets; rare).  Lane t probes the k-mer's t-th window bucket and votes once
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
    const uint64_t n_e = e_plane_len(ix.n_u);
    const uint64_t n_items = *a.n_deferred;   // the E counters are mapped by K2e

    // deferred items all passed the thresholds in K2a; one item per wave at a time, dealt round-robin so that a few
    // thousand items spread over the whole grid
    for (uint64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        {
            const uint64_t ci = n_e + a.deferred[item];
            unsigned long long v = a.counters[ci];
            v = v > a.cs ? a.cs : v;                              // kmc -cs: reported count saturates
            uint64_t c;
            uint32_t isrc;
            if (ci < n_e) {                                       // E: a reference k-mer itself
                c = ix.kmer_of[ci >> 1];
                isrc = (uint32_t)ci & 1u;
            } else {                                              // V: reference k-mer p with base bb at window position t
                const uint64_t vi = ci - n_e;
                isrc = (uint32_t)vi & 1u;
                const uint32_t bb = (uint32_t)(vi >> 1) & 3u;
                uint32_t t, p;
                row_owner(ix, vi >> 3, p, t);
                const int sh = 2 * (k - 1 - (ix.wstart + (int)t));
                c = (ix.kmer_of[p] & ~(3ull << sh)) | ((uint64_t)bb << sh);
            }

            if (lane < ix.W) {
                int s;
                if (ci < n_e) {
                    s = (int)ix.slot_of[(size_t)(ci >> 1) * ix.W + lane];   // a reference k-mer owns all its buckets
                } else {
                    const int sh = 2 * (k - 1 - (ix.wstart + lane));
                    s = probe_table(ix.table + (size_t)lane * S, ix.log2s, c & ~(3ull << sh));
                }
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
    if (lane == 0) { ntouched[0] = 0; ntouched[1] = 0; }
    __syncthreads();
    finalize_epilogue(a, lstats, 0u, 0u, ntouched, a.row_general + (int)blockIdx.x);
}

size_t finalize_lds_bytes(int n_files) { return ((size_t)n_files * 5 + 4) * sizeof(uint32_t); }
constexpr unsigned kFinVariantBlocks = 256 * 8, kFinExactBlocks = 256 * 8, kFinGeneralBlocks = 256 * 4;
size_t finalize_partial_rows() { return (size_t)kFinVariantBlocks + kFinExactBlocks + kFinGeneralBlocks; }

void launch_finalize(const FinalizeArgs& a0, hipStream_t stream) {
    if (a0.ix.W <= 0) return;
    FinalizeArgs a = a0;
    const size_t lds_stats = ((size_t)a.ix.n_files * 3 + 2) * sizeof(uint32_t);
    // K2a
    const uint64_t n_v = v_plane_len(a.ix.n_rows);
    const unsigned b_var = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_v + 255) / 256, kFinVariantBlocks));
    hipLaunchKernelGGL(finalize_variant_kernel, dim3(b_var), dim3(256), lds_stats, stream, a);
    // K2e
    const uint64_t n_work = e_plane_len(a.ix.n_u) * (uint64_t)a.ix.W;
    const unsigned b_ex = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_work + 255) / 256, kFinExactBlocks));
    a.row_exact = (int)b_var;
    hipLaunchKernelGGL(finalize_exact_kernel, dim3(b_ex), dim3(256), lds_stats, stream, a);
    // K2b (deferred k-mers only; the kernel reads their number on the device)
    const size_t lds = finalize_lds_bytes(a.ix.n_files);
    unsigned b_gen = (unsigned)std::min<size_t>(kFinGeneralBlocks, 256 * std::max<size_t>(1, (160u * 1024u) / lds));
    a.row_general = (int)(b_var + b_ex);
    hipLaunchKernelGGL(finalize_general_kernel, dim3(b_gen), dim3(64), lds, stream, a);
    if (a.partials) {
        const int cols = a.ix.n_files * 3 + 2;
        hipLaunchKernelGGL(finalize_reduce_kernel, dim3((unsigned)cols), dim3(256), 0, stream, a, (int)(b_var + b_ex + b_gen));
    }
}

}  // namespace bk
