// bk_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the k-mer -> pileup engine.
//
// K0  pack_reads       : ASCII sequence lines -> 2-bit fixed-stride records (KMC's read handling: split at non-ACGT, ...).
// K1  scan_count       : records -> for every k-mer, which distinct index-touching k-mer is it? -> its occurrence counter.
//                        Replaces the external KMC3 run of call.rs:1166-1211 for every k-mer that can touch the index.
//                        Counts are recorded per RUN of consecutive k-mers (difference arrays, see K1's comment): a
//                        word-parallel comparison of the read with the reference along its diagonal finds the runs; what
//                        it cannot settle is marked in a per-record bitmap.
// K1b level2           : the marked k-mers, one by one (rolling canonical k-mer, membership test, neighbour search).
// K1c fold             : per-workgroup per-cell counts (slabs) -> E counters of the u64 plane.
// K2a finalize_variant : V rows -> per-k-mer counts (prefix sums) -> KMC thresholds -> map_kmers vote.
// K2e finalize_exact   : E counters -> thresholds -> map_kmers vote, one thread per (reference k-mer, bucket).
// K2b finalize_general : the k-mers K2a defers (several buckets) -> map_kmers vote, one wave per k-mer.
//                        K2a/K2e/K2b together are call.rs:1286-1418 applied to KMC's kept k-mers (-ci/-cs/-cx).
//
// Counter naming (bk_device.h): a read k-mer equal to a reference k-mer u owns E[2*id(u) + rc]; a read k-mer at Hamming
// distance 1 from reference k-mers, differing at a window position, owns the V counter of the smallest (position,
// neighbour).  Both are functions of the k-mer alone, so every occurrence of a k-mer lands on the same counter and no k-mer
// owns two; a k-mer that touches no window bucket is not counted at all (map_kmers would ignore it: call.rs:1307).
// finalize re-derives the k-mer from the counter's coordinates and replays map_kmers on it with its exact count.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

#include "bk_device.h"
#include "bk_kernels.h"
#include "bk_scan_common.h"
#include "bk_finalize_common.h"

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


__device__ __forceinline__ HalfDir half_lookup(const HalfView& hv, uint64_t half) {
    const uint32_t pilot = hv.pilots[phf_bucket(half, hv.log2nb)];
    const uint4 e = *reinterpret_cast<const uint4*>(hv.dir + phf_pos(half, pilot, hv.m, hv.log2nb, hv.log2p));
    HalfDir d;
    d.key = e.x; d.off = e.y; d.cnt = (e.x == (uint32_t)half) ? e.z : 0u; d.pad = 0u;
    return d;
}

// Reference k-mers at Hamming distance exactly 1 from c whose differing position lies in the window.
// Calls f(j, p, valid) for each (pigeonhole: such a k-mer shares c's low half or c's high half).
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
        if (j >= wlo && j < whi && ((e.w >> (j - wlo)) & 1u)) f(j, e.z, e.w);
    }
    for (uint32_t i = 0; i < dh.cnt; ++i) {
        const uint4 e = *reinterpret_cast<const uint4*>(ix.hi.cand + dh.off + i);
        const int j = single_diff_pos((uint64_t)e.x | ((uint64_t)e.y << 32), c, k);
        if (j >= wlo && j < whi && ((e.w >> (j - wlo)) & 1u)) f(j, e.z, e.w);
    }
}

// ------------------------------------------------------------------------------------------------ K0
// ASCII reads -> fixed-stride 2-bit records (the host-side twin is bk_pack_reads): split at every non-ACGT/acgt symbol (KMC
// contract), drop runs shorter than k, cut runs longer than the stride into chunks overlapping by k-1 bases.
//
// pack_words_kernel (round 4)  one THREAD PER OUTPUT WORD.  A read that is one clean run that fits a record -- nearly every read --
//   goes to the record slot of its own index, so word q of read r is 16 letters at a known place: the block's sequence lines are
//   staged in LDS with coalesced 16-byte loads, every thread realigns its 16 letters (v_alignbyte), turns them into codes four at a
//   time (the 2-bit code of a letter is ((c >> 1) ^ (c >> 2)) & 3 in either case; the letters the codes stand for come back from
//   one v_perm_b32 and are compared with the input, case folded: "all four are ACGT" in three instructions) and stores one word --
//   consecutive threads, consecutive words: the 40 MB of records leave as whole cache lines.  (Round 3's kernel took a thread per
//   READ: 38 dependent steps of four letters each and ten word stores a record apart per lane -- 0.15 ms per million reads, the
//   longest kernel of the K0..K2 chain.)  A read with anything else in it (N, too short a run, longer than a record) leaves its slot
//   empty (length 0) and is taken byte by byte by the thread that holds its first word (pack_read_slow): its runs are appended
//   behind the n_reads slots through a device counter.  Records are unordered anyway.
// pack_slow_kernel   long reads (records of more than kPackMaxWords words): one thread per read, byte by byte, as before.
__device__ __forceinline__ int acgt_code(unsigned char c) {
    switch (c | 0x20) {
        case 'a': return 0;
        case 'c': return 1;
        case 'g': return 2;
        case 't': return 3;
        default: return -1;
    }
}

// One read, byte by byte: its runs of letters as records appended behind the n_reads slots (a.n_records[0]); returns how many.
__device__ __forceinline__ unsigned long long pack_read_slow(const PackArgs& a, uint64_t r) {
    const uint64_t maxb = min((uint64_t)a.stride_words * 16, (uint64_t)65535);
    const uint64_t o0 = a.offsets[r], len = a.offsets[r + 1] - o0;
    const uint8_t* s = a.bases + a.shift + o0;
    unsigned long long real = 0;
    uint64_t start = 0;
    for (uint64_t p = 0; p <= len; ++p) {
        if (p < len && acgt_code(s[p]) >= 0) continue;
        const uint64_t run = p - start;            // maximal ACGT run [start, p)
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
                    ++real;
                }
                if (pos + take >= run) break;
                pos += take - (uint64_t)(a.k - 1);
            }
        }
        start = p + 1;
    }
    return real;
}

constexpr int kPackBlock = 256;
constexpr uint32_t kPackMaxWords = 16;     // records of up to 256 bases take the word-per-thread kernel (a block holds at least 32 reads)
constexpr uint32_t kPackWpt = 2;           // output words (32 letters) per thread: the per-thread set-up is half of what a thread issues
constexpr uint32_t kPackLdsBytes = kPackBlock * kPackWpt * 16u + 96u;   // the block's lines + alignment slack

// 16 letters (four realigned words) -> 16 codes; ok: every one of the first nb is a letter.  What lies behind the nb letters is
// turned into codes and judged like the rest where it shares four bytes with them, and cut off afterwards: a read flagged for a
// symbol that is not its own only takes the byte-by-byte kernel, which looks at exactly its letters.
__device__ __forceinline__ uint32_t pack16(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t nb, bool& ok) {
    uint32_t out = 0u;
    const uint32_t x4[4] = {x0, x1, x2, x3};
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        const uint32_t x = x4[j];
        const uint32_t c = ((x >> 1) ^ (x >> 2)) & 0x03030303u;                       // four codes, one per byte
        ok &= (x & 0xdfdfdfdfu) == __builtin_amdgcn_perm(0x54474341u, 0x54474341u, c) || nb <= 4u * j;   // the letters A C G T the codes stand for
        const uint32_t c2 = c | (c >> 6);
        out |= ((c2 | (c2 >> 12)) & 0xffu) << (8u * j);
    }
    return nb < 16u ? out & ((1u << (2u * nb)) - 1u) : out;
}

__global__ __launch_bounds__(kPackBlock) void pack_words_kernel(PackArgs a, uint32_t rpb /* reads per block */, uint32_t tpr /* threads per read =
                                                                ceil(stride_words / kPackWpt) */, uint32_t tpr_recip /* ceil(2^32 / tpr) */) {
    __shared__ __attribute__((aligned(16))) unsigned char lines[kPackLdsBytes];
    __shared__ unsigned long long off_s[kPackBlock + 1];       // offsets of the block's reads (rpb + 1 of them)
    __shared__ unsigned int bad_s[kPackBlock];                 // per read of the block: some word met a symbol that is not a letter
    const uint32_t sw = a.stride_words;
    const uint64_t r0 = (uint64_t)blockIdx.x * rpb, r1 = min(r0 + (uint64_t)rpb, a.n_reads);
    const uint32_t nr = (uint32_t)(r1 - r0);
    // nr + 1 offsets: with records of one or two words a full block holds 256 reads, one offset more than it has threads
    for (uint32_t i = threadIdx.x; i <= nr; i += kPackBlock) off_s[i] = a.offsets[r0 + i] + a.shift;
    if (threadIdx.x < rpb) bad_s[threadIdx.x] = 0u;
    __syncthreads();
    const uint64_t b0 = off_s[0], b1 = off_s[nr], end = a.offsets[a.n_reads] + a.shift;
    const uint64_t a0 = b0 & ~15ull;                       // the block's lines from a 16-byte boundary (a.bases is 16-byte aligned)
    const uint64_t span = b1 - a0;                         // bytes of the block's lines from there
    const bool fits = span + 48u <= kPackLdsBytes;         // (reads no longer than their records: always; guards a malformed batch)
    if (fits) {
        // whole 16-byte units that lie inside the batch (two more than the lines take: the last read's last words are judged with
        // what follows them), then the batch's last few bytes one by one
        const uint64_t n16 = (span + 15) / 16 + 2, full16 = (end - a0) / 16;
        for (uint64_t i = threadIdx.x; i < min(n16, full16); i += kPackBlock) reinterpret_cast<uint4*>(lines)[i] = reinterpret_cast<const uint4*>(a.bases + a0)[i];
        if (n16 > full16) for (uint64_t i = full16 * 16 + threadIdx.x; i < min(end - a0, n16 * 16); i += kPackBlock) lines[i] = a.bases[a0 + i];
    }
    __syncthreads();
    const uint32_t t = threadIdx.x;
    const uint32_t rl = tpr == 1u ? t : __umulhi(t, tpr_recip), q0 = (t - rl * tpr) * kPackWpt;   // this thread's read (within the block) and first word
    const bool mine = rl < nr;
    const uint64_t o0 = mine ? off_s[rl] : b0, len64 = mine ? off_s[rl + 1] - o0 : 0ull;
    const uint64_t maxb = min((uint64_t)sw * 16, (uint64_t)65535);
    const bool simple = mine && fits && len64 >= (uint64_t)a.k && len64 <= maxb;   // a candidate for the slot of its own index
    const uint32_t len = simple ? (uint32_t)len64 : 0u;
    uint32_t out[kPackWpt];
    bool ok = true;
    {
        const uint32_t p = (uint32_t)(o0 - a0) + 16u * q0;  // where the thread's letters start in the staged lines
        const uint32_t* wsrc = reinterpret_cast<const uint32_t*>(lines) + ((simple ? p : 0u) >> 2);
        const uint32_t sh = p & 3u;
        uint32_t w[4 * kPackWpt + 1];
#pragma unroll
        for (uint32_t i = 0; i <= 4u * kPackWpt; ++i) w[i] = wsrc[i];   // (whatever lies behind the letters is cut off below)
#pragma unroll
        for (uint32_t u = 0; u < kPackWpt; ++u) {
            const uint32_t at = 16u * (q0 + u);
            const uint32_t nb = len > at ? min(len - at, 16u) : 0u;   // letters of this word
            out[u] = pack16(__builtin_amdgcn_alignbyte(w[4 * u + 1], w[4 * u], sh), __builtin_amdgcn_alignbyte(w[4 * u + 2], w[4 * u + 1], sh),
                            __builtin_amdgcn_alignbyte(w[4 * u + 3], w[4 * u + 2], sh), __builtin_amdgcn_alignbyte(w[4 * u + 4], w[4 * u + 3], sh), nb, ok);
            if (nb == 0u) out[u] = 0u;
        }
    }
    if (mine && !ok) atomicOr(&bad_s[rl], 1u);
    if (mine) {
        uint32_t* dst = a.words + (r0 + rl) * sw + q0;     // (a slot that stays empty holds whatever: its length says 0)
#pragma unroll
        for (uint32_t u = 0; u < kPackWpt; ++u) if (q0 + u < sw) dst[u] = out[u];
    }
    __syncthreads();
    if (mine && q0 == 0u) {
        const bool good = simple && !bad_s[rl];
        a.lens[r0 + rl] = good ? (uint16_t)len : (uint16_t)0;
        if (!good) {
            // (the launcher counted every read as a record that holds a run: one atomic per read that is not, none per block -- a
            // tally per block was 40,000 additions to one address, 0.5 ms per million reads)
            // anything that may still hold a run of k letters: byte by byte, by this thread, here (a kernel of its own behind this
            // one -- a work list, a launch whose grid found it empty on the benchmark -- was 4.6 us of K0's 49)
            const unsigned long long real = len64 >= (uint64_t)a.k ? pack_read_slow(a, r0 + rl) : 0ull;
            atomicAdd(a.n_real, real - 1ull);
        }
    }
}

// One thread per read (long reads: records of more than kPackMaxWords words): the slot of its index stays empty.
__global__ __launch_bounds__(kPackBlock) void pack_slow_kernel(PackArgs a) {
    for (uint64_t r = (uint64_t)blockIdx.x * kPackBlock + threadIdx.x; r < a.n_reads; r += (uint64_t)gridDim.x * kPackBlock) {
        a.lens[r] = 0;   // its records, if any, go behind the n_reads slots
        const unsigned long long real = pack_read_slow(a, r);
        if (real) atomicAdd(a.n_real, real);
    }
}

__global__ void add_u64_kernel(unsigned long long* dst, const unsigned long long* src) { *dst += *src; }
__global__ void add_const_u64_kernel(unsigned long long* dst, unsigned long long v) { *dst += v; }
void launch_add_const_u64(unsigned long long* dst, unsigned long long v, hipStream_t stream) {
    hipLaunchKernelGGL(add_const_u64_kernel, dim3(1), dim3(1), 0, stream, dst, v);
}
void launch_add_u64(unsigned long long* dst, const unsigned long long* src, hipStream_t stream) {
    hipLaunchKernelGGL(add_u64_kernel, dim3(1), dim3(1), 0, stream, dst, src);
}

// K0's counters before its kernel: dst[0] = record slots in use when it ends (the n_reads slots of the reads' own indices + what is
// appended), dst[1..2] unused, and the sample's tally of records that hold a run += v1 (the packer corrects it read by read)
__global__ void pack_begin_kernel(unsigned long long* dst, unsigned long long v0, unsigned long long* tally, unsigned long long v1) { dst[0] = v0; dst[1] = 0ull; dst[2] = 0ull; *tally += v1; }
// a.n_records[0] = record slots in use when the kernels end; `tally` (the sample's count of records that hold a run -- what KMC would
// call its input sequences) grows by this batch's: a launch of its own that added a batch-local count to it afterwards was 4.5 us
void launch_pack_reads(const PackArgs& a0, unsigned long long* tally, hipStream_t stream) {
    if (a0.n_reads == 0) return;
    PackArgs a = a0;
    a.n_real = tally;
    const bool by_word = a.stride_words <= kPackMaxWords;
    // (pack_words_kernel starts from "every read is a record that holds a run" and takes the others off)
    hipLaunchKernelGGL(pack_begin_kernel, dim3(1), dim3(1), 0, stream, a.n_records, (unsigned long long)a.n_reads, tally, by_word ? (unsigned long long)a.n_reads : 0ull);
    if (by_word) {
        const uint32_t tpr = (a.stride_words + kPackWpt - 1) / kPackWpt, rpb = (uint32_t)kPackBlock / tpr;
        const uint32_t recip = (uint32_t)(((1ull << 32) + tpr - 1) / tpr);
        hipLaunchKernelGGL(pack_words_kernel, dim3((unsigned)((a.n_reads + rpb - 1) / rpb)), dim3(kPackBlock), 0, stream, a, rpb, tpr, recip);
    } else {
        a.work = nullptr;   // long reads: every read byte by byte
        hipLaunchKernelGGL(pack_slow_kernel, dim3((unsigned)std::min<uint64_t>((a.n_reads + kPackBlock - 1) / kPackBlock, 65535)), dim3(kPackBlock), 0, stream, a);
    }
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

constexpr uint32_t kKtabFillWords = 1024;   // new-key tallies behind the overflow flag: overflow[4 + i], spread so that no word is hot
__device__ __forceinline__ void ktab_insert_key(const KmerTable& t, unsigned long long key, unsigned int n) {
    const uint64_t mask = (1ull << t.log2n) - 1ull;
    uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> (64 - t.log2n);
    for (uint32_t probes = 0; probes < 4096; ++probes) {
        const unsigned long long old = atomicCAS(t.keys + h, ~0ull, key);
        if (old == ~0ull || old == key) {
            if (t.cnt[h] < 0xf0000000u) atomicAdd(t.cnt + h, n);   // saturates far above any -cx
            if (old == ~0ull) atomicAdd(t.overflow + 4 + (h & (kKtabFillWords - 1u)), 1ull);   // the engine grows the table by its fill
            return;
        }
        h = (h + 1) & mask;
    }
    *t.overflow = 1ull;   // (the engine keeps the load below one half: unreachable unless the table cannot grow any more)
}
__device__ __forceinline__ void ktab_insert(const KmerTable& t, uint64_t c, uint32_t isrc, unsigned int n) {
    ktab_insert_key(t, c | ((unsigned long long)isrc << 63) | ((unsigned long long)t.mate << 62), n);
}

// ---- the statistics tables of ranks that shared one sample's reads (full_kmer_stats with a sharded finalize) ----------------
// A k-mer that touches no window bucket sits in the table of every rank whose reads held it; the sample's "unique k-mers" /
// "unique counted k-mers" (call.rs:1190-1199) need each k-mer once with its total count.  Every key has one owner rank (a hash
// of the key); a rank lists its entries grouped by owner (count, then scatter: one append per wave and owner), the host
// exchanges the groups (all-to-all), and the rank rebuilds its table from what it received -- equal keys add up.
__device__ __forceinline__ uint32_t ktab_owner(unsigned long long key, uint32_t n_parts) {
    return (uint32_t)((((key * 0xD6E8FEB86659FD93ull) >> 32) * (unsigned long long)n_parts) >> 32);
}
__global__ __launch_bounds__(256) void ktab_count_parts_kernel(const unsigned long long* __restrict__ keys, uint64_t n, uint32_t n_parts,
                                                               unsigned long long* counts /* [n_parts], zeroed */) {
    __shared__ unsigned int h[kMaxShards];
    for (uint32_t i = threadIdx.x; i < n_parts; i += 256) h[i] = 0u;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const unsigned long long key = keys[i];
        if (key != ~0ull) atomicAdd(&h[ktab_owner(key, n_parts)], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_parts; i += 256) if (h[i]) atomicAdd(counts + i, (unsigned long long)h[i]);
}
__global__ __launch_bounds__(256) void ktab_scatter_parts_kernel(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ cnt,
                                                                 uint64_t n, uint32_t n_parts, unsigned long long* cursors /* [n_parts]: first free slot */,
                                                                 unsigned long long* out_keys, unsigned int* out_cnt) {
    const uint64_t n_round = (n + 255) / 256 * 256;   // (whole waves stay in the loop: the appends are per wave)
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_round; i += (uint64_t)gridDim.x * 256) {
        const unsigned long long key = i < n ? keys[i] : ~0ull;
        const bool live = key != ~0ull;
        const uint32_t own = live ? ktab_owner(key, n_parts) : 0u;
        unsigned long long todo = __ballot(live);
        while (todo) {
            const uint32_t p = (uint32_t)__shfl((int)own, __builtin_ctzll(todo));
            const unsigned long long m = __ballot(live && own == p);
            unsigned long long base = 0;
            if ((int)(threadIdx.x & 63u) == __builtin_ctzll(m)) base = atomicAdd(cursors + p, (unsigned long long)__popcll(m));
            base = __shfl(base, __builtin_ctzll(m));
            if (live && own == p) { const uint64_t at = base + lane_prefix(m); out_keys[at] = key; out_cnt[at] = cnt[i]; }
            todo &= ~m;
        }
    }
}
__global__ __launch_bounds__(256) void ktab_import_kernel(const unsigned long long* __restrict__ in_keys, const unsigned int* __restrict__ in_cnt, uint64_t n,
                                                          KmerTable t) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) ktab_insert_key(t, in_keys[i], in_cnt[i]);
}
void launch_ktab_count_parts(const unsigned long long* keys, uint32_t log2n, uint32_t n_parts, unsigned long long* counts, hipStream_t stream) {
    const uint64_t n = 1ull << log2n;
    hipLaunchKernelGGL(ktab_count_parts_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 8)), dim3(256), 0, stream, keys, n, n_parts, counts);
}
void launch_ktab_scatter_parts(const unsigned long long* keys, const unsigned int* cnt, uint32_t log2n, uint32_t n_parts, unsigned long long* cursors,
                               unsigned long long* out_keys, unsigned int* out_cnt, hipStream_t stream) {
    const uint64_t n = 1ull << log2n;
    hipLaunchKernelGGL(ktab_scatter_parts_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 8)), dim3(256), 0, stream, keys, cnt, n, n_parts,
                       cursors, out_keys, out_cnt);
}
void launch_ktab_import(const unsigned long long* in_keys, const unsigned int* in_cnt, uint64_t n, unsigned long long* keys, unsigned int* cnt, uint32_t log2n,
                        unsigned long long* overflow, hipStream_t stream) {
    if (!n) return;
    const KmerTable t{keys, cnt, log2n, overflow, 0u};
    hipLaunchKernelGGL(ktab_import_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 8)), dim3(256), 0, stream, in_keys, in_cnt, n, t);
}
// sharded finalize: the table's totals join the device tallies that the ranks add up (kstats[mate][2], [3]); an overflowed table
// adds 2^56 to both -- no sum of real tallies reaches it, bk_sample_download reports "unavailable" on every rank
__global__ void ktab_totals_to_kstats_kernel(unsigned long long* ktab_out, unsigned long long* kstats, int n_mates) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int m = 0; m < n_mates; m++) {
        kstats[m * 4 + 2] += ktab_out[m * 2 + 0] + (ktab_out[4] ? 1ull << 56 : 0ull);
        kstats[m * 4 + 3] += ktab_out[m * 2 + 1] + (ktab_out[4] ? 1ull << 56 : 0ull);
        ktab_out[m * 2 + 0] = 0ull; ktab_out[m * 2 + 1] = 0ull;
    }
}
void launch_ktab_totals_to_kstats(unsigned long long* ktab_out, unsigned long long* kstats, int n_mates, hipStream_t stream) {
    hipLaunchKernelGGL(ktab_totals_to_kstats_kernel, dim3(1), dim3(64), 0, stream, ktab_out, kstats, n_mates);
}

// old table -> a larger one (same keys, same counts; the fill tallies stay as they are)
__global__ __launch_bounds__(256) void ktab_rehash_kernel(const unsigned long long* __restrict__ okeys, const unsigned int* __restrict__ ocnt, uint64_t on,
                                                          unsigned long long* nkeys, unsigned int* ncnt, uint32_t nlog2, unsigned long long* overflow) {
    const uint64_t mask = (1ull << nlog2) - 1ull;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < on; i += (uint64_t)gridDim.x * 256) {
        const unsigned long long key = okeys[i];
        if (key == ~0ull) continue;
        uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> (64 - nlog2);
        uint32_t probes = 0;
        for (; probes < 65536; ++probes) {
            const unsigned long long old = atomicCAS(nkeys + h, ~0ull, key);
            if (old == ~0ull) { ncnt[h] = ocnt[i]; break; }   // keys of the old table are distinct
            h = (h + 1) & mask;
        }
        if (probes == 65536) *overflow = 1ull;
    }
}
void launch_ktab_rehash(const unsigned long long* okeys, const unsigned int* ocnt, uint32_t olog2, unsigned long long* nkeys, unsigned int* ncnt,
                        uint32_t nlog2, unsigned long long* overflow, hipStream_t stream) {
    const uint64_t on = 1ull << olog2;
    hipLaunchKernelGGL(ktab_rehash_kernel, dim3((unsigned)std::min<uint64_t>((on + 255) / 256, 256 * 16)), dim3(256), 0, stream, okeys, ocnt, on, nkeys, ncnt, nlog2, overflow);
}
uint32_t ktab_fill_words() { return kKtabFillWords; }

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
// Persistent workgroups of 16 waves (one per CU: the LDS arrays take most of the CU's 160 KB); each wave takes
// tiles of 64 records, one record per lane.  Nothing is counted k-mer by k-mer unless it has to be.
//
// Diagonal.  A few seed k-mers of the read (evenly spaced, looked up in the perfect hash of U; a second round at the
// midpoints for the rare read whose seeds all carry an error) give its diagonal: read k-mer s <-> reference cell dg + s
// (same strand) or dg - s (opposite strand).
// Mismatches.  160 bases at a time (a whole 150 bp read), the read words are XORed with the reference words aligned to
// them (packed reference in LDS, funnel shifts; reversed and complemented for the opposite strand) and folded to one
// mismatch flag per base.  Everything else follows from the flags, mismatch by mismatch (a loop of as many passes as the
// lane with the most mismatches needs -- three or four per tile at 0.5 % errors):
//   E run   the k-mers between two mismatches (more than k apart) hold none: they ARE the reference k-mers of their cells.
//           Cells c0..c1 each seen once more in this read's direction: +1 at c0, -1 at c1 + 1 in a per-cell DIFFERENCE array
//           in LDS (low half-word: reads along the reference, high half-word: against it).  The workgroup turns it into
//           per-cell counts with one prefix sum at the end (epilogue) and writes them as a slab; fold adds slab cell c into
//           E[id_at[c]] (orientation = the cell's, flipped for the high half).
//   S run   a mismatch with no other within k - 1 bases on either side: the up to k k-mers that hold it hold nothing else.
//           If all their cells are "fast" (IndexView::cell_fast: clean, ids = cell + one constant) they are one row of the V
//           plane (bk_device.h): +1 at the first offset, -1 after the last, whatever the run's length -- two global atomics,
//           no table is read.
//   N run   everything else -- mismatches in reach of each other, cells that are not fast, reads off the LDS window or
//           without a diagonal: the run's k-mers are marked in n_bits[record] (one bit per k-mer, one bit per record in n_any)
//           and resolved by level2_kernel, which takes them run by run first and k-mer by k-mer for what that leaves.
// Until round 3 this kernel also queued the N runs in LDS and resolved them itself, 64 at a time ("N batch"), and told E
// from N k-mer by k-mer (a k-step shift-or over the flags, then one pass per run boundary): 2400 VALU instructions per tile
// and a dependent chain of global loads per batch against 1000 now; the batch code lives on in level2_kernel, where it
// sees a tenth of the runs.
//
// LDS difference array: one 32-bit word per cell, value = (runs starting - runs ending) of reads along the
// reference + 65536 * the same for reads against it, modulo 2^32.  The prefix sum S_c = F_c + 65536 R_c is exact
// as long as both counts stay below 65536: a launch gives a workgroup at most kMaxRecordsPerGroup records and a
// record covers a cell at most once on its diagonal.
//
// The LDS arrays cover a window of cells [win_lo, win_lo + n_lds_bins): everything for one genome of SARS-CoV-2 size;
// for a multi-genome index the engine puts it on the genome the sample looks like (pick_window_kernel) and the seeds
// land on that genome's copy of a k-mer (occ).  Exact hits outside the window (other genomes' copies) are found by
// Level 2's membership test and counted in the u64 plane directly.  A reference too large for LDS is read from
// global memory instead (REF_LDS = false).
constexpr int kScanBlock = 1024;
constexpr int kScanWaves = kScanBlock / 64;
constexpr int kNIters = 8;                  // mismatches of a piece the N batch (level2_kernel) resolves in one pass
constexpr uint32_t kMaxRecordsPerGroup = 16384;
// A sample's true variants put thousands of reads on the same few V counters -- every read that covers a fixed SNP adds 1 to one
// and the same counter --, and same-address global atomics serialise at ~12 ns each: 0.02 ms of the kernel on the benchmark's 40
// variant sites.  So a workgroup keeps the counters it meets a second time in a small LDS table (direct-mapped; a filter of one
// bit per hash value says "met before") and adds its totals once at the end; everything else goes out as before.
constexpr uint32_t kHotSlots = 256, kSeenWords = 1024, kBlkTouchWords = 32;   // (32 words: 1024 blocks of 64 cells, more than an LDS window holds)
constexpr size_t kScanLdsFixed = 16 + 64 + 4 * kScanBlock + 4 * (2 * kHotSlots + kSeenWords + kBlkTouchWords) + 8;   // k-mer tally, wave totals of the epilogue, the item owners, the hot counters, alignment of the block entries


// +1 on "reference k-mer id with base b (forward strand of the reference) at offset o, read in direction d": a
// single-k-mer run of its V row
// sparse finalize (ScanArgs::touch_v): set bit `i` of a touch bitmap (null: dense finalize, nothing to note)
__device__ __forceinline__ void touch(unsigned int* bm, uint32_t i) {
    // (looked at first: most rows are touched many times a sample and a stale 0 only costs the atomic it would have cost anyway)
    // (an agent-scope load: from L2, where the atomics land; a non-temporal one went to memory every time)
    if (bm && !(__hip_atomic_load(bm + (i >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (i & 31u) & 1u)) atomicOr(bm + (i >> 5), 1u << (i & 31u));
}

__device__ __forceinline__ void v_point(unsigned long long* __restrict__ v_counters, uint32_t id, uint32_t o, uint32_t alt, uint32_t d,
                                        int omin, int span, unsigned int* touch_v = nullptr) {
    const uint32_t oo = o - (uint32_t)omin;
    if (oo >= (uint32_t)span) return;   // offsets outside the layout touch no window bucket
    touch(touch_v, v_row_index(id + oo, alt, d));
    unsigned long long* row = v_counters + v_row_base(id + oo, alt, d, span) + oo;
    atomicAdd(row, 1ull);
    if (oo + 1u < (uint32_t)span) atomicAdd(row + 1, ~0ull);   // (slot `span` is never read)
}

// The slow path as a software pipeline.  A batch of up to 64 queued k-mers (one per lane) advances one stage per
// k-mer step of Level 2: stage 1 has the three pilots in flight (reference k-mer set, low half, high half),
// stage 2 the perfect-hash entry and the two directory entries, stage 3 the first candidate of each list (unless
// the k-mer turned out to be a reference k-mer: then it is counted and done); stage 4 resolves the V counter
// and issues the atomics.  Each stage only *issues* its loads; they are consumed one step later.
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
    unsigned int* touch_v = nullptr; unsigned int* touch_p = nullptr;   // sparse finalize (ScanArgs::touch_v); the caller's
                                                                        // count_exact notes reference k-mers itself
#ifdef BK_TESTING
    unsigned long long* dbg = nullptr;   // BK_L2_STATS tallies
#endif

    // queue entry: canonical k-mer | orientation << 62 | stat_only << 63
    __device__ __forceinline__ void start(const unsigned long long* q, uint32_t n, int lane, const IndexView& ix) {
        have = (uint32_t)lane < n;
        const unsigned long long e = have ? q[lane] : 0ull;
        c = e & 0x3fffffffffffffffull;
        isrc = (uint32_t)(e >> 62) & 1u;
        stat_only = (e >> 63) != 0;
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
            b0 = *reinterpret_cast<const uint4*>(ix.kmer_pos + phf_pos(c, pu, ix.m, ix.log2nb, ix.log2p));
            b1 = *reinterpret_cast<const uint4*>(ix.lo.dir + phf_pos(lo, pl, ix.lo.m, ix.lo.log2nb, ix.lo.log2p));
            b2 = *reinterpret_cast<const uint4*>(ix.hi.dir + phf_pos(hi, ph, ix.hi.m, ix.hi.log2nb, ix.hi.log2p));
            stage = 2;
        } else if (stage == 2) {
            const bool member = have && !stat_only && ((uint64_t)b0.x | ((uint64_t)b0.y << 32)) == c;   // a reference k-mer after all
            count_exact(member, b0.z, b0.w & kIdMask, isrc, b0.w >> 31);
            if (member) have = false;
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
            uint64_t best = ~0ull;   // (j << 32) | NbEntry::p, smallest wins
            uint32_t bvalid = 0, bref = 0;   // ... its valid mask and its own base at j
            auto consider = [&](const uint4& e) {
                const uint64_t eu = (uint64_t)e.x | ((uint64_t)e.y << 32);
                const int j = single_diff_pos(eu, c, k);
                if (j >= wlo && j < whi && ((e.w >> (j - wlo)) & 1u)) {   // the neighbour owns a bucket at j
                    const uint64_t key = ((uint64_t)j << 32) | e.z;
                    if (key < best) { best = key; bvalid = e.w; bref = (uint32_t)(eu >> (2 * (k - 1 - j))) & 3u; }
                }
            };
            if (cnt_lo) consider(b0);
            if (cnt_hi) consider(b1);
            // the rest of both lists (one candidate each with a single genome; dozens with a hundred strains, where every k-mer of
            // a strain shares its halves with the other strains' variants): four loads in flight at a time, not one after the other
            const uint32_t r_lo = cnt_lo ? cnt_lo - 1u : 0u, r_hi = cnt_hi ? cnt_hi - 1u : 0u, tot = r_lo + r_hi;
            for (uint32_t i = 0; i < tot; i += 4u) {
                uint4 e[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const uint32_t ii = min(i + u, tot - 1u);
                    e[u] = ii < r_lo ? *reinterpret_cast<const uint4*>(ix.lo.cand + off_lo + 1u + ii) : *reinterpret_cast<const uint4*>(ix.hi.cand + off_hi + 1u + (ii - r_lo));
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) if (i + u < tot) consider(e[u]);
            }
            if (best != ~0ull) {
                const int j = (int)(best >> 32);
                const uint32_t p = (uint32_t)best;
                const uint32_t b = (uint32_t)(c >> (2 * (k - 1 - j))) & 3u;   // base of the canonical k-mer at j
                if (p < ix.n_full) {
                    // in reference coordinates: offset from the k-mer's start, base on the forward strand, read direction
                    const uint32_t rcu = bvalid >> 31;
                    v_point(v_counters, p, (uint32_t)(rcu ? k - 1 - j : j), v_alt(b, bref), isrc ^ rcu, ix.v_omin, ix.v_span, touch_v);
                } else {
                    const uint32_t row = p - ix.n_full + (uint32_t)__popc(bvalid & ((1u << (j - wlo)) - 1u));
                    touch(touch_p, row);
                    atomicAdd(v_counters + v_real_len(ix.n_full, ix.v_span) + ((uint64_t)row * 4 + b) * 2 + isrc, 1ull);
                }
            } else if (have && kt.keys) {
                ktab_insert(kt, c, isrc, 1u);   // touches no window bucket: only KMC's distinct / counted totals see it
            }
#ifdef BK_TESTING
            if (dbg && have && !stat_only) atomicAdd(dbg + (best != ~0ull ? 9 : 10), 1ull);
#endif
            stage = 0;
        }
    }

    template <typename CountExact>
    __device__ __forceinline__ void finish(const IndexView& ix, unsigned long long* __restrict__ v_counters, CountExact&& count_exact,
                                           const KmerTable& kt) {
        while (stage) advance(ix, v_counters, count_exact, kt);
    }
};

// KT: k as a compile-time constant for the common sizes, 0 = any k
// SPARSE: the V rows written are noted in ScanArgs::touch_v (sparse finalize of a large index)
template <bool REF_LDS, int KT, bool SPARSE>
__global__ __launch_bounds__(kScanBlock) void scan_count_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned int* block_kmers = reinterpret_cast<unsigned int*>(smem);   // 16 B reserved
    unsigned int* scan_tmp = block_kmers + 4;       // 16 words: wave totals of the epilogue's prefix sum
    unsigned int* own_all = scan_tmp + 16;          // [16 waves][64] which lane owns each of the items of a pass (below)
    unsigned int* hot_key = own_all + kScanBlock;   // [kHotSlots] V counters this workgroup adds to again and again (below), ~0 = free
    unsigned int* hot_cnt = hot_key + kHotSlots;    // [kHotSlots] ... and what it has for them
    unsigned int* seen = hot_cnt + kHotSlots;       // [kSeenWords] one bit per hash value: a +1 for such a counter was issued before
    unsigned int* blk_t = seen + kSeenWords;        // [kBlkTouchWords] sparse planes: the window's blocks of 64 cells whose V rows this workgroup counted into
    unsigned int* bins = blk_t + kBlkTouchWords;    // [n_lds_bins + 1] the per-cell difference array
    unsigned int* lds_ref = bins + a.n_lds_bins + 1;   // REF_LDS: padded ref words, the padded fast-bit array, the block entries

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform by construction: tell the compiler
    unsigned int* const own_s = own_all + wave * 64;

    const uint32_t total = a.total_cells;
    // the LDS window (a multiple of 64 cells from the start): chosen on the device for a multi-genome index
    // (choose_window_kernel), else the engine's constant.  The scan only ever settles reads whose cells all lie in it, so that is
    // all the LDS copies of the per-cell arrays hold: front pad, the window's cells, back pad
    const uint32_t win_lo = a.win_dev ? a.win_dev[1] : a.win_lo;
    const uint32_t win_file = a.win_dev ? a.win_dev[0] : (uint32_t)a.win_file;
    const uint32_t lds_cells = min(total - win_lo, a.n_lds_bins);
    const uint32_t n_refw = kRefPadWords + (lds_cells + 15) / 16 + kRefBackWords;
    const uint32_t n_bitw = kBitPadWords + (lds_cells + 31) / 32 + kBitBackWords;
    const uint32_t n_blk = (lds_cells + 63) / 64 + 2;
    const uint32_t blk_w0 = (a.n_lds_bins + 1 + n_refw + 2u * n_bitw + 1u) & ~1u;   // (8-byte aligned: bins starts at a multiple of 16 bytes)
    for (uint32_t i = threadIdx.x; i <= a.n_lds_bins; i += kScanBlock) bins[i] = 0u;
    for (uint32_t i = threadIdx.x; i < kHotSlots; i += kScanBlock) { hot_key[i] = 0xffffffffu; hot_cnt[i] = 0u; }
    for (uint32_t i = threadIdx.x; i < kSeenWords; i += kScanBlock) seen[i] = 0u;
    if (threadIdx.x < kBlkTouchWords) blk_t[threadIdx.x] = 0u;
    if (REF_LDS) {
        for (uint32_t i = threadIdx.x; i < n_refw; i += kScanBlock) lds_ref[i] = a.ref_words[(win_lo >> 4) + i];
        for (uint32_t i = threadIdx.x; i < n_bitw; i += kScanBlock) lds_ref[n_refw + i] = a.cell_fast[(win_lo >> 5) + i];
        for (uint32_t i = threadIdx.x; i < n_bitw; i += kScanBlock) lds_ref[n_refw + n_bitw + i] = a.cell_clean3[(win_lo >> 5) + i];
        for (uint32_t i = threadIdx.x; i < 2 * n_blk; i += kScanBlock) bins[blk_w0 + i] = reinterpret_cast<const uint32_t*>(a.cell_blk + (win_lo >> 6))[i];
    }
    __syncthreads();
    // symbol / bit 0 is cell win_lo; negative positions down to -64 are readable (padding or earlier cells)
    const unsigned int* refw1 = (REF_LDS ? lds_ref : a.ref_words + (win_lo >> 4)) + kRefPadWords;
    const unsigned int* fastw = (REF_LDS ? lds_ref + n_refw : a.cell_fast + (win_lo >> 5)) + kBitPadWords;
    const unsigned int* c3w = (REF_LDS ? lds_ref + n_refw + n_bitw : a.cell_clean3 + (win_lo >> 5)) + kBitPadWords;
    const uint2* blkw = REF_LDS ? reinterpret_cast<const uint2*>(bins + blk_w0) : a.cell_blk + (win_lo >> 6);   // entry 0 = the block of cell win_lo

    const int k = KT ? KT : a.k;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;  // k <= 31
    const uint32_t km1 = (uint32_t)k - 1u;
    const int omin = a.v_omin, span = a.v_span;
    unsigned long long* const v_counters = a.counters + a.v_off;
    const uint32_t last_word = a.stride_words - 1u;

    uint32_t nkm = 0;  // k-mer occurrences of this lane's records
    const IndexView& ix = *a.ixp;
    // k-mers [sk, sk + n) of record `rec` (index within this launch) are an N run left to level2_kernel: set their bits (the
    // record's diagonal goes with the mark: nothing is written for the records without one)
    auto n_mark = [&](bool on, uint32_t rec, uint32_t sk, uint32_t n, int32_t dgm, uint32_t flm) {
        if (!on) return;
        unsigned int* row = a.n_bits + (size_t)rec * a.l2_words;
        atomicOr(a.n_any + (rec >> 5), 1u << (rec & 31u));
        a.l2_diag[rec] = make_uint2((uint32_t)dgm, flm);
        uint32_t w = sk >> 5, bit = sk & 31u, left = n;
        while (left) {
            const uint32_t take = min(left, 32u - bit);
            atomicOr(row + w, (take == 32u ? 0xffffffffu : (1u << take) - 1u) << bit);
            left -= take; ++w; bit = 0u;
        }
    };

    // ... or, one by one, to its second pass (l2_bits): the k-mers that hold two mismatches and cannot be discarded
    auto l2_mark = [&](bool on, uint32_t rec, uint32_t sk, uint32_t n, int32_t dgm, uint32_t flm) {
        if (!on) return;
        unsigned int* row = a.l2_bits + (size_t)rec * a.l2_words;
        atomicOr(a.l2_any + (rec >> 5), 1u << (rec & 31u));
        a.l2_diag[rec] = make_uint2((uint32_t)dgm, flm);
        uint32_t w = sk >> 5, bit = sk & 31u, left = n;
        while (left) {
            const uint32_t take = min(left, 32u - bit);
            atomicOr(row + w, (take == 32u ? 0xffffffffu : (1u << take) - 1u) << bit);
            left -= take; ++w; bit = 0u;
        }
    };
    const bool stats = a.ktab_keys != nullptr;   // full_kmer_stats: k-mers that touch nothing are still wanted by the statistics table (level2_kernel)

    uint64_t n_records = a.n_records;
    if (a.n_records_dev) {
        const uint64_t nd = *a.n_records_dev;
        n_records = nd > a.rec_base ? min(nd - a.rec_base, a.n_records) : 0ull;
    }
    const uint32_t* const words0 = a.words + a.rec_base * a.stride_words;
    const uint16_t* const lens0 = a.lens + a.rec_base;
    const uint64_t n_tiles = (n_records + 63) / 64;
    uint32_t pf_sink = 0;               // destination of the prefetch loads (never read)
    constexpr uint32_t kNoPos = 0x40000000u;
    // the seed table of the window's genome, and where the seeds sit: evenly spaced over the launch's first record
    const uint2* const seed_tab = a.seed_tab ? a.seed_tab + ((size_t)win_file << a.seed_log2) : nullptr;
    const uint32_t hint_len = n_records ? (uint32_t)__builtin_amdgcn_readfirstlane((int)lens0[0]) : 0u;
    const uint32_t hint_span = hint_len >= (uint32_t)k ? hint_len - (uint32_t)k : 0u;

    for (uint64_t tile = (uint64_t)blockIdx.x * kScanWaves + wave; tile < n_tiles; tile += (uint64_t)gridDim.x * kScanWaves) {
        const uint64_t r = tile * 64 + lane;
        const bool live = r < n_records;
        const uint32_t r32 = live ? (uint32_t)r : 0u;   // record index within this launch (a launch has < 2^32 records)
        uint32_t len = live ? (uint32_t)lens0[r32] : 0u;
        if (len < (uint32_t)k) len = 0u;   // no k-mer
        uint32_t maxlen = len;
#pragma unroll
        for (int off = 32; off; off >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, off));
        maxlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxlen);
        const uint32_t* __restrict__ w = words0 + (uint64_t)r32 * a.stride_words;
        const uint32_t nk = len ? len - km1 : 0u;   // k-mers of the record
        nkm += nk;
        {   // touch the next tile's records (one lane per 128-byte line) so that its seed loads find them in cache
            const uint64_t nt = tile + (uint64_t)gridDim.x * kScanWaves;
            const uint64_t first = nt * 64ull * a.stride_words, words_tile = 64ull * a.stride_words;
            const uint64_t at = first + (uint64_t)lane * 32ull;
            if (nt < n_tiles && (uint64_t)lane * 32ull < words_tile && at < n_records * a.stride_words)
                asm volatile("global_load_dword %0, %1, off" : "=v"(pf_sink) : "v"(words0 + at) : "memory");
        }
        if (!maxlen) continue;

        // ---- seeds -> diagonal ----------------------------------------------------------------------------------
        bool fwd = true, seeded = false, l1ok = false;   // seeded: diagonal known; l1ok: ... and all its cells are in the LDS window
        int32_t dg = 0;                                  // cell of the reference k-mer aligned with read k-mer 0: k-mer s <-> dg + s (fwd) / dg - s
        // a candidate diagonal: the whole read must lie on the reference (hi_cell + k <= total); to be settled here its cells must lie
        // in the LDS window and each of them must carry a reference k-mer (no sequence tail in between: cell_blk)
        uint32_t best_cell = 0xffffffffu;
        auto candidate = [&](bool hit, uint32_t scell, bool f, uint32_t s) {
            if (hit && scell < best_cell) {   // several seeds may hit (usually all, on one diagonal); prefer the lowest cell
                const int64_t d0 = f ? (int64_t)scell - (int64_t)s : (int64_t)scell + (int64_t)s;
                const int64_t lo_cell = f ? d0 : d0 - (int64_t)(len - (uint32_t)k);
                const int64_t hi_cell = f ? d0 + (int64_t)(len - (uint32_t)k) : d0;
                if (lo_cell >= 0 && hi_cell + k <= (int64_t)total) {
                    best_cell = scell; dg = (int32_t)d0; fwd = f; seeded = true;
                    l1ok = lo_cell >= (int64_t)win_lo && hi_cell < (int64_t)win_lo + (int64_t)a.n_lds_bins;
                    if (l1ok) l1ok = (uint32_t)hi_cell < blkw[((uint32_t)lo_cell >> 6) - (win_lo >> 6)].y;
                }
            }
        };
        // First the seed table of the window's genome (bk_device.h seed_hash): kSeeds k-mers at positions that do not depend on the
        // read's length (evenly spaced over the launch's first record: the length load and the seeds' word loads go out together),
        // one 8-byte bucket each, and the candidate it names is verified against the reference in LDS -- two trips to memory where
        // the perfect hash of U takes four (length, words, pilot, entry; a fifth for the window genome's copy, occ).
        if (seed_tab && !BK_ABLATE(a, 9) && !BK_ABLATE(a, 11)) {
            uint64_t sg[kSeeds], sff[kSeeds];
            uint32_t sh[kSeeds], sis[kSeeds];
            uint2 sb[kSeeds];
#pragma unroll
            for (int sq = 0; sq < kSeeds; ++sq) {
                const uint32_t s = (hint_span * (uint32_t)sq) / (uint32_t)(kSeeds - 1);   // (wave-uniform)
                const uint64_t g = read_symbols_at(w, s, last_word) & kmask;       // base t of the k-mer at bits 2t
                const uint64_t rr = ~g & kmask;                                      // its reverse complement, first base on top
                const uint64_t ff = rev2_64(g) >> (64 - 2 * k);                      // the k-mer, first base on top
                const bool lt = ff < rr;                                             // lcb.rs:90-94
                sg[sq] = g; sff[sq] = ff; sis[sq] = lt ? 0u : 1u;
                sh[sq] = seed_hash(lt ? ff : rr);
                sb[sq] = seed_tab[sh[sq] >> (32u - a.seed_log2)];
            }
#pragma unroll
            for (int sq = 0; sq < kSeeds; ++sq) {
                const uint32_t s = (hint_span * (uint32_t)sq) / (uint32_t)(kSeeds - 1);
                const uint32_t tag = sh[sq] & 15u;
                const uint32_t ent = (sb[sq].x != 0xffffffffu && (sb[sq].x >> 28) == tag) ? sb[sq].x : sb[sq].y;
                const uint32_t cell = ent & ((1u << kSeedCellBits) - 1u), rc = (ent >> kSeedCellBits) & 1u;
                // in reach of the staged reference?  (REF_LDS: the window's cells and 64 in front)
                const bool in_ref = ent != 0xffffffffu && (ent >> 28) == tag && len != 0u && s + (uint32_t)k <= len &&
                                    cell + 64u >= win_lo && cell + (uint32_t)k <= win_lo + lds_cells;
                const int32_t cw = in_ref ? (int32_t)cell - (int32_t)win_lo : 0;
                const uint64_t ref = symbols_at(refw1, cw) & kmask;                  // reference base cell + t at bits 2t
                const bool same = sis[sq] == rc;                                     // same strand as the reference?
                candidate(in_ref && ref == (same ? sg[sq] : (~sff[sq] & kmask)), cell, same, s);
            }
        }
        // The lanes that are still without a diagonal (an error in every seed, a k-mer that found its bucket full, a read shorter
        // than the first record, no seed table): the perfect hash of U.  Round 0: kSeeds k-mers evenly spaced from the read's first
        // to its last; round 1, only when some lane found nothing again: the midpoints between them.  A read without a diagonal
        // costs ~100 slow-path searches, so the rare rounds pay.  (Two seeds per round and more rounds -- 2.95 lookups per read
        // instead of 4.02 -- was measured in round 2 and is slower, 0.155 against 0.139 ms: half of the tiles then pay a second
        // chain of dependent loads; the seeds are latency, not instructions.)
        for (int round = 0; round < 2 && !BK_ABLATE(a, 9); ++round) {   // (9: no seeds at all, 7: nothing behind them, 6: no mismatch loop)
            if (!__ballot(len != 0u && !seeded)) break;
            const bool had = seeded;   // a round is for the lanes left without a diagonal so far
            uint64_t sc[kSeeds];
            uint32_t sisrc[kSeeds], spil[kSeeds], spos[kSeeds];
#pragma unroll
            for (int sq = 0; sq < kSeeds; ++sq) {
                const uint32_t span_k = len ? len - (uint32_t)k : 0u;
                const uint32_t s = round == 0 ? (span_k * (uint32_t)sq) / (uint32_t)(kSeeds - 1)
                                              : (span_k * (uint32_t)(2 * sq + 1)) / (uint32_t)(2 * kSeeds);
                spos[sq] = s;
                const uint64_t g = read_symbols_at(w, s, last_word) & kmask;       // base t of the k-mer at bits 2t
                const uint64_t rr = ~g & kmask;                                      // its reverse complement, first base on top
                const uint64_t ff = rev2_64(g) >> (64 - 2 * k);                      // the k-mer, first base on top
                const bool lt = ff < rr;                                             // lcb.rs:90-94
                sc[sq] = lt ? ff : rr;
                sisrc[sq] = lt ? 0u : 1u;
                spil[sq] = ix.pilots[phf_bucket(sc[sq], ix.log2nb)];
            }
            // the loads of the four seeds go out together: perfect-hash entries, then (multi-genome index) where each hit sits in
            // the genome the LDS window covers
            uint4 se[kSeeds];
#pragma unroll
            for (int sq = 0; sq < kSeeds; ++sq) se[sq] = *reinterpret_cast<const uint4*>(ix.kmer_pos + phf_pos(sc[sq], spil[sq], ix.m, ix.log2nb, ix.log2p));
            bool shit[kSeeds];
            uint32_t soc[kSeeds];
#pragma unroll
            for (int sq = 0; sq < kSeeds; ++sq) {
                shit[sq] = len && !had && ((uint64_t)se[sq].x | ((uint64_t)se[sq].y << 32)) == sc[sq];
                soc[sq] = 0xffffffffu;
                if (a.occ && shit[sq] && (se[sq].w & kIdMask) < ix.n_full)
                    soc[sq] = a.occ[(size_t)(se[sq].w & kIdMask) * (uint32_t)a.n_files + win_file];
            }
#pragma unroll
            for (int sq = 0; sq < kSeeds; ++sq) {
                uint32_t scell = se[sq].z, src_rc = se[sq].w >> 31;   // where the seed sits: the k-mer's first occurrence ...
                if (soc[sq] != 0xffffffffu) { scell = soc[sq] & 0x7fffffffu; src_rc = soc[sq] >> 31; }   // ... or its occurrence in the window's genome
                candidate(shit[sq], scell, sisrc[sq] == src_rc, spos[sq]);
            }
        }
        const uint32_t dfl = (fwd ? 1u : 0u) | (seeded ? 2u : 0u);
        if (BK_ABLATE(a, 9) || BK_ABLATE(a, 7)) continue;
        // a read that cannot be settled here is one N run
        n_mark(nk != 0u && !l1ok, r32, 0u, nk, dg, dfl);

        // ---- mismatch flags, 160 bases at a time; mismatch by mismatch --------------------------------------------------
        const int32_t dgw = dg - (int32_t)win_lo;   // the diagonal in window coordinates
        // mismatch flags of bases [i0, i0 + 32) (i0 a multiple of 32): read words vs the reference words aligned with them
        auto mism32 = [&](uint32_t i0) -> uint32_t {
            const bool act = l1ok && i0 < len;
            const uint32_t wi = i0 >> 4;
            const uint32_t x0 = w[min(wi, last_word)], x1 = w[min(wi + 1u, last_word)];
            const int32_t p0 = act ? (fwd ? dgw + (int32_t)i0 : dgw + (int32_t)km1 - (int32_t)i0 - 31) : 0;
            const uint32_t sh = 2u * ((uint32_t)p0 & 15u);
            const uint32_t r0 = refw1[p0 >> 4], r1 = refw1[(p0 >> 4) + 1], r2 = refw1[(p0 >> 4) + 2];
            const uint32_t ya = __builtin_amdgcn_alignbit(r1, r0, sh), yb = __builtin_amdgcn_alignbit(r2, r1, sh);   // 32 reference bases, rising
            // against the reference: read base i0 + t <-> complement of reference base p0 + 31 - t
            const uint32_t d0 = x0 ^ (fwd ? ya : ~rev2_32(yb)), d1 = x1 ^ (fwd ? yb : ~rev2_32(ya));
            const uint32_t m = even_bits(d0 | (d0 >> 1)) | (even_bits(d1 | (d1 >> 1)) << 16);
            const uint32_t hi = len > i0 ? min(len - i0, 32u) : 0u;   // bases of the record in these words
            return act ? m & (hi >= 32u ? 0xffffffffu : (1u << hi) - 1u) : 0u;
        };
        // E run over k-mers [g_lo, g_hi]: cells dg + g_lo .. dg + g_hi (fwd) / dg - g_hi .. dg - g_lo
        auto e_run = [&](bool on, uint32_t g_lo, uint32_t g_hi) {
            if (!on) return;
            const uint32_t n = g_hi - g_lo + 1u;
            const uint32_t c_lo = (uint32_t)(fwd ? dgw + (int32_t)g_lo : dgw - (int32_t)g_hi);
            const uint32_t inc = fwd ? 1u : 0x10000u;
            __hip_atomic_fetch_add(&bins[c_lo], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&bins[c_lo + n], 0u - inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        unsigned long long M01 = 0ull, M23 = 0ull;   // flags of the chunk's bases 0..63, 64..127
        uint32_t M4 = 0u;                            // ... 128..159
        int32_t tp = -0x20000000;    // the lane's last resolved mismatch (absolute base), far away before the first
        auto peek3 = [&](unsigned long long m01, unsigned long long m23, uint32_t m4) -> uint32_t {   // lowest flag of the chunk
            return m01 ? (uint32_t)__builtin_ctzll(m01) : m23 ? 64u + (uint32_t)__builtin_ctzll(m23) : m4 ? 128u + (uint32_t)__builtin_ctz(m4) : kNoPos;
        };
        auto pop3 = [&](unsigned long long& m01, unsigned long long& m23, uint32_t& m4) {   // ... taken off
            const bool z01 = m01 == 0ull, z23 = m23 == 0ull;
            m4 = (z01 && z23) ? m4 & (m4 - 1u) : m4;
            m23 = z01 ? m23 & (m23 - 1ull) : m23;
            m01 &= m01 - 1ull;
        };
        for (uint32_t cb = 0;; cb += 128u) {   // (wave-uniform) the chunk covers bases [cb, cb + 160); its first word is carried over
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                if (cb != 0u && j == 0) continue;
                const uint32_t f = mism32(cb + 32u * (uint32_t)j);
                if (j == 0) M01 |= (unsigned long long)f; else if (j == 1) M01 |= (unsigned long long)f << 32;
                else if (j == 2) M23 |= (unsigned long long)f; else if (j == 3) M23 |= (unsigned long long)f << 32;
                else M4 = f;
            }
            const uint32_t scanned = cb + 160u;
            // A mismatch is resolved once the k - 1 bases behind it are scanned (or the read ends): that far reach the k-mers that
            // hold it, and whatever they hold of its successors is then known.  What is not resolved lies in the chunk's last
            // word (k <= 31): the word that is carried over.
            const bool all_res = scanned >= len;
            const uint32_t r4 = all_res ? M4 : M4 & ((1u << (32u - km1)) - 1u);   // (flags 128 .. 159 - (k - 1); words 0..3 are always resolved)
            if (BK_ABLATE(a, 6)) { M01 = 0ull; M23 = 0ull; M4 = 0u; }
            // ---- one mismatch per lane: the tile's resolved mismatches are dealt out over the wave (a read has 0.75 of them on the
            // benchmark, the busiest of 64 reads four or five: taken read by read, five passes would run at a sixth of the lanes) ----
            const uint32_t cnt = BK_ABLATE(a, 6) ? 0u : (uint32_t)__popcll(M01) + (uint32_t)__popcll(M23) + (uint32_t)__popc(r4);
            BK_DBG(a, 23, cnt != 0u, cnt);
            uint32_t pin = cnt;   // inclusive prefix sum over the lanes
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)pin, off); if (lane >= off) pin += x; }
            const uint32_t pex = pin - cnt;
            const uint32_t n_items = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)pin, 63));
            for (uint32_t it0 = 0; it0 < n_items; it0 += 64u) {
                // owner of item it0 + lane: every lane with items writes its number where its first item of this pass sits, a running
                // maximum spreads it over the items behind
                own_s[lane] = 0u;
                __builtin_amdgcn_wave_barrier();
                if (cnt && pin > it0 && pex < it0 + 64u) own_s[pex > it0 ? pex - it0 : 0u] = (uint32_t)lane;
                __builtin_amdgcn_wave_barrier();
                uint32_t ow = own_s[lane];
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)ow, off); if (lane >= off) ow = max(ow, x); }
                const bool on = it0 + (uint32_t)lane < n_items;
                const int src = on ? (int)ow : lane;
                // the owner's read: flags, diagonal, length, record, last mismatch before this chunk's
                unsigned long long m01 = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(M01 >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)M01, src);
                unsigned long long m23 = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(M23 >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)M23, src);
                uint32_t m4 = (uint32_t)__shfl((int)M4, src);
                const int32_t o_dgw = __shfl(dgw, src);
                const uint32_t o_fl = (uint32_t)__shfl((int)dfl, src);
                const uint32_t o_nk = (uint32_t)__shfl((int)nk, src);
                const uint32_t o_rec = (uint32_t)__shfl((int)r32, src);
                int32_t tq = __shfl(tp, src);                      // becomes the mismatch before mine
                const uint32_t o_pex = (uint32_t)__shfl((int)pex, src);   // (every shuffle outside the conditions: a lane that is off may be another's owner)
                uint32_t jj = on ? it0 + (uint32_t)lane - o_pex : 0u;   // mine is the owner's jj-th of this chunk
                while (__ballot(jj != 0u)) {
                    if (jj) { tq = (int32_t)(cb + peek3(m01, m23, m4)); pop3(m01, m23, m4); --jj; }
                }
                const bool ofwd = o_fl & 1u;
                const int32_t t = on ? (int32_t)(cb + peek3(m01, m23, m4)) : 0;
                pop3(m01, m23, m4);
                const uint32_t nb = peek3(m01, m23, m4);
                const int32_t tn = (on && nb != kNoPos) ? (int32_t)(cb + nb) : 0x20000000;   // (an unseen one is out of reach)
                pop3(m01, m23, m4);
                const uint32_t nb2 = peek3(m01, m23, m4);
                const int32_t tn2 = (on && nb2 != kNoPos) ? (int32_t)(cb + nb2) : 0x20000000;
                BK_DBG(a, 24, on, 1);
                // The k-mers between the previous mismatch and this one hold none: an E run (they start behind the previous one and
                // end before this one)
                const uint32_t done = tq < 0 ? 0u : min((uint32_t)tq, o_nk - 1u) + 1u;
                {
                    const bool eg = on && t >= k && (uint32_t)(t - k) >= done && done < o_nk;
                    const uint32_t g_hi = min((uint32_t)(t - k), o_nk - 1u);
                    BK_DBG(a, 25, eg, 1);
                    if (eg) {
                        const uint32_t c_lo = (uint32_t)(ofwd ? o_dgw + (int32_t)done : o_dgw - (int32_t)g_hi);
                        const uint32_t inc = ofwd ? 1u : 0x10000u;
                        __hip_atomic_fetch_add(&bins[c_lo], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(&bins[c_lo + (g_hi - done + 1u)], 0u - inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                // The k-mers whose FIRST mismatch is t: they start behind the previous one and hold t.  Every k-mer that holds a
                // mismatch belongs to exactly one such range.
                const uint32_t o_lo = (uint32_t)max(max(t - (int32_t)km1, tq + 1), 0), o_hi = min((uint32_t)t, o_nk - 1u);
                const bool own = on && o_lo <= o_hi;
                // Their cells, lowest first (window coordinates), and which of them are "fast" (clean, ids = cell + one constant).
                // Usually all: one stretch.  Otherwise the range is taken stretch by stretch -- fast cells as below, the k-mers at the
                // others are left to level2_kernel one by one (its answer table knows the cells that are not clean).
                const int32_t ca = own ? (ofwd ? o_dgw + (int32_t)o_lo : o_dgw - (int32_t)o_hi) : 0;
                const uint32_t n_own = own ? o_hi - o_lo + 1u : 0u;           // (at most k <= 31)
                uint32_t pat = bits32_at(fastw, ca) & ((1u << n_own) - 1u);   // bit p: the cell ca + p is fast
                // which of the three other bases: read XOR reference at the mismatch, the same on either strand (bk_device.h)
                const uint32_t tt = own ? (uint32_t)t : 0u;
                const uint32_t rw = (words0 + (uint64_t)o_rec * a.stride_words)[min(tt >> 4, last_word)];
                const int32_t pr = own ? (ofwd ? o_dgw + (int32_t)tt : o_dgw + (int32_t)km1 - (int32_t)tt) : 0;
                const uint32_t refb = (refw1[pr >> 4] >> (2u * ((uint32_t)pr & 15u))) & 3u;
                const uint32_t alt = ((((rw >> (2u * (tt & 15u))) & 3u) ^ (ofwd ? refb : 3u - refb)) & 3u) - 1u;
                // Cells that are not fast (no cell of a many-genome index is clean): per (position of the mismatch, other base) the
                // offsets at which the k-mer still takes its own row (IndexView::cell_nat) -- one word for all the k-mers of the range,
                // bit o <-> cell pr - o
                // Without touch lists the bits stand for one V row of their own (cell_natrow: id + offset), whatever cell_blk says.
                bool by_row = false;
                uint32_t nat_row = 0u;
                if (a.cell_nat && own && pat != (1u << n_own) - 1u && !BK_ABLATE(a, 12)) {
                    const uint32_t nm = (__brev(a.cell_nat[((size_t)(pr + (int32_t)win_lo)) * 3u + alt]) >> (31u - (uint32_t)(pr - ca))) & ((1u << n_own) - 1u);
                    if (!SPARSE && a.cell_natrow) { pat = nm; by_row = true; nat_row = a.cell_natrow[(size_t)(pr + (int32_t)win_lo)]; }
                    else pat |= nm;
                }
                if (BK_ABLATE(a, 5)) pat = 0u;                                // (5: nothing is settled here)
                uint32_t used = 0u;   // cells of the range already dealt with
                while (__ballot(used < n_own)) {
                    const bool go = used < n_own;
                    const uint32_t rest = pat >> used;
                    const bool ones = rest & 1u;
                    // the stretch of equal bits at `used`: cells ca + used .. ca + used + ln - 1
                    const uint32_t ln = go ? min((uint32_t)__builtin_ctz((ones ? ~rest : rest) | (1u << (n_own - used))), n_own - used) : 0u;
                    const int32_t c0 = ca + (int32_t)used, c1 = c0 + (int32_t)ln - 1;
                    // ... are the k-mers [x_lo, x_hi] (a read against the reference meets the cells from the top)
                    const uint32_t x_lo = ofwd ? o_lo + used : o_hi + 1u - used - ln, x_hi = x_lo + ln - 1u;
                    const uint2 ba = blkw[(go ? c0 : 0) >> 6], bz = blkw[(go ? c1 : 0) >> 6];
                    const bool fast = go && ones && (by_row || ba.x == bz.x);
                    // of these, [x_lo, xs_hi] hold nothing but t -- an S run -- and [xm_lo, x_hi] also hold the next mismatch
                    const int32_t xs_hi_i = min((int32_t)x_hi, tn - k);
                    const bool has_s = fast && xs_hi_i >= (int32_t)x_lo;
                    const uint32_t xs_hi = has_s ? (uint32_t)xs_hi_i : x_lo;
                    const uint32_t xm_lo = (uint32_t)max((int32_t)x_lo, tn - (int32_t)km1);
                    const bool has_m = fast && xm_lo <= x_hi;
                    {
                        const uint32_t tpos = tt - x_lo, nm1 = xs_hi - x_lo;   // offset of the differing base in the run's first k-mer
                        // offsets (along the reference, from each k-mer's start) the run's k-mers have the difference at
                        const uint32_t of_first = ofwd ? tpos : km1 - tpos;
                        const uint32_t of_lo = ofwd ? tpos - nm1 : of_first;       // fwd: later k-mers start later, the offset shrinks
                        const uint32_t of_hi = ofwd ? tpos : of_first + nm1;
                        const int lo2 = max((int)of_lo, omin), hi2 = min((int)of_hi, omin + span - 1);
                        if (has_s && lo2 <= hi2 && !BK_ABLATE(a, 2)) {
                            const uint32_t idS = by_row ? nat_row - of_first                       // (cell_natrow: id + offset of every k-mer of the run)
                                                        : (uint32_t)(ofwd ? c0 : c1) + win_lo + ba.x;   // id of the cell of k-mer x_lo (cell_fast: ids = cell + constant)
                            const uint64_t ci = v_row_base(idS + of_first - (uint32_t)omin, alt, ofwd ? 0u : 1u, span) + (uint32_t)(lo2 - omin);
                            if constexpr (SPARSE) {
                                // the row is touched: noted per block of 64 cells (a bit per row in device memory, looked at and set
                                // for every mismatch, took two thirds of this kernel on a 100-strain index), handed on in the epilogue
                                if (BK_ABLATE(a, 13)) touch(a.touch_v, v_row_index(idS + of_first - (uint32_t)omin, alt, ofwd ? 0u : 1u));   // (13: the row's own bit as well)
                                const uint32_t tb = (uint32_t)(ofwd ? c0 : c1) >> 6;
                                if (!(blk_t[tb >> 5] >> (tb & 31u) & 1u)) __hip_atomic_fetch_or(&blk_t[tb >> 5], 1u << (tb & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                            // the +1: into the workgroup's table when this counter was met before and its slot is free or its own
                            const uint32_t hk = (uint32_t)ci * 0x9E3779B1u;
                            const uint32_t sbit = 1u << (hk >> 27);
                            bool kept = false;
                            if (!BK_ABLATE(a, 10) && (__hip_atomic_fetch_or(&seen[(hk >> 17) & (kSeenWords - 1u)], sbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & sbit)) {
                                const uint32_t hs = (hk >> 9) & (kHotSlots - 1u);
                                unsigned int expect = 0xffffffffu;
                                kept = __hip_atomic_compare_exchange_strong(&hot_key[hs], &expect, (uint32_t)ci, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) || expect == (uint32_t)ci;
                                if (kept) __hip_atomic_fetch_add(&hot_cnt[hs], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                            if (!kept && !BK_ABLATE(a, 14)) atomicAdd(v_counters + ci, 1ull);
                            if (hi2 - omin + 1 < span && !BK_ABLATE(a, 14)) atomicAdd(v_counters + ci + (uint32_t)(hi2 - lo2 + 1), ~0ull);   // (slot `span` is never read)
                        }
                    }
                    {
                        // The k-mers that hold t and its successor.  If none of them reaches the successor after that and each of their
                        // cells has no other reference k-mer form within Hamming distance 3, they hold exactly two differences from a
                        // reference k-mer that is isolated up to distance 3: neither a reference k-mer nor one base away from one
                        // (triangle inequality) -- they touch nothing.  Otherwise level2_kernel looks at them one by one.
                        const int32_t ma = has_m ? (ofwd ? o_dgw + (int32_t)xm_lo : o_dgw - (int32_t)x_hi) : 0;
                        const uint32_t needm = has_m ? 0xffffffffu >> (31u - (x_hi - xm_lo)) : 0u;
                        const bool dead = !stats && tn2 - (int32_t)km1 > (int32_t)x_hi && (bits32_at(c3w, ma) & needm) == needm;
                        l2_mark(has_m && !dead, o_rec, xm_lo, x_hi + 1u - xm_lo, o_dgw + (int32_t)win_lo, o_fl);
                    }
                    // cells that are not fast
                    l2_mark(go && !fast, o_rec, x_lo, ln, o_dgw + (int32_t)win_lo, o_fl);
                    used += ln;
                }
            }
            // each lane: its last resolved mismatch; what is not resolved stays
            if (cnt) tp = (int32_t)cb + (r4 ? 159 - (int32_t)__builtin_clz(r4) : M23 ? 127 - (int32_t)__builtin_clzll(M23) : 63 - (int32_t)__builtin_clzll(M01));
            if (scanned >= maxlen) break;
            M01 = (unsigned long long)(M4 & ~r4); M23 = 0ull; M4 = 0u;   // the next chunk starts 128 bases on
        }
        const uint32_t done = tp < 0 ? 0u : min((uint32_t)tp, nk - 1u) + 1u;
        BK_DBG(a, 26, l1ok && done < nk, 1);
        e_run(l1ok && done < nk, done, nk - 1u);   // behind the last mismatch
    }
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(pf_sink) : "memory");   // the prefetch register stays reserved up to here

    // ---- epilogue: difference array -> per-cell counts (prefix sum over the workgroup), written as this workgroup's slab ----
    if (threadIdx.x == 0) *block_kmers = 0;
    __syncthreads();
    if (threadIdx.x < kHotSlots && hot_cnt[threadIdx.x]) atomicAdd(v_counters + hot_key[threadIdx.x], (unsigned long long)hot_cnt[threadIdx.x]);   // the table's totals
    if constexpr (SPARSE) {
        // blocks whose rows were counted into -> ScanArgs::touch_b (expanded into touch_v's row bits before the rows are listed)
        const uint32_t b = threadIdx.x;
        if (b < 32u * kBlkTouchWords && (blk_t[b >> 5] >> (b & 31u) & 1u)) {
            const uint32_t gb = (win_lo >> 6) + b;
            if (!(a.touch_b[gb >> 5] >> (gb & 31u) & 1u)) atomicOr(a.touch_b + (gb >> 5), 1u << (gb & 31u));
        }
    }
    if (!BK_ABLATE(a, 8)) {   // (8: without the prefix sum and the slab)
        const uint32_t nb = a.n_lds_bins;
        const uint32_t per = (nb + kScanBlock - 1) / kScanBlock;
        const uint32_t b0 = min(threadIdx.x * per, nb), b1 = min(b0 + per, nb);
        uint32_t sum = 0;
        for (uint32_t i = b0; i < b1; ++i) sum += bins[i];
        uint32_t inc = sum;   // inclusive scan of the per-thread sums across the wave, then across waves
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, off); if (lane >= off) inc += t; }
        if (lane == 63) scan_tmp[wave] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (int wv = 0; wv < wave; ++wv) run += scan_tmp[wv];
        for (uint32_t i = b0; i < b1; ++i) { run += bins[i]; bins[i] = run; }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nb; i += kScanBlock) a.slabs[(size_t)blockIdx.x * nb + i] = bins[i];
    }
    uint32_t tot = nkm;
#pragma unroll
    for (int off = 32; off; off >>= 1) tot += (uint32_t)__shfl_xor((int)tot, off);
    if (lane == 0 && tot) atomicAdd(block_kmers, tot);
    __syncthreads();
    if (threadIdx.x == 0 && *block_kmers && a.kmer_total) atomicAdd(a.kmer_total, (unsigned long long)*block_kmers);
}

// ------------------------------------------------------------------------------------------------ K1b
// What the scan did not settle, in two kernels.
//
// nbatch_kernel -- the N runs (ScanArgs::n_bits, one bit per k-mer of each record; n_any, one bit per record): k-mers at cells
// that are not fast, of reads off the LDS window or without a diagonal.  Cut into pieces of at most 32 k-mers (a piece lies in
// 64 read bases) and taken 64 pieces at a time, one per lane ("N batch"): the piece's bases are compared with the reference
// along the diagonal; its leading k-mers whose cells are clean and continue one id sequence are resolved mismatch by mismatch
// -- the k-mers that hold one mismatch and nothing else are an S run (one row of the V plane, two atomics), the k-mers that
// hold exactly two at cells isolated up to Hamming distance 3 touch nothing -- and what that cannot settle is marked in
// l2_bits[record] (one bit per k-mer; l2_any, one bit per record) like the k-mers the scan itself leaves to Level 2.
//
// level2_kernel -- the marked k-mers, one per lane, nothing rolls: every lane extracts its k-mer and the reference along the
// record's diagonal straight from the packed words, so the 64 k-mers of a batch are independent and all their loads are in flight
// together (this kernel is latency, not arithmetic: about one k-mer in a thousand on the benchmark, one in forty with four
// strains).
//   discovery   l2_any -> marked records (LDS queue) -> their bitmap words -> the marked k-mers, listed
//               (record, k-mer index) in an LDS queue in record order: the k-mers of a marked run sit in neighbouring lanes.
//               Every bit taken is cleared: the bitmaps are all zero again when the kernel ends.
//   per k-mer   differences with the reference at its cell along the diagonal, the cell's id and flags (bk_device.h):
//               0 differences, a reference k-mer starts there     -> it IS that k-mer: +1 on its E counter
//               1 difference, clean cell                          -> a single-k-mer S run in the V row of (id, offset, base)
//               1 difference, cell not clean                      -> the precomputed answer (DirtyAns): one load
//               2 differences, isolated up to distance 3          -> touches nothing (full_kmer_stats: statistics table)
//               everything else (no diagonal, no k-mer at the cell, more differences) -> the slow path: compacted into a
//               per-wave LDS queue, batches of the SlowPipe (perfect-hash membership, then the neighbour search over the two
//               half-k-mer directories, smallest (position, NbEntry::p) wins).
//   V atomics   a single-k-mer S run is +1 at its counter, -1 at the next.  The k-mers that cover one sequencing error are
//               consecutive and land in one row at consecutive offsets, so the -1 of one is the +1 of its neighbour: lanes
//               compare with their neighbours (shuffles) and only the ends of a run reach memory.
constexpr int kL2Block = 256;
constexpr int kL2Waves = kL2Block / 64;
constexpr int kL2QueueCap = 128;            // record / slow queues: a batch is taken at 64 pending, a round adds <= 64
constexpr int kL2KmerCap = 256;             // k-mer queue
constexpr int kAnyWords = 8;                // words of l2_any a wave takes at a time (256 records)
constexpr int kNbBlocksPerWave = 4;        // blocks of 256 records a wave of nbatch_kernel takes when there are enough of them
template <bool STATS, int KT, bool SPARSE>
__global__ __launch_bounds__(kL2Block) void nbatch_kernel(ScanArgs a) {
    __shared__ unsigned int rec_q[kL2Waves * kL2QueueCap];
    __shared__ unsigned long long piece_q[kL2Waves * kL2QueueCap];   // record << 24 | first k-mer << 8 | k-mers
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned int* const rq = rec_q + wave * kL2QueueCap;
    unsigned long long* const pq = piece_q + wave * kL2QueueCap;
    const unsigned int* refw = a.ref_words + kRefPadWords;
    const int k = KT ? KT : a.k;
    const uint32_t km1 = (uint32_t)k - 1u;
    const int omin = a.v_omin, span = a.v_span;
    unsigned long long* const v_counters = a.counters + a.v_off;
    const uint32_t last_word = a.stride_words - 1u;
    uint64_t n_records = a.n_records;
    if (a.n_records_dev) {
        const uint64_t nd = *a.n_records_dev;
        n_records = nd > a.rec_base ? min(nd - a.rec_base, a.n_records) : 0ull;
    }
    const uint32_t* const words0 = a.words + a.rec_base * a.stride_words;
    const uint32_t nw = a.l2_words;
    const uint64_t n_any = (n_records + 31) / 32;                    // words of n_any
    const uint64_t n_blk = (n_any + kAnyWords - 1) / kAnyWords;
    {
        const unsigned int* yfw = a.cell_yf + kRefPadWords;
        const unsigned int* yrw = a.cell_yr + kRefPadWords;
        const unsigned int* c3w = a.cell_clean3 + kBitPadWords;
        const uint32_t piece = 65u - (uint32_t)k;                   // most k-mers of a run the N batch looks at together (64 bases)
        // k-mers [sk, sk + n) of record `rec` are left to the second pass: set their bits
        auto l2_mark = [&](bool on, uint32_t rec, uint32_t sk, uint32_t n) {
            if (!on) return;
            unsigned int* row = a.l2_bits + (size_t)rec * nw;
            atomicOr(a.l2_any + (rec >> 5), 1u << (rec & 31u));
            uint32_t wd = sk >> 5, bit = sk & 31u, left = n;
            while (left) {
                const uint32_t take = min(left, 32u - bit);
                atomicOr(row + wd, (take == 32u ? 0xffffffffu : (1u << take) - 1u) << bit);
                left -= take; ++wd; bit = 0u;
            }
        };
        uint32_t qp = 0;   // wave-uniform fill of the piece queue
        // The N batch: up to 64 queued pieces, one per lane.  The piece's n <= 65 - k k-mers lie in 64 read bases.  Its leading
        // k-mers whose cells are clean and continue one id sequence are resolved here, mismatch by mismatch (t_1 < t_2 < ...): the
        // k-mers that hold only t_i are an S run (two atomics); those that hold t_i and t_i+1 go to the second pass as a chunk --
        // unless they hold exactly these two and every one of their cells has no other reference k-mer form within Hamming
        // distance 3: then they are neither reference k-mers nor one base away from one and touch nothing (full_kmer_stats: the
        // statistics table still wants them).  K-mers without a mismatch (a read off the LDS window), at dirty cells or without a
        // diagonal go to the second pass as a chunk; what is left of the piece comes back into the queue.
        auto n_batch = [&]() {
            const uint32_t nb2 = min(qp, 64u);
            BK_DBG(a, 20, lane == 0, 1); BK_DBG(a, 21, lane == 0, nb2); BK_DBG(a, 22, lane == 0 && qp < 64u, 1);
            const bool have = (uint32_t)lane < nb2;
            const unsigned long long ent = have ? pq[lane] : 0ull;
            {   // move the rest of the queue down
                const uint32_t rest = qp - nb2;
                const unsigned long long t = (uint32_t)lane < rest ? pq[64 + lane] : 0ull;
                __builtin_amdgcn_wave_barrier();
                if ((uint32_t)lane < rest) pq[lane] = t;
                __builtin_amdgcn_wave_barrier();
                qp = rest;
            }
            const uint32_t rec2 = (uint32_t)(ent >> 24), s_first = (uint32_t)(ent >> 8) & 0xffffu, n2 = (uint32_t)ent & 0xffu;
            const uint2 dgf = have ? a.l2_diag[rec2] : make_uint2(0u, 0u);
            const int32_t dg2 = (int32_t)dgf.x;
            const uint32_t fl2 = dgf.y;
            const bool fwd2 = fl2 & 1u;
            const bool an = have && (fl2 & 2u) && n2 + km1 <= 64u;   // analysable: a diagonal, and 64 bases hold it
            const uint32_t* __restrict__ w2 = words0 + (uint64_t)rec2 * a.stride_words;
            const int32_t c_first = an ? (fwd2 ? dg2 + (int32_t)s_first : dg2 - (int32_t)s_first) : 0;
            const uint32_t ddir = fwd2 ? 1u : 0xffffffffu;
            const uint64_t ga = read_symbols_at(w2, s_first, last_word), gb = read_symbols_at(w2, s_first + 32u, last_word);
            const uint32_t id_first = a.id_at[c_first];
            // how many leading k-mers sit at clean cells that continue one id sequence (the first needs no follow bit)
            int n1;          // ... that many; 0: the first cell is dirty
            int head = 0;    // leading k-mers that go to the second pass as one chunk
            {
                const uint64_t y_lo = fwd2 ? symbols_at(yfw, c_first) : rev2_64(symbols_at(yrw, c_first - 31));      // symbol j: cell of k-mer j
                const uint64_t y_hi = fwd2 ? symbols_at(yfw, c_first + 32) : rev2_64(symbols_at(yrw, c_first - 63));
                const uint64_t e5 = 0x5555555555555555ull;
                const uint64_t bad_lo = ~(y_lo & ((y_lo >> 1) | 1ull)) & e5, bad_hi = ~(y_hi & (y_hi >> 1)) & e5;
                const int good_len = bad_lo ? (__builtin_ctzll(bad_lo) >> 1) : 32 + (bad_hi ? (__builtin_ctzll(bad_hi) >> 1) : 32);
                const uint64_t cl_lo = y_lo & e5, cl_hi = y_hi & e5;
                const int dirty_len = cl_lo ? (__builtin_ctzll(cl_lo) >> 1) : 32 + (cl_hi ? (__builtin_ctzll(cl_hi) >> 1) : 32);
                n1 = an ? min(good_len, (int)n2) : 0;
                if (n1 == 0) head = an ? min(dirty_len, (int)n2) : (int)n2;
            }
            uint64_t F;   // mismatch flags of the read bases s_first + [0, 64) the first n1 k-mers cover
            uint64_t xa, xb;   // read XOR reference along those bases (2 bits per base): which other base stands at a mismatch
            {
                // read base s_first + t <-> reference base c_first + t (fwd) / complement of base c_first + k - 1 - t
                const uint64_t ra = fwd2 ? symbols_at(refw, c_first) : ~rev2_64(symbols_at(refw, c_first + (int32_t)km1 - 31));
                const uint64_t rb = fwd2 ? symbols_at(refw, c_first + 32) : ~rev2_64(symbols_at(refw, c_first + (int32_t)km1 - 63));
                const uint64_t da = ga ^ ra, db = gb ^ rb;
                xa = da; xb = db;
                const uint32_t f_lo = even_bits((uint32_t)(da | (da >> 1))) | (even_bits((uint32_t)((da | (da >> 1)) >> 32)) << 16);
                const uint32_t f_hi = even_bits((uint32_t)(db | (db >> 1))) | (even_bits((uint32_t)((db | (db >> 1)) >> 32)) << 16);
                const uint32_t L = n1 ? (uint32_t)n1 + km1 : 0u;   // bases they cover
                F = (((uint64_t)f_hi << 32) | f_lo) & (L >= 64u ? ~0ull : (1ull << L) - 1ull);
            }
            const int kk = (int)km1;
            int cutn = n1;   // k-mers [0, cutn) of the piece are resolved in this pass
            uint64_t c3 = 0;   // bit j: the cell of k-mer j has no other reference k-mer form within Hamming distance 3
            if (!STATS) c3 = fwd2 ? ((uint64_t)bits32_at(c3w, c_first + 32) << 32) | bits32_at(c3w, c_first)
                                  : ((uint64_t)__builtin_bitreverse32(bits32_at(c3w, c_first - 63)) << 32) | __builtin_bitreverse32(bits32_at(c3w, c_first - 31));
            if (n1) {
                const int t1 = F ? __builtin_ctzll(F) : 1000;
                if (t1 > kk) { head = min(t1 - kk, n1); cutn = 0; F = 0ull; }   // leading k-mers without a mismatch
            }
            // ---- the head ----
            BK_DBG(a, !(have && (fl2 & 2u) && n2 + km1 <= 64u) ? 0 : n1 == 0 ? 1 : 2, have && head > 0, head);
            l2_mark(have && head > 0, rec2, s_first, (uint32_t)head);
            if (have && head > 0) cutn = head;
            // ---- mismatch by mismatch ----
            int tprev = -1000, gprev = -1;   // the previous mismatch; the last k-mer already sent to the second pass
            for (int it = 0; it < kNIters; ++it) {
                const bool act = F != 0ull;
                if (!__ballot(act)) break;
                const int ti = act ? __builtin_ctzll(F) : 0;
                F &= F - 1ull;
                int tn = F ? __builtin_ctzll(F) : 1000;
                const uint64_t F2 = F & (F - 1ull);
                const int tn2 = F2 ? __builtin_ctzll(F2) : 1000;
                if (act && (tn - ti > k || it == kNIters - 1)) {
                    // k-mers without a mismatch follow (or the piece has more mismatches than passes): the piece is cut
                    // after the last k-mer that holds t_i
                    cutn = min(cutn, ti + 1);
                    F = 0ull;
                    if (tn - ti > k) tn = 1000;
                }
                // S run: the k-mers that hold t_i and nothing else
                const int s_lo = max(max(ti - kk, tprev + 1), 0), s_hi = min(min(ti, tn - k), cutn - 1);
                if (act && s_hi >= s_lo) {
                    const uint32_t tpos = (uint32_t)(ti - s_lo);    // offset of the differing base in the run's first k-mer
                    // which of the three other bases: read XOR reference at the mismatch, the same on either strand (bk_device.h)
                    const uint32_t alt = ((uint32_t)((ti < 32 ? xa : xb) >> (2u * ((uint32_t)ti & 31u))) & 3u) - 1u;
                    const uint32_t nm1 = (uint32_t)(s_hi - s_lo);
                    // offsets (along the reference, from each k-mer's start) the run's k-mers have the difference at
                    const uint32_t o_first = fwd2 ? tpos : km1 - tpos;
                    const uint32_t o_lo = fwd2 ? tpos - nm1 : o_first;       // fwd: later k-mers start later, the offset shrinks
                    const uint32_t o_hi = fwd2 ? tpos : o_first + nm1;
                    const int lo2 = max((int)o_lo, omin), hi2 = min((int)o_hi, omin + span - 1);
                    if (lo2 <= hi2 && !BK_ABLATE(a, 2)) {
                        const uint32_t idS = id_first + ddir * (uint32_t)s_lo;
                        unsigned long long* row = v_counters + v_row_base(idS + o_first - (uint32_t)omin, alt, fwd2 ? 0u : 1u, span);
                        if constexpr (SPARSE) touch(a.touch_v, v_row_index(idS + o_first - (uint32_t)omin, alt, fwd2 ? 0u : 1u));
                        atomicAdd(row + (lo2 - omin), 1ull);
                        if (hi2 - omin + 1 < span) atomicAdd(row + (hi2 - omin + 1), ~0ull);   // (slot `span` is never read)
                    }
                }
                // the k-mers that hold t_i and t_i+1 (those that also hold t_i-1 went with the previous pair)
                const int g_lo = max(max(tn - kk, gprev + 1), 0), g_hi = min(ti, cutn - 1);
                const bool pair = act && tn - ti <= kk && g_hi >= g_lo;
                bool dead = false;
                if (!STATS) {
                    const uint32_t need = pair ? 0xffffffffu >> (31 - (g_hi - g_lo)) : 0u;
                    dead = tn2 - kk > g_hi && ((uint32_t)(c3 >> (pair ? g_lo : 0)) & need) == need;   // none of them reaches t_i+2
                }
                BK_DBG(a, 3, pair && !dead, g_hi - g_lo + 1);
                l2_mark(pair && !dead, rec2, s_first + (uint32_t)(pair ? g_lo : 0), (uint32_t)(g_hi - g_lo + 1));
                if (pair) gprev = g_hi;
                tprev = ti;
            }
            // ---- what is left of the piece comes back ----
            const bool requeue = have && (uint32_t)cutn < n2;
            const unsigned long long rm = __ballot(requeue);
            if (rm) {
                if (requeue) pq[qp + lane_prefix(rm)] = ((unsigned long long)rec2 << 24) | ((unsigned long long)(s_first + (uint32_t)cutn) << 8) | (n2 - (uint32_t)cutn);
                qp += (uint32_t)__popcll(rm);
            }
            __builtin_amdgcn_wave_barrier();
        };
        // 64 marked records, one per lane: the runs of set bits of their n_bits rows (cleared as they are taken), word by word --
        // a run that crosses a word boundary is two pieces; every piece is a sub-range of a run, any cut gives the same counts
        auto take_n_records = [&](uint32_t n) {
            const uint32_t rec = (uint32_t)lane < n ? rq[lane] : 0xffffffffu;
            unsigned int* row = a.n_bits + (size_t)(rec == 0xffffffffu ? 0u : rec) * nw;
            for (uint32_t w0 = 0; w0 < nw; w0 += 4u) {   // (four words' loads in flight together)
                uint32_t bw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bw[j] = (rec != 0xffffffffu && w0 + j < nw) ? row[w0 + j] : 0u;
                    if (bw[j]) row[w0 + j] = 0u;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t bits = bw[j];
                    while (__ballot(bits != 0u)) {
                        const uint32_t st = bits ? (uint32_t)__builtin_ctz(bits) : 0u;
                        const uint32_t ones = bits ? min((uint32_t)__builtin_ctzll(~(unsigned long long)(bits >> st)), piece) : 0u;   // (length of the run of ones at st, at most to the word's end)
                        const unsigned long long pm = __ballot(bits != 0u);
                        if (bits) pq[qp + lane_prefix(pm)] = ((unsigned long long)rec << 24) | ((unsigned long long)((w0 + (uint32_t)j) * 32u + st) << 8) | ones;
                        qp += (uint32_t)__popcll(pm);
                        bits &= ~((ones >= 32u ? 0xffffffffu : (1u << ones) - 1u) << st);
                        __builtin_amdgcn_wave_barrier();
                        while (qp >= 64u) n_batch();   // (a batch may put up to 64 leftovers back: below 64 again before the next round adds its own)
                    }
                }
            }
        };
        uint32_t qrn = 0;   // wave-uniform fill of the record queue
        // a wave takes kNbBlocksPerWave consecutive blocks of 256 records per turn: their n_any words in one load, 32 of the lanes
        const uint64_t n_turns = (n_blk + kNbBlocksPerWave - 1) / kNbBlocksPerWave;
        for (uint64_t turn = (uint64_t)blockIdx.x * kL2Waves + wave; turn < n_turns; turn += (uint64_t)gridDim.x * kL2Waves) {
            const uint64_t i = turn * (kNbBlocksPerWave * kAnyWords) + lane;
            const uint32_t anyw = (lane < kNbBlocksPerWave * kAnyWords && i < n_any) ? a.n_any[i] : 0u;
            if (anyw) a.n_any[i] = 0u;     // taken: n_any and n_bits are all zero again when this kernel ends
            if (!__ballot(anyw != 0u)) continue;
#pragma unroll
            for (int j = 0; j < kNbBlocksPerWave * kAnyWords / 2; ++j) {   // record 64 j + lane of the turn: bit (lane & 31) of word 2 j + (lane >> 5)
                const uint32_t wv = (uint32_t)__shfl((int)anyw, 2 * j + (lane >> 5));
                const bool marked = (wv >> (lane & 31)) & 1u;
                const unsigned long long hm = __ballot(marked);
                if (hm) {
                    if (marked) rq[qrn + lane_prefix(hm)] = (uint32_t)(turn * (kNbBlocksPerWave * kAnyWords * 32) + 64u * (uint32_t)j + (uint32_t)lane);
                    qrn += (uint32_t)__popcll(hm);
                    __builtin_amdgcn_wave_barrier();
                    if (qrn >= 64u) {
                        take_n_records(64u);
                        const uint32_t rest = qrn - 64u;
                        const uint32_t t = (uint32_t)lane < rest ? rq[64 + lane] : 0u;
                        __builtin_amdgcn_wave_barrier();
                        if ((uint32_t)lane < rest) rq[lane] = t;
                        __builtin_amdgcn_wave_barrier();
                        qrn = rest;
                    }
                }
            }
        }
        if (qrn) take_n_records(qrn);
        while (qp) n_batch();
    }
}

template <bool STATS, int KT, bool SPARSE>
__global__ __launch_bounds__(kL2Block) void level2_kernel(ScanArgs a) {
    __shared__ unsigned long long queue_c[kL2Waves * kL2QueueCap];   // slow-path queue: canonical k-mer | orientation << 62 | stat_only << 63
    __shared__ unsigned int rec_q[kL2Waves * kL2QueueCap];
    __shared__ unsigned long long kmer_q[kL2Waves * kL2KmerCap];     // record << 16 | k-mer index in the record
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned long long* const q = queue_c + wave * kL2QueueCap;
    unsigned int* const rq = rec_q + wave * kL2QueueCap;
    unsigned long long* const kq = kmer_q + wave * kL2KmerCap;

    const unsigned int* refw = a.ref_words + kRefPadWords;
    const int k = KT ? KT : a.k;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;  // k <= 31
    const uint32_t km1 = (uint32_t)k - 1u;
    const int omin = a.v_omin, span = a.v_span;
    unsigned long long* const v_counters = a.counters + a.v_off;
    const uint32_t last_word = a.stride_words - 1u;
    const IndexView& ix = *a.ixp;
    const KmerTable kt{STATS ? a.ktab_keys : nullptr, a.ktab_cnt, a.ktab_log2, a.ktab_overflow, a.mate};   // STATS = full_kmer_stats
    const DirtyAns* const answers = ix.dirty_ans;
    const uint8_t* const cflags = ix.cell_flags;

    // slow path: +1 on the E counter of reference k-mer `id` read in orientation `isrc`
    auto count_exact = [&](bool hit, uint32_t /*cell*/, uint32_t id, uint32_t isrc, uint32_t /*rc_first*/) __attribute__((always_inline)) {
        if (hit) { if constexpr (SPARSE) touch(a.touch_e, id); atomicAdd(a.counters + 2 * (size_t)id + isrc, 1ull); }
        BK_DBG(a, 8, hit, 1);
    };
    uint32_t qn = 0;   // wave-uniform fill of the slow-path queue
    SlowPipe pipe;
    pipe.touch_v = a.touch_v; pipe.touch_p = a.touch_p;
#ifdef BK_TESTING
    pipe.dbg = a.dbg;
#endif
    // resolve up to 64 queued k-mers: every stage's loads of the whole batch are issued together
    auto slow_batch = [&]() __attribute__((always_inline)) {
        const uint32_t nb = min(qn, 64u);
        pipe.start(q, nb, lane, ix);
        const uint32_t rest = qn - nb;
        const unsigned long long tc = ((uint32_t)lane < rest) ? q[64 + lane] : 0ull;
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < rest) q[lane] = tc;
        __builtin_amdgcn_wave_barrier();
        qn = rest;
        pipe.finish(ix, v_counters, count_exact, kt);
    };

    uint64_t n_records = a.n_records;
    if (a.n_records_dev) {
        const uint64_t nd = *a.n_records_dev;
        n_records = nd > a.rec_base ? min(nd - a.rec_base, a.n_records) : 0ull;
    }
    const uint32_t* const words0 = a.words + a.rec_base * a.stride_words;
    const uint32_t nw = a.l2_words;

    // the first n (<= 64) k-mers of the queue, one per lane
    auto process_kmers = [&](uint32_t n) __attribute__((always_inline)) {
        const bool on = (uint32_t)lane < n;
        const unsigned long long e = on ? kq[lane] : 0ull;
        const uint32_t rec = (uint32_t)(e >> 16), s = (uint32_t)e & 0xffffu;
        const uint2 dgf = on ? a.l2_diag[rec] : make_uint2(0u, 0u);
        const int32_t dg = (int32_t)dgf.x;
        const bool fwd = dgf.y & 1u, ok = on && (dgf.y & 2u) != 0u;
        const uint32_t* __restrict__ w = words0 + (uint64_t)rec * a.stride_words;
        const uint64_t g = read_symbols_at(w, s, last_word) & kmask;             // base t of the k-mer at bits 2t
        const uint64_t rr = ~g & kmask;                                          // its reverse complement, first base on top
        const uint64_t ff = rev2_64(g) >> (64 - 2 * k);                          // the k-mer, first base on top
        const bool lt = ff < rr;                                                 // lcb.rs:90-94
        const uint64_t c = lt ? ff : rr;
        const uint32_t isrc = lt ? 0u : 1u;
        // along the diagonal: the reference as the read sees it at the k-mer's cell, the cell's id and flags
        const int32_t cell = ok ? (fwd ? dg + (int32_t)s : dg - (int32_t)s) : 0;
        const uint64_t ra = fwd ? symbols_at(refw, cell) : ~rev2_64(symbols_at(refw, cell + (int32_t)km1 - 31));
        const uint32_t id = a.id_at[cell];
        const uint32_t fl = ok ? (uint32_t)cflags[cell] : 0u;
        const uint64_t da = (g ^ ra) & kmask;
        const uint64_t dbits = (da | (da >> 1)) & 0x5555555555555555ull;
        const uint32_t n_diff = ok ? (uint32_t)__popcll(dbits) : 99u;
        const bool has = (fl & 3u) != 0u;                                        // a reference k-mer starts at the cell (id valid)
        BK_DBG(a, 4, on, 1);
        bool slow = on;
        bool stat_only = false;
        // a V counter of the reference k-mers' part: +1 at vp, -1 at vm (0xffffffff: none)
        uint32_t vp = 0xffffffffu, vm = 0xffffffffu;
        if (has && n_diff == 0u) {
            if constexpr (SPARSE) touch(a.touch_e, id);
            atomicAdd(a.counters + 2 * (size_t)id + isrc, 1ull);                // the read k-mer IS the cell's reference k-mer
            slow = false;
            BK_DBG(a, 8, true, 1);
        } else if (has && n_diff == 1u) {
            const uint32_t t = (uint32_t)__builtin_ctzll(dbits) >> 1;           // position of the difference in the read k-mer
            const uint32_t br = (uint32_t)(g >> (2u * t)) & 3u;                 // ... and the read's base there
            const uint32_t o = fwd ? t : km1 - t;                               // offset along the reference from the cell
            const uint32_t bfw = fwd ? br : 3u - br;                            // base on the reference's forward strand
            const uint32_t alt = ((uint32_t)(da >> (2u * t)) & 3u) - 1u;        // which of the three other bases: read XOR reference (bk_device.h)
            if (fl & kCellClean) {
                // a clean reference k-mer of known id: provably not a reference k-mer, and that k-mer is its only possible
                // neighbour (bk_device.h, amb) -- a single-k-mer S run (v_point)
                const uint32_t oo = o - (uint32_t)omin;
                if (oo < (uint32_t)span && !BK_ABLATE(a, 2)) {                  // offsets outside the layout touch no window bucket
                    vp = (uint32_t)(v_row_base(id + oo, alt, fwd ? 0u : 1u, span) + oo);
                    if (oo + 1u < (uint32_t)span) vm = vp + 1u;                 // (slot `span` is never read)
                }
                slow = false;
                BK_DBG(a, 5, true, 1);
            } else if (answers) {
                // not clean (other reference k-mers nearby, or the other orientation of a repeat): worked out at create
                // (in the coordinates of the k-mer's first occurrence; a cell on the other strand of a reverse-complement repeat mirrors)
                const bool first_ori = (fl & kCellFirstOri) != 0u;
                const uint2 ans = *reinterpret_cast<const uint2*>(answers + ans_index(id, first_ori ? o : km1 - o, first_ori ? bfw : 3u - bfw, k));
                const uint32_t kind = ans.y & 3u;
                if (!(ans.y & kAnsNone)) {
                    slow = false;
                    if (kind == 1u) { if constexpr (SPARSE) touch(a.touch_e, ans.x >> 1); atomicAdd(a.counters + ans.x + isrc, 1ull); }
                    else if (kind == 2u) {
                        vp = ans.x + (((isrc ^ (ans.y >> 2)) & 1u) ? (uint32_t)span + 1u : 0u);
                        if (ans.y & 8u) vm = vp + 1u;
                    } else if (kind == 3u) {
                        if constexpr (SPARSE) touch(a.touch_p, (uint32_t)((ans.x - v_real_len(ix.n_full, span)) >> 3));
                        atomicAdd(v_counters + ans.x + isrc, 1ull);
                    }
                    else if (STATS) { slow = true; stat_only = true; }          // touches nothing: only the statistics table wants it
                    BK_DBG(a, 16, true, 1);
                }
            }
        } else if (has && n_diff == 2u && !(fl & kCellClean3) && (fl & kCellIso23) && answers && !STATS && ix.n_u == ix.n_full) {
            // (no pseudo k-mers -- k = 31 --: they are stored as they come, not canonical, and the answer table's "is a reference
            // k-mer" goes by the stored value: "u with one difference" can BE a pseudo k-mer's other strand and not say so)
            // two differences from a reference k-mer u that has no other form at distance 2 or 3 (kCellIso23): the read k-mer can only
            // equal or neighbour "u with one of the two differences".  If the answer table says neither of those two is a reference
            // k-mer, it touches nothing; if one is, the slow path sorts it out (rare: it is that k-mer's neighbour).
            const bool first_ori = (fl & kCellFirstOri) != 0u;
            bool member = false, none = false;
            uint64_t dleft = dbits;
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const uint32_t t = (uint32_t)__builtin_ctzll(dleft) >> 1;
                dleft &= dleft - 1ull;
                const uint32_t br = (uint32_t)(g >> (2u * t)) & 3u;
                const uint32_t o = fwd ? t : km1 - t;
                const uint32_t bfw = fwd ? br : 3u - br;
                const uint2 ans = *reinterpret_cast<const uint2*>(answers + ans_index(id, first_ori ? o : km1 - o, first_ori ? bfw : 3u - bfw, k));
                none |= (ans.y & kAnsNone) != 0u;
                member |= (ans.y & 3u) == 1u;
            }
            if (!none && !member) { slow = false; BK_DBG(a, 6, true, 1); }
        } else if (has && n_diff == 2u && (fl & kCellClean3)) {
            // two differences from a reference k-mer that has no other reference k-mer form within distance 3: neither a
            // reference k-mer nor one base away from one (triangle inequality) -- it touches nothing
            slow = STATS;
            stat_only = true;
            BK_DBG(a, 6, true, 1);
        }
        {
            // +1 / -1 pairs on one counter cancel: the -1 of a k-mer is the +1 of the next k-mer of its run (a read along the
            // reference: offsets fall as the k-mers advance) or of the previous one (against it).  Only k-mers of one record pair
            // up, each +1 / -1 with at most one partner, and both lanes of a pair come to the same conclusion.
            const uint32_t vp_up = (uint32_t)__shfl_up((int)vp, 1), vm_up = (uint32_t)__shfl_up((int)vm, 1), rec_up = (uint32_t)__shfl_up((int)rec, 1);
            const uint32_t vp_dn = (uint32_t)__shfl_down((int)vp, 1), vm_dn = (uint32_t)__shfl_down((int)vm, 1), rec_dn = (uint32_t)__shfl_down((int)rec, 1);
            const bool same_up = lane > 0 && rec_up == rec, same_dn = lane < 63 && rec_dn == rec;
            const bool p_gone = fwd ? (same_dn && vm_dn == vp) : (same_up && vm_up == vp);
            const bool m_gone = fwd ? (same_up && vp_up == vm) : (same_dn && vp_dn == vm);
            const bool p_live = vp != 0xffffffffu && !p_gone, m_live = vm != 0xffffffffu && !m_gone;
            if (SPARSE && (p_live || m_live))      // (vp and vm lie in one row; row = counter / row length, by the reciprocal)
                touch(a.touch_v, (uint32_t)__umul64hi((unsigned long long)(p_live ? vp : vm), a.rl_recip));
            if (p_live) atomicAdd(v_counters + vp, 1ull);
            if (m_live) atomicAdd(v_counters + vm, ~0ull);
        }
        if (BK_ABLATE(a, 3)) slow = false;
        // Before the slow path: can it be a reference k-mer, or one base from one, at all?  Only if one of its halves is a reference
        // k-mer's half (pigeonhole; the two directories the slow path walks are keyed by exactly these halves of the canonical k-mer):
        // two bit tests in the halves' presence filters (HalfView::bits) -- both "absent" is proof that it touches nothing.  What
        // comes here are reads that are not from the reference (every k-mer of a read without a diagonal) and k-mers with two
        // sequencing errors where no cell is isolated (many related genomes: three in four of the slow path's k-mers found nothing
        // there after eight random lines each).
        if (slow && !stat_only && ix.lo.bits && ix.hi.bits && !BK_ABLATE(a, 21)) {
            const int lo_bits = 2 * ix.lo_bases;
            const uint64_t lo = c & ((1ull << lo_bits) - 1ull), hi = c >> lo_bits;
            const uint32_t bl = half_bit_index(lo, ix.lo.bits_log2, ix.lo.bits_exact), bh = half_bit_index(hi, ix.hi.bits_log2, ix.hi.bits_exact);
            const uint32_t wl = ix.lo.bits[bl >> 5], wh = ix.hi.bits[bh >> 5];
            if (!(((wl >> (bl & 31u)) | (wh >> (bh & 31u))) & 1u)) {
                stat_only = true;
                slow = STATS;          // (full_kmer_stats still wants it in the statistics table)
                BK_DBG(a, 18, true, 1);
            }
        }
        const unsigned long long mm = __ballot(slow);
        if (mm) {
            BK_DBG(a, 7, slow, 1);
            if (slow) q[qn + lane_prefix(mm)] = c | ((unsigned long long)isrc << 62) | (stat_only ? 1ull << 63 : 0ull);
            qn += (uint32_t)__popcll(mm);
            __builtin_amdgcn_wave_barrier();
            if (qn >= 64u) slow_batch();
        }
    };

    uint32_t qr = 0, qk = 0;   // wave-uniform fills of the record and k-mer queues
    const bool roll_ok = !STATS && ix.lo.bits && ix.hi.bits && !BK_ABLATE(a, 21) && !BK_ABLATE(a, 22);   // (take_records: reads marked whole are rolled)
    constexpr uint32_t kRollMin = 16;   // ... when a batch of 64 marked records holds at least this many
    auto take_kmers = [&]() __attribute__((always_inline)) {
        const uint32_t n = min(qk, 64u);
        process_kmers(n);   // reads kq[0, n)
        const uint32_t rest = qk - n;
        unsigned long long t[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) t[j] = (uint32_t)(64 * j + lane) < rest ? kq[64 * (j + 1) + lane] : 0ull;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 3; ++j) if ((uint32_t)(64 * j + lane) < rest) kq[64 * j + lane] = t[j];
        __builtin_amdgcn_wave_barrier();
        qk = rest;
    };
    // 64 marked records, one per lane: their bitmap words (cleared as they are taken); the marked k-mers of a word go into the
    // k-mer queue lane after lane, each lane's in rising order
    auto take_records = [&]() __attribute__((always_inline)) {
        const uint32_t n = min(qr, 64u);
        const uint32_t rec = (uint32_t)lane < n ? rq[lane] : 0xffffffffu;
        {
            const uint32_t rest = qr - n;
            const uint32_t t = (uint32_t)lane < rest ? rq[64 + lane] : 0u;
            __builtin_amdgcn_wave_barrier();
            if ((uint32_t)lane < rest) rq[lane] = t;
            __builtin_amdgcn_wave_barrier();
            qr = rest;
        }
        unsigned int* row = a.l2_bits + (size_t)(rec == 0xffffffffu ? 0u : rec) * nw;
        // Reads marked whole -- reads without a diagonal: not from the reference (the host's, a contaminant's: most of a real sample),
        // or the few the seeds missed -- cost the path below a dozen loads per k-mer to find that nearly all touch nothing.  When
        // a batch of records holds enough of them (their first 64 marks are all set: no run of marks of a read with a diagonal is
        // that long), they are ROLLED first, each by its lane: the canonical k-mer from its predecessor in a handful of
        // instructions, its two halves looked up in the presence filters (HalfView::bits; see process_kmers) -- a k-mer neither
        // of whose halves is a reference k-mer's half touches nothing and is not listed.  A sample of the reference's own reads
        // never gets here (a wave would walk a read's length for a handful of records); full_kmer_stats wants every k-mer in its
        // table and takes no short cut.
        if (roll_ok && nw >= 2u) {
            const bool valid = rec != 0xffffffffu;
            const bool whole = valid && row[0] == 0xffffffffu && row[1] == 0xffffffffu;
            if ((uint32_t)__popcll(__ballot(whole)) >= kRollMin) {
                const uint32_t rec0 = valid ? rec : 0u;
                const uint32_t len = whole ? (uint32_t)(a.lens + a.rec_base)[rec0] : 0u;
                const uint32_t* __restrict__ w = words0 + (uint64_t)rec0 * a.stride_words;
                const uint32_t max_len = wave_max(len);
                const int lo_bits = 2 * ix.lo_bases;
                const uint64_t lo_mask = (1ull << lo_bits) - 1ull;
                uint64_t g = 0ull, ff = 0ull;   // the k-mer that ends at base i: base t at bits 2 t / first base on top (process_kmers' g and ff)
                uint32_t keep = 0u, x = 0u;
                // four bases at a time: their eight filter words are asked for together and looked at together (the filters' words come
                // one cache line per lane: the loads are bound by the CU's line rate, not by their latency -- a dozen waves keep them going)
                for (uint32_t i0 = 0; i0 < max_len; i0 += 4u) {   // (wave-uniform)
                    if ((i0 & 15u) == 0u) x = w[min(i0 >> 4, last_word)];
                    uint32_t wl[4], wh[4], sl = 0u, shh = 0u;   // the filter words; which bit of each (5 bits a k-mer)
#pragma unroll
                    for (uint32_t t = 0; t < 4u; ++t) {
                        const uint32_t b = (x >> (2u * ((i0 & 12u) + t))) & 3u;
                        g = (g >> 2) | ((uint64_t)b << (2 * (k - 1)));
                        ff = ((ff << 2) | b) & kmask;
                        const uint64_t rr = ~g & kmask;
                        const uint64_t c = ff < rr ? ff : rr;                                      // lcb.rs:90-94
                        const uint64_t lo = c & lo_mask, hi = c >> lo_bits;
                        const uint32_t bl = half_bit_index(lo, ix.lo.bits_log2, ix.lo.bits_exact), bh = half_bit_index(hi, ix.hi.bits_log2, ix.hi.bits_exact);
                        wl[t] = ix.lo.bits[bl >> 5]; wh[t] = ix.hi.bits[bh >> 5];                  // (any k-mer's: the filters are readable all over; the first k - 1 of a read are not looked at)
                        sl |= (bl & 31u) << (5u * t); shh |= (bh & 31u) << (5u * t);
                    }
#pragma unroll
                    for (uint32_t t = 0; t < 4u; ++t) {
                        const uint32_t i = i0 + t;                  // the base that came in: k-mer s = i - (k - 1) ends there
                        if (i < km1) continue;                      // (wave-uniform)
                        const uint32_t s2 = i - km1;
                        const uint32_t pl = wl[t] >> ((sl >> (5u * t)) & 31u), ph = wh[t] >> ((shh >> (5u * t)) & 31u);
                        keep |= ((pl | ph) & 1u) << (s2 & 31u);
                        if ((s2 & 31u) == 31u || i + 1u >= max_len) {   // (wave-uniform) a word of the record's marks is through
                            const uint32_t wi = s2 >> 5;
                            if (whole && wi < nw) {   // (the lane's own record: it reads these words again below, in program order)
                                const uint32_t old = row[wi];
                                BK_DBG(a, 18, (old & ~keep) != 0u, __popc(old & ~keep));
                                if (old & ~keep) row[wi] = old & keep;
                            }
                            keep = 0u;
                            if (i + 1u >= max_len) break;
                        }
                    }
                }
            }
        }
        for (uint32_t w0 = 0; w0 < nw; w0 += 4u) {
            uint32_t bw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bw[j] = (rec != 0xffffffffu && w0 + j < nw) ? row[w0 + j] : 0u;
                if (bw[j]) row[w0 + j] = 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t bits = bw[j];
                while (__ballot(bits != 0u)) {
                    // as many whole lanes' k-mers as the queue has room for (at least 64 entries are free here)
                    const uint32_t p = (uint32_t)__popc(bits);
                    uint32_t incl = p;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off); if (lane >= off) incl += t; }
                    const uint32_t room = (uint32_t)kL2KmerCap - qk;
                    const bool fits = p != 0u && incl <= room;
                    const uint32_t base = qk + incl - p;
                    if (fits) {
                        uint32_t x = bits, i = 0;
                        while (x) {
                            const uint32_t bpos = (uint32_t)__builtin_ctz(x);
                            x &= x - 1u;
                            kq[base + i++] = ((unsigned long long)rec << 16) | ((w0 + (uint32_t)j) * 32u + bpos);
                        }
                        bits = 0u;
                    }
                    const unsigned long long fm = __ballot(fits);
                    // total taken = inclusive sum of the last fitting lane (fitting lanes form a prefix of the lanes with bits)
                    const uint32_t taken = fm ? (uint32_t)__shfl((int)incl, 63 - __builtin_clzll(fm)) : 0u;
                    qk += taken;
                    __builtin_amdgcn_wave_barrier();
                    while (qk >= 64u) take_kmers();
                }
            }
        }
    };

    const uint64_t n_any = (n_records + 31) / 32;                    // words of l2_any / n_any
    const uint64_t n_blk = (n_any + kAnyWords - 1) / kAnyWords;

    // (ScanArgs::l2_plan: with few records marked, few workgroups -- the others leave at once)
    const uint32_t g_work = a.l2_plan ? min((uint32_t)gridDim.x, max(*a.l2_plan, 1u)) : (uint32_t)gridDim.x;
    if (blockIdx.x >= g_work) return;
    for (uint64_t blk = (uint64_t)blockIdx.x * kL2Waves + wave; blk < n_blk; blk += (uint64_t)g_work * kL2Waves) {
        const uint64_t i = blk * kAnyWords + lane;
        const uint32_t anyw = (lane < kAnyWords && i < n_any) ? a.l2_any[i] : 0u;
        if (anyw) a.l2_any[i] = 0u;     // taken: l2_any and l2_bits are all zero again when this kernel ends
        if (!__ballot(anyw != 0u)) continue;
#pragma unroll
        for (int j = 0; j < kAnyWords / 2; ++j) {   // record 64 j + lane of the block: bit (lane & 31) of word 2 j + (lane >> 5)
            const uint32_t wv = (uint32_t)__shfl((int)anyw, 2 * j + (lane >> 5));
            const bool marked = (wv >> (lane & 31)) & 1u;
            const unsigned long long hm = __ballot(marked);
            if (hm) {
                if (marked) rq[qr + lane_prefix(hm)] = (uint32_t)(blk * (kAnyWords * 32) + 64u * (uint32_t)j + (uint32_t)lane);
                qr += (uint32_t)__popcll(hm);
                __builtin_amdgcn_wave_barrier();
                if (qr >= 64u) take_records();
            }
        }
    }
    while (qr) take_records();
    while (qk) take_kmers();
    while (qn) slow_batch();
}

// Which genome does the sample look like?  One k-mer (the middle one) of each of the first records: votes[f] += 1 for every
// genome file the k-mer occurs in.  The engine puts the LDS window on the genome with the most votes.
__global__ __launch_bounds__(256) void pick_window_kernel(ScanArgs a, uint64_t n_probe, unsigned int* votes) {
    extern __shared__ unsigned int lvotes[];   // [n_files] this workgroup's tallies (thousands of k-mers vote for the same few genomes)
    for (int f = threadIdx.x; f < a.n_files; f += 256) lvotes[f] = 0u;
    __syncthreads();
    const IndexView& ix = *a.ixp;
    uint64_t n_records = a.n_records;
    if (a.n_records_dev) n_records = min((uint64_t)*a.n_records_dev, a.n_records);
    n_records = min(n_records, n_probe);
    const int k = a.k;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n_records; r += (uint64_t)gridDim.x * 256) {
        const uint32_t len = a.lens[r];
        if (len < (uint32_t)k) continue;
        const uint64_t g = read_symbols_at(a.words + r * a.stride_words, (len - (uint32_t)k) / 2u, a.stride_words - 1u) & kmask;
        const uint64_t rr = ~g & kmask, ff = rev2_64(g) >> (64 - 2 * k);
        const uint64_t c = ff < rr ? ff : rr;
        const uint4 e = *reinterpret_cast<const uint4*>(ix.kmer_pos + phf_pos(c, ix.pilots[phf_bucket(c, ix.log2nb)], ix.m, ix.log2nb, ix.log2p));
        if (((uint64_t)e.x | ((uint64_t)e.y << 32)) != c || (e.w & kIdMask) >= ix.n_full) continue;
        const uint32_t* oc = a.occ + (size_t)(e.w & kIdMask) * (uint32_t)a.n_files;
        for (int f = 0; f < a.n_files; ++f) if (oc[f] != 0xffffffffu) atomicAdd(&lvotes[f], 1u);
    }
    __syncthreads();
    for (int f = threadIdx.x; f < a.n_files; f += 256) if (lvotes[f]) atomicAdd(votes + f, lvotes[f]);
}
// ... and the choice itself, on the device (no host round trip between a sample's first push and its scan): the genome with the
// most votes (lowest id on ties), its first cell rounded down to a multiple of 64; `forced` >= 0 overrides (testing build)
__global__ void choose_window_kernel(const unsigned int* votes, int n_files, const uint32_t* file_cell_lo, int forced, uint32_t* win) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int best = 0;
    for (int f = 1; f < n_files; ++f) if (votes[f] > votes[best]) best = f;
    if (forced >= 0) best = forced < n_files ? forced : n_files - 1;
    win[0] = (uint32_t)best;
    win[1] = file_cell_lo[best] & ~63u;
}
void launch_pick_window(const ScanArgs& a, uint64_t n_probe, unsigned int* votes, const uint32_t* file_cell_lo, int forced, uint32_t* win,
                        hipStream_t stream) {
    hipLaunchKernelGGL(pick_window_kernel, dim3(64), dim3(256), (size_t)a.n_files * sizeof(unsigned int), stream, a, n_probe, votes);
    hipLaunchKernelGGL(choose_window_kernel, dim3(1), dim3(64), 0, stream, votes, a.n_files, file_cell_lo, forced, win);
}

// Empty window (n_fixed * 2 + 1 >= k, call.rs:1291-1300): no k-mer can touch the index.  KMC's total is still wanted, and with
// full_kmer_stats its distinct / kept k-mer counts: every k-mer of every record goes into the statistics table (one thread per
// record, rolling canonical k-mer -- this configuration maps nothing, speed is not a concern).
__global__ __launch_bounds__(256) void count_kmers_kernel(ScanArgs a) {
    uint64_t n_records = a.n_records;
    if (a.n_records_dev) n_records = min((uint64_t)*a.n_records_dev, a.n_records);
    const KmerTable kt{a.ktab_keys, a.ktab_cnt, a.ktab_log2, a.ktab_overflow, a.mate};
    const int k = a.k;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;
    unsigned long long sum = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n_records; r += (uint64_t)gridDim.x * 256) {
        const uint32_t len = a.lens[r];
        if (len < (uint32_t)k) continue;
        sum += len - (uint32_t)k + 1u;
        if (!kt.keys) continue;
        const uint32_t* w = a.words + r * a.stride_words;
        uint64_t f = 0, rc = 0;
        for (uint32_t i = 0; i < len; ++i) {
            const uint64_t b = (w[i >> 4] >> (2u * (i & 15u))) & 3u;
            f = ((f << 2) | b) & kmask;
            rc = (rc >> 2) | ((3ull - b) << (2 * (k - 1)));
            if (i + 1u >= (uint32_t)k) ktab_insert(kt, f < rc ? f : rc, f < rc ? 0u : 1u, 1u);   // lcb.rs:90-94
        }
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0 && sum && a.kmer_total) atomicAdd(a.kmer_total, sum);
}
void launch_count_kmers(const ScanArgs& a, hipStream_t stream) {
    if (a.n_records == 0) return;
    hipLaunchKernelGGL(count_kmers_kernel, dim3((unsigned)std::min<uint64_t>(1024, (a.n_records + 255) / 256)), dim3(256), 0, stream, a);
}

// (12 KB of the CU's 160 KB are left free: a workgroup of another stream's finalize kernels fits next to the scan's)
size_t scan_lds_budget() { return 148u * 1024u - 64u - kScanLdsFixed - sizeof(unsigned int); }
// LDS bytes of the per-cell arrays the scan stages for `cells` cells (reference 2 bits, two 1-bit arrays, one block entry per 64
// cells; paddings)
size_t scan_ref_lds_bytes(uint32_t cells) {
    return ((size_t)(kRefPadWords + (cells + 15) / 16 + kRefBackWords) + 2u * (size_t)(kBitPadWords + (cells + 31) / 32 + kBitBackWords) +
            2u * ((size_t)(cells + 63) / 64 + 2u)) * sizeof(unsigned int);
}
size_t scan_lds_bytes(uint32_t n_lds_bins, bool ref_in_lds, uint32_t total_cells) {
    return kScanLdsFixed + ((size_t)n_lds_bins + 1) * sizeof(unsigned int) +
           (ref_in_lds ? scan_ref_lds_bytes(std::min(n_lds_bins, total_cells)) : 0);
}
int scan_ref_pad_words() { return kRefPadWords; }
int scan_ref_back_words() { return kRefBackWords; }
int scan_bit_pad_words() { return kBitPadWords; }
int scan_bit_back_words() { return kBitBackWords; }

uint32_t scan_grid(uint64_t n_records, int n_cus) {
    const uint64_t want = (n_records + kScanBlock - 1) / kScanBlock;
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)n_cus));
}
// records one launch may take so that no workgroup sees more than kMaxRecordsPerGroup of them
uint32_t scan_l2_words(uint32_t stride_words, int k) { const uint32_t b = stride_words * 16u; return b >= (uint32_t)k ? (b - (uint32_t)k + 1u + 31u) / 32u : 1u; }
uint64_t scan_max_records(uint32_t grid) { return (uint64_t)grid * (kMaxRecordsPerGroup - kScanBlock - 64); }

hipError_t launch_scan_count(const ScanArgs& a, uint32_t grid, hipStream_t stream) {
    if (a.n_records == 0 || a.W <= 0) return hipSuccess;
    if (a.n_records > scan_max_records(grid)) return hipErrorInvalidValue;
    const size_t lds = scan_lds_bytes(a.n_lds_bins, a.ref_in_lds != 0, a.total_cells);
    void (*kern)(ScanArgs);
#define BK_PICK(KT) (!a.touch_v ? (a.ref_in_lds ? scan_count_kernel<true, KT, false> : scan_count_kernel<false, KT, false>) \
                                : (a.ref_in_lds ? scan_count_kernel<true, KT, true> : scan_count_kernel<false, KT, true>))
    kern = a.k == 21 ? BK_PICK(21) : a.k == 31 ? BK_PICK(31) : BK_PICK(0);
#undef BK_PICK
    {   // the dynamic-LDS limit of a kernel variant is raised once (per process and device), not at every launch
        static std::mutex mu;
        static std::vector<std::pair<std::pair<const void*, int>, size_t>> have;   // ((kernel, device), limit set)
        int dev = 0;
        (void)hipGetDevice(&dev);
        const void* fn = reinterpret_cast<const void*>(kern);
        std::lock_guard<std::mutex> lock(mu);
        size_t* cur = nullptr;
        for (auto& h : have) if (h.first.first == fn && h.first.second == dev) cur = &h.second;
        if (!cur || *cur < lds) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            if (cur) *cur = lds; else have.push_back({{fn, dev}, lds});
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kScanBlock), lds, stream, a);
    return hipGetLastError();
}

// How many workgroups of level2_kernel a launch's marks are worth: a wave gathers 64 marked records before it looks at their
// k-mers, and a sample of the reference's own reads marks one read in a hundred -- two thousand waves each end up with a
// handful, and every one of them holds a quarter of its SIMD's registers for the whole chain of loads, on a CU that a sibling
// sample's scan workgroup (which needs the CU to itself) is waiting for.  One workgroup per 256 marked records instead, between
// ScanArgs::l2_min_grid and the grid: config 2 with four samples in flight 12.3 -> 12.9 G reads/s; reads with 5 % errors or not
// from the reference mark nearly every record and keep the whole grid (a fixed small grid cost them 25-35 %).
__global__ __launch_bounds__(1024) void l2_plan_kernel(const unsigned int* __restrict__ l2_any, uint64_t n_words, unsigned int* __restrict__ plan, uint32_t g_min) {
    __shared__ unsigned int part[16];
    unsigned int n = 0;
    for (uint64_t i = threadIdx.x; i < n_words; i += 1024u) n += (unsigned int)__popc(l2_any[i]);
#pragma unroll
    for (int off = 32; off; off >>= 1) n += (unsigned int)__shfl_xor((int)n, off);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int t = 0;
        for (int w = 0; w < 16; ++w) t += part[w];
        plan[0] = max(g_min, (t + 255u) / 256u);
    }
}

hipError_t launch_level2(const ScanArgs& a, int n_cus, hipStream_t stream) {
    if (a.n_records == 0 || a.W <= 0) return hipSuccess;
    const bool stats = a.ktab_keys != nullptr;
    void (*kern)(ScanArgs);
    if (!a.n_direct) {   // the N runs first (they mark k-mers for the kernel below); ScanArgs::n_direct: the scan left none
#define BK_PICKN(KT) (!a.touch_v ? (stats ? nbatch_kernel<true, KT, false> : nbatch_kernel<false, KT, false>) \
                                 : (stats ? nbatch_kernel<true, KT, true> : nbatch_kernel<false, KT, true>))
        kern = a.k == 21 ? BK_PICKN(21) : a.k == 31 ? BK_PICKN(31) : BK_PICKN(0);
#undef BK_PICKN
        const uint64_t turns = (a.n_records + 32 * kAnyWords * kNbBlocksPerWave - 1) / (32 * kAnyWords * kNbBlocksPerWave);
        const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((turns + kL2Waves - 1) / kL2Waves, (uint64_t)n_cus * 8));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kL2Block), 0, stream, a);
    }
#define BK_PICK(KT) (!a.touch_v ? (stats ? level2_kernel<true, KT, false> : level2_kernel<false, KT, false>) \
                                : (stats ? level2_kernel<true, KT, true> : level2_kernel<false, KT, true>))
    kern = a.k == 21 ? BK_PICK(21) : a.k == 31 ? BK_PICK(31) : BK_PICK(0);
#undef BK_PICK
    const uint64_t blks = (a.n_records + 32 * kAnyWords - 1) / (32 * kAnyWords);     // a wave takes kAnyWords words of l2_any at a time
    // One genome file: a thousandth of the k-mers are marked, and four thousand waves that find nothing to do still take their turn
    // on the CUs the sibling samples' kernels are waiting for (three samples in flight: 10.5 -> 10.8 G reads/s with a quarter of the
    // waves; alone the kernel takes what it took).  Several files: the marks are Level 2's real work (config 3 lost 8 % that way).
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((blks + kL2Waves - 1) / kL2Waves, (uint64_t)n_cus * (a.n_files == 1 ? 2 : 8)));
    if (a.l2_plan) hipLaunchKernelGGL(l2_plan_kernel, dim3(1), dim3(1024), 0, stream, (const unsigned int*)a.l2_any, (a.n_records + 31) / 32, a.l2_plan, a.l2_min_grid);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kL2Block), 0, stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K1c
// slab[b][c] = (reads of workgroup b that contain the reference k-mer of cell c, along the reference) + 65536 * (against it)
// E[2 id_at[c] + rc]     += sum over slabs of the low half   (rc = the cell's k-mer was reverse-complemented to become canonical:
// E[2 id_at[c] + 1 - rc] += sum over slabs of the high half    a read along the reference has the k-mer as written), and
__global__ __launch_bounds__(256) void fold_kernel(FoldArgs f) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * 256;
    // blockIdx.y splits the slabs into groups so that the 31 MB of slabs are streamed by the whole chip; each
    // group adds its partial sums with one u64 atomic per non-zero half
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
            const uint64_t cell = (f.win_dev ? f.win_dev[1] : f.win_lo) + i;
            const uint32_t id = f.id_at[cell];   // a counted cell always has a reference k-mer
            const uint32_t rc = ((f.cell_codes[cell >> 4] >> (2 * (cell & 15))) & 3u) == 2u ? 1u : 0u;
            touch(f.touch_e, id);
            if (s0) atomicAdd(f.counters + 2 * (size_t)id + rc, s0);
            if (s1) atomicAdd(f.counters + 2 * (size_t)id + (1u - rc), s1);
        }
    }
}

void launch_fold(const FoldArgs& f, hipStream_t stream) {
    const uint64_t work = f.n_lds_bins;
    if (work == 0) return;
    uint64_t blocks = (work + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    const unsigned groups = std::max(1u, std::min(8u, f.n_slabs / 8));
    hipLaunchKernelGGL(fold_kernel, dim3((unsigned)blocks, groups), dim3(256), 0, stream, f);
}

// ------------------------------------------------------------------------------------------------ K2
// Second pass of bk_params.pileup_selected_only: the entries of a bucket are sorted by genome file -- the first one of `file`
// (or cnt), by bisection
__device__ __forceinline__ uint32_t first_of_file(const DevEntry* __restrict__ ent, uint32_t cnt, int file) {
    uint32_t lo = 0, hi = cnt;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((int)ent[mid].file < file) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// File bitmaps (IndexView::ent_files) of a wave's k-mers -> lstats[file * 3 + COL] += number of lanes whose bitmap holds the file:
// one ballot per file, the count written into the file's own lane (v_writelane), one LDS add per word of 32 files.  Every lane of
// the wave calls it (lanes without a k-mer with an all-zero bitmap).
template <int B>
__device__ __forceinline__ void write_lane(int& acc, int nb) { asm("v_writelane_b32 %0, %1, %2" : "+v"(acc) : "s"(nb), "n"(B)); }
template <int... B>
__device__ __forceinline__ void tally_word(uint32_t wd, int& acc, std::integer_sequence<int, B...>) {
    (write_lane<B>(acc, __popcll(__ballot((wd & (1u << B)) != 0u))), ...);
}
template <bool ATOMIC, uint32_t COL = 1u>
__device__ __forceinline__ void tally_files(const uint4& fb, uint32_t* lstats, uint32_t lane, uint32_t n_files) {
    if (!__ballot(files_any(fb))) return;
#pragma unroll
    for (uint32_t w = 0; w < 4u; ++w) {
        const uint32_t wd = w == 0u ? fb.x : w == 1u ? fb.y : w == 2u ? fb.z : fb.w;
        if (w * 32u >= n_files || !__ballot(wd != 0u)) continue;
        int acc = 0;
        tally_word(wd, acc, std::make_integer_sequence<int, 32>{});
        const uint32_t f = w * 32u + lane;
        if (lane < 32u && acc && f < n_files) {
            if (ATOMIC) atomicAdd(&lstats[f * 3u + COL], (uint32_t)acc); else lstats[f * 3u + COL] += (uint32_t)acc;
        }
    }
}

// stats / present / kept / distinct += column sums of the partials rows written by the finalize workgroups
__global__ __launch_bounds__(256) void finalize_reduce_kernel(FinalizeArgs a, int n_rows, unsigned long long* zero_p, size_t zero_n) {
    __shared__ unsigned long long wave_sums[4];
    const int n3 = a.ix.n_files * 3, cols = n3 + 2;
    // (the last kernel of a sample's last pass also zeroes the E part of the plane behind the sample: dense planes, see K2a's clear_v)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < zero_n; i += (size_t)gridDim.x * 256) zero_p[i] = 0ull;
    // (... and empties the regional finalize's list of reference k-mers that are not simple for the next pass: its readers are done)
    if (a.lean_n_list && blockIdx.x == 0 && threadIdx.x < 8) a.lean_n_list[threadIdx.x] = 0u;
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

// Votes of one workgroup gathered in LDS before they go to the pileup: many k-mers vote for the same cell (the k-mers that
// cover one reference position), and scattered 64-bit global atomics are what bounds finalize.  A small open-addressing
// table keyed by (strand, cell, base): #k-mers add up, depth takes the max; vt_flush writes every used slot with one pair of
// global atomics.  A vote that finds its neighbourhood of the table full goes to the pileup directly.
constexpr int kVoteSlots = 512;
constexpr uint32_t kVoteMaxEntries = 4;   // buckets with more BucketInfos than this vote directly
constexpr size_t kVoteLdsBytes = (size_t)kVoteSlots * (8 + 8 + 4 + 2) + 16;
struct VoteTable {
    unsigned long long* keys;   // [kVoteSlots], ~0 = free
    unsigned long long* mx;     // [kVoteSlots]
    unsigned int* cnt;          // [kVoteSlots]
    unsigned int* n_used;       // [2] slots taken since the last flush, alternating with the round's parity
    unsigned short* used;       // [kVoteSlots] ... and which
};
__device__ __forceinline__ VoteTable vt_make(unsigned char* lds) {
    VoteTable vt;
    vt.keys = reinterpret_cast<unsigned long long*>(lds);
    vt.mx = vt.keys + kVoteSlots;
    vt.cnt = reinterpret_cast<unsigned int*>(vt.mx + kVoteSlots);
    vt.n_used = vt.cnt + kVoteSlots;
    vt.used = reinterpret_cast<unsigned short*>(vt.n_used + 4);
    return vt;
}
__device__ __forceinline__ void vt_clear(const VoteTable& vt) {   // whole workgroup; caller synchronises
    for (int i = threadIdx.x; i < kVoteSlots; i += blockDim.x) { vt.keys[i] = ~0ull; vt.mx[i] = 0ull; vt.cnt[i] = 0u; }
    if (threadIdx.x == 0) { vt.n_used[0] = 0u; vt.n_used[1] = 0u; }
}
// the vote of call.rs:1327-1384 (see vote()), into the table
__device__ __forceinline__ void vt_vote(const VoteTable& vt, uint32_t par, const FinalizeArgs& a, const DevEntry& e, uint64_t c, uint32_t isrc,
                                        int k, unsigned long long v) {
    if (a.mode == 1 || (a.mode == 2 && (int)e.file != a.sel_file)) return;   // statistics pass / votes for the selected genome only
    uint32_t bit_idx;
    bool forward;
    if (e.canonical) { bit_idx = ((uint32_t)(c >> (2 * e.idx)) & 3u) ^ 3u; forward = isrc != 0; }
    else { bit_idx = (uint32_t)(c >> (2 * (k - 1 - e.idx))) & 3u; forward = isrc == 0; }
    const unsigned long long key = (((unsigned long long)e.cell * 4 + bit_idx) << 1) | (forward ? 0ull : 1ull);
    uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 55) & (kVoteSlots - 1);
    for (int probe = 0; probe < 8; ++probe) {
        const unsigned long long old = atomicCAS(&vt.keys[h], ~0ull, key);
        if (old == ~0ull) vt.used[atomicAdd(vt.n_used + par, 1u)] = (unsigned short)h;
        if (old == ~0ull || old == key) { atomicAdd(&vt.cnt[h], 1u); atomicMax(&vt.mx[h], v); return; }
        h = (h + 1) & (kVoteSlots - 1);
    }
    vote(a, e, c, isrc, k, v);
}
// whole workgroup, between two barriers; par = parity of the round that just voted (the next round uses the other counter)
__device__ __forceinline__ void vt_flush(const VoteTable& vt, uint32_t par, const FinalizeArgs& a) {
    __syncthreads();
    const unsigned int nu = vt.n_used[par];
    if (threadIdx.x == 0) vt.n_used[par ^ 1u] = 0u;
    for (unsigned int u = threadIdx.x; u < nu; u += blockDim.x) {
        const int i = vt.used[u];
        const unsigned long long key = vt.keys[i];
        const size_t cell = (size_t)(key >> 1);
        atomicAdd(a.pileup + ((key & 1ull) ? 3 : 2) * a.plane + cell, (unsigned long long)vt.cnt[i]);   // #kmers
        atomicMax(a.pileup + ((key & 1ull) ? 1 : 0) * a.plane + cell, vt.mx[i]);                        // depth
        vt.keys[i] = ~0ull; vt.mx[i] = 0ull; vt.cnt[i] = 0u;
    }
    __syncthreads();
    if (threadIdx.x == 0) { vt.n_used[0] = 0u; vt.n_used[1] = 0u; }
}

// The k-mer a V counter stands for: canonical form c (in the orientation of its reference neighbour), orientation the reads
// had it in, and its occurrence count n (reference k-mers: prefix sum of the row's difference array, bk_device.h).
__device__ __forceinline__ void v_kmer_of_counter(const IndexView& ix, const unsigned long long* __restrict__ vc, uint64_t vi,
                                                  uint32_t& p, uint32_t& t, uint64_t& c, uint32_t& isrc, unsigned long long& n) {
    const int k = ix.k;
    const uint64_t real_len = v_real_len(ix.n_full, ix.v_span);
    uint32_t bb;
    if (vi < real_len) {
        const uint32_t rl = (uint32_t)ix.v_span + 1u;
        const uint64_t row = vi / rl;
        const uint32_t oo = (uint32_t)(vi % rl);
        n = 0;
        if (vc) for (uint32_t x = 0; x <= oo; ++x) n += vc[row * rl + x];   // (null: the caller has the count)
        const uint32_t r6 = (uint32_t)(row % kVRowsPerPos);           // (alternative << 1) | direction
        p = (uint32_t)(row / kVRowsPerPos) - oo;
        const uint32_t rcid = (ix.amb[p] >> 1) & 1u;
        const int o = (int)oo + ix.v_omin;
        const int j = rcid ? k - 1 - o : o;
        const uint32_t own = (uint32_t)(ix.kmer_of[p] >> (2 * (k - 1 - j))) & 3u;   // the reference k-mer's own base there (canonical form)
        bb = own ^ ((r6 >> 1) + 1u);                                                // the alternative: XOR, the same on either strand
        isrc = (r6 & 1u) ^ rcid;
        t = (uint32_t)(j - ix.wstart);
    } else {
        const uint64_t x = vi - real_len;
        isrc = (uint32_t)x & 1u;
        bb = (uint32_t)(x >> 1) & 3u;
        p = ix.prow_id[x >> 3];
        t = ix.prow_t[x >> 3];
        n = vc ? vc[vi] : 0ull;
    }
    const int sh = 2 * (k - 1 - (ix.wstart + (int)t));
    c = (ix.kmer_of[p] & ~(3ull << sh)) | ((uint64_t)bb << sh);
}

// K2a: the V counters.  v_span lanes per row of the reference k-mers' part (lane = offset; the running sums of the row's
// difference array -- each k-mer's count -- by shuffles) and one thread per counter of the pseudo k-mers' part.  A kept non-reference k-mer almost
// always touches exactly one window bucket (the one its name says); then the whole of map_kmers for it is: vote once per
// BucketInfo of that bucket, and per genome file "variant" (or "perfect" if the file has exactly W entries there, which
// needs W == 1 or repeats).  K-mers that touch several buckets need cross-bucket per-genome totals; they are appended
// (by counter index) to `deferred` and mapped by K2b.
// A row also holds k-mers that cannot touch the index: the difference outside the window, or a k-mer whose canonical form
// lies on the other strand than its neighbour's (scan_count records what the reads contain, not what it means).  They are
// skipped -- with full_kmer_stats they join the k-mer statistics table, like every other k-mer that touches nothing.
// MANY: an index with file bitmaps (IndexView::slot_files: several genome files) -- the code for them is compiled into that
// instantiation only (the kernel sits at its register limit; the single-genome form is the benchmark's).
template <bool MANY>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void finalize_variant_kernel(FinalizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (a.mode == 2) { a.sel_file = *a.sel; if (a.sel_file < 0 && !a.clear_v) return; }   // (no genome selected: no votes, but the plane is still to be cleared)   // second pass: votes for the selected genome only
    const bool do_stats = a.mode < 2;                                        // (its statistics were tallied by the first pass; mode 3: votes only, every genome)
    const IndexView& ix = a.ix;
    uint32_t* lstats = reinterpret_cast<uint32_t*>(smem);   // [n_files][3] block-local tallies + 2 scratch words
    const VoteTable vt = vt_make(smem + (((size_t)ix.n_files * 3 + 2) * sizeof(uint32_t) + 15) / 16 * 16);
    for (int g = threadIdx.x; g < ix.n_files * 3 + 2; g += 256) lstats[g] = 0;
    vt_clear(vt);
    __syncthreads();

    const int k = ix.k;
    const uint64_t n_rows = v_real_rows(ix.n_full, ix.v_span);
    const uint64_t real_len = v_real_len(ix.n_full, ix.v_span);
    const uint32_t rl = (uint32_t)ix.v_span + 1u;
    unsigned long long* __restrict__ vc = const_cast<unsigned long long*>(a.counters) + ix.v_off;   // (written only under clear_v)
    const KmerTable kt{a.ktab_keys, a.ktab_cnt, a.ktab_log2, a.ktab_overflow, a.mate};
    unsigned int kept = 0, distinct = 0;
    // this shard's rows and pseudo counters (elem_lo / elem_hi are multiples of the row length, like v_off)
    const uint64_t vlo = a.elem_lo > ix.v_off ? a.elem_lo - ix.v_off : 0ull, vhi = a.elem_hi > ix.v_off ? a.elem_hi - ix.v_off : 0ull;
    const uint64_t row_lo = min(vlo / rl, n_rows), row_hi = min(vhi / rl, n_rows);
    const uint64_t px_lo = min(vlo > real_len ? vlo - real_len : 0ull, ix.n_prows * 8ull), px_hi = min(vhi > real_len ? vhi - real_len : 0ull, ix.n_prows * 8ull);

    // The k-mers of one row mostly vote for the same pileup cell (the reference position of the differing base): votes are
    // gathered per (plane, cell) and written when the target changes -- #k-mers += votes, depth = max(depth, counts).
    size_t acc_at[2] = {~(size_t)0, ~(size_t)0};
    unsigned long long acc_n[2] = {0, 0}, acc_max[2] = {0, 0};
    auto flush = [&](int f) {
        if (acc_at[f] != ~(size_t)0) {
            atomicAdd(a.pileup + (f ? 3 : 2) * a.plane + acc_at[f], acc_n[f]);   // #kmers
            atomicMax(a.pileup + (f ? 1 : 0) * a.plane + acc_at[f], acc_max[f]); // depth
        }
        acc_at[f] = ~(size_t)0; acc_n[f] = 0; acc_max[f] = 0;
    };
    // the vote of call.rs:1327-1384 (see vote()), gathered
    auto vote_acc = [&](const DevEntry& e, uint64_t c, uint32_t isrc, unsigned long long v) {
        if (a.mode == 1 || (a.mode == 2 && (int)e.file != a.sel_file)) return;
        uint32_t bit_idx;
        bool forward;
        if (e.canonical) { bit_idx = ((uint32_t)(c >> (2 * e.idx)) & 3u) ^ 3u; forward = isrc != 0; }
        else { bit_idx = (uint32_t)(c >> (2 * (k - 1 - e.idx))) & 3u; forward = isrc == 0; }
        const size_t cell = (size_t)e.cell * 4 + bit_idx;
        const int f = forward ? 0 : 1;
        if (acc_at[f] != cell) { flush(f); acc_at[f] = cell; }
        acc_n[f] += 1;
        acc_max[f] = acc_max[f] > v ? acc_max[f] : v;
    };

    // one kept k-mer: c = reference k-mer p with base changed at window position t, read in orientation isrc, n times
    auto map_one = [&](uint32_t p, uint32_t t, uint64_t c, uint32_t isrc, unsigned long long n, uint64_t vi) {
        if (do_stats) distinct += 1;
        if (n < a.ci || n > a.cx) return;                        // kmc -ci / -cx act on the true count
        if (do_stats) ++kept;
        const unsigned long long v = n > a.cs ? a.cs : n;       // kmc -cs: reported count saturates
        // It can touch a second window bucket only if another reference k-mer lies at Hamming distance 2 from u
        // (amb[p], precomputed); otherwise its one bucket is u's own bucket at t.  Ambiguous u: enumerate the
        // neighbours; several buckets -> general path (K2b).
        if (ix.amb[p] & 1u) {
            uint32_t jmask = 0;   // window positions at which c has a neighbouring reference k-mer
            for_each_neighbour(ix, c, [&](int jj, uint32_t, uint32_t) { jmask |= 1u << (jj - ix.wstart); });
            if (jmask != (1u << t)) {
                if (do_stats) { const unsigned int at = atomicAdd(a.n_deferred, 1u); a.deferred[at] = (uint32_t)vi; if (a.deferred_n) a.deferred_n[at] = n; }
                return;
            }
        }
        uint32_t off, cnt;
        DevEntry first{};
        if (p < ix.n_full) {
            const uint4 r = *reinterpret_cast<const uint4*>(ix.slot_rec + (size_t)p * ix.W + t);
            off = r.x; cnt = r.y;
            first.cell = r.z; first.file = (uint16_t)(r.w & 0xffffu); first.idx = (uint8_t)(r.w >> 16); first.canonical = (uint8_t)(r.w >> 24);
        } else {
            const uint32_t s = ix.slot_of[(size_t)p * ix.W + t];
            // (FinalizeArgs::gather: the votes through the buckets' own keys were gathered cell by cell -- only an alias slot is left)
            if (a.gather && !(ix.slot_alias && ((ix.slot_alias[s >> 5] >> (s & 31u)) & 1u))) return;
            off = ix.ent_off[s]; cnt = ix.ent_len[s];
            if (cnt) first = ix.entries[off];
        }
        // entries of one bucket are grouped by file (index build appends file by file): run lengths = hits per file
        uint32_t n_perfect = 0, perfect_file = 0;
        for (uint32_t q = 0; q < cnt;) {
            DevEntry en = q ? ix.entries[off + q] : first;
            const uint32_t file = en.file;
            uint32_t run = 0;
            for (;;) {
                vote_acc(en, c, isrc, v); ++run; ++q;
                if (q >= cnt) break;
                en = ix.entries[off + q];
                if (en.file != file) break;
            }
            if (!do_stats) continue;
            if (run == (uint32_t)ix.W) { atomicAdd(&lstats[file * 3 + 0], 1u); ++n_perfect; perfect_file = file; }
            else atomicAdd(&lstats[file * 3 + 1], 1u);
        }
        if (do_stats && n_perfect == 1) atomicAdd(&lstats[perfect_file * 3 + 2], 1u);
    };

    // v_span lanes per row, lane oo = offset: the row's prefix sums by shuffles, then every lane maps its own
    // k-mer.  A workgroup takes 8 rows with the same base and direction and q, q + 2, ..., q + 14: the k-mers of a row that
    // are canonical as written all vote for one pileup cell (the reference position the row stands for), and those that
    // were reverse-complemented vote for cells that rows q and q + 2 share (their vote mirrors the offset, call.rs:1331-1357)
    // -- the workgroup's vote table (LDS) merges both kinds before anything goes to the pileup.
    // A row holds v_span counts (the slot behind them is never read): as many rows share a wave as fit, lanes [g * v_span,
    // (g + 1) * v_span) take row g (17 counts: three rows per wave, one lane idle).
    const uint32_t lpr = (uint32_t)ix.v_span;                    // lanes per row (<= 32: k <= 31)
    const uint32_t gpw = 64u / lpr;                              // rows per wave
    const uint32_t lane64 = threadIdx.x & 63u;
    const uint32_t grp = lane64 / lpr;
    const uint32_t oo = lane64 - grp * lpr;
    const bool lane_on = grp < gpw;
    const uint32_t hw = (threadIdx.x >> 6) * gpw + grp;          // which of the workgroup's rows
    const uint32_t RPW = 4u * gpw;                               // rows per workgroup and unit
    const uint64_t nq = (uint64_t)ix.n_full + (uint32_t)ix.v_span;
    // sparse finalize: the touched rows from the list, RPW at a time, instead of every row of the plane
    const uint64_t n_listed = a.v_list ? a.n_list[0] : 0ull;
    const uint64_t n_units = a.gather ? 0ull                                      // (the reference k-mers' rows voted cell by cell: bk_gather.hip)
                             : a.v_list ? (n_listed + RPW - 1) / RPW
                                      : ((nq + 2 * RPW - 1) / (2 * RPW)) * 12;   // unit u: q block u / 12 (2 RPW values of q), parity (u / 6) & 1, (alternative, direction) u % 6
    uint32_t par = 0;
    const bool sparse_plane = ix.n_files > 1 && !a.v_list;
    for (uint64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        // (row indices fit 32 bits -- a plane of 2^32 counters is refused at create --: divisions by 6 on 32-bit words, not 64)
        uint32_t q, r6;
        uint64_t wk;
        bool in_row;
        if (a.v_list) {
            const uint64_t li = u * RPW + hw;
            in_row = lane_on && li < n_listed;
            const uint32_t w32 = in_row ? a.v_list[li] : 0u;
            q = w32 / kVRowsPerPos; r6 = w32 - q * kVRowsPerPos;
            wk = w32;
        } else {
            const uint32_t u32 = (uint32_t)u, u6 = u32 / 6u;    // (wave-uniform)
            r6 = u32 - u6 * 6u;
            const uint64_t qrow = (uint64_t)(u6 >> 1) * (2 * RPW) + (u6 & 1u) + 2ull * hw;
            wk = qrow * kVRowsPerPos + r6;
            in_row = lane_on && qrow < nq && wk >= row_lo && wk < row_hi;
            q = (uint32_t)qrow;
        }
        unsigned long long n = in_row ? vc[wk * rl + oo] : 0ull;
        if (a.clear_v && n) vc[wk * rl + oo] = 0ull;   // (every counter is read by exactly one lane of one pass)
        // What this lane needs besides its count depends on the row's coordinates only: the reference k-mer's record (k-mer,
        // first cell, flags -- one 16-byte load, consecutive ids across the lanes) goes out together with the row's load.
        const uint32_t d = r6 & 1u, alt = r6 >> 1;
        const bool inq = in_row && oo < (uint32_t)ix.v_span && q >= oo && q - oo < ix.n_full;
        const uint32_t p = inq ? q - oo : 0u;
        const uint4 idr = *reinterpret_cast<const uint4*>(ix.id_rec + p);
        const uint64_t kmer_p = (uint64_t)idr.x | ((uint64_t)idr.y << 32);
        const uint32_t ambp = idr.w;
        const int o = (int)oo + ix.v_omin;
        // a unit whose rows are all empty (most of them, with a large index and one sample) costs its loads and this vote only
        // (asked only where it pays: a single genome's plane is three quarters full and the extra barrier costs 5 %)
        if (sparse_plane && !__syncthreads_or(n != 0ull)) continue;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const unsigned long long t = __shfl_up(n, off, 64);   // (lane - off is in the same row whenever oo >= off)
            if (oo >= (uint32_t)off) n += t;
        }
        bool act = inq && n != 0;
        const uint32_t rcid = (ambp >> 1) & 1u;
        const bool dirty = ambp & 1u;
        const int j = rcid ? k - 1 - o : o;
        const int sh = 2 * (k - 1 - (act ? j : 0));
        // the base that stands there instead of the reference k-mer's own: own XOR (alternative + 1), on either strand
        const uint32_t bb = ((uint32_t)(kmer_p >> sh) & 3u) ^ (alt + 1u);
        const uint32_t isrc = d ^ rcid;
        const uint64_t c = (kmer_p & ~(3ull << sh)) | ((uint64_t)bb << sh);
        const uint64_t rc = revcomp_kmer(c, k);
        const bool alive = j >= ix.wstart && j < ix.wstart + ix.W && c < rc;
        if (act && !alive) {
            if (kt.keys && do_stats) ktab_insert(kt, c < rc ? c : rc, c < rc ? isrc : isrc ^ 1u, (unsigned int)(n > 0xf0000000ull ? 0xf0000000ull : n));
            act = false;
        }
        if (act) { distinct += do_stats; if (n < a.ci || n > a.cx) act = false; else kept += do_stats; }   // kmc -ci / -cx act on the true count
        const unsigned long long v = n > a.cs ? a.cs : n;       // kmc -cs: reported count saturates
        const uint32_t t = (uint32_t)(j - ix.wstart);
        bool defer = false;
        if (act && dirty) {
            // u has another reference k-mer within Hamming distance 2: c may touch a second window bucket.  Whether its
            // neighbours sit at several window positions was worked out with its answer (DirtyAns); without the table,
            // enumerate them.  Several buckets -> general path (K2b)
            bool multi;
            const uint2 ans = ix.dirty_ans ? *reinterpret_cast<const uint2*>(ix.dirty_ans + ans_index(p, (uint32_t)o, rcid ? 3u - bb : bb, k)) : make_uint2(0u, kAnsNone);
            if (!(ans.y & kAnsNone)) multi = (ans.y & kAnsMulti) != 0u;
            else {
                uint32_t jmask = 0;
                for_each_neighbour(ix, c, [&](int jj, uint32_t, uint32_t) { jmask |= 1u << (jj - ix.wstart); });
                multi = jmask != (1u << t);
            }
            if (multi) { defer = do_stats; act = false; }
        }
        {
            // the deferred k-mers of the wave are appended together: one returning atomic on the list's counter per wave, not one per
            // k-mer (with 100 strains one variant k-mer in eight comes here)
            const unsigned long long dm = __ballot(defer);
            if (dm) {
                unsigned int base = 0;
                if (lane64 == (uint32_t)__builtin_ctzll(dm)) base = atomicAdd(a.n_deferred, (unsigned int)__popcll(dm));
                base = (unsigned int)__shfl((int)base, __builtin_ctzll(dm));
                if (defer) {
                    const unsigned int at = base + (unsigned int)__popcll(dm & ((1ull << lane64) - 1ull));
                    a.deferred[at] = (uint32_t)(wk * rl + oo);
                    if (a.deferred_n) a.deferred_n[at] = n;
                }
            }
        }
        // the bucket (p, t): a "simple" k-mer's is its own single occurrence, known from the record; otherwise the table says
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        if (act) {
            if (ambp & kIdSimple) r = make_uint4(0u, 1u, idr.z + (uint32_t)j, (ambp >> 16) | ((uint32_t)j << 16) | (rcid << 24));
            else r = *reinterpret_cast<const uint4*>(ix.slot_rec + (size_t)p * ix.W + t);
        }
        const uint32_t cnt = r.y;
        DevEntry first;
        first.cell = r.z; first.file = (uint16_t)(r.w & 0xffffu); first.idx = (uint8_t)(r.w >> 16); first.canonical = (uint8_t)(r.w >> 24);
        // statistics pass, many genomes: which of them the bucket holds, as a bitmap (IndexView::slot_files) -- the wave tallies the
        // genomes of all its k-mers together below, one ballot per genome, instead of every lane walking ~100 entries
        uint4 sfb = make_uint4(0u, 0u, 0u, 0u);
        bool big = false;   // a bucket with a file bitmap whose votes go through the table chunk by chunk (below)
        if constexpr (MANY) {
            if (a.mode == 1) {
                if (act && cnt > 1u) sfb = ix.slot_files[(size_t)p * ix.W + t];
                tally_files<true>(sfb, lstats, lane64, (uint32_t)ix.n_files);   // one hit in each of its genomes (W > 1: "variant"), never perfect
            } else if (a.mode == 0) {
                // Every genome's rows: the k-mers of a V row vote for the same cell in each genome that holds their bucket (the
                // position of the differing base), ~27 votes on one counter pair, and with 100 strains a sample casts 300 M of them
                // -- the whole of its finalize.  So the workgroup's lanes walk their buckets' genomes in step, 16 genome files at a
                // time, through the vote table (which merges them) and flush it after every chunk.
                if (act && cnt > kVoteMaxEntries) { sfb = ix.slot_files[(size_t)p * ix.W + t]; big = files_any(sfb); }
                if (__syncthreads_or(big)) {
                    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
                    tally_files<true>(big ? sfb : z, lstats, lane64, (uint32_t)ix.n_files);   // (W > 1, one BucketInfo per genome: "variant")
                    uint32_t ri = 0;   // entries of the lane's bucket taken so far (they are sorted by genome file)
                    for (uint32_t f0 = 0; f0 < (uint32_t)ix.n_files; f0 += 16u) {
                        const uint32_t wd = f0 < 32u ? sfb.x : f0 < 64u ? sfb.y : f0 < 96u ? sfb.z : sfb.w;
                        for (uint32_t bits = big ? (wd >> (f0 & 31u)) & 0xffffu : 0u; bits; bits &= bits - 1u) vt_vote(vt, par, a, ix.entries[r.x + ri++], c, isrc, k, v);
                        vt_flush(vt, par, a);
                        par ^= 1u;
                    }
                }
            }
        }
        // single-entry bucket: the vote of call.rs:1327-1384 (see vote()), merged across the row when possible
        const bool single = act && cnt == 1u;
        if (single) vt_vote(vt, par, a, first, c, isrc, k, v);
        {
            // one hit in that genome: "variant" unless the window is a single bucket.  Tallied once per wave for the genome of
            // the wave's first voter (64 lanes adding 1 to the same LDS word would serialise), one by one for the others
            const unsigned long long sm = __ballot(single && do_stats);
            if (sm) {
                const int lf = __builtin_ctzll(sm);
                const bool same = single && do_stats && first.file == (uint16_t)__shfl((int)first.file, lf);
                const uint32_t n_same = (uint32_t)__popcll(__ballot(same));
                const uint32_t add = same ? ((int)(threadIdx.x & 63u) == lf ? n_same : 0u) : (single && do_stats ? 1u : 0u);
                if (add) {
                    if (ix.W == 1) { atomicAdd(&lstats[first.file * 3 + 0], add); atomicAdd(&lstats[first.file * 3 + 2], add); }
                    else atomicAdd(&lstats[first.file * 3 + 1], add);
                }
            }
        }
        if (act && cnt > 1u && !do_stats) {
            // second pass: the selected genome's entries of the bucket only
            const uint4 fb = ix.slot_files ? ix.slot_files[(size_t)p * ix.W + t] : make_uint4(0u, 0u, 0u, 0u);
            if (files_any(fb)) {                               // (one entry per genome: the selected genome's by its rank, if it is there)
                if (files_has(fb, (uint32_t)a.sel_file)) vt_vote(vt, par, a, ix.entries[r.x + files_rank(fb, (uint32_t)a.sel_file)], c, isrc, k, v);
            } else
            for (uint32_t x = first_of_file(ix.entries + r.x, cnt, a.sel_file); x < cnt; ++x) {
                const DevEntry en = ix.entries[r.x + x];
                if ((int)en.file != a.sel_file) break;
                vt_vote(vt, par, a, en, c, isrc, k, v);
            }
        } else if (act && cnt > 1u && a.mode == 1 && !files_any(sfb)) {
            // statistics pass, no votes: the genomes of the bucket and how often each is there.  Four entries are asked for at a
            // time (the walk over ~100 genomes' entries is a chain of dependent loads otherwise: latency, not bandwidth)
            uint32_t n_perfect = 0, perfect_file = 0, cur = first.file, run = 1;
            auto close = [&]() {
                const bool perfect = run == (uint32_t)ix.W;
                tally(lstats, cur * 3 + (perfect ? 0u : 1u));
                if (perfect) { ++n_perfect; perfect_file = cur; }
            };
            for (uint32_t x = 1; x < cnt; x += 4u) {
                uint32_t f[4];
#pragma unroll
                for (uint32_t i = 0; i < 4u; i++) f[i] = ix.entries[r.x + min(x + i, cnt - 1u)].file;
#pragma unroll
                for (uint32_t i = 0; i < 4u; i++) {
                    if (x + i >= cnt) break;
                    if (f[i] != cur) { close(); cur = f[i]; run = 0; }
                    ++run;
                }
            }
            close();
            if (n_perfect == 1) atomicAdd(&lstats[perfect_file * 3 + 2], 1u);
        } else if (act && cnt > 1u && a.mode != 1 && !big) {
            // entries of one bucket are grouped by file (index build appends file by file): run lengths = hits per file
            uint32_t n_perfect = 0, perfect_file = 0;
            for (uint32_t x = 0; x < cnt;) {
                DevEntry en = x ? ix.entries[r.x + x] : first;
                const uint32_t file = en.file;
                uint32_t run = 0;
                for (;;) {
                    if (cnt <= kVoteMaxEntries) vt_vote(vt, par, a, en, c, isrc, k, v);   // a few genomes: still worth the table
                    else vote(a, en, c, isrc, k, v);                                        // many: as many cells, it would overflow
                    ++run; ++x;
                    if (x >= cnt) break;
                    en = ix.entries[r.x + x];
                    if (en.file != file) break;
                }
                const bool perfect = run == (uint32_t)ix.W;
                tally(lstats, file * 3 + (perfect ? 0u : 1u));
                if (perfect) { ++n_perfect; perfect_file = file; }
            }
            if (n_perfect == 1) atomicAdd(&lstats[perfect_file * 3 + 2], 1u);
        }
        vt_flush(vt, par, a);
        par ^= 1u;
    }
    // the pseudo k-mers' counters (k = 31 only), one thread each (sparse finalize: the 8 counters of each touched row)
    const uint64_t px_n = a.p_list ? (uint64_t)a.n_list[4] * 8ull : px_hi;
    for (uint64_t xi = (a.p_list ? 0ull : px_lo) + (uint64_t)blockIdx.x * 256 + threadIdx.x; xi < px_n; xi += (uint64_t)gridDim.x * 256) {
        const uint64_t x = a.p_list ? (uint64_t)a.p_list[xi >> 3] * 8ull + (xi & 7ull) : xi;
        const uint64_t vi = real_len + x;
        if (vc[vi] == 0) continue;
        uint32_t p, t, isrc; uint64_t c; unsigned long long n;
        v_kmer_of_counter(ix, vc, vi, p, t, c, isrc, n);
        if (a.clear_v) vc[vi] = 0ull;
        map_one(p, t, c, isrc, n, vi);
        flush(0); flush(1);
    }
    __syncthreads();
    if (do_stats) finalize_epilogue(a, lstats, kept, distinct, lstats + ix.n_files * 3, (int)blockIdx.x);
}

// K2e: the E counters (reference k-mers).  A reference k-mer owns all W of its window buckets (slot_of), and its
// per-genome hit totals -- hence perfect / variant / unique -- depend on the index alone, so the host precomputed
// them (estat).  That makes the map embarrassingly parallel: one thread per (E counter, window bucket) votes for
// the BucketInfos of that bucket; the bucket-0 thread also tallies the statistics.
__global__ __launch_bounds__(256) void finalize_exact_kernel(FinalizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (a.mode == 2) { a.sel_file = *a.sel; if (a.sel_file < 0) return; }   // second pass: votes for the selected genome only
    const bool do_stats = a.mode < 2;
    const IndexView& ix = a.ix;
    uint32_t* lstats = reinterpret_cast<uint32_t*>(smem);   // [n_files][3] block-local tallies + 2 scratch words
    const VoteTable vt = vt_make(smem + (((size_t)ix.n_files * 3 + 2) * sizeof(uint32_t) + 15) / 16 * 16);
    for (int g = threadIdx.x; g < ix.n_files * 3 + 2; g += 256) lstats[g] = 0;
    vt_clear(vt);
    __syncthreads();
    const int k = ix.k;
    const uint32_t W = (uint32_t)ix.W;
    const uint64_t c_lo = min(a.elem_lo, e_plane_len(ix.n_u)), c_hi = min(a.elem_hi, e_plane_len(ix.n_u));   // this shard's E counters
    const uint64_t r_hi = min(c_hi, 2ull * ix.n_full);          // E counters of reference k-mers proper end here
    // sparse finalize: the two counters of every touched id (reference k-mers first in the loop below, pseudo k-mers after)
    const uint64_t n_ids_listed = a.e_list ? a.n_list[2] : 0ull, n_pseudo_listed = a.e_list ? a.n_list[3] : 0ull;
    // ... and its statistics pass casts no votes: 8 lanes per counter share the walk over the k-mer's genomes (estat)
    const bool stats_walk = a.e_list && a.mode == 1;
    const uint32_t Wd = stats_walk ? (ix.estat_files ? 1u : 8u) : W;   // lanes per counter
    const uint64_t n_work = a.e_list ? n_ids_listed * 2ull * Wd : r_hi * W;
    unsigned int kept = 0, distinct = 0;
    uint32_t par = 0;
    // every genome's rows (mode 0) over a list of touched k-mers of a many-genome index: the votes of the genomes that hold a k-mer
    // as it is were cast by finalize_exact_own_kernel, cell by cell
    const bool own_all = a.mode == 0 && !a.gather && a.e_list && a.file_cell_lo && ix.id_own_files && ix.cell_file && ix.estat_files && ix.id_rest_off;
    // one (counter, window bucket) pair
    auto pair_body = [&](uint64_t cidx, uint32_t t) {
      do {
        const unsigned long long n = a.counters[cidx];
        distinct += (n != 0 && t == 0 && do_stats && !own_all);
        if (n == 0 || n < a.ci || n > a.cx) break;              // kmc -ci / -cx act on the true count
        const unsigned long long v = n > a.cs ? a.cs : n;       // kmc -cs: reported count saturates
        const uint32_t id = (uint32_t)(cidx >> 1), isrc = (uint32_t)cidx & 1u;
        const uint4 idr = *reinterpret_cast<const uint4*>(ix.id_rec + id);
        const uint64_t c = (uint64_t)idr.x | ((uint64_t)idr.y << 32);
        // second pass, many genomes: a k-mer whose BucketInfos for the selected genome are its own occurrence there, bucket after
        // bucket (IndexView::id_own_files), has been voted for by finalize_exact_own_kernel
        const bool by0 = a.mode == 2 && a.file_cell_lo && ix.id_own_files && files_has(ix.id_own_files[id], (uint32_t)a.sel_file);
        if (by0) {
        } else if (own_all) {
            // every genome's rows, the k-mers of a list: finalize_exact_own_kernel has cast the votes of the genomes that hold the
            // k-mer as it is (id_own_files); what is left of this bucket are the BucketInfos of the others
            const uint4 r = *reinterpret_cast<const uint4*>(ix.slot_rec + (size_t)id * W + t);
            const uint4 fb = ix.slot_files[(size_t)id * W + t], own = ix.id_own_files[id];
            if (files_any(fb)) {
                const uint32_t rest[4] = {fb.x & ~own.x, fb.y & ~own.y, fb.z & ~own.z, fb.w & ~own.w};
#pragma unroll
                for (uint32_t w4 = 0; w4 < 4u; ++w4)
                    for (uint32_t bits = rest[w4]; bits; bits &= bits - 1u)
                        vt_vote(vt, par, a, ix.entries[r.x + files_rank(fb, w4 * 32u + (uint32_t)__builtin_ctz(bits))], c, isrc, k, v);   // (neighbouring k-mers meet at the other genomes' differences: the table merges them)
            } else {
                for (uint32_t q = 0; q < r.y; ++q) vote(a, ix.entries[r.x + q], c, isrc, k, v);   // (no bitmap: own is all zero)
            }
        } else if (a.mode != 1) {   // (the statistics pass casts no votes: a reference k-mer's statistics come from estat below)
            const uint32_t j = (uint32_t)ix.wstart + t;
            const uint4 r = (idr.w & kIdSimple) ? make_uint4(0u, 1u, idr.z + j, (idr.w >> 16) | (j << 16) | (((idr.w >> 1) & 1u) << 24))
                                                : *reinterpret_cast<const uint4*>(ix.slot_rec + (size_t)id * W + t);
            DevEntry first;
            first.cell = r.z; first.file = (uint16_t)(r.w & 0xffffu); first.idx = (uint8_t)(r.w >> 16); first.canonical = (uint8_t)(r.w >> 24);
            if (r.y && r.y <= kVoteMaxEntries) {               // a few genomes: still worth the table
                vt_vote(vt, par, a, first, c, isrc, k, v);
                for (uint32_t q = 1; q < r.y; ++q) vt_vote(vt, par, a, ix.entries[r.x + q], c, isrc, k, v);
            } else if (r.y && !do_stats) {                     // second pass: the selected genome's entries only
                const uint4 fb = ix.slot_files ? ix.slot_files[(size_t)id * W + t] : make_uint4(0u, 0u, 0u, 0u);
                if (files_any(fb)) {                           // (one entry per genome: the selected genome's by its rank, if it is there)
                    if (files_has(fb, (uint32_t)a.sel_file)) vt_vote(vt, par, a, ix.entries[r.x + files_rank(fb, (uint32_t)a.sel_file)], c, isrc, k, v);
                } else
                for (uint32_t q = first_of_file(ix.entries + r.x, r.y, a.sel_file); q < r.y; ++q) {
                    const DevEntry en = ix.entries[r.x + q];
                    if ((int)en.file != a.sel_file) break;
                    vt_vote(vt, par, a, en, c, isrc, k, v);
                }
            } else if (r.y) {                                  // many: as many cells, it would overflow
                vote(a, first, c, isrc, k, v);
                for (uint32_t q = 1; q < r.y; ++q) vote(a, ix.entries[r.x + q], c, isrc, k, v);
            }
        }
        if (t == 0 && do_stats && !own_all) {
            ++kept;
            uint32_t n_perfect = 0, perfect_file = 0;
            for (uint32_t q = ix.estat_off[id]; q < ix.estat_off[id + 1]; ++q) {   // (file << 1) | perfect
                const uint32_t e = ix.estat[q];
                tally(lstats, (e >> 1) * 3 + ((e & 1u) ? 0u : 1u));
                if (e & 1u) { ++n_perfect; perfect_file = e >> 1; }
            }
            if (n_perfect == 1) atomicAdd(&lstats[perfect_file * 3 + 2], 1u);
        }
      } while (false);
    };
    // Voting pass of pileup_selected_only over a list of touched k-mers: nearly all of them were voted for by
    // finalize_exact_own_kernel (IndexView::id_own_files) or fail the thresholds -- one lane per counter finds the few that are
    // left (k-mers the selected genome does not hold as they are), and only those are spread over W lanes each
    const bool own_done = (a.mode == 2 && !a.gather && a.e_list && a.file_cell_lo && ix.id_own_files) || own_all;
    if (own_done) {
        __shared__ unsigned long long ownq[256], restq[256];
        __shared__ unsigned int ownq_n, restq_n;
        for (uint64_t g0 = (uint64_t)blockIdx.x * 256; g0 < n_ids_listed * 2ull; g0 += (uint64_t)gridDim.x * 256) {
            if (threadIdx.x == 0) { ownq_n = 0u; restq_n = 0u; }
            __syncthreads();
            const uint64_t g = g0 + threadIdx.x;
            uint4 pf = make_uint4(0u, 0u, 0u, 0u), vf = pf;
            if (g < n_ids_listed * 2ull) {
                const uint64_t cidx = 2ull * a.e_list[g >> 1] + (g & 1ull);
                const unsigned long long n = a.counters[cidx];
                const bool keep = n != 0 && n >= a.ci && n <= a.cx;
                const uint32_t id = (uint32_t)(cidx >> 1);
                if (own_all) {   // (also the statistics: one lane per counter, the k-mer's genomes as two bitmaps)
                    distinct += (n != 0);
                    kept += keep;
                    if (keep) { pf = ix.estat_files[(size_t)id * 2]; vf = ix.estat_files[(size_t)id * 2 + 1]; }
                    if (keep && !(ix.id_rec[id].flags & kIdAllOwn)) {   // with own files: its list of the others (id_rest); without: bucket by bucket
                        if (files_any(ix.id_own_files[id])) restq[atomicAdd(&restq_n, 1u)] = cidx; else ownq[atomicAdd(&ownq_n, 1u)] = cidx;
                    }
                } else if (keep && !files_has(ix.id_own_files[id], (uint32_t)a.sel_file)) ownq[atomicAdd(&ownq_n, 1u)] = cidx;
            }
            if (own_all) {
                tally_files<true, 0u>(pf, lstats, threadIdx.x & 63u, (uint32_t)ix.n_files);
                tally_files<true, 1u>(vf, lstats, threadIdx.x & 63u, (uint32_t)ix.n_files);
                if (__popc(pf.x) + __popc(pf.y) + __popc(pf.z) + __popc(pf.w) == 1) {   // perfect in exactly one genome: unique to it
                    const uint32_t f = pf.x ? (uint32_t)__builtin_ctz(pf.x) : pf.y ? 32u + (uint32_t)__builtin_ctz(pf.y) : pf.z ? 64u + (uint32_t)__builtin_ctz(pf.z) : 96u + (uint32_t)__builtin_ctz(pf.w);
                    atomicAdd(&lstats[f * 3u + 2u], 1u);
                }
            }
            __syncthreads();
            // the listed BucketInfos (IndexView::id_rest) of 32 counters at a time, 8 lanes per counter: neighbouring k-mers meet at
            // the other genomes' differences, the vote table merges them
            for (uint32_t i0 = 0; i0 < restq_n; i0 += 32u) {
                const uint32_t item = i0 + (threadIdx.x >> 3);
                if (item < restq_n) {
                    const uint64_t cidx = restq[item];
                    const uint32_t id = (uint32_t)(cidx >> 1);
                    const unsigned long long n = a.counters[cidx];
                    const unsigned long long v = n > a.cs ? a.cs : n;
                    const uint64_t c = ix.id_rec[id].kmer;
                    for (uint32_t q = ix.id_rest_off[id] + (threadIdx.x & 7u), qe = ix.id_rest_off[id + 1]; q < qe; q += 8u)
                        vt_vote(vt, par, a, ix.entries[ix.id_rest[q]], c, (uint32_t)cidx & 1u, k, v);
                }
                vt_flush(vt, par, a);
                par ^= 1u;
            }
            const uint32_t nq = ownq_n * W;
            for (uint32_t w0 = 0; w0 < nq; w0 += 256u) {
                const uint32_t w = w0 + threadIdx.x;
                if (w < nq) pair_body(ownq[w / W], w % W);
                vt_flush(vt, par, a);
                par ^= 1u;
            }
            __syncthreads();
        }
    }
    // 256 consecutive (counter, bucket) pairs per workgroup and round: ~8 consecutive reference k-mers, whose votes fall on
    // ~25 pileup cells -- gathered in the workgroup's vote table (LDS) before they go to the pileup
    for (uint64_t g0 = (a.e_list ? 0ull : c_lo * W) + (uint64_t)blockIdx.x * 256; g0 < n_work && !own_done && !a.gather; g0 += (uint64_t)gridDim.x * 256) {   // (gather: the reference k-mers voted cell by cell)
      const uint64_t g = g0 + threadIdx.x;
      // the counter of this (counter, bucket) pair; a listed id may be a pseudo k-mer: those are mapped by the loop below
      uint64_t cidx = g < n_work ? g / Wd : 0ull;
      const bool mine = g < n_work;
      if (a.e_list && mine) cidx = 2ull * a.e_list[cidx >> 1] + (cidx & 1ull);
      if (stats_walk && ix.estat_files) {
          // one lane per counter: the genomes in which the k-mer is perfect / a variant as two bitmaps, tallied for the whole wave
          const unsigned long long n = mine ? a.counters[cidx] : 0ull;
          distinct += (n != 0);
          const bool keep = n != 0 && n >= a.ci && n <= a.cx;
          kept += keep;
          const uint32_t id = (uint32_t)(cidx >> 1);
          const uint4 z = make_uint4(0u, 0u, 0u, 0u);
          const uint4 pf = keep ? ix.estat_files[(size_t)id * 2] : z, vf = keep ? ix.estat_files[(size_t)id * 2 + 1] : z;
          tally_files<true, 0u>(pf, lstats, threadIdx.x & 63u, (uint32_t)ix.n_files);
          tally_files<true, 1u>(vf, lstats, threadIdx.x & 63u, (uint32_t)ix.n_files);
          if (__popc(pf.x) + __popc(pf.y) + __popc(pf.z) + __popc(pf.w) == 1) {   // perfect in exactly one genome: unique to it
              const uint32_t f = pf.x ? (uint32_t)__builtin_ctz(pf.x) : pf.y ? 32u + (uint32_t)__builtin_ctz(pf.y) : pf.z ? 64u + (uint32_t)__builtin_ctz(pf.z) : 96u + (uint32_t)__builtin_ctz(pf.w);
              atomicAdd(&lstats[f * 3u + 2u], 1u);
          }
          continue;
      }
      if (stats_walk) {
          // (the 8 lanes of a counter are neighbours in one wave and take every branch together)
          const unsigned long long n = mine ? a.counters[cidx] : 0ull;
          const uint32_t t = (uint32_t)(g & 7ull);
          distinct += (n != 0 && t == 0);
          if (n == 0 || n < a.ci || n > a.cx) continue;
          kept += (t == 0);
          const uint32_t id = (uint32_t)(cidx >> 1);
          uint32_t n_perfect = 0, perfect_file = 0;
          for (uint32_t q = ix.estat_off[id] + t, qe = ix.estat_off[id + 1]; q < qe; q += 8u) {   // (file << 1) | perfect
              const uint32_t e = ix.estat[q];
              tally(lstats, (e >> 1) * 3 + ((e & 1u) ? 0u : 1u));
              if (e & 1u) { ++n_perfect; perfect_file = e >> 1; }
          }
          uint32_t tot = n_perfect;
          tot += (uint32_t)__shfl_xor((int)tot, 1); tot += (uint32_t)__shfl_xor((int)tot, 2); tot += (uint32_t)__shfl_xor((int)tot, 4);
          if (tot == 1 && n_perfect == 1) atomicAdd(&lstats[perfect_file * 3 + 2], 1u);
          continue;
      }
      // a round whose counters are all zero (most of them, with a large index and one sample) votes for nothing: skip its flush
      if (ix.n_files > 1 && !a.e_list && !__syncthreads_or(mine && a.counters[cidx] != 0ull)) continue;
      if (mine) pair_body(cidx, (uint32_t)(g % W));
      vt_flush(vt, par, a);
      par ^= 1u;
    }
    // pseudo k-mers (k = 31): nearly all of their counters are zero.  With a list of touched ids, 8 lanes per (counter, window
    // bucket) share the bucket's BucketInfos (one thread per counter walked W buckets of ~100 entries one after the other: a
    // chain of thousands of dependent steps, 4 ms of a 100-strain sample's finalize); without one, one thread per counter.
    const uint64_t pc_n = a.e_list ? n_pseudo_listed * 2ull : c_hi;
    const uint32_t pl = a.e_list ? 8u : 1u;                                  // lanes per (counter, bucket)
    const uint64_t pw_n = a.e_list ? pc_n * W * pl : pc_n;
    for (uint64_t wi = (a.e_list ? 0ull : max(c_lo, 2ull * ix.n_full)) + (uint64_t)blockIdx.x * 256 + threadIdx.x; wi < pw_n; wi += (uint64_t)gridDim.x * 256) {
        const uint64_t ci = a.e_list ? wi / (W * pl) : wi;
        const uint32_t rem = a.e_list ? (uint32_t)(wi - ci * (W * pl)) : 0u, t0 = rem / pl, sub = rem - t0 * pl;
        const bool head = t0 == 0u && sub == 0u;                             // (the counter's first lane: its tallies)
        const uint64_t cidx = a.e_list ? 2ull * a.e_list[ix.n_u - 1u - (uint32_t)(ci >> 1)] + (ci & 1ull) : ci;   // (the tail of the list)
        const unsigned long long n = a.counters[cidx];
        if (n == 0) continue;
        distinct += do_stats && head;
        if (n < a.ci || n > a.cx) continue;
        kept += do_stats && head;
        const unsigned long long v = n > a.cs ? a.cs : n;
        const uint32_t id = (uint32_t)(cidx >> 1), isrc = (uint32_t)cidx & 1u;
        const uint64_t c = ix.kmer_of[id];
        if (a.mode != 1) for (uint32_t t = a.e_list ? t0 : 0u; t < (a.e_list ? t0 + 1u : W); ++t) {
            const uint32_t s = ix.slot_of[(size_t)id * W + t];
            if (a.gather && !(ix.slot_alias && ((ix.slot_alias[s >> 5] >> (s & 31u)) & 1u))) continue;   // (gathered votes: only alias slots are left)
            const uint32_t off = ix.ent_off[s], cnt = ix.ent_len[s];
            if (a.mode == 2 && ix.ent_files) {   // (one entry per genome: the selected genome's by its rank -- no bisection in this serial loop)
                const uint4 fb = ix.ent_files[s];
                if (files_any(fb)) {
                    if (sub == 0u && files_has(fb, (uint32_t)a.sel_file)) vote(a, ix.entries[off + files_rank(fb, (uint32_t)a.sel_file)], c, isrc, k, v);
                    continue;
                }
            }
            if (a.mode == 2) {
                if (sub != 0u) continue;
                for (uint32_t q = first_of_file(ix.entries + off, cnt, a.sel_file); q < cnt; ++q) {
                    const DevEntry en = ix.entries[off + q];
                    if ((int)en.file != a.sel_file) break;
                    vote(a, en, c, isrc, k, v);
                }
            } else {
                for (uint32_t q = sub; q < cnt; q += pl) vote(a, ix.entries[off + q], c, isrc, k, v);
            }
        }
        if (!do_stats || !head) continue;
        uint32_t n_perfect = 0, perfect_file = 0;
        for (uint32_t q = ix.estat_off[id]; q < ix.estat_off[id + 1]; ++q) {   // (file << 1) | perfect
            const uint32_t e = ix.estat[q];
            atomicAdd(&lstats[(e >> 1) * 3 + ((e & 1u) ? 0 : 1)], 1u);
            if (e & 1u) { ++n_perfect; perfect_file = e >> 1; }
        }
        if (n_perfect == 1) atomicAdd(&lstats[perfect_file * 3 + 2], 1u);
    }
    __syncthreads();
    if (do_stats) finalize_epilogue(a, lstats, kept, distinct, lstats + ix.n_files * 3, a.row_exact + (int)blockIdx.x);
}

// K2e, voting pass of pileup_selected_only with many genomes: the reference k-mers whose BucketInfos for the selected genome are
// their own occurrence there, bucket after bucket (IndexView::id_own_files) -- nearly every k-mer the sample's reads hold.  One
// thread per CELL of the selected genome instead of one per (counter, bucket) of every touched k-mer: the k-mer that starts at the
// cell, its two counters, one BucketInfo (bucket 0's; bucket t's is that one t further on), and its 2 W votes (vote()) into a
// table in LDS over the positions the workgroup's 256 cells reach, which is then added to the pileup.  finalize_exact_kernel
// skips these k-mers in that pass.
constexpr uint32_t kOwnCells = 256, kOwnSpan = kOwnCells + 32;
__global__ __launch_bounds__(256) void finalize_exact_own_kernel(FinalizeArgs a) {
    __shared__ unsigned long long mx[8][kOwnSpan];
    __shared__ unsigned int cnt[8][kOwnSpan];
    const IndexView& ix = a.ix;
    const int sel = a.mode == 2 ? *a.sel : 0;   // (mode 0, every genome's rows: all cells, each for the genome it lies in)
    if (sel < 0) return;
    const uint32_t c_lo = a.mode == 2 ? a.file_cell_lo[sel] : 0u;
    const uint32_t c_hi = a.mode == 2 ? (sel + 1 < ix.n_files ? a.file_cell_lo[sel + 1] : ix.total_cells) : ix.total_cells;
    const uint32_t c0 = c_lo + blockIdx.x * kOwnCells;
    if (c0 >= c_hi) return;
    for (uint32_t i = threadIdx.x; i < 8u * kOwnSpan; i += 256u) { (&mx[0][0])[i] = 0ull; (&cnt[0][0])[i] = 0u; }
    __syncthreads();
    const int k = ix.k;
    const uint32_t W = (uint32_t)ix.W;
    const uint32_t c = c0 + threadIdx.x;
    const uint32_t id = c < c_hi ? ix.id_at[c] : 0xffffffffu;
    const uint32_t fsel = a.mode == 2 ? (uint32_t)sel : (c < c_hi ? (uint32_t)ix.cell_file[c] : 0u);
    unsigned long long n2[2] = {0ull, 0ull};   // the k-mer's two counters, those that pass the thresholds (kmc -ci / -cx act on the true count)
    if (id < ix.n_full) {
#pragma unroll
        for (int isrc = 0; isrc < 2; ++isrc) { const unsigned long long n = a.counters[2 * (size_t)id + isrc]; n2[isrc] = (n >= a.ci && n <= a.cx) ? n : 0ull; }
    }
    if ((n2[0] | n2[1]) != 0ull && files_has(ix.id_own_files[id], fsel)) {
        const uint4 fb = ix.slot_files[(size_t)id * W];
        const DevEntry e0 = ix.entries[ix.slot_rec[(size_t)id * W].off + files_rank(fb, fsel)];   // (cell = c + idx: checked at create, and below)
        const uint4 idr = *reinterpret_cast<const uint4*>(ix.id_rec + id);
        const uint64_t km = (uint64_t)idr.x | ((uint64_t)idr.y << 32);
        const int dir = (idr.w & kIdOwnMirror) ? -1 : 1;                        // bucket t's BucketInfo: bucket 0's, t further on / back
        const uint32_t q0 = e0.cell - c0;                                       // position of bucket 0's vote within the table
        for (uint32_t isrc = 0; isrc < 2u && e0.cell == c + (uint32_t)e0.idx; ++isrc) {   // (its one occurrence in the genome is at this very cell)
            const unsigned long long n = n2[isrc];
            if (n == 0) continue;
            const unsigned long long v = n > a.cs ? a.cs : n;                   // kmc -cs: reported count saturates
            for (uint32_t t = 0; t < W; ++t) {
                const uint32_t idx = (uint32_t)((int)e0.idx + dir * (int)t);     // (vote(): call.rs:1327-1384)
                uint32_t bit_idx;
                bool forward;
                if (e0.canonical) { bit_idx = ((uint32_t)(km >> (2 * idx)) & 3u) ^ 3u; forward = isrc != 0; }
                else { bit_idx = (uint32_t)(km >> (2 * (k - 1 - (int)idx))) & 3u; forward = isrc == 0; }
                const uint32_t row = (forward ? 0u : 4u) + bit_idx;
                const uint32_t q = (uint32_t)((int)q0 + dir * (int)t);           // (cell c + idx: within [0, 256 + k))
                atomicAdd(&cnt[row][q], 1u);
                atomicMax(&mx[row][q], v);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 8u * kOwnSpan; i += 256u) {
        const uint32_t row = i / kOwnSpan, q = i - row * kOwnSpan;
        const unsigned int nv = cnt[row][q];
        if (!nv) continue;
        const size_t cell = ((size_t)c0 + q) * 4 + (row & 3u);
        atomicAdd(a.pileup + (row < 4u ? 2 : 3) * a.plane + cell, (unsigned long long)nv);   // #kmers
        atomicMax(a.pileup + (row < 4u ? 0 : 1) * a.plane + cell, mx[row][q]);               // depth
    }
}

// K2b: one wave per workgroup and per k-mer, for the V counters K2a deferred (k-mers that touch several window
// buckets; rare).  Lane t probes the k-mer's t-th window bucket and votes once
// per BucketInfo found there.  Per-genome hit totals live in LDS (hits[n_files]); genomes touched by the
// current k-mer are listed so that only they are classified and re-zeroed.  Per-genome statistics are tallied
// in LDS and flushed once per workgroup (millions of k-mers voting for the same genome would otherwise
// serialise on one global atomic word).
__global__ __launch_bounds__(64) void finalize_general_kernel(FinalizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (a.mode == 2) { a.sel_file = *a.sel; if (a.sel_file < 0) return; }   // second pass: votes for the selected genome only
    const bool do_stats = a.mode < 2;
    const IndexView& ix = a.ix;
    // gathered votes (bk_gather.hip): this statistics pass notes which of its k-mers reach a bucket through an alias key
    auto note_alias = [&](int sb, uint64_t c, uint32_t isrc, unsigned long long n) {
        if (a.alias_hits && ix.slot_alias && sb >= 0 && ((ix.slot_alias[(uint32_t)sb >> 5] >> ((uint32_t)sb & 31u)) & 1u)) {
            const unsigned int at = atomicAdd(a.n_alias_hits, 1u);
            if (at < a.alias_cap) { a.alias_hits[3ull * at] = c | ((unsigned long long)isrc << 62); a.alias_hits[3ull * at + 1ull] = n; a.alias_hits[3ull * at + 2ull] = (unsigned long long)(uint32_t)sb; }
        }
    };
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

    if (a.mode == 2) {
        // Second pass (votes for the selected genome only, no statistics): nothing crosses lanes, so every lane takes a k-mer of
        // its own -- 64 chains of dependent loads in flight per wave instead of one -- and walks the window positions: probe the
        // bucket, find the selected genome's BucketInfos in it (sorted by genome: bisection), vote.
        for (uint64_t item = (uint64_t)blockIdx.x * 64 + (uint32_t)lane; item < n_items; item += (uint64_t)gridDim.x * 64) {
            unsigned long long v;
            uint64_t c;
            uint32_t isrc, p_, t_;
            v_kmer_of_counter(ix, a.deferred_n ? nullptr : a.counters + ix.v_off, a.deferred[item], p_, t_, c, isrc, v);
            if (a.deferred_n) v = a.deferred_n[item];
            v = v > a.cs ? a.cs : v;
            // (the window positions at which the first pass found a bucket, when it noted them)
            for (uint32_t tm = a.deferred_mask ? a.deferred_mask[item] : 0xffffffffu >> (32 - ix.W); tm; tm &= tm - 1u) {
                const int t = __builtin_ctz(tm);
                const int sh = 2 * (k - 1 - (ix.wstart + t));
                const int sb = probe_table(ix.table + (size_t)t * S, ix.log2s, c & ~(3ull << sh));
                if (sb < 0) continue;
                const uint32_t off = ix.ent_off[sb];
                const uint4 fb = ix.ent_files ? ix.ent_files[sb] : make_uint4(0u, 0u, 0u, 0u);
                if (files_any(fb)) {   // one entry per genome: the selected genome's by its rank, if it is there
                    if (files_has(fb, (uint32_t)a.sel_file)) vote(a, ix.entries[off + files_rank(fb, (uint32_t)a.sel_file)], c, isrc, k, v);
                    continue;
                }
                const uint32_t cnt = ix.ent_len[sb];
                for (uint32_t q = first_of_file(ix.entries + off, cnt, a.sel_file); q < cnt; ++q) {
                    const DevEntry e = ix.entries[off + q];
                    if ((int)e.file != a.sel_file) break;
                    vote(a, e, c, isrc, k, v);
                }
            }
        }
        return;
    }
    // (the regional finalize's list of reference k-mers that are not simple -- repeats: a few dozen -- rides with this kernel: two
    // items per listed k-mer behind the deferred ones, its two E counters; a launch of finalize_exact_kernel for them was 14 us)
    const uint64_t n_tail = (a.tail_e_list && a.mode == 0) ? 2ull * a.tail_n_list[2] : 0ull;
    unsigned int tail_kept = 0, tail_distinct = 0;
    auto wave_item = [&](uint64_t item) {
        {
            uint64_t ci, c;
            unsigned long long v;
            uint32_t isrc, p_, t_;
            if (item < n_items) {
                ci = n_e + a.deferred[item];                      // a V counter (n_e: the offset that tells it from an E counter below)
                v_kmer_of_counter(ix, a.deferred_n ? nullptr : a.counters + ix.v_off, ci - n_e, p_, t_, c, isrc, v);
                if (a.deferred_n) v = a.deferred_n[item];         // (K2a may have zeroed the row since)
            } else {
                const uint64_t e = item - n_items;
                const uint32_t id = a.tail_e_list[e >> 1];
                isrc = (uint32_t)e & 1u;
                ci = 2ull * id + isrc;                            // an E counter: the reference k-mer itself, read in orientation isrc
                v = a.counters[ci];
                const uint4 idr = *reinterpret_cast<const uint4*>(ix.id_rec + id);
                c = (uint64_t)idr.x | ((uint64_t)idr.y << 32);
                if (lane == 0) tail_distinct += v != 0;
                if (v == 0 || v < a.ci || v > a.cx) return;       // kmc -ci / -cx act on the true count (wave-uniform)
                if (lane == 0) tail_kept += 1;
            }
            v = v > a.cs ? a.cs : v;                              // kmc -cs: reported count saturates

            int s = -1;   // lane t: the k-mer's bucket at window position t, if the index has it
            if (lane < ix.W) {
                if (ci < n_e) {
                    s = (int)ix.slot_of[(size_t)(ci >> 1) * ix.W + lane];   // a reference k-mer owns all its buckets
                } else {
                    const int sh = 2 * (k - 1 - (ix.wstart + lane));
                    s = probe_table(ix.table + (size_t)lane * S, ix.log2s, c & ~(3ull << sh));
                }
            }
            // the buckets found, one after the other; the BucketInfos of a bucket (one per genome that has the k-mer: up to
            // hundreds) spread over the lanes
            if (a.mode == 1) note_alias(s, c, isrc, v);
            for (unsigned long long bm = __ballot(s >= 0); bm; bm &= bm - 1ull) {
                const int sb = __shfl(s, __builtin_ctzll(bm));
                const uint32_t off = ix.ent_off[sb], cnt = ix.ent_len[sb];
                for (uint32_t q = (uint32_t)lane; q < cnt; q += 64u) {
                    const DevEntry e = ix.entries[off + q];
                    // call.rs:1316-1318 per_genome_bucket_hits
                    if (atomicAdd(&hits[e.file], 1u) == 0u) touched[atomicAdd(ntouched, 1u)] = e.file;
                    vote(a, e, c, isrc, k, v);
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
                    if (do_stats) lstats[g * 3 + (perfect ? 0 : 1)] += 1;   // g is distinct per lane within one k-mer
                    if (perfect) my_perfect = (int)g;
                }
                n_perfect += (uint32_t)__popcll(__ballot(perfect));
            }
            if (do_stats && n_perfect == 1 && my_perfect >= 0) lstats[my_perfect * 3 + 2] += 1;
            __syncthreads();
            if (lane == 0) *ntouched = 0;
            __syncthreads();
        }
    };
    if (a.mode == 1 && ix.ent_files) {
        // Statistics pass with file bitmaps (IndexView::ent_files; many related genomes): every lane takes a k-mer of its own and
        // walks the window positions -- probe the bucket, OR its genomes into the k-mer's set.  With one BucketInfo per genome and
        // bucket, and fewer buckets than the window is long, no genome can be "perfect": each genome of the set is one "variant",
        // tallied for the whole wave with one ballot per genome.  The positions found are noted for the voting pass.  A k-mer
        // with a bucket that has no bitmap (a genome twice in it), or with all W buckets, takes the wave-per-k-mer path below.
        for (uint64_t item0 = (uint64_t)blockIdx.x * 64; item0 < n_items; item0 += (uint64_t)gridDim.x * 64) {
            const uint64_t item = item0 + (uint32_t)lane;
            const bool on = item < n_items;
            unsigned long long v = 0;
            uint64_t c = 0;
            uint32_t isrc = 0, p_ = 0, t_ = 0;
            if (on) v_kmer_of_counter(ix, nullptr, a.deferred[item], p_, t_, c, isrc, v);
            uint4 acc = make_uint4(0u, 0u, 0u, 0u);
            uint32_t found = 0, alias_n = 0;
            int alias_sb = -1;
            bool generic = false;
            for (int t = 0; t < ix.W; ++t) {
                const int sh = 2 * (k - 1 - (ix.wstart + t));
                const int sb = on ? probe_table(ix.table + (size_t)t * S, ix.log2s, c & ~(3ull << sh)) : -1;
                if (sb < 0) continue;
                const uint4 fb = ix.ent_files[sb];
                generic |= !files_any(fb);
                acc.x |= fb.x; acc.y |= fb.y; acc.z |= fb.z; acc.w |= fb.w;
                found |= 1u << t;
                alias_sb = (a.alias_hits && ix.slot_alias && ((ix.slot_alias[(uint32_t)sb >> 5] >> ((uint32_t)sb & 31u)) & 1u)) ? (alias_n++ ? -2 : sb) : alias_sb;
            }
            generic |= found == 0xffffffffu >> (32 - ix.W);
            generic |= alias_sb == -2;   // (more than one alias bucket: the wave-per-k-mer path notes them all)
            if (on && !generic && alias_sb >= 0) {   // (rare) the k-mer's count, from the plane
                unsigned long long n_; uint64_t c2; uint32_t i2, p2, t2;
                v_kmer_of_counter(ix, a.counters + ix.v_off, a.deferred[item], p2, t2, c2, i2, n_);
                note_alias(alias_sb, c, isrc, n_);
            }
            if (on && a.deferred_mask) a.deferred_mask[item] = found;
            if (!on || generic) acc = make_uint4(0u, 0u, 0u, 0u);
            tally_files<false>(acc, lstats, (uint32_t)lane, (uint32_t)ix.n_files);   // (one wave per workgroup: no atomic needed)
            for (unsigned long long gm = __ballot(on && generic); gm; gm &= gm - 1ull) wave_item(item0 + (uint64_t)__builtin_ctzll(gm));
        }
    } else
    // deferred items all passed the thresholds in K2a; one item per wave at a time, dealt round-robin so that a few
    // thousand items spread over the whole grid
    for (uint64_t item = blockIdx.x; item < n_items + n_tail; item += gridDim.x) wave_item(item);
    __syncthreads();
    if (lane == 0) { ntouched[0] = 0; ntouched[1] = 0; }
    __syncthreads();
    if (do_stats) finalize_epilogue(a, lstats, tail_kept, tail_distinct, ntouched, a.row_general + (int)blockIdx.x);
}

// Sparse finalize: the set bits of a touch bitmap -> a list of indices (order is irrelevant); every word read is cleared, so the
// bitmap is all zero again for the next sample.  Indices below `split` are appended at the front of the list (n_out[0] of them),
// the others from its last slot `cap - 1` downwards (n_out[1]): K2e maps reference k-mers and pseudo k-mers differently.  One
// append per wave and side: popcounts, a prefix sum over the lanes, one atomic by the last lane.
__global__ __launch_bounds__(256) void compact_touched_kernel(unsigned int* bm, uint64_t n_bits, unsigned int* list, unsigned int* n_out,
                                                              uint64_t split, uint64_t cap) {
    const uint64_t n_words = (n_bits + 31) / 32;
    const int lane = threadIdx.x & 63;
    for (uint64_t w0 = (uint64_t)blockIdx.x * 256; w0 < n_words; w0 += (uint64_t)gridDim.x * 256) {
        const uint64_t w = w0 + threadIdx.x;
        const uint32_t bits = w < n_words ? bm[w] : 0u;
        if (!__ballot(bits != 0u)) continue;
        if (bits) bm[w] = 0u;
        const uint32_t lo_mask = w * 32 + 32 <= split ? ~0u : w * 32 >= split ? 0u : (1u << (uint32_t)(split - w * 32)) - 1u;
        uint32_t lo = bits & lo_mask, hi = bits & ~lo_mask;
        const uint32_t n_lo = (uint32_t)__popc(lo), n_hi = (uint32_t)__popc(hi);
        uint32_t inc = n_lo | (n_hi << 16);   // (<= 32 * 64 each: both sums in one word)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)inc, d); if (lane >= d) inc += up; }
        uint32_t base_lo = 0, base_hi = 0;
        if (lane == 63) {
            if (inc & 0xffffu) base_lo = atomicAdd(n_out, inc & 0xffffu);
            if (inc >> 16) base_hi = atomicAdd(n_out + 1, inc >> 16);
        }
        base_lo = (uint32_t)__shfl((int)base_lo, 63) + (inc & 0xffffu) - n_lo;
        base_hi = (uint32_t)__shfl((int)base_hi, 63) + (inc >> 16) - n_hi;
        for (; lo; lo &= lo - 1u) list[base_lo++] = (uint32_t)(w * 32) + (uint32_t)__builtin_ctz(lo);
        for (; hi; hi &= hi - 1u) list[cap - 1 - base_hi++] = (uint32_t)(w * 32) + (uint32_t)__builtin_ctz(hi);
    }
}
// The scan notes the V rows it counted into per block of 64 cells (ScanArgs::touch_b): the rows of block gb are q = cell +
// cell_blk[gb].x + offset, |offset| < k -- all of them are marked here (an over-approximation: rows that hold nothing are
// read as zeros by finalize and zeroed again).  One wave per block; the block bitmap is cleared.
__global__ __launch_bounds__(256) void expand_touched_blocks_kernel(unsigned int* touch_b, uint32_t n_blocks, const uint2* cell_blk, unsigned int* touch_v,
                                                                    uint32_t span, uint64_t n_q) {
    const uint32_t gb = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;   // one wave per block of 64 cells
    if (gb >= n_blocks || !((touch_b[gb >> 5] >> (gb & 31u)) & 1u)) return;
    if (lane == 0) atomicAnd(touch_b + (gb >> 5), ~(1u << (gb & 31u)));
    // id of the block's first cell, were it one of the majority: negative where the ids restart inside the block (the head of a
    // genome whose k-mers an earlier genome holds).  (|id - cell| < 2^31: a plane of 2^32 counters is refused at create.)
    const long long q_0 = (long long)gb * 64 + (long long)(int32_t)cell_blk[gb].x;
    const long long q_lo = max(q_0 - (long long)span, 0ll);            // (a run's row is its first k-mer's id + an offset in (-span, span))
    const long long q_hi = min(q_0 + 64 + (long long)span, (long long)n_q);   // (exclusive)
    if (q_lo >= q_hi) return;
    const uint64_t r0 = (uint64_t)q_lo * kVRowsPerPos, r1 = (uint64_t)q_hi * kVRowsPerPos;   // the bits [r0, r1), a word per lane
    for (uint64_t w = (r0 >> 5) + lane; w * 32u < r1; w += 64u) {
        const uint64_t lo = max(w * 32u, r0), hi = min(w * 32u + 32u, r1);
        const uint32_t take = (uint32_t)(hi - lo);
        atomicOr(touch_v + w, (take == 32u ? 0xffffffffu : (1u << take) - 1u) << (uint32_t)(lo & 31u));
    }
}
void launch_expand_touched_blocks(unsigned int* touch_b, uint32_t n_blocks, const uint2* cell_blk, unsigned int* touch_v, uint32_t span, uint64_t n_q, hipStream_t stream) {
    if (!n_blocks) return;
    hipLaunchKernelGGL(expand_touched_blocks_kernel, dim3((n_blocks + 3u) / 4u), dim3(256), 0, stream, touch_b, n_blocks, cell_blk, touch_v, span, n_q);
}
void launch_compact_touched(unsigned int* touch_v, uint64_t n_rows, unsigned int* touch_p, uint64_t n_prows, unsigned int* touch_e, uint64_t n_ids,
                            uint64_t n_full, unsigned int* v_list, unsigned int* p_list, unsigned int* e_list, unsigned int* n_list, hipStream_t stream) {
    auto go = [&](unsigned int* bm, uint64_t n, unsigned int* list, unsigned int* cnt, uint64_t split) {
        if (!n) return;
        const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(((n + 31) / 32 + 255) / 256, 2048));
        hipLaunchKernelGGL(compact_touched_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, bm, n, list, cnt, split, n);
    };
    // n_list: [0] V rows, [1] (unused), [2] reference k-mer ids (front of e_list), [3] pseudo k-mer ids (its tail), [4] pseudo rows
    go(touch_v, n_rows, v_list, n_list + 0, ~0ull);
    go(touch_p, n_prows, p_list, n_list + 4, ~0ull);
    go(touch_e, n_ids, e_list, n_list + 2, n_full);
}
// ... and when a sample's maps are done: the listed rows / counters are zeroed again (the plane holds nothing else)
__global__ __launch_bounds__(256) void clear_touched_kernel(unsigned long long* counters, uint64_t v_off, uint64_t v_real, uint32_t rl,
                                                            const unsigned int* v_list, const unsigned int* p_list, const unsigned int* e_list,
                                                            const unsigned int* n_list, uint64_t n_ids) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x, nt = (uint64_t)gridDim.x * 256;
    unsigned long long* vc = counters + v_off;
    for (uint64_t i = t; i < (uint64_t)n_list[0] * rl; i += nt) vc[(uint64_t)v_list[i / rl] * rl + i % rl] = 0ull;
    for (uint64_t i = t; i < (uint64_t)n_list[4] * 8ull; i += nt) vc[v_real + (uint64_t)p_list[i >> 3] * 8ull + (i & 7ull)] = 0ull;
    for (uint64_t i = t; i < (uint64_t)n_list[2] * 2ull; i += nt) counters[2ull * e_list[i >> 1] + (i & 1ull)] = 0ull;
    for (uint64_t i = t; i < (uint64_t)n_list[3] * 2ull; i += nt) counters[2ull * e_list[n_ids - 1 - (i >> 1)] + (i & 1ull)] = 0ull;
}
void launch_clear_touched(unsigned long long* counters, uint64_t v_off, uint64_t v_real_len, uint32_t rl, const unsigned int* v_list,
                          const unsigned int* p_list, const unsigned int* e_list, const unsigned int* n_list, uint64_t n_ids, hipStream_t stream) {
    hipLaunchKernelGGL(clear_touched_kernel, dim3(2048), dim3(256), 0, stream, counters, v_off, v_real_len, rl, v_list, p_list, e_list, n_list, n_ids);
}

size_t finalize_lds_bytes(int n_files) { return ((size_t)n_files * 5 + 4) * sizeof(uint32_t); }
constexpr unsigned kFinVariantBlocks = 256 * 8, kFinExactBlocks = 256 * 8, kFinGeneralBlocks = 256 * 32;
size_t finalize_partial_rows() { return (size_t)kFinVariantBlocks + kFinExactBlocks + kFinGeneralBlocks; }

static unsigned finalize_general_blocks(const FinalizeArgs& a) {
    // (few genomes: multi-bucket k-mers are rare, a small grid starts and ends quickly; many: one k-mer in eight takes this path)
    return (unsigned)std::min<size_t>(a.ix.n_files >= 8 ? kFinGeneralBlocks : kFinGeneralBlocks / 32, 256 * std::max<size_t>(1, (160u * 1024u) / finalize_lds_bytes(a.ix.n_files)));
}
bool finalize_runs_by_region(const FinalizeArgs& a) {
    return finalize_lean_ok(a) && (uint64_t)(a.ix.n_full + (uint32_t)a.ix.v_span) / 64 + (uint64_t)a.ix.total_cells / 256 + 2 + 64 + finalize_general_blocks(a) <= finalize_partial_rows();
}
void launch_finalize(const FinalizeArgs& a0, hipStream_t stream) {
    if (a0.ix.W <= 0) return;
    FinalizeArgs a = a0;
    const size_t lds_stats = ((size_t)a.ix.n_files * 3 + 2) * sizeof(uint32_t);
    // K2a
    const uint64_t rows_per_group = 4ull * (64ull / (uint64_t)std::max(a.ix.v_span, 1));   // rows a workgroup takes at a time
    const uint64_t n_v = std::max<uint64_t>((v_real_rows(a.ix.n_full, a.ix.v_span) + rows_per_group - 1) / rows_per_group * 256ull, a.ix.n_prows * 8ull);   // threads K2a can use
    const unsigned b_var = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_v + 255) / 256, kFinVariantBlocks));
    const size_t lds_votes = (lds_stats + 15) / 16 * 16 + kVoteLdsBytes;
    const size_t lds = finalize_lds_bytes(a.ix.n_files);
    if (lds_votes > 64 * 1024 || lds > 64 * 1024) {   // thousands of genome files: beyond the default dynamic LDS limit
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(finalize_variant_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_votes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(finalize_variant_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_votes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(finalize_exact_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_votes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(finalize_general_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const unsigned b_gen = finalize_general_blocks(a);
    if (a.gather && a.mode >= 2) {
        // the voting pass behind gather_votes_kernel (bk_gather.hip): what reaches a bucket through an alias key -- the pseudo rows
        // (K2a's last loop), the pseudo k-mers' E counters (K2e's), the alias hits the statistics pass noted
        if (a.ix.n_prows) {
            if (a.ix.slot_files) hipLaunchKernelGGL(finalize_variant_kernel<true>, dim3(64), dim3(256), lds_votes, stream, a);
            else hipLaunchKernelGGL(finalize_variant_kernel<false>, dim3(64), dim3(256), lds_votes, stream, a);
        }
        if (a.ix.n_u > a.ix.n_full) hipLaunchKernelGGL(finalize_exact_kernel, dim3(64), dim3(256), lds_votes, stream, a);
        if (a.alias_hits) launch_alias_votes(a, stream);
        return;
    }
    unsigned b_v = b_var, b_e = 0;
    const bool lean = finalize_runs_by_region(a);
    if (lean) {
        // one genome file, dense planes: K2a and K2e by region of the reference (bk_finalize_lean.hip)
        b_v = launch_finalize_lean_variant(a, stream);   // (V rows and the E counters of the same region)
        // ... and the reference k-mers it listed (repeats: not "simple") go with K2b's deferred k-mers
        a.tail_e_list = a.lean_e_list; a.tail_n_list = a.lean_n_list;
        b_e = 0;
    } else {
    if (a.ix.slot_files) hipLaunchKernelGGL(finalize_variant_kernel<true>, dim3(b_var), dim3(256), lds_votes, stream, a);
    else hipLaunchKernelGGL(finalize_variant_kernel<false>, dim3(b_var), dim3(256), lds_votes, stream, a);
    // K2e
    const uint64_t n_work = e_plane_len(a.ix.n_u) * (uint64_t)a.ix.W;
    const unsigned b_ex = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_work + 255) / 256, kFinExactBlocks));
    a.row_exact = (int)b_var;
    if (a.mode == 2 && a.file_cell_lo && a.ix.id_own_files && a.max_file_cells)
        hipLaunchKernelGGL(finalize_exact_own_kernel, dim3((a.max_file_cells + kOwnCells - 1) / kOwnCells), dim3(256), 0, stream, a);
    else if (a.mode == 0 && a.e_list && a.file_cell_lo && a.ix.id_own_files && a.ix.cell_file && a.ix.estat_files && a.ix.id_rest_off && a.ix.total_cells)   // (finalize_exact_kernel: own_all)
        hipLaunchKernelGGL(finalize_exact_own_kernel, dim3((a.ix.total_cells + kOwnCells - 1) / kOwnCells), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(finalize_exact_kernel, dim3(b_ex), dim3(256), lds_votes, stream, a);
    b_e = b_ex;
    }
    // K2b (deferred k-mers only; the kernel reads their number on the device)
    a.row_general = (int)(b_v + b_e);
    hipLaunchKernelGGL(finalize_general_kernel, dim3(b_gen), dim3(64), lds, stream, a);
    if (a.partials && a.mode < 2) {
        const int cols = a.ix.n_files * 3 + 2;
        hipLaunchKernelGGL(finalize_reduce_kernel, dim3((unsigned)cols), dim3(256), 0, stream, a, (int)(b_v + b_e + b_gen), a.zero_e, a.zero_e_n);
    }
}

// one launch zeroes everything a sample starts from: the pileup arrays (`big`, 16 bytes per thread and step) and the small buffers
__global__ __launch_bounds__(256) void zero_small_kernel(unsigned long long* a, size_t na, unsigned long long* b, size_t nb, unsigned long long* c, size_t nc,
                                                         unsigned char* d, size_t nd, unsigned int* e, size_t ne, unsigned long long* big, size_t nbig) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
    for (size_t i = t; i < na; i += n) a[i] = 0;
    for (size_t i = t; i < nb; i += n) b[i] = 0;
    for (size_t i = t; i < nc; i += n) c[i] = 0;
    for (size_t i = t; i < nd; i += n) d[i] = 0;
    for (size_t i = t; i < ne; i += n) e[i] = 0;
    ulonglong2* big2 = reinterpret_cast<ulonglong2*>(big);   // (device allocations are aligned far beyond 16 bytes)
    for (size_t i = t; i < nbig / 2; i += n) big2[i] = make_ulonglong2(0ull, 0ull);
    if (t == 0 && (nbig & 1)) big[nbig - 1] = 0;
}
void launch_zero_small(unsigned long long* a, size_t na, unsigned long long* b, size_t nb, unsigned long long* c, size_t nc,
                       unsigned char* d, size_t nd, unsigned int* e, size_t ne, unsigned long long* big, size_t nbig, hipStream_t stream) {
    const unsigned grid = (unsigned)std::max<size_t>(16, std::min<size_t>((nbig / 2 + 255) / 256, 2048));
    hipLaunchKernelGGL(zero_small_kernel, dim3(grid), dim3(256), 0, stream, a, na, b, nb, c, nc, d, nd, e, ne, big, nbig);
}

// ---- sharded finalize: the small additive results as one u64 vector [stats 2*n_files*3 | present 2*n_files | kstats 8 | flag]
// (flag: a transport packer of this sample met a counter too large for its width, see xport_pack_kernel; summed over the ranks
// like the rest, so that every rank knows)
__global__ void pack_sums_kernel(unsigned long long* sums, const unsigned long long* stats, const unsigned char* present,
                                 const unsigned long long* kstats, int n_files, unsigned long long* xflag) {
    const int n_s = 2 * n_files * 3, n_p = 2 * n_files;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_s + n_p + 9; i += gridDim.x * blockDim.x)
        sums[i] = i < n_s ? stats[i] : i < n_s + n_p ? (unsigned long long)present[i - n_s] : i < n_s + n_p + 8 ? kstats[i - n_s - n_p] : xflag[0];
}
__global__ void unpack_sums_kernel(const unsigned long long* sums, unsigned long long* stats, unsigned char* present,
                                   unsigned long long* kstats, int n_files, unsigned long long* xflag) {
    const int n_s = 2 * n_files * 3, n_p = 2 * n_files;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_s + n_p + 9; i += gridDim.x * blockDim.x) {
        if (i < n_s) stats[i] = sums[i];
        else if (i < n_s + n_p) present[i - n_s] = sums[i] ? 1 : 0;
        else if (i < n_s + n_p + 8) kstats[i - n_s - n_p] = sums[i];
        else { if (sums[i]) xflag[1] = 1ull; xflag[0] = 0ull; }   // [1]: sticky until bk_transport_overflow reads it
    }
}
void launch_pack_sums(unsigned long long* sums, const unsigned long long* stats, const unsigned char* present, const unsigned long long* kstats,
                      int n_files, unsigned long long* xflag, hipStream_t stream) {
    hipLaunchKernelGGL(pack_sums_kernel, dim3(8), dim3(256), 0, stream, sums, stats, present, kstats, n_files, xflag);
}
void launch_unpack_sums(const unsigned long long* sums, unsigned long long* stats, unsigned char* present, unsigned long long* kstats,
                        int n_files, unsigned long long* xflag, hipStream_t stream) {
    hipLaunchKernelGGL(unpack_sums_kernel, dim3(8), dim3(256), 0, stream, sums, stats, present, kstats, n_files, xflag);
}

// ---- sharded finalize: a counter plane on its way to the other ranks (bk_shard_measure / bk_shard_transport / bk_shard_received)
// The plane's u64 elements are counts (E part, pseudo rows) and differences of counts (V rows) that wrap modulo 2^64; read as
// signed numbers they are small.  A reduce-scatter(sum) over n ranks of a narrower copy gives the same sums as long as every
// true sum fits the narrower type:
//   width 32   every element as int32.
//   width 16   two 16-bit lanes per int32 word (RCCL has no 16-bit integer type, and none is needed: lanes that hold unsigned
//              numbers whose sums stay below 2^16 add up inside a 32-bit addition without carrying into each other).  A V
//              element v, |v| <= L = 32767 / n, travels as v + L; the receiver takes n L off the sum.  An E count c < 2^32
//              travels as four 8-bit digits (c >> 8 q) & 255, one per lane -- sums of at most 64 digits stay below 2^14 -- and
//              the receiver puts sum_q digit_q << 8 q back together.
// Part p of the transport buffer holds the plane's elements [p P, (p + 1) P) (P = plane length / n; never cuts a V row) and is
// padded to the size of part 0, the one with the most E elements.  The packers check the sufficient local condition
// |element| <= (type's maximum) / n and raise *flag otherwise: the flag travels with the sample's statistics to every rank.
struct XportGeom {
    uint64_t plane_len, part_len, v_off;   // v_off: elements below it are E counts (or the zero padding behind them)
    uint32_t n_parts;
    uint64_t part_elems;                   // transport elements per part (16-bit lanes or int32)
};
BK_HD uint64_t xport_e_in_part(const XportGeom& g, uint64_t p) {
    const uint64_t lo = p * g.part_len;
    return g.v_off > lo ? (g.v_off - lo < g.part_len ? g.v_off - lo : g.part_len) : 0ull;
}
__global__ __launch_bounds__(256) void xport_measure_kernel(const unsigned long long* __restrict__ plane, XportGeom g, unsigned long long* out /* [2] max E, max |V| */) {
    unsigned long long me = 0, mv = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < g.plane_len; i += (uint64_t)gridDim.x * 256) {
        const unsigned long long v = plane[i];
        if (i < g.v_off) me = max(me, v);
        else { const long long sv = (long long)v; mv = max(mv, (unsigned long long)(sv < 0 ? -sv : sv)); }
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) { me = max(me, (unsigned long long)__shfl_xor(me, off)); mv = max(mv, (unsigned long long)__shfl_xor(mv, off)); }
    if ((threadIdx.x & 63) == 0) { if (me) atomicMax(out, me); if (mv) atomicMax(out + 1, mv); }
}
template <int WIDTH>
__global__ __launch_bounds__(256) void xport_pack_kernel(const unsigned long long* __restrict__ plane, XportGeom g, void* buf, unsigned long long* flag) {
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < g.plane_len; i += (uint64_t)gridDim.x * 256) {
        const uint64_t p = i / g.part_len, j = i - p * g.part_len;
        const unsigned long long v = plane[i];
        const long long sv = (long long)v;
        if (WIDTH == 32) {
            bad |= sv > (long long)(0x7fffffffll / g.n_parts) || sv < -(long long)(0x7fffffffll / g.n_parts);
            static_cast<int*>(buf)[p * g.part_elems + j] = (int)sv;
        } else {
            const uint64_t n_e = xport_e_in_part(g, p);
            unsigned short* part = static_cast<unsigned short*>(buf) + p * g.part_elems;
            const long long lim = 32767 / g.n_parts;
            if (j < n_e) {
                bad |= v >> 32 != 0ull;
                *reinterpret_cast<uint2*>(part + 4 * j) = make_uint2((uint32_t)(v & 255u) | ((uint32_t)((v >> 8) & 255u) << 16),
                                                                      (uint32_t)((v >> 16) & 255u) | ((uint32_t)((v >> 24) & 255u) << 16));
            } else {
                bad |= sv > lim || sv < -lim;
                part[4 * n_e + (j - n_e)] = (unsigned short)(sv + lim);
            }
        }
    }
    if (WIDTH == 16) {   // the padding lanes behind each part's elements
        for (uint64_t p = blockIdx.x; p < g.n_parts; p += gridDim.x) {
            const uint64_t used = g.part_len + 3 * xport_e_in_part(g, p);
            unsigned short* part = static_cast<unsigned short*>(buf) + p * g.part_elems;
            for (uint64_t x = used + threadIdx.x; x < g.part_elems; x += 256) part[x] = 0;
        }
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1ull);
}
template <int WIDTH>
__global__ __launch_bounds__(256) void xport_unpack_kernel(const void* __restrict__ recv, XportGeom g, uint32_t shard, unsigned long long* reduced) {
    const uint64_t n_e = xport_e_in_part(g, shard);
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < g.part_len; j += (uint64_t)gridDim.x * 256) {
        long long v;
        if (WIDTH == 32) v = static_cast<const int*>(recv)[j];
        else if (j < n_e) {
            const unsigned short* d = static_cast<const unsigned short*>(recv) + 4 * j;
            v = (long long)d[0] + ((long long)d[1] << 8) + ((long long)d[2] << 16) + ((long long)d[3] << 24);
        } else v = (long long)static_cast<const unsigned short*>(recv)[4 * n_e + (j - n_e)] - (long long)(32767 / g.n_parts) * g.n_parts;
        reduced[j] = (unsigned long long)v;
    }
}
static XportGeom xport_geom(uint64_t plane_len, uint64_t v_off, uint32_t n_parts, int width) {
    XportGeom g{plane_len, plane_len / n_parts, v_off, n_parts, 0};
    g.part_elems = width == 16 ? (g.part_len + 3 * xport_e_in_part(g, 0) + 3) / 4 * 4 : g.part_len;   // (parts stay 8-byte aligned)
    return g;
}
uint64_t xport_part_bytes(uint64_t plane_len, uint64_t v_off, uint32_t n_parts, int width) {
    return width == 64 ? plane_len / n_parts * 8 : xport_geom(plane_len, v_off, n_parts, width).part_elems * (uint64_t)(width / 8);
}
void launch_xport_measure(const unsigned long long* plane, uint64_t plane_len, uint64_t v_off, unsigned long long* out, hipStream_t stream) {
    hipLaunchKernelGGL(xport_measure_kernel, dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((plane_len + 255) / 256, 2048))), dim3(256), 0, stream, plane,
                       xport_geom(plane_len, v_off, 1, 32), out);
}
void launch_xport_pack(const unsigned long long* plane, uint64_t plane_len, uint64_t v_off, uint32_t n_parts, int width, void* buf, unsigned long long* flag,
                       hipStream_t stream) {
    const XportGeom g = xport_geom(plane_len, v_off, n_parts, width);
    const dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((plane_len + 255) / 256, 4096)));
    if (width == 16) hipLaunchKernelGGL(xport_pack_kernel<16>, grid, dim3(256), 0, stream, plane, g, buf, flag);
    else hipLaunchKernelGGL(xport_pack_kernel<32>, grid, dim3(256), 0, stream, plane, g, buf, flag);
}
void launch_xport_unpack(const void* recv, uint64_t plane_len, uint64_t v_off, uint32_t n_parts, uint32_t shard, int width, unsigned long long* reduced,
                         hipStream_t stream) {
    const XportGeom g = xport_geom(plane_len, v_off, n_parts, width);
    const dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((g.part_len + 255) / 256, 4096)));
    if (width == 16) hipLaunchKernelGGL(xport_unpack_kernel<16>, grid, dim3(256), 0, stream, recv, g, shard, reduced);
    else hipLaunchKernelGGL(xport_unpack_kernel<32>, grid, dim3(256), 0, stream, recv, g, shard, reduced);
}

}  // namespace bk
