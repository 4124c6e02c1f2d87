// bk_gather.hip -- the votes of map_kmers (call.rs:1305-1384) GATHERED cell by cell instead of scattered k-mer by k-mer: the
// voting pass of a many-genome index (sparse planes; BASELINE config 5: 100 strains, k = 31).
//
// map_kmers walks the distinct read k-mers and, for each, the BucketInfos of its window buckets: with a hundred related genomes
// a bucket holds ~70 of them, a sample casts 300 M votes, and the general kernels (bk_kernels.hip K2a / K2e / K2b) spend their time
// on scattered 64-bit atomics (87 M + 51 M + 13 M per sample, profiles/r05_a_config5_pmc_literal.json).  Turned around: a BucketInfo
// {location c0, idx j, canonical} of the index is an OCCURRENCE of a reference k-mer u (at cell c0, orientation `canonical`) seen
// through wildcard position j, and the k-mers that vote for it are exactly the strand-specific k-mers "u with any base at j" --
// four bases x two read orientations = eight counters, all of them known without a search:
//     the base u has there      the E counters of u                                              (E[2 id + isrc])
//     another base b            what the answer table says "u with b at j" is (DirtyAns, built at create for Level 2): another
//                               reference k-mer (its E counters), a k-mer counted in a V row (its own or, by the naming rule of
//                               bk_device.h, a neighbour's), one counted under a pseudo k-mer (k = 31), or -- u clean -- u's own row.
// Its vote goes to pileup row c0 + j (call.rs:1334 / :1361).  So pileup position P collects the BucketInfos (P - j, j), j in the
// window: W occurrences x 8 counters, and because ids follow the reference, all of them read ONE stretch of the answer table and
// ONE position q of the V plane.  A workgroup owns 64 positions: it casts their votes into a table in LDS and STORES the four
// pileup arrays' rows -- no atomics, no BucketInfo lists, no file bitmaps, nothing to zero beforehand; every genome's rows
// (bk_params.pileup_selected_only = 0) are the same kernel over all cells.
//
// What makes it exact (IndexView::gather_ok, checked at create): every window bucket sits under one key and holds every
// occurrence of its k-mers exactly once (what `bronko build` writes), and every dirty reference k-mer has its answers.  V rows are
// difference arrays: prefix_rows_kernel turns the sample's touched rows into counts first (they are zeroed behind the sample
// anyway).  What the gather cannot see are votes that reach a bucket through its ALIAS key (k = 31: the other exact rank that
// wraps onto the bucket's id; IndexView::slot_alias) -- k-mers that match a pseudo k-mer: the general kernels still cast those,
// and only those (FinalizeArgs::gather), from the pseudo rows, the pseudo k-mers' E counters and the alias hits the statistics
// pass noted among its deferred k-mers (alias_votes_kernel).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "bk_device.h"
#include "bk_kernels.h"
#include "bk_scan_common.h"
#include "bk_finalize_common.h"

namespace bk {

#ifdef BK_TESTING
#define BK_ABLATE2(a, x) ((a).gather_ablate == (x))
#else
#define BK_ABLATE2(a, x) false
#endif
constexpr uint32_t kGatherPos = 64;      // pileup positions per workgroup
constexpr int kGatherBlock = 256;

// the listed V rows of the reference k-mers: difference arrays -> counts, in place (lanes per row = v_span, as in K2a)
// (row_bits, if not null: one bit per V row, set for the listed ones -- voter_table_kernel asks it before it goes to the plane)
__global__ __launch_bounds__(256) void prefix_rows_kernel(unsigned long long* __restrict__ vc, const unsigned int* __restrict__ v_list,
                                                          const unsigned int* __restrict__ n_list, uint32_t span, unsigned int* __restrict__ row_bits) {
    const uint32_t rl = span + 1u, gpw = 64u / span, lane64 = threadIdx.x & 63u;
    const uint32_t grp = lane64 / span, oo = lane64 - grp * span;
    const bool lane_on = grp < gpw;
    const uint64_t n_listed = n_list[0];
    const uint32_t rpb = 4u * gpw;   // rows per workgroup and round
    for (uint64_t r0 = (uint64_t)blockIdx.x * rpb; r0 < n_listed; r0 += (uint64_t)gridDim.x * rpb) {
        const uint64_t li = r0 + (threadIdx.x >> 6) * gpw + grp;
        const bool on = lane_on && li < n_listed;
        const size_t at = on ? (size_t)v_list[li] * rl + oo : 0;
        unsigned long long n = on ? vc[at] : 0ull;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const unsigned long long t = __shfl_up(n, off, 64);   // (lane - off is in the same row whenever oo >= off)
            if (oo >= (uint32_t)off) n += t;
        }
        if (on) vc[at] = n;
        if (on && oo == 0u && row_bits) atomicOr(&row_bits[v_list[li] >> 5], 1u << (v_list[li] & 31u));
    }
}

// Where the two counters (one per read orientation) of the k-mer "reference k-mer `id` (canonical form u) with base b at canonical
// position j" are -- kind 1: E, at + isrc; kind 2: a V row's counter, v_off + at for direction 0, + the row length for direction 1
// (direction = isrc ^ rcu); kind 3: a pseudo k-mer's, v_off + at + isrc; 0: there is no such strand-specific k-mer in this bucket
// (its canonical form lies on the other strand) or it touches nothing
// (ur: the reverse complement of u -- that of "u with b at j" is ur with the complement of b at the mirrored position: one
// reversal per occurrence instead of one per voter)
__device__ __forceinline__ uint32_t voter_counters(const IndexView& ix, uint32_t id, uint32_t flags, uint64_t u, uint64_t ur, uint32_t j, uint32_t b, uint64_t& z,
                                                   uint64_t& at, uint32_t& rcu) {
    const int k = ix.k;
    const int sh = 2 * (k - 1 - (int)j);
    const uint32_t own = (uint32_t)(u >> sh) & 3u, rcid = (flags >> 1) & 1u;
    z = (u & ~(3ull << sh)) | ((uint64_t)b << sh);
    rcu = 0u;
    if (b == own) { at = 2ull * id; return 1u; }
    if (!(z < ((ur & ~(3ull << (2 * (int)j))) | ((uint64_t)(3u - b) << (2 * (int)j))))) return 0u;
    const uint32_t o = rcid ? (uint32_t)k - 1u - j : j;          // offset in the coordinates of the k-mer's first occurrence (answer table, V rows)
    if (!(flags & kIdDirty)) {                                    // a clean k-mer's neighbours are counted in its own rows
        const int oo = (int)o - ix.v_omin;
        if (oo < 0 || oo >= ix.v_span) return 0u;
        rcu = rcid;
        at = v_row_base(id + (uint32_t)oo, v_alt(b, own), 0u, ix.v_span) + (uint32_t)oo;
        return 2u;
    }
    const uint2 ans = *reinterpret_cast<const uint2*>(ix.dirty_ans + ans_index(id, o, rcid ? 3u - b : b, k));
    rcu = (ans.y >> 2) & 1u; at = ans.x;
    return ans.y & 3u;
}

// the counter of voter_counters()' answer for read orientation isrc, as an index into a mate file's plane -- selects, one load behind
// them: as three loads in the arms of an if / else-if / else, hipcc 7.2 left the address of the third arm's second orientation
// undefined when all lanes of a wave took it (the k = 31 fuzz case profiles/r05_fuzz.txt names; a memory fault)
__device__ __forceinline__ uint64_t counter_index(uint32_t kind, uint64_t at, uint32_t rcu, uint32_t isrc, uint64_t v_off, uint32_t rl) {
    const uint64_t in_row = kind == 2u ? ((((isrc ^ rcu) & 1u) != 0u) ? (uint64_t)rl : 0ull) : (uint64_t)isrc;
    return (kind == 1u ? 0ull : v_off) + at + in_row;
}

// WIDE: counts are capped above 2^32 (bk_params.cs; KMC's default is 10^6): the maxima need 64 bits -- a compare-and-swap loop in
// LDS, which the W lanes of a position fight over (7.4 ms for 0.9 M cells where the 32-bit form, one ds_max_u32 each, takes a tenth)
template <bool WIDE>
__global__ __launch_bounds__(kGatherBlock) void gather_votes_kernel(FinalizeArgs a, const unsigned long long* __restrict__ counters1 /* the second mate file's plane, or null */) {
    typedef typename std::conditional<WIDE, unsigned long long, unsigned int>::type max_t;
    __shared__ max_t mx[8][kGatherPos];
    __shared__ unsigned int cnt[8][kGatherPos];
    const IndexView& ix = a.ix;
    // votes for the selected genome only (mode 2): its cells; every genome's rows (mode 3): all cells
    uint32_t c_lo = 0u, c_hi = ix.total_cells;
    if (a.mode == 2) {
        const int sel = *a.sel;
        if (sel < 0) return;
        c_lo = a.file_cell_lo[sel];
        c_hi = sel + 1 < ix.n_files ? a.file_cell_lo[sel + 1] : ix.total_cells;
    }
    const uint64_t p0_64 = (uint64_t)c_lo + (uint64_t)blockIdx.x * kGatherPos;
    if (p0_64 >= c_hi) return;
    const uint32_t P0 = (uint32_t)p0_64;
    for (uint32_t i = threadIdx.x; i < 8u * kGatherPos; i += kGatherBlock) { (&mx[0][0])[i] = (max_t)0; (&cnt[0][0])[i] = 0u; }
    __syncthreads();
    const uint32_t W = (uint32_t)ix.W, span = (uint32_t)ix.v_span, rl = span + 1u;
    const unsigned long long* const planes[2] = {a.counters, counters1};
    const int n_planes = counters1 ? 2 : 1;
    const uint64_t v_off = ix.v_off;
    // pair (position, window position): the W pairs of a position sit in neighbouring lanes -- their answers and V counters are
    // neighbours in memory (one diagonal of the answer table, one position q of the plane)
    for (uint32_t w = threadIdx.x; w < kGatherPos * W; w += kGatherBlock) {
        const uint32_t pi = w / W, t = w - pi * W;
        const uint32_t P = P0 + pi, j = (uint32_t)ix.wstart + t;
        if (P >= c_hi || P < j) continue;
        const uint32_t c0 = P - j;                                   // the occurrence: a reference k-mer starts here
        const uint32_t id = ix.id_at[c0];
        if (id >= ix.n_full) continue;                               // (none: a sequence's last k - 1 cells; an occurrence lies in one sequence, so does P)
        const bool canon = (ix.cell_flags[c0] & 3u) == 2u;           // BucketInfo::canonical: this occurrence was reverse-complemented to become canonical
        const uint4 idr = *reinterpret_cast<const uint4*>(ix.id_rec + id);
        const uint64_t u = (uint64_t)idr.x | ((uint64_t)idr.y << 32);
        const uint64_t ur = revcomp_kmer(u, ix.k);
#pragma unroll
        for (uint32_t b = 0; b < 4u; ++b) {
            uint64_t z, at;
            uint32_t rcu;
            uint32_t kind = voter_counters(ix, id, (BK_ABLATE2(a, 3) ? idr.w & ~kIdDirty : idr.w), u, ur, j, b, z, at, rcu);
            if (kind == 0u) continue;
            if (BK_ABLATE2(a, 2) || BK_ABLATE2(a, 3)) { kind = 1u; at = 2ull * id; }
            uint32_t bit_idx;                                        // (vote(): call.rs:1327-1384, BucketInfo {cell c0 + j, idx j, canonical})
            if (canon) bit_idx = ((uint32_t)(z >> (2u * j)) & 3u) ^ 3u; else bit_idx = b;
            for (int m = 0; m < n_planes; ++m) {
                const unsigned long long* const pl = planes[m];
#pragma unroll
                for (uint32_t isrc = 0; isrc < 2u; ++isrc) {
                    const unsigned long long n = pl[counter_index(kind, at, rcu, isrc, v_off, rl)];
                    if (n == 0ull || n < a.ci || n > a.cx) continue;     // kmc -ci / -cx act on the true count
                    const unsigned long long v = n > a.cs ? a.cs : n;   // kmc -cs: reported count saturates
                    const bool forward = canon ? isrc != 0u : isrc == 0u;
                    const uint32_t row = (forward ? 0u : 4u) + bit_idx;
                    if (BK_ABLATE2(a, 1)) { if (v == 0x123456789ull) cnt[row][pi] = 1u; continue; }
                    __hip_atomic_fetch_add(&cnt[row][pi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if constexpr (WIDE) atomicMax(&mx[row][pi], v);
                    else __hip_atomic_fetch_max(&mx[row][pi], (unsigned int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    __syncthreads();
    // the four arrays' rows of the workgroup's positions: 64 x 4 values each, stored (nothing else has written them in this sample)
    for (uint32_t i = threadIdx.x; i < kGatherPos * 4u; i += kGatherBlock) {
        const uint32_t pi = i >> 2, base = i & 3u;
        if (P0 + pi >= c_hi) continue;
        const size_t cell = ((size_t)P0 + pi) * 4 + base;
        a.pileup[0 * a.plane + cell] = (unsigned long long)mx[base][pi];
        a.pileup[1 * a.plane + cell] = (unsigned long long)mx[4u + base][pi];
        a.pileup[2 * a.plane + cell] = (unsigned long long)cnt[base][pi];
        a.pileup[3 * a.plane + cell] = (unsigned long long)cnt[4u + base][pi];
    }
}

// Every genome's rows (mode 3) with many related genomes: a reference k-mer has an occurrence in nearly every one of them, and the
// eight voters of (k-mer, window position) are the same for all of its occurrences -- 100 strains: 0.75 M k-mers, 3.0 M occurrences.
// voter_table_kernel finds them once per (k-mer, window position) -- answers, counters, thresholds -- and writes eight words;
// gather_table_kernel is the gather with two 16-byte loads in the place of all that.  A word: bits 0-27 the largest reported count
// of the mate files that pass -ci / -cx (so -cs < 2^28: KMC's default is 10^6; otherwise the direct kernel), bits 28-29 how many
// pass; word 0 also carries, in bits 30-31, the k-mer's base at the mirrored position (what a reverse-complemented occurrence votes
// for: vote()'s `canonical` branch).
constexpr uint32_t kVtCountBits = 28;
constexpr uint32_t kVoterBlock = 512;   // = the (k-mer, window position) pairs of a workgroup's tile: each thread has one
__global__ __launch_bounds__(kVoterBlock) void voter_table_kernel(FinalizeArgs a, const unsigned long long* __restrict__ counters1, uint32_t* __restrict__ tab,
                                                          const unsigned int* __restrict__ row_bits /* which V rows the sample touched (prefix_rows_kernel) */,
                                                          unsigned long long rl_recip /* ceil(2^64 / row length) */) {
    const IndexView& ix = a.ix;
    const uint32_t W = (uint32_t)ix.W, rl = (uint32_t)ix.v_span + 1u;
    const uint64_t v_real = v_real_len(ix.n_full, ix.v_span);
    // The table is laid out [half of the entry][t][id]: the lanes of gather_table_kernel are neighbouring positions, their k-mers
    // neighbouring ids, and a load of theirs is one stretch of 1 KB (laid out [id][t] every lane's entry was a line of its own, and
    // the texture unit takes a divergent load line by line: 0.50 -> 0.34 ms).  The threads here stay (id, t), the window positions
    // of a k-mer side by side -- their voters' counters are neighbours (a pseudo k-mer's rows follow (id, t)); with the ids side by
    // side this kernel took 1.46 ms instead of 0.55 -- so a workgroup takes 512 / W k-mers, a thread per pair, turns their entries round in LDS and
    // stores rows of them (16 bytes a lane straight from the registers: 0.65 ms).
    extern __shared__ __attribute__((aligned(16))) unsigned char vt_smem[];
    uint4* const rows = reinterpret_cast<uint4*>(vt_smem);       // [2 W][tile + 1]
    const uint32_t tile = kVoterBlock / W, stride = tile + 1u;
    const uint32_t id0 = blockIdx.x * tile, n_here = min(tile, ix.n_full - id0);
    const unsigned long long* const planes[2] = {a.counters, counters1};
    const int n_planes = counters1 ? 2 : 1;
    for (uint32_t x = threadIdx.x; x < n_here * W; x += kVoterBlock) {
        const uint32_t idl = x / W, t = x - idl * W, id = id0 + idl, j = (uint32_t)ix.wstart + t;
        const uint4 idr = *reinterpret_cast<const uint4*>(ix.id_rec + id);
        const uint64_t u = (uint64_t)idr.x | ((uint64_t)idr.y << 32);
        const uint64_t ur = revcomp_kmer(u, ix.k);
        uint32_t out[8];
#pragma unroll
        for (uint32_t b = 0; b < 4u; ++b) {
            uint64_t z, at;
            uint32_t rcu;
            const uint32_t kind = voter_counters(ix, id, idr.w, u, ur, j, b, z, at, rcu);
            // a V row of a reference k-mer (most voters are: "u with b at j" is no reference k-mer and takes its own row) that the
            // sample did not touch is all zero: one bit of a 0.5 MB map instead of a line of the 1 GB plane (a sample touches 6 % of them)
            uint32_t row0 = 0u;
            const bool by_row = kind == 2u && at < v_real;
            if (by_row) row0 = (uint32_t)__umul64hi(at, rl_recip);
#pragma unroll
            for (uint32_t isrc = 0; isrc < 2u; ++isrc) {
                uint32_t np = 0u, mxv = 0u;
                bool look = kind != 0u;
                if (by_row) { const uint32_t row = row0 + ((isrc ^ rcu) & 1u); look = (row_bits[row >> 5] >> (row & 31u)) & 1u; }
                if (look)
                    for (int m = 0; m < n_planes; ++m) {
                        const unsigned long long n = planes[m][counter_index(kind, at, rcu, isrc, ix.v_off, rl)];
                        if (n == 0ull || n < a.ci || n > a.cx) continue;
                        ++np;
                        mxv = max(mxv, (uint32_t)(n > a.cs ? a.cs : n));
                    }
                out[2u * b + isrc] = (np << kVtCountBits) | mxv;
            }
        }
        out[0] |= ((uint32_t)(u >> (2u * j)) & 3u) << 30;
        rows[t * stride + idl] = make_uint4(out[0], out[1], out[2], out[3]);
        rows[(W + t) * stride + idl] = make_uint4(out[4], out[5], out[6], out[7]);
    }
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < 2u * W * n_here; x += kVoterBlock) {
        const uint32_t r = x / n_here, idl = x - r * n_here;     // r = half * W + t
        *reinterpret_cast<uint4*>(tab + ((uint64_t)r * ix.n_full + id0 + idl) * 4u) = rows[r * stride + idl];
    }
}

// A lane per position and a wave per window position (every fourth), the lane's sixteen sums and maxima in registers.  Where an
// entry goes: an occurrence as written (not reverse-complemented to become canonical) sends voter (b, isrc) to base b, forward for
// isrc = 0 -- the same register for every window position; a reverse-complemented one sends all four bases' voters to ONE base,
// the complement of the k-mer's base at the mirrored position (vote()'s `canonical` branch) -- which is the reference's own base
// at the pileup position, the same for every window position of the lane: they are summed apart and added to that base at the
// end (the middle position of an odd k, where the mirrored position is the voter's own, goes base by base).  No branch and no
// load behind a branch: the compiler sends the loads of all the wave's window positions ahead of the arithmetic.
// (Round 5's form kept the sums in LDS, 32 read-modify-writes per entry behind one another, asked for an entry only after the one
// before was done and walked the genomes one after the other: 0.59 ms per sample at 100 strains, 2.9 GB from the fabric, 42 M LDS
// instructions; this one 0.50 ms, 0.70 GB, 3 M -- what is left is its 141 M vector instructions: at this coverage every entry of
// the table has a count, 624 M sums and maxima per sample.)
constexpr uint32_t kGatherWaves = kGatherBlock / 64;
constexpr uint32_t kGatherTMax = (32u + kGatherWaves - 1u) / kGatherWaves;   // window positions per wave (W <= k <= 31 ... 32)
__global__ __launch_bounds__(kGatherBlock) void gather_table_kernel(FinalizeArgs a, const uint32_t* __restrict__ tab) {
    constexpr uint32_t kWaves = kGatherWaves;
    __shared__ unsigned int red[kWaves][16][kGatherPos];
    const IndexView& ix = a.ix;
    // Related genomes are collinear: stretch s of every one of them reads the same ~50 KB of the table.  Workgroups are handed to
    // the eight XCDs in turn, so blockIdx = (s / 8, genome, s % 8): the genomes' copies of a stretch follow each other on ONE XCD
    // and find the table's lines in its L2.
    const uint32_t nf = (uint32_t)ix.n_files, n = blockIdx.x >> 3;
    const uint32_t g = n % nf, s = (n / nf) * 8u + (blockIdx.x & 7u);
    const uint32_t g_lo = a.file_cell_lo[g], c_hi = g + 1u < nf ? a.file_cell_lo[g + 1u] : ix.total_cells;
    const uint64_t p0_64 = (uint64_t)g_lo + (uint64_t)s * kGatherPos;
    if (p0_64 >= c_hi) return;
    const uint32_t P0 = (uint32_t)p0_64;
    const uint32_t W = (uint32_t)ix.W, km1 = (uint32_t)ix.k - 1u;
    const uint32_t pi = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t P = P0 + pi;
    constexpr uint32_t kMask = (1u << kVtCountBits) - 1u;
    uint32_t cnt[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, mx[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // [0..3] forward, [4..7] reverse
    uint32_t rc_cnt[2] = {0u, 0u}, rc_mx[2] = {0u, 0u}, rc_base = 0u;                                     // the reverse-complemented occurrences'
    uint32_t ids[kGatherTMax];
    uint32_t w_bits = 0u, c_bits = 0u;                           // bit x: the lane has an occurrence at the wave's x-th window position, as written / reverse-complemented
#pragma unroll
    for (uint32_t x = 0; x < kGatherTMax; ++x) {
        const uint32_t t = wave + x * kWaves, j = (uint32_t)ix.wstart + t;
        const bool in = t < W && P < c_hi && P >= j;
        const uint32_t c0 = in ? P - j : 0u;
        const uint32_t id = ix.id_at[c0];
        const bool canon = (ix.cell_flags[c0] & 3u) == 2u, on = in && id < ix.n_full;
        ids[x] = on ? id : 0u;
        w_bits |= (on && !canon ? 1u : 0u) << x;
        c_bits |= (on && canon ? 1u : 0u) << x;
    }
#pragma unroll
    for (uint32_t x = 0; x < kGatherTMax; ++x) {
        const uint32_t t = wave + x * kWaves, j = (uint32_t)ix.wstart + t;
        if (t >= W) break;                                       // (wave-uniform)
        const uint64_t at = (uint64_t)t * ix.n_full + ids[x];                        // (voter_table_kernel: [half][t][id])
        const uint4 lo = *reinterpret_cast<const uint4*>(tab + at * 4u), hi = *reinterpret_cast<const uint4*>(tab + ((uint64_t)ix.n_full * W + at) * 4u);
        const uint32_t vals[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const uint32_t cb = lo.x >> 30;
        // (masks, not selects: a select per sum and maximum was two thirds of the kernel's instructions)
        const uint32_t m_w = (uint32_t)__builtin_amdgcn_sbfe((int)w_bits, x, 1u), m_c = (uint32_t)__builtin_amdgcn_sbfe((int)c_bits, x, 1u);   // 0 or ~0
        uint32_t np[8], v[8];
#pragma unroll
        for (uint32_t e = 0; e < 8u; ++e) { np[e] = (vals[e] >> kVtCountBits) & 3u; v[e] = vals[e] & kMask; }
        if (2u * j == km1) {                                     // (wave-uniform; odd k only) the mirrored position is the voter's own
#pragma unroll
            for (uint32_t b = 0; b < 4u; ++b)
#pragma unroll
                for (uint32_t isrc = 0; isrc < 2u; ++isrc) {
                    const uint32_t e = 2u * b + isrc;
                    // as written: base b, forward for isrc 0; reverse-complemented: base 3 - b, forward for isrc 1
                    const uint32_t r_w = isrc * 4u + b, r_c = (1u - isrc) * 4u + (3u - b);
                    cnt[r_w] += np[e] & m_w; mx[r_w] = max(mx[r_w], v[e] & m_w);
                    cnt[r_c] += np[e] & m_c; mx[r_c] = max(mx[r_c], v[e] & m_c);
                }
            continue;
        }
#pragma unroll
        for (uint32_t e = 0; e < 8u; ++e) {
            const uint32_t r_w = (e & 1u) * 4u + (e >> 1);
            cnt[r_w] += np[e] & m_w;
            mx[r_w] = max(mx[r_w], v[e] & m_w);
        }
        // reverse-complemented: forward for isrc = 1 (the odd entries)
        const uint32_t s1 = np[1] + np[3] + np[5] + np[7], s0 = np[0] + np[2] + np[4] + np[6];
        const uint32_t m1 = max(max(v[1], v[3]), max(v[5], v[7])), m0 = max(max(v[0], v[2]), max(v[4], v[6]));
        rc_cnt[0] += s1 & m_c; rc_mx[0] = max(rc_mx[0], m1 & m_c);
        rc_cnt[1] += s0 & m_c; rc_mx[1] = max(rc_mx[1], m0 & m_c);
        rc_base = (rc_base & ~m_c) | ((cb ^ 3u) & m_c);
    }
#pragma unroll
    for (uint32_t r = 0; r < 4u; ++r) {
        const bool here = rc_base == r;
        cnt[r] += here ? rc_cnt[0] : 0u; mx[r] = max(mx[r], here ? rc_mx[0] : 0u);
        cnt[4u + r] += here ? rc_cnt[1] : 0u; mx[4u + r] = max(mx[4u + r], here ? rc_mx[1] : 0u);
    }
#pragma unroll
    for (uint32_t r = 0; r < 8u; ++r) { red[wave][r][pi] = cnt[r]; red[wave][8u + r][pi] = mx[r]; }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kGatherPos * 4u; i += kGatherBlock) {
        const uint32_t p = i >> 2, base = i & 3u;
        if (P0 + p >= c_hi) continue;
        unsigned int m0 = 0u, m1 = 0u, n0 = 0u, n1 = 0u;
#pragma unroll
        for (uint32_t w = 0; w < kWaves; ++w) {
            n0 += red[w][base][p]; n1 += red[w][4u + base][p];
            m0 = max(m0, red[w][8u + base][p]); m1 = max(m1, red[w][12u + base][p]);
        }
        const size_t cell = ((size_t)P0 + p) * 4 + base;
        a.pileup[0 * a.plane + cell] = (unsigned long long)m0;
        a.pileup[1 * a.plane + cell] = (unsigned long long)m1;
        a.pileup[2 * a.plane + cell] = (unsigned long long)n0;
        a.pileup[3 * a.plane + cell] = (unsigned long long)n1;
    }
}

// the alias hits the statistics pass noted (FinalizeArgs::alias_hits: {canonical k-mer | isrc << 62, count, slot}): the k-mer
// votes for the BucketInfos of that slot (vote() keeps to the selected genome in mode 2)
__global__ __launch_bounds__(256) void alias_votes_kernel(FinalizeArgs a) {
    if (a.mode == 2) { a.sel_file = *a.sel; if (a.sel_file < 0) return; }
    const IndexView& ix = a.ix;
    const unsigned int n = min(*a.n_alias_hits, a.alias_cap);
    for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const unsigned long long c_ = a.alias_hits[3ull * i], v0 = a.alias_hits[3ull * i + 1ull], sb = a.alias_hits[3ull * i + 2ull];
        const uint64_t c = c_ & 0x3fffffffffffffffull;
        const uint32_t isrc = (uint32_t)(c_ >> 62) & 1u;
        const unsigned long long v = v0 > a.cs ? a.cs : v0;
        const uint32_t off = ix.ent_off[sb], cnt = ix.ent_len[sb];
        for (uint32_t q = 0; q < cnt; ++q) vote(a, ix.entries[off + q], c, isrc, ix.k, v);
    }
}

// Buckets that hold several keys (k = 31: two reference buckets whose ids wrapped onto each other; FinalizeArgs::merged_slots): what
// votes through one key also votes for the BucketInfos of the others (call.rs:1307-1309 walks the bucket's whole Vec) -- and those
// the gather, which goes by occurrence and key, does not see.  One wave per (merged bucket, window key): the key's eight voters,
// found as the gather finds them from one k-mer of the key, vote for every BucketInfo of the bucket that is not under that key.
__global__ __launch_bounds__(64) void merged_votes_kernel(FinalizeArgs a) {
    if (a.mode == 2) { a.sel_file = *a.sel; if (a.sel_file < 0) return; }
    const IndexView& ix = a.ix;
    const int k = ix.k;
    const uint32_t lane = threadIdx.x, rl = (uint32_t)ix.v_span + 1u;
    for (uint32_t mi = blockIdx.x; mi < a.n_merged_slots; mi += gridDim.x) {
        const uint64_t w0 = a.merged_slots[2ull * mi], key = a.merged_slots[2ull * mi + 1ull];
        const uint32_t sb = (uint32_t)w0, j = (uint32_t)ix.wstart + (uint32_t)(w0 >> 32);
        const uint32_t off = ix.ent_off[sb], cnt = ix.ent_len[sb];
        const int sh = 2 * (k - 1 - (int)j);
        // a k-mer of this key: the first BucketInfo of the bucket that stands under it
        uint32_t rep_id = 0xffffffffu;
        for (uint32_t q0 = 0; q0 < cnt && rep_id == 0xffffffffu; q0 += 64u) {
            const uint32_t q = q0 + lane;
            uint32_t id = 0xffffffffu;
            if (q < cnt) {
                const DevEntry e = ix.entries[off + q];
                if ((uint32_t)e.idx == j && e.cell >= j) {
                    const uint32_t i2 = ix.id_at[e.cell - j];
                    if (i2 < ix.n_full && (ix.id_rec[i2].kmer & ~(3ull << sh)) == key) id = i2;
                }
            }
            const unsigned long long bm = __ballot(id != 0xffffffffu);
            if (bm) rep_id = (uint32_t)__shfl((int)id, __builtin_ctzll(bm));
        }
        if (rep_id == 0xffffffffu) continue;   // (cannot be: the key came from one of the bucket's BucketInfos)
        const uint4 idr = *reinterpret_cast<const uint4*>(ix.id_rec + rep_id);
        const uint64_t u = (uint64_t)idr.x | ((uint64_t)idr.y << 32);
        const uint64_t ur = revcomp_kmer(u, k);
        for (uint32_t b = 0; b < 4u; ++b) {
            uint64_t z, at;
            uint32_t rcu;
            const uint32_t kind = voter_counters(ix, rep_id, idr.w, u, ur, j, b, z, at, rcu);
            if (kind == 0u) continue;
            for (uint32_t isrc = 0; isrc < 2u; ++isrc) {
                const unsigned long long n = a.counters[counter_index(kind, at, rcu, isrc, ix.v_off, rl)];
                if (n == 0ull || n < a.ci || n > a.cx) continue;
                const unsigned long long v = n > a.cs ? a.cs : n;
                for (uint32_t q = lane; q < cnt; q += 64u) {
                    const DevEntry e = ix.entries[off + q];
                    bool own_key = false;
                    if ((uint32_t)e.idx == j && e.cell >= j) {
                        const uint32_t i2 = ix.id_at[e.cell - j];
                        own_key = i2 < ix.n_full && (ix.id_rec[i2].kmer & ~(3ull << sh)) == key;
                    }
                    if (!own_key) vote(a, e, z, isrc, k, v);   // (the key's own BucketInfos were voted for cell by cell)
                }
            }
        }
    }
}

// pileup_selected_only: the rows of the genome the previous sample voted for are all that a new sample finds written
__global__ __launch_bounds__(256) void zero_genome_rows_kernel(unsigned long long* pileup, size_t plane, const uint32_t* file_cell_lo, int n_files, uint32_t total_cells,
                                                               const int* last_sel) {
    const int sel = *last_sel;
    if (sel < 0 || sel >= n_files) return;
    const size_t lo = (size_t)file_cell_lo[sel] * 4, hi = (size_t)(sel + 1 < n_files ? file_cell_lo[sel + 1] : total_cells) * 4;
    for (size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (size_t)gridDim.x * 256) {
        pileup[i] = 0ull; pileup[plane + i] = 0ull; pileup[2 * plane + i] = 0ull; pileup[3 * plane + i] = 0ull;
    }
}
__global__ void copy_int_kernel(int* dst, const int* src) { *dst = *src; }
void launch_zero_genome_rows(unsigned long long* pileup, size_t plane, const uint32_t* file_cell_lo, int n_files, uint32_t total_cells, const int* last_sel,
                             hipStream_t stream) {
    hipLaunchKernelGGL(zero_genome_rows_kernel, dim3(128), dim3(256), 0, stream, pileup, plane, file_cell_lo, n_files, total_cells, last_sel);
}
void launch_copy_int(int* dst, const int* src, hipStream_t stream) { hipLaunchKernelGGL(copy_int_kernel, dim3(1), dim3(1), 0, stream, dst, src); }

void launch_prefix_rows(unsigned long long* counters, const IndexView& ix, const unsigned int* v_list, const unsigned int* n_list, unsigned int* row_bits, hipStream_t stream) {
    if (ix.v_span <= 0) return;
    hipLaunchKernelGGL(prefix_rows_kernel, dim3(2048), dim3(256), 0, stream, counters + ix.v_off, v_list, n_list, (uint32_t)ix.v_span, row_bits);
}
void launch_gather_votes(const FinalizeArgs& a, const unsigned long long* counters1, hipStream_t stream) {
    const uint64_t cells = a.mode == 2 ? (uint64_t)a.max_file_cells : (uint64_t)a.ix.total_cells;
    if (!cells) return;
    const dim3 grid((unsigned)((cells + kGatherPos - 1) / kGatherPos));
    if (a.cs < (1ull << 32)) hipLaunchKernelGGL(gather_votes_kernel<false>, grid, dim3(kGatherBlock), 0, stream, a, counters1);
    else hipLaunchKernelGGL(gather_votes_kernel<true>, grid, dim3(kGatherBlock), 0, stream, a, counters1);
}
bool vote_table_fits(const FinalizeArgs& a) { return a.mode == 3 && a.cs < (1ull << kVtCountBits); }
size_t vote_table_words(const IndexView& ix) { return (size_t)ix.n_full * (size_t)ix.W * 8u; }
void launch_gather_votes_table(const FinalizeArgs& a, const unsigned long long* counters1, uint32_t* tab, const unsigned int* row_bits, hipStream_t stream) {
    const uint64_t n_pairs = (uint64_t)a.ix.n_full * (uint64_t)a.ix.W;
    if (!n_pairs || !a.ix.total_cells) return;
    const uint32_t tile = kVoterBlock / (uint32_t)a.ix.W;        // (W <= 32: at least 16 k-mers)
    hipLaunchKernelGGL(voter_table_kernel, dim3((a.ix.n_full + tile - 1u) / tile), dim3(kVoterBlock), (size_t)2 * a.ix.W * (tile + 1u) * sizeof(uint4), stream, a, counters1, tab, row_bits,
                       ~0ull / (unsigned long long)(a.ix.v_span + 1) + 1ull);
    const uint64_t stretches = ((uint64_t)a.max_file_cells + kGatherPos - 1) / kGatherPos;   // of the longest genome
    hipLaunchKernelGGL(gather_table_kernel, dim3((unsigned)((stretches + 7) / 8 * 8 * (uint64_t)a.ix.n_files)), dim3(kGatherBlock), 0, stream, a, (const uint32_t*)tab);
}
void launch_merged_votes(const FinalizeArgs& a, hipStream_t stream) {
    if (!a.n_merged_slots) return;
    hipLaunchKernelGGL(merged_votes_kernel, dim3(std::min<uint32_t>(a.n_merged_slots, 1024u)), dim3(64), 0, stream, a);
}
void launch_alias_votes(const FinalizeArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(alias_votes_kernel, dim3(64), dim3(256), 0, stream, a);
}

}  // namespace bk
