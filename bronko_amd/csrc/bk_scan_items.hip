// bk_scan_items.hip -- the binned scan (K1 of round 4): reads -> 16-bit items grouped by region of the reference -> plane.
//
// scan_items_kernel   settles every read exactly as scan_count_kernel (bk_kernels.hip) does -- seeds -> diagonal, mismatch flags 160
//                     bases at a time, one mismatch per lane, E runs / S runs / marks for nbatch_kernel and level2_kernel -- but
//                     what that kernel ADDS (two LDS atomics per E run into a whole-genome difference array that pins one
//                     workgroup to a CU; two global atomics per S run into the V plane) this one EMITS as 16-bit items into small
//                     per-bin buckets in LDS (bk_kernels.h ItemGeom).  No 117 KB array, no slab, no prefix sum over the genome, no
//                     hot-counter table.  One workgroup of sixteen waves per CU (kItemBlock = 1024): the window's reference, its
//                     reverse complement and the flag arrays are staged once per CU, the bucket area (45 KB) is one, and the kernel
//                     allocates all 128 VGPRs so that nothing shares its CU (three workgroups of eight waves were slower: smaller
//                     sets of buckets overflow earlier).
// bin_count_kernel    one workgroup per bin: the bin's items of every scan workgroup (+ the overflow list) added up in LDS -- an E
//                     bin in two 384-cell difference arrays (reads along / against the reference), a V bin in the bin's counters --
//                     and what is not zero added to the u64 plane: E[2 id_at[cell] + orientation] as fold_kernel did, V counters
//                     one atomic each, neighbouring counters by neighbouring lanes.  Every count of a sample still lands on the
//                     same counter of the same plane as before: nbatch / level2 / finalize / the sharded transport are untouched.
//                     BinArgs::part: every bin, the E bins only (a mate file's first launch: its V items wait for the regional
//                     finalize, bk_finalize_lean.hip FinalizeArgs::f_items) or the V bins only (they go to the plane after all).
// Replaces call.rs:1152-1255 (what KMC does: bin, then count) for the k-mers of reads on the window genome; SURVEY.md 7.2 K1.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

#include "bk_device.h"
#include "bk_kernels.h"
#include "bk_scan_common.h"

namespace bk {

constexpr int kItemBlock = 1024;            // one workgroup of 16 waves per CU: the window's reference is staged once per CU
constexpr int kItemWaves = kItemBlock / 64;
constexpr int kItemGroupsPerCu = 1;
constexpr uint32_t kListCap = 256;          // entries of a wave's list of mismatch positions (a tile with more goes round by round of lanes)
constexpr uint32_t kListWords = kListCap + 4u;   // ... with an empty entry in front and two behind
constexpr uint32_t kDealRing = 32;          // the chunks of tiles a workgroup has been dealt, by ordinal (modulo this): {ordinal, chunk}
constexpr size_t kItemLdsFixed = (4 + 4 + 2 * kDealRing + kItemWaves * kListWords) * sizeof(unsigned int);   // k-mer tally, the workgroup's tile counter, its chunks, the waves' lists
constexpr int kBinBlock = 256;
constexpr uint32_t kMaxChunkMismatches = 24;   // more differences than this in a read's first 160 bases: not a read of that diagonal
constexpr uint32_t kStageMaxWords = 12;    // records of up to 192 bases are staged in LDS (48 KB for the workgroup's 16 waves)

// An item that found neither room in its bucket nor in the overflow list: straight to the plane (never lost, never fast).
__device__ __forceinline__ void item_direct(const __attribute__((address_space(4))) ScanArgs* ap, uint32_t bin, uint32_t item, uint32_t win_lo) {
    const __attribute__((address_space(4))) ScanArgs& a = *ap;
    if (bin < a.ig.n_ebins) {
        const uint32_t c0 = win_lo + (bin << kEBinLog2) + (item & 127u), n = ((item >> 7) & 255u) + 1u, rev = item >> 15;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t cell = c0 + i;
            const uint32_t id = a.id_at[cell];
            const uint32_t rc = ((a.cell_codes[kRefPadWords + (cell >> 4)] >> (2u * (cell & 15u))) & 3u) == 2u ? 1u : 0u;
            atomicAdd(a.counters + 2 * (size_t)id + (rc ^ rev), 1ull);
        }
    } else {
        const uint64_t at = (uint64_t)(bin - a.ig.n_ebins) * ((6ull << a.ig.vq_log2) * (uint32_t)(a.v_span + 1)) + (item & 0x7fffu);
        atomicAdd(a.counters + a.v_off + at, (item & 0x8000u) ? ~0ull : 1ull);
    }
}

// KT: k as a compile-time constant for the common sizes, 0 = any k.  Dense planes, the window's reference in LDS.
// STAGED: every wave keeps the 64 records of its tile in LDS and has the NEXT tile's copied there (LDS-DMA: global_load_lds, 16
// bytes per lane and instruction, no register in between) while it works on this one's mismatches -- a lane reads its record's
// words from LDS, not at a 40-byte stride from memory, and the chain length -> words -> hash -> seed table is one trip to memory
// per tile instead of three.  Records of more than kStageMaxWords words (long reads) are read from memory as before.
template <int KT, bool STAGED>
__global__ __launch_bounds__(kItemBlock) void scan_items_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned int* block_kmers = reinterpret_cast<unsigned int*>(smem);   // 16 B reserved
    unsigned int* wg_ctr = block_kmers + 4;         // [1] the workgroup's next tile that no wave has taken (16 B reserved)
    uint2* ring = reinterpret_cast<uint2*>(wg_ctr + 4);   // [kDealRing] {ordinal, chunk of 16 tiles}: what this workgroup was dealt (below)
    unsigned int* lst_all = wg_ctr + 4 + 2 * kDealRing;   // [waves][kListWords] the positions of a tile's mismatches, lane after lane (below)
    unsigned int* cnt = lst_all + kItemWaves * kListWords;   // [n_bins] items of each bin (the first cap in its bucket, the rest in its extension in device memory)
    const uint32_t n_eb = a.ig.n_ebins, n_bins = n_eb + a.ig.n_vbins, cap_e = a.ig.cap_e, cap_v = a.ig.cap_v;
    const uint32_t nb_pad = (n_bins + 3u) & ~3u;
    unsigned short* buck = reinterpret_cast<unsigned short*>(cnt + nb_pad);   // E bin b: [b * cap_e, + cap_e); V bins behind them, cap_v each
    unsigned int* lds_ref = reinterpret_cast<unsigned int*>(smem + ((kItemLdsFixed + (size_t)nb_pad * 4u + (size_t)a.ig.wg_stride * 2u + 7u) & ~(size_t)7u));

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform by construction: tell the compiler
    unsigned int* const lst = lst_all + (uint32_t)wave * kListWords;
    // The arguments only rare paths read (overflow, marks for Level 2, the fallback seeds, start and end of the workgroup) are read
    // again from the kernel-argument segment where they are used, through a pointer the compiler cannot see through: held in
    // scalar registers for the whole tile loop they cost ~60 of the 102 there are, and the loop paid for it in v_readlane /
    // v_writelane spill traffic (617 of its 5,500 instructions).
    typedef const __attribute__((address_space(4))) ScanArgs* ColdArgs;
    const ColdArgs cold0 = (ColdArgs)__builtin_amdgcn_kernarg_segment_ptr();
    auto cold = [&]() __attribute__((always_inline)) -> ColdArgs { ColdArgs p = cold0; asm volatile("" : "+s"(p)); return p; };

    // The kernel keeps its CU to itself: 16 waves of 128 registers are a SIMD's whole file.  (At the 116 it needs, a wave of a small
    // kernel of another sample -- K2b, the reduce, the zeroing -- fits next to four of its own, lands on a scan CU instead of a free
    // one and runs a third slower there: 2.5 % of the headline.)
    asm volatile("v_mov_b32 v127, 0" ::: "v127");
    BK_DBG_CLOCK(a, 0);
    const uint32_t total = a.total_cells;
    // the window (a multiple of 64 cells from the start): chosen on the device for a multi-genome index (choose_window_kernel), else
    // the engine's constant.  Only reads whose cells all lie in it are settled here: front pad, the window's cells, back pad
    const uint32_t win_lo = a.win_dev ? a.win_dev[1] : a.win_lo;
    const uint32_t win_file = a.win_dev ? a.win_dev[0] : (uint32_t)a.win_file;
    const uint32_t lds_cells = min(total - win_lo, a.n_lds_bins);
    const uint32_t n_refw = kRefPadWords + (lds_cells + 15) / 16 + kRefBackWords;
    const uint32_t n_bitw = kBitPadWords + (lds_cells + 31) / 32 + kBitBackWords;
    const uint32_t n_blk = (lds_cells + 63) / 64 + 2;
    const uint32_t blk_w0 = (n_refw + 2u * n_bitw + 1u) & ~1u;   // (8-byte aligned: lds_ref is)
    // the same cells of the reverse-complemented reference (ScanArgs::rc_words): window cell p is its symbol rc_base + lds_cells - 1 - p
    const uint32_t rc_lo = total - win_lo - lds_cells, rc_base = rc_lo & 15u;
    const uint32_t rc_w0 = blk_w0 + 2u * n_blk;
    uint64_t n_records = a.n_records;
    if (a.n_records_dev) {
        const uint64_t nd = *a.n_records_dev;
        n_records = nd > a.rec_base ? min(nd - a.rec_base, a.n_records) : 0ull;
    }
    const uint32_t* const words0 = a.words + a.rec_base * a.stride_words;
    const uint16_t* const lens0 = a.lens + a.rec_base;
    const uint64_t n_tiles = (n_records + 63) / 64;
    // Tiles are dealt in chunks of sixteen (one per wave): a workgroup starts on chunk blockIdx.x and takes every further chunk
    // from a counter in device memory (one returning atomic per sixteen tiles: a thousand per launch; per TILE the atomics queue up
    // on their one address, 20 ns each), a chunk's tiles go to the workgroup's waves one at a time (a counter in LDS) -- a tile with
    // many mismatches keeps one wave busy while the others take what is left, and a workgroup that started late (the last of a
    // launch's 256 start 8 us after the first) or shares its CU's memory path with a sibling's kernels takes fewer chunks instead of
    // making the launch wait (until round 6 every workgroup had a fixed 61 tiles: the mean workgroup ended 6 us before the last).
    unsigned int* const deal_ctr = reinterpret_cast<unsigned int*>(cold()->ov_n + 2);   // [2] by the launch's parity: this launch's counter; the other is zeroed for the next
    const uint64_t t_hi = n_tiles;
    // STAGED: this wave's record buffer, and the copy of a tile into it.  Returns the lane's record length of that tile.
    const uint32_t sw = a.stride_words;
    unsigned int* const rec_buf = reinterpret_cast<unsigned int*>(smem + a.stage_off) + (uint32_t)wave * 64u * sw;
    auto stage = [&](uint64_t t) -> uint32_t {
        if (t >= t_hi) return 0u;
        const uint64_t rr = t * 64 + (uint32_t)lane;
        const uint32_t ln = rr < n_records ? (uint32_t)lens0[rr] : 0u;
        if ((t + 1) * 64 <= n_records) {   // (wave-uniform) a whole tile: 64 * sw words, 16 * sw units of 16 bytes
            const unsigned char* g = reinterpret_cast<const unsigned char*>(words0 + t * 64 * sw);
            unsigned char* l = reinterpret_cast<unsigned char*>(rec_buf);
            for (uint32_t u0 = 0; u0 < 16u * sw; u0 += 64u) {
                const uint32_t u = u0 + (uint32_t)lane;
                if (u < 16u * sw)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)u * 16u),
                                                     (__attribute__((address_space(3))) void*)(l + (size_t)u0 * 16u), 16, 0, 0);
            }
        } else if (rr < n_records) {       // the launch's last tile: every lane copies its own record
            const uint32_t* src = words0 + rr * sw;
            for (uint32_t j = 0; j < sw; ++j) rec_buf[(uint32_t)lane * sw + j] = src[j];
        }
        return ln;
    };
    // (the wave's first tile is sent on its way into LDS before the workgroup stages the reference: the two copies run side by side)
    uint32_t len_pf = 0u;
    const uint64_t tile0 = (uint64_t)blockIdx.x * kItemWaves + (uint32_t)wave;   // the wave's first tile: of the workgroup's own chunk
    if constexpr (STAGED) len_pf = stage(tile0);
    BK_DBG_CLOCK2(a, 0);
    uint32_t chunk1 = 0u;
    if (threadIdx.x == 0) {
        const uint32_t par = cold()->ov_par;
        if (blockIdx.x == 0) deal_ctr[par ^ 1u] = 0u;
        // the workgroup's second chunk is asked for now (the answer is picked up behind the staging below)
        chunk1 = gridDim.x + __hip_atomic_fetch_add(deal_ctr + par, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    for (uint32_t i = threadIdx.x; i < n_bins; i += kItemBlock) cnt[i] = 0u;
    if (lane == 0) lst[0] = 0u;   // (no entry in front of a list's first)
    if (threadIdx.x == 0) wg_ctr[1] = (unsigned int)kItemWaves;
    {
        // The window's arrays, five of them, as ONE stretch of words dealt to the threads eight at a time: all of a thread's loads are
        // on their way before the first is stored (five loops, each a trip to memory of its own, took 5 us of every workgroup's life).
        const ColdArgs c = cold();
        const uint32_t* const g_ref = c->ref_words + (win_lo >> 4);
        const uint32_t* const g_fast = c->cell_fast + (win_lo >> 5);
        const uint32_t* const g_c3 = c->cell_clean3 + (win_lo >> 5);
        const uint32_t* const g_blk = reinterpret_cast<const uint32_t*>(c->cell_blk + (win_lo >> 6));
        const uint32_t* const g_rc = c->rc_words + (rc_lo >> 4);
        const uint32_t c0 = n_refw, c1 = c0 + n_bitw, c2 = c1 + n_bitw, c3 = c2 + 2u * n_blk, c4 = c3 + n_refw + 1u;   // (the copy: one word more, its slice starts rc_base symbols into its first word)
        for (uint32_t b0 = 0; b0 < c4; b0 += 8u * kItemBlock) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) {
                const uint32_t i = b0 + u * kItemBlock + threadIdx.x;
                const uint32_t* src = i < c0 ? g_ref + i : i < c1 ? g_fast + (i - c0) : i < c2 ? g_c3 + (i - c1) : i < c3 ? g_blk + (i - c2) : g_rc + (i - c3);
                v[u] = i < c4 ? *src : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) {
                const uint32_t i = b0 + u * kItemBlock + threadIdx.x;
                if (i < c4) lds_ref[i < c2 ? i : i < c3 ? blk_w0 + (i - c2) : rc_w0 + (i - c3)] = v[u];
            }
        }
    }
    BK_DBG_CLOCK2(a, 1);
    if (threadIdx.x == 0) { ring[0] = make_uint2(0u, blockIdx.x); ring[1] = make_uint2(1u, chunk1); }
    __syncthreads();
    BK_DBG_CLOCK(a, 1);
    // symbol / bit 0 is cell win_lo; negative positions down to -64 are readable (padding or earlier cells)
    const unsigned int* refw1 = lds_ref + kRefPadWords;
    const unsigned int* fastw = lds_ref + n_refw + kBitPadWords;
    const unsigned int* c3w = lds_ref + n_refw + n_bitw + kBitPadWords;
    const uint2* blkw = reinterpret_cast<const uint2*>(lds_ref + blk_w0);   // entry 0 = the block of cell win_lo
    const int32_t rc_sym0 = (int32_t)(rc_w0 * 16u);   // symbol x of the reverse-complemented copy is symbol rc_sym0 + x of refw1's array

    const int k = KT ? KT : a.k;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;  // k <= 31
    const uint32_t km1 = (uint32_t)k - 1u;
    const int omin = a.v_omin, span = a.v_span;
    const uint32_t rl = (uint32_t)span + 1u, vq_log2 = a.ig.vq_log2, vqm = (1u << vq_log2) - 1u;
    const uint32_t v_buck0 = n_eb * cap_e;
    const uint32_t last_word = a.stride_words - 1u;

    // ---- the sinks: an item into its bin's bucket (slot s of the bin, counted in LDS); past the bucket's capacity into the bin's
    // extension in device memory; past that into the device-wide overflow list, and past the list's end straight to the plane ----
    auto put_ext = [&](uint32_t bin, uint32_t g /* s - capacity */, uint32_t item) {
        const ColdArgs c = cold();
        if (g < kItemGCap) { c->gext[((size_t)blockIdx.x * n_bins + bin) * kItemGCap + g] = (unsigned short)item; return; }
        const unsigned long long i = atomicAdd(c->ov_n + c->ov_par, 1ull);
        if (i < (unsigned long long)c->ov_cap) c->ov[i] = (bin << 16) | item;
        else item_direct(c, bin, item, win_lo);
    };
    // cells [c_lo, c_lo + n) (window coordinates) each seen once more by a read along (f) / against the reference
    auto emit_e = [&](bool on, uint32_t c_lo, uint32_t n, bool f) {
        uint32_t c = c_lo, left = (on && !BK_ABLATE(a, 14)) ? n : 0u;
        do {   // (a run of more than 255 cells -- a long read -- goes out in pieces)
            const uint32_t take = min(left, kERunMax);
            if (left) {
                const uint32_t bin = c >> kEBinLog2;
                const uint32_t item = (c & 127u) | ((take - 1u) << 7) | (f ? 0u : 0x8000u);
                const uint32_t s = __hip_atomic_fetch_add(&cnt[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (s < cap_e) buck[bin * cap_e + s] = (unsigned short)item; else put_ext(bin, s - cap_e, item);
            }
            c += take; left -= take;
        } while (__ballot(left != 0u));
    };

    uint32_t nkm = 0;  // k-mer occurrences of this lane's records
    // k-mers [sk, sk + n) of record `rec` (index within this launch) are an N run left to nbatch_kernel: set their bits (the
    // record's diagonal goes with the mark: nothing is written for the records without one)
    auto n_mark = [&](bool on, uint32_t rec, uint32_t sk, uint32_t n, int32_t dgm, uint32_t flm) {
        if (!on) return;
        const ColdArgs c = cold();
        unsigned int* row = c->n_bits + (size_t)rec * c->l2_words;
        atomicOr(c->n_any + (rec >> 5), 1u << (rec & 31u));
        c->l2_diag[rec] = make_uint2((uint32_t)dgm, flm);
        uint32_t w = sk >> 5, bit = sk & 31u, left = n;
        while (left) {
            const uint32_t take = min(left, 32u - bit);
            atomicOr(row + w, (take == 32u ? 0xffffffffu : (1u << take) - 1u) << bit);
            left -= take; ++w; bit = 0u;
        }
    };
    // ... or, one by one, to level2_kernel (l2_bits): the k-mers that hold two mismatches and cannot be discarded
    auto l2_mark = [&](bool on, uint32_t rec, uint32_t sk, uint32_t n, int32_t dgm, uint32_t flm) {
        if (!on) return;
        const ColdArgs c = cold();
        unsigned int* row = c->l2_bits + (size_t)rec * c->l2_words;
        atomicOr(c->l2_any + (rec >> 5), 1u << (rec & 31u));
        c->l2_diag[rec] = make_uint2((uint32_t)dgm, flm);
        uint32_t w = sk >> 5, bit = sk & 31u, left = n;
        while (left) {
            const uint32_t take = min(left, 32u - bit);
            atomicOr(row + w, (take == 32u ? 0xffffffffu : (1u << take) - 1u) << bit);
            left -= take; ++w; bit = 0u;
        }
    };
    const bool stats = a.ktab_keys != nullptr;   // full_kmer_stats: k-mers that touch nothing are still wanted by the statistics table (level2_kernel)

    // the wave's next tile: the workgroup's s-th is tile s % 16 of its (s / 16)-th chunk; whoever draws a chunk's first tile asks
    // for the chunk after it (a round of tiles before anybody needs it) and leaves it in the ring
    auto take_tile = [&]() __attribute__((always_inline)) -> uint64_t {
        uint32_t t = 0u;
        if (lane == 0) {
            const uint32_t s = __hip_atomic_fetch_add(wg_ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t q = s / kItemWaves, i = s % kItemWaves;
            volatile uint2* const rg = ring;
            if (i == 0u) {
                const uint32_t cn = gridDim.x + __hip_atomic_fetch_add(deal_ctr + cold()->ov_par, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                rg[(q + 1u) % kDealRing].y = cn;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                rg[(q + 1u) % kDealRing].x = q + 1u;
            }
            while (rg[q % kDealRing].x != q) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            t = rg[q % kDealRing].y * kItemWaves + i;
        }
        return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    };
    uint32_t pf_sink = 0;               // destination of the prefetch loads (never read)
    // the seed table of the window's genome, and where the seeds sit: evenly spaced over the launch's first record
    const uint2* const seed_tab = a.seed_tab2 ? a.seed_tab2 + ((size_t)win_file << a.seed2_log2) : nullptr;
    const uint32_t hint_len = n_records ? (uint32_t)__builtin_amdgcn_readfirstlane((int)lens0[0]) : 0u;
    const uint32_t hint_span = hint_len >= (uint32_t)k ? hint_len - (uint32_t)k : 0u;
    // chunk flags that are not resolved at a chunk's end (long reads): those of its last k - 1 bases, i.e. bits 64 - 2 (k - 1) and
    // up of the 64 flag bits of its last 32 bases (words 8 and 9, two bits per base)
    const uint32_t um_sh = 64u - 2u * km1;
    const uint32_t um8 = um_sh >= 32u ? 0u : 0xffffffffu << um_sh, um9 = um_sh >= 32u ? 0xffffffffu << (um_sh - 32u) : 0xffffffffu;

    // The seeds of a tile: three k-mers of every read (its first, its last, one in between) looked up in the seed table of the
    // window's genome (bk_device.h seed_hash) -- k-mers at positions that do not depend on the read's length (evenly spaced over the
    // launch's first record), one 8-byte bucket each.  (Asking for the NEXT tile's seeds before this tile's mismatches are dealt with
    // was tried in round 6: the carried state cost registers and instructions and the kernel was 2 % slower -- its waves wait on
    // dependent LDS steps and branches all along a tile, not on the table.)
    struct SeedSet { uint64_t g[3]; uint32_t h[3]; uint2 b[3]; };
    auto seed_pos = [&](int sround, int j) __attribute__((always_inline)) -> uint32_t {
        const int qj = sround ? kSeeds - 2 : (j == 0 ? 0 : j == 1 ? kSeeds - 1 : 1);
        return (hint_span * (uint32_t)qj) / (uint32_t)(kSeeds - 1);   // (wave-uniform)
    };
    auto seed_load = [&](const uint32_t* __restrict__ wr, int sround, int ns, SeedSet& S) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j >= ns) continue;
            S.g[j] = read_symbols_at(wr, seed_pos(sround, j), last_word) & kmask;   // the k-mer as the read shows it: base t at bits 2t
            S.h[j] = seed_hash(S.g[j]);                                              // (no reverse complement, no canonical form: the table holds both strands)
            S.b[j] = seed_tab[S.h[j] >> (32u - a.seed2_log2)];
        }
    };
    const bool use_tab = seed_tab && !BK_ABLATE(a, 9) && !BK_ABLATE(a, 11);

    for (uint64_t tile = tile0, next_tile = 0; tile < t_hi; tile = next_tile) {
        const uint64_t r = tile * 64 + lane;
        next_tile = take_tile();
        const bool live = r < n_records;
        const uint32_t r32 = live ? (uint32_t)r : 0u;   // record index within this launch (a launch has < 2^32 records)
        uint32_t len;
        if constexpr (STAGED) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the tile's records have landed in the wave's buffer (and its lengths in len_pf)
            len = len_pf;
        } else {
            len = live ? (uint32_t)lens0[r32] : 0u;
        }
        if (len < (uint32_t)k) len = 0u;   // no k-mer
        if (tile == tile0) BK_DBG_CLOCK2(a, 2);
        const uint32_t maxlen = wave_max(len);
        const uint32_t* __restrict__ w = STAGED ? rec_buf + (uint32_t)lane * sw : words0 + (uint64_t)r32 * a.stride_words;
        const uint32_t nk = len ? len - km1 : 0u;   // k-mers of the record
        nkm += nk;
        if constexpr (!STAGED) {   // touch the next tile's records (one lane per 128-byte line) so that its seed loads find them in cache
            const uint64_t nt = next_tile;
            const uint64_t first = nt * 64ull * a.stride_words, words_tile = 64ull * a.stride_words;
            const uint64_t at = first + (uint64_t)lane * 32ull;
            if (nt < t_hi && (uint64_t)lane * 32ull < words_tile && at < n_records * a.stride_words)
                asm volatile("global_load_dword %0, %1, off" : "=v"(pf_sink) : "v"(words0 + at) : "memory");
        }
        SeedSet S;
        if (use_tab && maxlen) seed_load(w, 0, 3, S);
        // STAGED: the next tile is sent on its way into the wave's buffer once this tile's words are all read (behind the flag pass:
        // nothing after it reads a record).  Every path through a tile does it exactly once.
        bool next_sent = false;
        auto send_next = [&]() __attribute__((always_inline)) {
            if constexpr (STAGED) { if (!next_sent) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); len_pf = stage(next_tile); next_sent = true; } }
        };
        auto ask_next = [&]() __attribute__((always_inline)) { send_next(); };
        if (!maxlen) { ask_next(); continue; }

        // ---- seeds -> diagonal ----------------------------------------------------------------------------------
        // Which diagonal a read is settled on decides how much of it is settled HERE, never what is counted: k-mers without a
        // difference from the reference along any diagonal are the reference k-mers of those cells, single differences at clean
        // cells belong to that cell's k-mer whatever brought the read there, everything else goes to Level 2.
        bool fwd = true, seeded = false, l1ok = false;   // seeded: diagonal known; l1ok: ... and all its cells are in the window
        int32_t dg = 0;                                  // cell of the reference k-mer aligned with read k-mer 0: k-mer s <-> dg + s (fwd) / dg - s
        // a diagonal: the whole read must lie on the reference (hi_cell + k <= total); to be settled here its cells must lie in the
        // window and each of them must carry a reference k-mer (no sequence tail in between: cell_blk).  Branch-free: one LDS read.
        // (cells stay below 2^27 -- kSeedCellBits --, read positions below 2^16: 32-bit signed arithmetic holds everything)
        auto accept = [&](bool hit, int32_t d0, bool f) __attribute__((always_inline)) {
            const int32_t span_k = (int32_t)(len - (uint32_t)k);
            const int32_t lo_cell = f ? d0 : d0 - span_k;
            const int32_t hi_cell = f ? d0 + span_k : d0;
            const bool ok = hit && len != 0u && lo_cell >= 0 && (uint32_t)hi_cell + (uint32_t)k <= total;
            const bool inw = ok && (uint32_t)lo_cell >= win_lo && (uint32_t)hi_cell < win_lo + a.n_lds_bins;
            const uint32_t tail = blkw[inw ? ((uint32_t)lo_cell >> 6) - (win_lo >> 6) : 0u].y;
            dg = ok ? d0 : dg; fwd = ok ? f : fwd; seeded = seeded || ok;
            l1ok = ok ? (inw && (uint32_t)hi_cell < tail) : l1ok;
        };
        // A candidate the table names is verified against the reference in LDS.  The first seed that verifies names the diagonal
        // (they agree unless the read is a chimera), and the diagonal is tested once per round.  The fourth seed only for the
        // lanes the three left without a diagonal.
        for (int sround = 0; sround < 2 && use_tab; ++sround) {
            if (sround == 1) {
                if (!__ballot(len != 0u && !seeded)) break;
                seed_load(w, 1, 1, S);
            }
            const int ns = sround ? 1 : 3;
            bool hit = false, hf = true;
            int32_t hd0 = 0;
#pragma unroll
            for (int j = 2; j >= 0; --j) {   // (the later seeds first: an earlier one that verifies overrides them)
                if (j >= ns) continue;
                const uint32_t s = seed_pos(sround, j);
                const uint32_t tag = S.h[j] & 15u;
                const uint64_t sg = S.g[j];
                const uint32_t ent = (S.b[j].x != 0xffffffffu && (S.b[j].x >> 28) == tag) ? S.b[j].x : S.b[j].y;
                const uint32_t cell = ent & ((1u << kSeedCellBits) - 1u), strand = (ent >> kSeedCellBits) & 1u;
                // in reach of the staged reference?  (the window's cells and 64 in front)
                const bool in_ref = !seeded && ent != 0xffffffffu && (ent >> 28) == tag && s + (uint32_t)k <= len &&
                                    cell + 64u >= win_lo && cell + (uint32_t)k <= win_lo + lds_cells;
                const int32_t cw = in_ref ? (int32_t)cell - (int32_t)win_lo : 0;
                // the reference k-mer of that cell as a read on that strand shows it: from the reference, or from its reverse-
                // complemented copy (one LDS array, one base pointer: the copy's symbols lie rc_sym0 symbols behind the reference's)
                const int32_t pos = strand ? rc_sym0 + (int32_t)(rc_base + lds_cells) - k - cw : cw;
                const uint64_t ref = symbols_at(refw1, pos) & kmask;
                const bool h = in_ref && ref == sg;
                hd0 = h ? (strand ? (int32_t)cell + (int32_t)s : (int32_t)cell - (int32_t)s) : hd0;
                hf = h ? strand == 0u : hf;
                hit = hit || h;
            }
            accept(hit, hd0, hf);
        }
        // The lanes that are still without a diagonal (an error in every seed, a k-mer that found its bucket full, a read shorter
        // than the first record, no seed table): the perfect hash of U, two rounds (scan_count_kernel has the why).
        uint32_t best_cell = 0xffffffffu;
        auto candidate = [&](bool hit, uint32_t scell, bool f, uint32_t s) __attribute__((always_inline)) {
            if (hit && scell < best_cell) {   // several seeds may hit (usually all, on one diagonal); prefer the lowest cell
                const int32_t span_k = (int32_t)(len - (uint32_t)k);
                const int32_t d0 = f ? (int32_t)scell - (int32_t)s : (int32_t)scell + (int32_t)s;
                const int32_t lo_cell = f ? d0 : d0 - span_k;
                const int32_t hi_cell = f ? d0 + span_k : d0;
                if (lo_cell >= 0 && (uint32_t)hi_cell + (uint32_t)k <= total) {
                    best_cell = scell; dg = d0; fwd = f; seeded = true;
                    l1ok = (uint32_t)lo_cell >= win_lo && (uint32_t)hi_cell < win_lo + a.n_lds_bins;
                    if (l1ok) l1ok = (uint32_t)hi_cell < blkw[((uint32_t)lo_cell >> 6) - (win_lo >> 6)].y;
                }
            }
        };
        for (int round = 0; round < 2 && !BK_ABLATE(a, 9); ++round) {   // (9: no seeds at all, 7: nothing behind them, 6: no mismatch loop)
            if (!__ballot(len != 0u && !seeded)) break;
            const ColdArgs c = cold();
            const IndexView& ix = *c->ixp;
            const uint32_t* const occ = c->occ;
            const uint32_t n_files = (uint32_t)c->n_files;
            const bool had = seeded;   // a round is for the lanes left without a diagonal so far
            uint64_t sc[kSeeds];
            uint32_t sisrc[kSeeds], spil[kSeeds], spos[kSeeds];
#pragma unroll
            for (int sq = 0; sq < kSeeds; ++sq) {
                const uint32_t span_k = len ? len - (uint32_t)k : 0u;
                const uint32_t s = round == 0 ? (span_k * (uint32_t)sq) / (uint32_t)(kSeeds - 1)
                                              : (span_k * (uint32_t)(2 * sq + 1)) / (uint32_t)(2 * kSeeds);
                spos[sq] = s;
                const uint64_t g = read_symbols_at(w, s, last_word) & kmask;
                const uint64_t rr = ~g & kmask;
                const uint64_t ff = rev2_64(g) >> (64 - 2 * k);
                const bool lt = ff < rr;                                             // lcb.rs:90-94
                sc[sq] = lt ? ff : rr;
                sisrc[sq] = lt ? 0u : 1u;
                spil[sq] = ix.pilots[phf_bucket(sc[sq], ix.log2nb)];
            }
            uint4 se[kSeeds];
#pragma unroll
            for (int sq = 0; sq < kSeeds; ++sq) se[sq] = *reinterpret_cast<const uint4*>(ix.kmer_pos + phf_pos(sc[sq], spil[sq], ix.m, ix.log2nb, ix.log2p));
            bool shit[kSeeds];
            uint32_t soc[kSeeds];
#pragma unroll
            for (int sq = 0; sq < kSeeds; ++sq) {
                shit[sq] = len && !had && ((uint64_t)se[sq].x | ((uint64_t)se[sq].y << 32)) == sc[sq];
                soc[sq] = 0xffffffffu;
                if (occ && shit[sq] && (se[sq].w & kIdMask) < ix.n_full)
                    soc[sq] = occ[(size_t)(se[sq].w & kIdMask) * n_files + win_file];
            }
#pragma unroll
            for (int sq = 0; sq < kSeeds; ++sq) {
                uint32_t scell = se[sq].z, src_rc = se[sq].w >> 31;   // where the seed sits: the k-mer's first occurrence ...
                if (soc[sq] != 0xffffffffu) { scell = soc[sq] & 0x7fffffffu; src_rc = soc[sq] >> 31; }   // ... or its occurrence in the window's genome
                candidate(shit[sq], scell, sisrc[sq] == src_rc, spos[sq]);
            }
        }
        uint32_t dfl = (fwd ? 1u : 0u) | (seeded ? 2u : 0u);
        if (BK_ABLATE(a, 9) || BK_ABLATE(a, 7)) { ask_next(); continue; }

        // ---- mismatch flags, 160 bases at a time; mismatch by mismatch --------------------------------------------------
        const int32_t dgw = dg - (int32_t)win_lo;   // the diagonal in window coordinates
        const int32_t rc_off = rc_sym0 + (int32_t)(rc_base + lds_cells) - 1 - (int32_t)km1;
        // A chunk's 160 bases stay as the XOR with the reference along the diagonal leaves them, two bits per base: bits 2 i, 2 i + 1 of
        // D[j] = read base cb + 16 j + i XOR the reference's (zero where the lane has no base).  A base differs where either bit is
        // set (flags: (d | d << 1) at the odd bits); the flags are counted where they are (popcount) and turned into positions one
        // by one (find-first-bit >> 1), and the two bits themselves say WHICH other base the read has there (bk_device.h v_alt) --
        // compressing the flags to one bit per base cost a third of the kernel's instructions and bought nothing.
        auto flags_of = [&](uint32_t d) __attribute__((always_inline)) -> uint32_t {
            uint32_t f;
            asm("v_lshl_or_b32 %0, %1, 1, %1" : "=v"(f) : "v"(d));   // d | d << 1 (the compiler folds a neighbouring op into a three-input one instead and pays a shift more)
            return f & 0xaaaaaaaau;
        };
        uint32_t D[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) D[j] = 0u;
        int32_t tp = -0x20000000;    // the lane's last resolved mismatch (absolute base), far away before the first
        for (uint32_t cb = 0;; cb += 128u) {   // (wave-uniform) the chunk covers bases [cb, cb + 160); its first 32 are carried over
            const bool act = l1ok && cb < len;
            const uint32_t vlen = act ? min(len - cb, 160u) : 0u;   // the lane's bases in this chunk
            // every active lane has at least min_v bases in the chunk, none more than max_v: the words all of them fill need no
            // mask of their own, the words none of them reaches are not looked at
            const uint32_t min_v = wave_min(act ? vlen : 0xffffffffu);
            const uint32_t max_v = maxlen > cb ? min(maxlen - cb, 160u) : 0u;
            // read base i <-> reference base dgw + i along the reference; against it <-> complement of reference base dgw + k - 1 - i,
            // which is symbol rc_off - dgw + i of the reverse-complemented copy: the same walk on the other array
            const int32_t p0 = act ? (fwd ? dgw + (int32_t)cb : rc_off - dgw + (int32_t)cb) : 0;
            const uint32_t rsh = 2u * ((uint32_t)p0 & 15u);
            const unsigned int* const rp = refw1 + (p0 >> 4);
            const uint32_t lmask = act ? 0xffffffffu : 0u;
            const uint32_t wi0 = cb >> 4;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                if (cb != 0u && j < 2) continue;
                if (16u * (uint32_t)j >= max_v) { D[j] = 0u; continue; }
                // (STAGED: words past the record's end belong to the next lane's record or to the padding behind the buffers; what
                // they give is masked away below)
                const uint32_t x = STAGED ? w[wi0 + (uint32_t)j] : w[min(wi0 + (uint32_t)j, last_word)];
                const uint32_t y = __builtin_amdgcn_alignbit(rp[j + 1], rp[j], rsh);   // 16 bases of that strand, in reading order
                uint32_t d = (x ^ y) & lmask;
                if (16u * (uint32_t)(j + 1) > min_v) {   // (wave-uniform) some lane's bases end in this word or before it
                    asm volatile("" ::: "memory");       // (a branch, not a select: nine words in ten of a launch of equal reads skip this)
                    const int32_t amt = min(max(2 * (int32_t)vlen - 32 * j, 0), 32);
                    d &= (uint32_t)(1ull << amt) - 1u;     // (amt = 32: the low word of 2^32 is 0, minus one = all ones)
                }
                D[j] = d;
            }
            const uint32_t scanned = cb + 160u;
            const bool last_chunk = scanned >= maxlen;
            // STAGED: the record's words are all read: the wave's next tile is sent on its way into the same buffer.  (Nothing below
            // reads a record: what an item needs of the read's base at its mismatch is in the flag pass's XOR.)
            if (last_chunk && !BK_ABLATE(a, 6)) send_next();
            uint32_t nfl = 0u;   // flags of the lane in this chunk
#pragma unroll
            for (int j = 0; j < 10; ++j) nfl += (uint32_t)__popc(flags_of(D[j]));
            if (cb == 0u) {
                // A read that differs from the reference all along its diagonal -- a chimera, an adapter, a diagonal that a repeat's
                // k-mer vouched for -- is no read of this diagonal: it goes to Level 2 whole, like a read without one.  (Settled
                // here it would keep its wave busy for a hundred mismatches while the workgroup's other waves run out of tiles: six
                // such reads in a million set the kernel's time.)
                const bool off_diag = l1ok && nfl > kMaxChunkMismatches;
                if (off_diag) {
                    l1ok = false; dfl &= ~2u; nfl = 0u;
#pragma unroll
                    for (int j = 0; j < 10; ++j) D[j] = 0u;
                }
                // a read that cannot be settled here is one N run
                if (cold()->n_direct) l2_mark(nk != 0u && !l1ok, r32, 0u, nk, dg, dfl);   // (ScanArgs::n_direct: straight to Level 2's marks)
                else n_mark(nk != 0u && !l1ok, r32, 0u, nk, dg, dfl);
            }
            // A mismatch is resolved once the k - 1 bases behind it are scanned (or the read ends): that far reach the k-mers that
            // hold it, and whatever they hold of its successors is then known.  What is not resolved lies in the chunk's last
            // 32 bases (k <= 31): the words that are carried over.
            const bool all_res = scanned >= len;
            if (BK_ABLATE(a, 6)) {
                nfl = 0u;
#pragma unroll
                for (int j = 0; j < 10; ++j) D[j] = 0u;
            }
            // resolved flags of the lane: all of them, or all but those of words 8 and 9 above the line
            const uint32_t cr = (last_chunk || all_res) ? nfl : nfl - (uint32_t)__popc(flags_of(D[8] & um8)) - (uint32_t)__popc(flags_of(D[9] & um9));
            // ---- one mismatch per lane: the tile's flags are dealt out over the wave ----
            // Every lane writes the positions of its flags into the wave's list (lst: position | lane << 16 | 1 << 23 | XOR << 24,
            // the lanes' stretches one behind the other); lane i of a pass then takes entry i: its mismatch t, the entries next to
            // it -- the same read's previous and next two, if the owner is the same -- and the owner's diagonal, length and record
            // through three shuffles.  (Until round 6 the owner's 160 flags travelled to every item lane through five shuffles,
            // were popped one by one there, and a running maximum over the lanes found the owner.)  A tile with more flags than
            // the list holds goes round by round of whole lanes.
            const uint32_t pin = wave_incl_add(nfl), pex = pin - nfl;
            const uint32_t n_all = (uint32_t)__builtin_amdgcn_readlane((int)pin, 63);
            uint32_t last_pos = 0u;   // absolute position of the lane's last flag (low 16 bits)
            const uint32_t own_w0 = ((uint32_t)dgw << 3) | (all_res ? 4u : 0u) | dfl;   // what an item needs of its owner, word 0
            for (uint32_t lane0 = 0u; n_all != 0u && lane0 < 64u;) {
                const bool one = n_all <= kListCap;   // (wave-uniform) everything fits: the common case
                const uint32_t base = one ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)pex, (int)lane0);
                const bool in_r = one || ((uint32_t)lane >= lane0 && pin - base <= kListCap);   // (a lane has at most 160 flags: lane0 always is)
                const uint32_t lane1 = one ? 64u : lane0 + (uint32_t)__popcll(__ballot(in_r));
                const uint32_t n_r = one ? n_all : (uint32_t)__builtin_amdgcn_readlane((int)pin, (int)lane1 - 1) - base;
                {
                    unsigned int* slot = lst + 1u + (pex - base);
                    const uint32_t tag = ((uint32_t)lane << 16) | (1u << 23) | cb;   // (cb is a multiple of 128 below 2^16: | is +)
#pragma unroll
                    for (int j = 0; j < 10; ++j) {
                        uint32_t f = in_r ? flags_of(D[j]) : 0u;
                        for (;;) {
                            const bool has = f != 0u;
                            if (!__ballot(has)) break;
                            const uint32_t bit = (uint32_t)__builtin_ctz(f | 0x80000000u);            // 2 i + 1 (31 where f is 0: flags sit at odd bits)
                            const uint32_t x2 = (D[j] >> ((bit - 1u) & 31u)) & 3u;                      // read base XOR reference base there
                            const uint32_t e = (tag + 16u * (uint32_t)j + (bit >> 1)) | (x2 << 24);
                            if (has) { *slot = e; ++slot; last_pos = e; }
                            f &= f - 1u;
                        }
                    }
                    if (lane == 0) { lst[n_r + 1u] = 0u; lst[n_r + 2u] = 0u; }   // (no entry: bit 23 is clear)
                }
                __builtin_amdgcn_wave_barrier();
                for (uint32_t it0 = 0; it0 < n_r; it0 += 64u) {
                    const uint32_t ix0 = it0 + (uint32_t)lane;
                    const bool in_list = ix0 < n_r;
                    const uint32_t e_m = lst[ix0], e_0 = lst[ix0 + 1u], e_1 = lst[ix0 + 2u], e_2 = lst[ix0 + 3u];
                    const int src = in_list ? (int)((e_0 >> 16) & 63u) : lane;
                    // the owner's read: diagonal, flags, length, record, last mismatch before this chunk's
                    // (every shuffle outside the conditions: a lane that is off may be another's owner)
                    const uint32_t o_w0 = (uint32_t)__shfl((int)own_w0, src);
                    const uint32_t o_nk = (uint32_t)__shfl((int)nk, src);
                    const uint32_t o_rec = (uint32_t)__shfl((int)r32, src);
                    int32_t tq = cb ? __shfl(tp, src) : -0x20000000;   // becomes the mismatch before mine
                    const int32_t o_dgw = (int32_t)o_w0 >> 3;
                    const uint32_t o_fl = o_w0 & 3u;
                    const bool ofwd = o_fl & 1u;
                    auto same = [&](uint32_t e) { return ((e ^ e_0) & 0x00bf0000u) == 0u; };   // an entry, and of the same lane
                    const int32_t t = in_list ? (int32_t)(e_0 & 0xffffu) : 0;
                    // mine is resolved?  (an unresolved one waits for the next chunk; its lane idles)
                    const bool on = in_list && ((o_w0 & 4u) || (uint32_t)t - cb + km1 < 160u);
                    if (same(e_m)) tq = (int32_t)(e_m & 0xffffu);
                    const int32_t tn = (in_list && same(e_1)) ? (int32_t)(e_1 & 0xffffu) : 0x20000000;   // (an unseen one is out of reach)
                    const int32_t tn2 = (in_list && same(e_2)) ? (int32_t)(e_2 & 0xffffu) : 0x20000000;
                    // The k-mers between the previous mismatch and this one hold none: an E run (they start behind the previous one and
                    // end before this one)
                    const uint32_t done = tq < 0 ? 0u : min((uint32_t)tq, o_nk - 1u) + 1u;
                    {
                        const bool eg = on && t >= k && (uint32_t)(t - k) >= done && done < o_nk;
                        const uint32_t g_hi = min((uint32_t)(t - k), o_nk - 1u);
                        emit_e(eg, (uint32_t)(ofwd ? o_dgw + (int32_t)done : o_dgw - (int32_t)g_hi), g_hi - done + 1u, ofwd);
                    }
                    // The k-mers whose FIRST mismatch is t: they start behind the previous one and hold t.  Every k-mer that holds a
                    // mismatch belongs to exactly one such range.
                    const uint32_t o_lo = (uint32_t)max(max(t - (int32_t)km1, tq + 1), 0), o_hi = min((uint32_t)t, o_nk - 1u);
                    const bool own = on && o_lo <= o_hi;
                    // Their cells, lowest first (window coordinates), and which of them are "fast" (clean, ids = cell + one constant).
                    const int32_t ca = own ? (ofwd ? o_dgw + (int32_t)o_lo : o_dgw - (int32_t)o_hi) : 0;
                    const uint32_t n_own = own ? o_hi - o_lo + 1u : 0u;           // (at most k <= 31)
                    uint32_t pat = bits32_at(fastw, ca) & ((1u << n_own) - 1u);   // bit p: the cell ca + p is fast
                    // which of the three other bases: read XOR reference at the mismatch, the same on either strand (bk_device.h)
                    const uint32_t tt = own ? (uint32_t)t : 0u;
                    const int32_t pr = own ? (ofwd ? o_dgw + (int32_t)tt : o_dgw + (int32_t)km1 - (int32_t)tt) : 0;   // the reference position of the mismatch
                    const uint32_t alt = ((e_0 >> 24) & 3u) - 1u;   // (the flag pass's XOR: read against reference along, or complement against it -- the same)
                    // Cells that are not fast: per (position of the mismatch, other base) the offsets at which the k-mer still takes its own
                    // row (IndexView::cell_nat); with cell_natrow the bits stand for one V row of their own, whatever cell_blk says.
                    bool by_row = false;
                    uint32_t nat_row = 0u;
                    if (a.cell_nat && own && pat != (1u << n_own) - 1u && !BK_ABLATE(a, 12)) {
                        const uint32_t nm = (__brev(a.cell_nat[((size_t)(pr + (int32_t)win_lo)) * 3u + alt]) >> (31u - (uint32_t)(pr - ca))) & ((1u << n_own) - 1u);
                        if (a.cell_natrow) { pat = nm; by_row = true; nat_row = a.cell_natrow[(size_t)(pr + (int32_t)win_lo)]; }
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
                            if (has_s && lo2 <= hi2 && !BK_ABLATE(a, 2) && !BK_ABLATE(a, 14)) {
                                const uint32_t idS = by_row ? nat_row - of_first                       // (cell_natrow: id + offset of every k-mer of the run)
                                                            : (uint32_t)(ofwd ? c0 : c1) + win_lo + ba.x;   // id of the cell of k-mer x_lo (cell_fast: ids = cell + constant)
                                // one row of the V plane: +1 at the first offset, -1 after the last (slot `span` is never read) -- two items
                                // of the row's bin (rows never straddle bins)
                                const uint32_t q = idS + of_first - (uint32_t)omin;
                                const uint32_t bin = n_eb + (q >> vq_log2);
                                const uint32_t o0 = (((q & vqm) * 3u + alt) * 2u + (ofwd ? 0u : 1u)) * rl + (uint32_t)(lo2 - omin);
                                const bool tail = hi2 - omin + 1 < span;
                                const uint32_t s = __hip_atomic_fetch_add(&cnt[bin], tail ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                unsigned short* const bk = buck + v_buck0 + (bin - n_eb) * cap_v;
                                if (s < cap_v) bk[s] = (unsigned short)o0; else put_ext(bin, s - cap_v, o0);
                                if (tail) {
                                    const uint32_t o1 = (o0 + (uint32_t)(hi2 - lo2 + 1)) | 0x8000u;
                                    if (s + 1u < cap_v) bk[s + 1u] = (unsigned short)o1; else put_ext(bin, s + 1u - cap_v, o1);
                                }
                            }
                        }
                        {
                            // The k-mers that hold t and its successor.  If none of them reaches the successor after that and each of their
                            // cells has no other reference k-mer form within Hamming distance 3, they hold exactly two differences from a
                            // reference k-mer that is isolated up to distance 3: neither a reference k-mer nor one base away from one
                            // (triangle inequality) -- they touch nothing.  Otherwise level2_kernel looks at them one by one.
                            // (those of them that stop short of tn2 -- [xm_lo, xd_hi] -- are judged on their own: in a stretch of three
                            // close mismatches only the k-mers that hold all three are Level 2's)
                            const int32_t xd_hi_i = min((int32_t)x_hi, tn2 - k);
                            const bool has_d = has_m && !stats && xd_hi_i >= (int32_t)xm_lo;
                            const uint32_t xd_hi = has_d ? (uint32_t)xd_hi_i : xm_lo;
                            const int32_t ma = has_d ? (ofwd ? o_dgw + (int32_t)xm_lo : o_dgw - (int32_t)xd_hi) : 0;
                            const uint32_t needm = has_d ? 0xffffffffu >> (31u - (xd_hi - xm_lo)) : 0u;
                            const bool dead = has_d && (bits32_at(c3w, ma) & needm) == needm;
                            const uint32_t xl = dead ? xd_hi + 1u : xm_lo;   // first k-mer that is marked
                            l2_mark(has_m && xl <= x_hi, o_rec, xl, x_hi + 1u - xl, o_dgw + (int32_t)win_lo, o_fl);
                        }
                        // cells that are not fast
                        l2_mark(go && !fast, o_rec, x_lo, ln, o_dgw + (int32_t)win_lo, o_fl);
                        used += ln;
                    }
                }
                lane0 = lane1;
            }
            if (last_chunk) ask_next();   // (a tile without a flag: nothing above did)
            // each lane: its last resolved mismatch; what is not resolved stays
            if (cr) {
                if (last_chunk || all_res) tp = (int32_t)(last_pos & 0xffffu);
                else {   // (long reads) the highest flag below the line
                    uint32_t hi = 0u, hj = 0u;
#pragma unroll
                    for (int j = 0; j < 10; ++j) { const uint32_t f = flags_of(j == 8 ? D[8] & ~um8 : j == 9 ? D[9] & ~um9 : D[j]); if (f) { hi = f; hj = (uint32_t)j; } }
                    tp = (int32_t)(cb + 16u * hj + ((31u - (uint32_t)__builtin_clz(hi)) >> 1));
                }
            }
            if (scanned >= maxlen) break;
            // the next chunk starts 128 bases on: what was not resolved is its first 32 bases' flags
            D[0] = all_res ? 0u : D[8] & um8; D[1] = all_res ? 0u : D[9] & um9;
        }
        {   // behind the last mismatch: k-mers [done, nk - 1]
            const uint32_t done = tp < 0 ? 0u : min((uint32_t)tp, nk - 1u) + 1u;
            const bool eg = l1ok && done < nk;
            emit_e(eg, (uint32_t)(fwd ? dgw + (int32_t)done : dgw - (int32_t)(nk - 1u)), nk - done, fwd);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(pf_sink) : "memory");   // the prefetch register stays reserved up to here

    // ---- the buckets, as they are, into this workgroup's region: the whole bucket area in 16-byte units (slots that hold nothing
    // go out as they are -- tab says how many of a bin's slots count), then the table ----
    if (threadIdx.x == 0) *block_kmers = 0;
    __syncthreads();
    BK_DBG_CLOCK(a, 2);
    {
        const ColdArgs c = cold();
        // (bin-major in device memory: a bin's buckets of all workgroups are one stretch that bin_count reads front to back -- with a
        // region per workgroup its threads fetched 96 bytes each from 256 places 45 KB apart, three times the bytes they used)
        uint4* const out4 = reinterpret_cast<uint4*>(c->items);
        const uint4* const buck4 = reinterpret_cast<const uint4*>(buck);
        const uint32_t ue = cap_e / 8u, uv = cap_v / 8u, n_eu = n_eb * ue, G = a.ig.grid_max;   // 16-byte units per bucket
        for (uint32_t i = threadIdx.x; i < a.ig.wg_stride / 8u; i += kItemBlock) {
            size_t dst;
            if (i < n_eu) { const uint32_t bin = i / ue; dst = ((size_t)bin * G + blockIdx.x) * ue + (i - bin * ue); }
            else { const uint32_t i2 = i - n_eu, bin = i2 / uv; dst = (size_t)n_eu * G + ((size_t)bin * G + blockIdx.x) * uv + (i2 - bin * uv); }
            out4[dst] = buck4[i];
        }
        for (uint32_t bin = threadIdx.x; bin < n_bins; bin += kItemBlock)
            c->tab[(size_t)bin * G + blockIdx.x] = (unsigned short)min(cnt[bin], (bin < n_eb ? cap_e : cap_v) + kItemGCap);
    }
    BK_DBG_CLOCK2(a, 3);
    uint32_t tot = nkm;
#pragma unroll
    for (int off = 32; off; off >>= 1) tot += (uint32_t)__shfl_xor((int)tot, off);
    if (lane == 0 && tot) atomicAdd(block_kmers, tot);
    __syncthreads();
    if (threadIdx.x == 0 && *block_kmers) { unsigned long long* kt = cold()->kmer_total; if (kt) atomicAdd(kt, (unsigned long long)*block_kmers); }
    BK_DBG_CLOCK(a, 3);
}

// One workgroup per bin.  E bin b: window cells [128 b, 128 b + 383) -- its items start in its 128 cells and reach at most 255
// further; cells counted by two bins simply receive two additions.  V bin: counters [bin * size, + size) of the plane's V part.
__global__ __launch_bounds__(kBinBlock) void bin_count_kernel(BinArgs b) {
    extern __shared__ __attribute__((aligned(16))) unsigned int acc[];   // E: [2][kEBinSpan + 1] difference arrays (along / against); V: the bin's counters
    const uint32_t n_eb = b.ig.n_ebins, n_bins = n_eb + b.ig.n_vbins, bin = blockIdx.x + (b.part == 2 ? n_eb : 0u);   // (BinArgs::part)
    const bool is_e = bin < n_eb;
    const uint32_t vsize = (6u << b.ig.vq_log2) * b.rl;
    const uint32_t n_acc = is_e ? 2u * (kEBinSpan + 1u) : vsize;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (BK_ABLATE(b, 5)) return;
    const uint32_t cap = is_e ? b.ig.cap_e : b.ig.cap_v;
    const uint32_t G = b.ig.grid_max;
    // this bin's buckets of all scan workgroups: one stretch of device memory, workgroup after workgroup
    const unsigned short* const bin_items = b.items + (is_e ? (size_t)bin * G * b.ig.cap_e : (size_t)n_eb * G * b.ig.cap_e + (size_t)(bin - n_eb) * G * b.ig.cap_v);
    // A thread per scan workgroup: this bin's bucket in that workgroup's region.  The bucket's first 48 / 24 slots are asked for
    // together with the table entry that says how many of them hold items, before anything else is done -- one trip to memory,
    // not two (every bin's workgroup reads what 255 others wrote: nothing of it is in this XCD's L2).
    const bool mine = threadIdx.x < b.n_wg && !BK_ABLATE(b, 1);
    const unsigned short* reg0 = bin_items + (size_t)(mine ? threadIdx.x : 0u) * cap;
    uint32_t hdr0 = 0u;
    uint4 v0 = make_uint4(0u, 0u, 0u, 0u), v1 = v0, v2 = v0, v3 = v0, v4 = v0, v5 = v0;
    if (mine) {
        const uint4* q = reinterpret_cast<const uint4*>(reg0);
        hdr0 = b.tab[(size_t)bin * G + threadIdx.x];
        v0 = q[0]; v1 = q[1]; v2 = q[2];
        if (is_e) { v3 = q[3]; v4 = q[4]; v5 = q[5]; }
    }
    for (uint32_t i = threadIdx.x; i < n_acc; i += kBinBlock) acc[i] = 0u;
    if (BK_ABLATE(b, 6)) return;
    if (bin == 0 && threadIdx.x == 0) b.ov_n[b.ov_par ^ 1u] = 0ull;   // the next launch's overflow count starts at zero (part 2 follows a launch with part 1: done there)
    __syncthreads();
    auto take = [&](uint32_t it) __attribute__((always_inline)) {
        if (it == 0xffffu) return;
        if (is_e) {
            const uint32_t c = it & 127u, n = ((it >> 7) & 255u) + 1u;
            unsigned int* d = acc + (it >> 15) * (kEBinSpan + 1u);
            __hip_atomic_fetch_add(d + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(d + c + n, 0u - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add(acc + (it & 0x7fffu), (it & 0x8000u) ? 0u - 1u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // the first `left` of the eight items of one 16-byte unit
    auto take8 = [&](const uint4& v, uint32_t left) __attribute__((always_inline)) {
        if (left > 0u) take(v.x & 0xffffu);
        if (left > 1u) take(v.x >> 16);
        if (left > 2u) take(v.y & 0xffffu);
        if (left > 3u) take(v.y >> 16);
        if (left > 4u) take(v.z & 0xffffu);
        if (left > 5u) take(v.z >> 16);
        if (left > 6u) take(v.w & 0xffffu);
        if (left > 7u) take(v.w >> 16);
    };
    // n items (whole 16-byte units are readable): four units in flight at a time
    auto take_items = [&](const unsigned short* p, uint32_t n) __attribute__((always_inline)) {
        const uint4* q = reinterpret_cast<const uint4*>(p);
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t i0 = 0; i0 < n; i0 += 32u) {
            const uint32_t u0 = i0 >> 3;
            const uint4 v0 = q[u0];
            const uint4 v1 = i0 + 8u < n ? q[u0 + 1u] : z;
            const uint4 v2 = i0 + 16u < n ? q[u0 + 2u] : z;
            const uint4 v3 = i0 + 24u < n ? q[u0 + 3u] : z;
            take8(v0, n - i0);
            take8(v1, n > i0 + 8u ? n - i0 - 8u : 0u);
            take8(v2, n > i0 + 16u ? n - i0 - 16u : 0u);
            take8(v3, n > i0 + 24u ? n - i0 - 24u : 0u);
        }
    };
    const uint32_t spec = is_e ? 48u : 24u;   // slots asked for above
    uint32_t g_mine = 0u;   // items of this bin in my scan workgroup's extension
    if (mine) {
        const uint32_t n_all = hdr0, n = min(n_all, cap);
        g_mine = min(n_all - n, kItemGCap);
        take8(v0, n);
        take8(v1, n > 8u ? n - 8u : 0u);
        take8(v2, n > 16u ? n - 16u : 0u);
        if (is_e) {
            take8(v3, n > 24u ? n - 24u : 0u);
            take8(v4, n > 32u ? n - 32u : 0u);
            take8(v5, n > 40u ? n - 40u : 0u);
        }
        if (n > spec) take_items(reg0 + spec, n - spec);   // (larger buckets than this kernel was written for)
    }
    for (uint32_t wg = threadIdx.x + kBinBlock; wg < b.n_wg && !BK_ABLATE(b, 1); wg += kBinBlock) {   // (more scan workgroups than threads here: never on this chip)
        const uint32_t n_all = b.tab[(size_t)bin * G + wg], n = min(n_all, cap);
        take_items(bin_items + (size_t)wg * cap, n);
        take_items(b.gext + ((size_t)wg * n_bins + bin) * kItemGCap, min(n_all - n, kItemGCap));
    }
    // The extensions: a hot bin (a true variant site: thousands of reads on the same counters) has a hundred items in every scan
    // workgroup's -- every thread reads its own workgroup's, sixteen 16-byte units in flight at a time (the counts are alike from
    // workgroup to workgroup: the threads finish together).  (Sharing the concatenation of all extensions among the threads with a
    // bisection per item was twice as slow: eight dependent LDS reads per item.)
    if (g_mine) {
        const uint4* q = reinterpret_cast<const uint4*>(b.gext + ((size_t)threadIdx.x * n_bins + bin) * kItemGCap);
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t i0 = 0; i0 < g_mine; i0 += 128u) {
            uint4 u[16];
#pragma unroll
            for (uint32_t j = 0; j < 16u; ++j) u[j] = i0 + 8u * j < g_mine ? q[(i0 >> 3) + j] : z;
#pragma unroll
            for (uint32_t j = 0; j < 16u; ++j) take8(u[j], g_mine > i0 + 8u * j ? g_mine - i0 - 8u * j : 0u);
        }
    }
    if (!BK_ABLATE(b, 7)) {   // the overflow list: everything there that names this bin
        const unsigned long long n_all = b.ov_n[b.ov_par];
        const uint32_t n_ov = (uint32_t)(n_all < (unsigned long long)b.ov_cap ? n_all : (unsigned long long)b.ov_cap);
        for (uint32_t i = threadIdx.x; i < n_ov; i += kBinBlock) {
            const uint32_t e = b.ov[i];
            if ((e >> 16) == bin) take(e & 0xffffu);
        }
    }
    __syncthreads();
    if (is_e) {
        // difference arrays -> per-cell counts: wave 0 the reads along the reference, wave 1 those against it, six entries per lane
        if (wave < 2) {
            unsigned int* d = acc + (uint32_t)wave * (kEBinSpan + 1u) + (uint32_t)lane * 6u;
            const uint32_t d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4], d5 = d[5];
            const uint32_t sum = d0 + d1 + d2 + d3 + d4 + d5;
            uint32_t inc = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, off); if (lane >= off) inc += t; }
            uint32_t run = inc - sum;
            run += d0; d[0] = run; run += d1; d[1] = run; run += d2; d[2] = run; run += d3; d[3] = run; run += d4; d[4] = run; run += d5; d[5] = run;
        }
        __syncthreads();
        const uint32_t win_lo = b.win_dev ? b.win_dev[1] : b.win_lo;
        for (uint32_t i = threadIdx.x; i < kEBinSpan && !BK_ABLATE(b, 2) && !BK_ABLATE(b, 3); i += kBinBlock) {
            const uint32_t s0 = acc[i], s1 = acc[(kEBinSpan + 1u) + i];
            if (s0 | s1) {
                const uint32_t cell = win_lo + (bin << kEBinLog2) + i;
                const uint32_t id = b.id_at[cell];   // a counted cell always has a reference k-mer
                const uint32_t rc = ((b.cell_codes[cell >> 4] >> (2u * (cell & 15u))) & 3u) == 2u ? 1u : 0u;
                if (s0) atomicAdd(b.counters + 2 * (size_t)id + rc, (unsigned long long)s0);
                if (s1) atomicAdd(b.counters + 2 * (size_t)id + (1u - rc), (unsigned long long)s1);
            }
        }
    } else {
        const uint64_t first = (uint64_t)(bin - n_eb) * vsize;
        unsigned long long* const vc = b.counters + b.v_off + first;
        const uint32_t n_here = (uint32_t)min((uint64_t)vsize, b.v_real_len > first ? b.v_real_len - first : 0ull);
        // a difference: sign-extended, the plane wraps modulo 2^64.  Nothing else adds to this mate file's plane while this kernel runs
        // (the stream orders it against nbatch / level2 / finalize) and a counter belongs to one thread of one workgroup: plain
        // read-modify-write of whole lines instead of a million scattered atomics; v_mode 2: the V part is known to be all zero
        // (a sample's first launch into a clean plane) -- stores only
        if (BK_ABLATE(b, 2) || BK_ABLATE(b, 4)) {
        } else if (b.v_mode == 2 && b.ov_n[b.ov_par] <= (unsigned long long)b.ov_cap) {   // (past the list's end the scan added to the plane itself: it is not all zero then)
            for (uint32_t i = threadIdx.x; i < n_here; i += kBinBlock) vc[i] = (unsigned long long)(long long)(int32_t)acc[i];
        } else if (b.v_mode == 1 || b.v_mode == 2) {
            for (uint32_t i = threadIdx.x; i < n_here; i += kBinBlock) vc[i] += (unsigned long long)(long long)(int32_t)acc[i];
        } else {
            for (uint32_t i = threadIdx.x; i < n_here; i += kBinBlock) {
                const uint32_t v = acc[i];
                if (v) atomicAdd(vc + i, (unsigned long long)(long long)(int32_t)v);
            }
        }
    }
}

bool item_geometry(uint32_t win_cells, uint32_t n_full, int v_span, ItemGeom* g) {
    if (v_span <= 0 || win_cells == 0 || n_full == 0) return false;
    const uint32_t rl = (uint32_t)v_span + 1u;
    const uint64_t n_q = (uint64_t)n_full + (uint32_t)v_span;
    uint32_t vq_log2 = 0, nv = 0;
    for (uint32_t l = 6; l <= 8; ++l) {   // the smallest bins that keep their number at 512 or below
        if ((6u << l) * rl > 32767u) break;
        vq_log2 = l;
        nv = (uint32_t)((n_q + (1u << l) - 1u) >> l);
        if (nv <= 512u) break;
    }
    if (!vq_log2 || nv > 1024u) return false;
    const uint32_t ne = (win_cells + (1u << kEBinLog2) - 1u) >> kEBinLog2;
    if (ne + nv > 2u * (uint32_t)kItemBlock) return false;
    g->n_ebins = ne; g->n_vbins = nv; g->vq_log2 = vq_log2;
    // a workgroup sees 1 / n_cus of a launch: 3,900 of a million reads, 6,800 E items and 4,500 V items over these bins; a bucket
    // holds the mean of a uniform sample and three to four standard deviations (what goes beyond continues in the bin's extension in device memory)
    g->cap_e = 48u; g->cap_v = 24u;
    g->wg_items = ne * g->cap_e + nv * g->cap_v;
    g->wg_stride = g->wg_items;
    g->grid_max = 0;   // (set by the engine: items_max_grid of its device)
    return true;
}

uint32_t items_max_grid(int n_cus) { return (uint32_t)n_cus * kItemGroupsPerCu; }
uint32_t items_grid(uint64_t n_records, int n_cus) {
    const uint64_t tiles = (n_records + 63) / 64;
    const uint64_t want = (tiles + kItemWaves - 1) / kItemWaves;
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, items_max_grid(n_cus)));
}
size_t items_lds_bytes(const ItemGeom& g, uint32_t win_cells) {
    const size_t nb_pad = ((size_t)g.n_ebins + g.n_vbins + 3u) & ~(size_t)3u;
    return ((kItemLdsFixed + nb_pad * 4u + (size_t)g.wg_stride * 2u + 7u) & ~(size_t)7u) + scan_ref_lds_bytes(win_cells) + 8u +
           ((size_t)(kRefPadWords + (win_cells + 15) / 16 + kRefBackWords) + 1u) * sizeof(unsigned int);   // (+ the reverse-complemented reference)
}

hipError_t raise_lds_limit(const void* fn, size_t lds) {
    // the dynamic-LDS limit of a kernel is raised once (per process and device), not at every launch
    static std::mutex mu;
    static std::vector<std::pair<std::pair<const void*, int>, size_t>> have;   // ((kernel, device), limit set)
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    size_t* cur = nullptr;
    for (auto& h : have) if (h.first.first == fn && h.first.second == dev) cur = &h.second;
    if (!cur || *cur < lds) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (cur) *cur = lds; else have.push_back({{fn, dev}, lds});
    }
    return hipSuccess;
}

hipError_t launch_scan_items(const ScanArgs& a, uint32_t grid, hipStream_t stream) {
    if (a.n_records == 0 || a.W <= 0) return hipSuccess;
    size_t lds = items_lds_bytes(a.ig, std::min(a.n_lds_bins, a.total_cells));
    ScanArgs b = a;
    // the records in LDS when they are short enough and a tile starts at a 16-byte boundary (LDS-DMA moves 16 bytes per lane)
    const uintptr_t first = reinterpret_cast<uintptr_t>(a.words + a.rec_base * a.stride_words);
    // (+ 128 bytes behind the buffers: the flag pass reads a chunk's ten words whatever the record's length; 160 KB per workgroup)
    const size_t stage_off = (lds + 15u) & ~(size_t)15u, stage_end = stage_off + (size_t)kItemWaves * 64u * a.stride_words * sizeof(unsigned int) + 128u;
    const bool staged = a.stride_words <= kStageMaxWords && (first & 15u) == 0 && stage_end <= 160u * 1024u && !BK_ABLATE(a, 16);
    b.stage_off = 0;
    if (staged) {
        b.stage_off = (uint32_t)stage_off;
        lds = stage_end;
    }
    void (*kern)(ScanArgs);
    if (staged) kern = a.k == 21 ? scan_items_kernel<21, true> : a.k == 31 ? scan_items_kernel<31, true> : scan_items_kernel<0, true>;
    else kern = a.k == 21 ? scan_items_kernel<21, false> : a.k == 31 ? scan_items_kernel<31, false> : scan_items_kernel<0, false>;
    if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kern), lds)) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kItemBlock), lds, stream, b);
    return hipGetLastError();
}

hipError_t launch_bin_count(const BinArgs& b, hipStream_t stream) {
    const uint32_t n_bins = b.part == 1 ? b.ig.n_ebins : b.part == 2 ? b.ig.n_vbins : b.ig.n_ebins + b.ig.n_vbins;
    if (n_bins == 0 || b.n_wg == 0) return hipSuccess;
    const size_t lds = std::max<size_t>(2u * (kEBinSpan + 1u), (size_t)(6u << b.ig.vq_log2) * b.rl) * sizeof(unsigned int);
    if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(bin_count_kernel), lds)) return e;
    hipLaunchKernelGGL(bin_count_kernel, dim3(n_bins), dim3(kBinBlock), lds, stream, b);
    return hipGetLastError();
}

}  // namespace bk
