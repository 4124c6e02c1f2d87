// bk_scan_common.h -- device helpers shared by the scan kernels (bk_kernels.hip: scan_count / nbatch / level2; bk_scan_items.hip:
// the binned scan).  Wave64 throughout.
#pragma once
#include <hip/hip_runtime.h>

#include "bk_device.h"
#include "bk_kernels.h"

namespace bk {

// Measurement aids (ScanArgs::ablate) exist in the -DBK_TESTING build only; in the release library the tests fold away.
#ifdef BK_TESTING
#define BK_ABLATE(a, x) ((a).ablate == (x))
// BK_L2_STATS tallies (ScanArgs::dbg): k-mers marked by the scan [0] without a diagonal, [1] at a dirty / id-breaking head,
// [2] mismatch-free head, [3] close pairs; Level 2: [4] k-mers looked at, [5] single-k-mer S runs, [6] dropped as dead,
// [7] queued for the slow pipeline, [8] ... reference k-mers after all, [9] ... a neighbour found, [10] ... nothing; [11] chunks
#define BK_DBG(a, idx, pred, cnt) do { if ((a).dbg && (pred)) atomicAdd((a).dbg + (idx), (unsigned long long)(cnt)); } while (0)
// ... and behind the 32 tallies, per workgroup of scan_items_kernel: [32 + 4 b] clock at its start, [+ 1] after the reference is
// staged, [+ 2] after its last tile, [+ 3] at its end (wall_clock64: 100 MHz)
#define BK_DBG_CLOCK(a, slot) do { if ((a).dbg && threadIdx.x == 0) (a).dbg[32 + 4 * blockIdx.x + (slot)] = wall_clock64(); } while (0)
// ... four more per workgroup (grids of up to 512) for the prologue: [2080 + 4 b]: first tile's copy sent, [+ 1] the window's loads
// stored, [+ 2] wave 0 has its first tile, [+ 3] the buckets are written out
#define BK_DBG_CLOCK2(a, slot) do { if ((a).dbg && threadIdx.x == 0 && blockIdx.x < 512) (a).dbg[2080 + 4 * blockIdx.x + (slot)] = wall_clock64(); } while (0)
#else
#define BK_ABLATE(a, x) false
#define BK_DBG(a, idx, pred, cnt) do { } while (0)
#define BK_DBG_CLOCK(a, slot) do { } while (0)
#define BK_DBG_CLOCK2(a, slot) do { } while (0)
#endif

constexpr int kSeeds = 4;
constexpr int kRefPadWords = 4;             // words of padding in front of the 2-bit per-cell arrays (64 cells)
constexpr int kRefBackWords = 6;            // ... and behind them (96 cells)
constexpr int kBitPadWords = 2;             // the same 64 cells for the 1-bit per-cell arrays
constexpr int kBitBackWords = 3;

// number of set bits of a wave mask below this lane
__device__ __forceinline__ uint32_t lane_prefix(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// Wave64 scans over the lanes in registers (DPP: row_shr 1 / 2 / 4 / 8 inside each row of 16 lanes, then row_bcast:15 into rows 1
// and 3 and row_bcast:31 into rows 2 and 3): six VALU instructions and no trip through the LDS crossbar, where a ladder of
// __shfl_up steps is six dependent ds_bpermute round trips with a compare, a select and an address shift each.
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {   // inclusive prefix sum
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v) {   // inclusive running maximum (unsigned)
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}
// ... and the whole wave's maximum / minimum as a wave-uniform value (lane 63 of the running one)
__device__ __forceinline__ uint32_t wave_max(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_max(v), 63); }
__device__ __forceinline__ uint32_t wave_min(uint32_t v) { return ~wave_max(~v); }
// reverse the order of the sixteen 2-bit groups of a word
__device__ __forceinline__ uint32_t rev2_32(uint32_t x) {
    const uint32_t t = __builtin_bitreverse32(x);
    return ((t >> 1) & 0x55555555u) | ((t & 0x55555555u) << 1);
}
__device__ __forceinline__ uint64_t rev2_64(uint64_t x) {
    return ((uint64_t)rev2_32((uint32_t)x) << 32) | rev2_32((uint32_t)(x >> 32));
}
// the even bits of a word, packed
__device__ __forceinline__ uint32_t even_bits(uint32_t x) {
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0f0f0f0fu;
    x = (x | (x >> 4)) & 0x00ff00ffu;
    return (x | (x >> 8)) & 0xffffu;
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
// 32 bits of a 1-bit-per-cell array starting at cell `pos`
__device__ __forceinline__ uint32_t bits32_at(const uint32_t* __restrict__ w, int32_t pos) {
    return __builtin_amdgcn_alignbit(w[(pos >> 5) + 1], w[pos >> 5], (uint32_t)pos & 31u);
}

}  // namespace bk
