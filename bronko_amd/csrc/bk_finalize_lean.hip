// bk_finalize_lean.hip -- K2a / K2e for ONE genome file with dense planes (BASELINE configs 2 and 4), organised by REGION of the
// reference like the binned scan: a workgroup owns 64 row positions of the V plane and the 64 reference k-mers of the same
// ids (finalize_vbin_kernel), casts the votes of call.rs:1327-1384 into a dense table in LDS over the pileup positions its k-mers can
// reach -- plain 32-bit LDS atomics, no hashing, no barrier inside the loop -- and adds the table to the pileup ONCE.
//
// The general kernels (bk_kernels.hip) take rows 12 at a time through a 512-slot hash table with two barriers and a flush per
// unit: ~1.2 M scattered 64-bit global atomics and 15 k barrier pairs per sample bound them (0.049 + 0.023 ms per 1 M reads), not
// bytes and not instructions.  Here a pileup cell receives one pair of global atomics per workgroup that reaches it (~1.5 per
// cell) and a workgroup synchronises twice in all.
//
// What these kernels do not settle themselves goes where the general path takes it: a V k-mer whose reference neighbour is not
// "simple" (repeats) or whose answer was not worked out, or that touches several buckets, is appended to the deferred list of
// finalize_general_kernel (K2b), exactly as K2a defers its multi-bucket k-mers; a reference k-mer that is not simple walks its
// buckets' BucketInfos here, vote by vote.  launch_finalize (bk_kernels.hip) decides: one genome file, statistics and votes in
// one pass over whole dense planes, no k-mer statistics table, no pseudo k-mers, an answer table, W > 1, cs < 2^32.
#include <hip/hip_runtime.h>

#include "bk_device.h"
#include "bk_kernels.h"
#include "bk_scan_common.h"
#include "bk_finalize_common.h"

namespace bk {

constexpr int kLeanVBlock = 1024;
constexpr uint32_t kLeanVq = 64;       // row positions q of the V plane per workgroup (6 rows each)
constexpr uint32_t kLeanWin = 256;     // pileup positions of its vote table: 64 + v_span reference k-mers' cells and k - 1 behind, with slack

// one vote into the dense table (or, out of its reach, straight to the pileup)
__device__ __forceinline__ void lean_vote(unsigned int* cnt, unsigned int* mxv, uint32_t span, uint32_t p0, const FinalizeArgs& a, uint32_t cell,
                                          uint32_t idx, bool canonical, uint64_t c, uint32_t isrc, int k, uint32_t v) {
    uint32_t bit_idx;
    bool forward;
    if (canonical) { bit_idx = ((uint32_t)(c >> (2 * idx)) & 3u) ^ 3u; forward = isrc != 0; }        // (vote(): call.rs:1327-1384)
    else { bit_idx = (uint32_t)(c >> (2 * (k - 1 - (int)idx))) & 3u; forward = isrc == 0; }
    const uint32_t pos = cell - p0;
    if (cell >= p0 && pos < span) {
        const uint32_t at = ((forward ? 0u : 4u) + bit_idx) * span + pos;
        __hip_atomic_fetch_add(cnt + at, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(mxv + at, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        const size_t pc = (size_t)cell * 4 + bit_idx;
        atomicAdd(a.pileup + (forward ? 2 : 3) * a.plane + pc, 1ull);
        atomicMax(a.pileup + (forward ? 0 : 1) * a.plane + pc, (unsigned long long)v);
    }
}
// the table -> pileup: one pair of global atomics per cell that received a vote
__device__ __forceinline__ void lean_flush(const unsigned int* cnt, const unsigned int* mxv, uint32_t span, uint32_t p0, const FinalizeArgs& a) {
    for (uint32_t i = threadIdx.x; i < 8u * span; i += blockDim.x) {
        const unsigned int nv = cnt[i];
        if (!nv) continue;
        const uint32_t row = i / span, pos = i - row * span;
        const size_t pc = ((size_t)p0 + pos) * 4 + (row & 3u);
        atomicAdd(a.pileup + (row < 4u ? 2 : 3) * a.plane + pc, (unsigned long long)nv);   // #kmers
        atomicMax(a.pileup + (row < 4u ? 0 : 1) * a.plane + pc, (unsigned long long)mxv[i]);   // depth
    }
}

// K2a, one workgroup per 64 row positions.  Lanes as in finalize_variant_kernel: v_span lanes per row (lane = offset), the row's
// prefix sums by shuffles, every lane maps the k-mer its count belongs to.
// FUSED (FinalizeArgs::f_items): the counts are not in the plane -- the mate file's reads were one scan launch, whose V items are
// still where the scan left them.  The workgroup's 64 row positions are one V bin of the binned scan (ItemGeom::vq_log2 = 6): it
// adds the bin's items of every scan workgroup up in LDS, as bin_count_kernel does, and never stores them; a row that Level 2
// wrote to meanwhile (f_touch: a bit per row) is read from the plane as well, zeroed, its bit cleared.  28 MB of stores, 27 MB of
// loads and 25 MB of zeroing per 1 M-read sample of SARS-CoV-2 become 6 MB of item loads.
template <bool FUSED>
__global__ __launch_bounds__(kLeanVBlock) void finalize_vbin_kernel(FinalizeArgs a) {
    __shared__ unsigned int cnt[8 * kLeanWin], mxv[8 * kLeanWin];
    __shared__ uint32_t lstats[3 + 2];
    __shared__ unsigned int tmask[kLeanVq * kVRowsPerPos / 32 + 1];   // FUSED: the touch bits of the workgroup's 384 rows; [12]: all rows (the scan's item_direct wrote to the plane)
    extern __shared__ __attribute__((aligned(16))) unsigned int acc[];   // FUSED: [384 * (v_span + 1)] the bin's counters (differences), from the items
    const IndexView& ix = a.ix;
    const int k = ix.k;
    const uint32_t span = (uint32_t)ix.v_span, rl = span + 1u;
    const uint32_t q0 = blockIdx.x * kLeanVq;
    const uint32_t nq = ix.n_full + span;
    // the lowest id a row of this workgroup can belong to, and the cell its k-mer starts at: nothing here votes below it
    const uint32_t id_lo = min(q0 >= span - 1u ? q0 - (span - 1u) : 0u, ix.n_full - 1u);
    const uint32_t p0 = ix.id_rec[id_lo].cell;
    for (uint32_t i = threadIdx.x; i < 8u * kLeanWin; i += kLeanVBlock) { cnt[i] = 0u; mxv[i] = 0u; }
    if (threadIdx.x < 5) lstats[threadIdx.x] = 0u;
    if constexpr (FUSED) {
        const uint32_t n_eb = a.f_ig.n_ebins, n_bins = n_eb + a.f_ig.n_vbins, bin = n_eb + blockIdx.x, cap = a.f_ig.cap_v, G = a.f_ig.grid_max;
        const uint32_t upb = cap / 8u;                                     // 16-byte units per bucket
        const uint32_t vsize = kLeanVq * kVRowsPerPos * rl;
        // bucket unit u of scan workgroup wg: asked for together with the table entry that says how many of the bucket's slots count,
        // before anything else is done (one trip to memory, as in bin_count_kernel); 1024 threads, 256 workgroups x 3 units
        const uint32_t n_units = a.f_n_wg * upb;
        const unsigned short* const bin_items = a.f_items + (size_t)n_eb * G * a.f_ig.cap_e + (size_t)blockIdx.x * G * cap;
        const bool mine = threadIdx.x < n_units;
        const uint32_t wg0 = mine ? threadIdx.x / upb : 0u, un0 = threadIdx.x - wg0 * upb;
        uint32_t hdr0 = 0u;
        uint4 v0 = make_uint4(0u, 0u, 0u, 0u);
        if (mine) { hdr0 = a.f_tab[(size_t)bin * G + wg0]; v0 = reinterpret_cast<const uint4*>(bin_items + (size_t)wg0 * cap)[un0]; }
        const unsigned long long ov_all = a.f_ov_n[a.f_ov_par];
        if (threadIdx.x <= kLeanVq * kVRowsPerPos / 32) {
            unsigned int* tw = a.f_touch + (size_t)blockIdx.x * (kLeanVq * kVRowsPerPos / 32) + threadIdx.x;
            unsigned int m = 0u;
            if (threadIdx.x < kLeanVq * kVRowsPerPos / 32) { m = *tw; if (m) *tw = 0u; }
            else m = ov_all > (unsigned long long)a.f_ov_cap ? 1u : 0u;   // past the list's end the scan added to the plane itself: every row is read
            tmask[threadIdx.x] = m;
        }
        for (uint32_t i = threadIdx.x; i < vsize; i += kLeanVBlock) acc[i] = 0u;
        __syncthreads();
        auto take = [&](uint32_t it) __attribute__((always_inline)) {
            __hip_atomic_fetch_add(acc + (it & 0x7fffu), (it & 0x8000u) ? 0u - 1u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        auto take8 = [&](const uint4& v, uint32_t left) __attribute__((always_inline)) {   // the first `left` of the eight items of one unit
            if (left > 0u) take(v.x & 0xffffu);
            if (left > 1u) take(v.x >> 16);
            if (left > 2u) take(v.y & 0xffffu);
            if (left > 3u) take(v.y >> 16);
            if (left > 4u) take(v.z & 0xffffu);
            if (left > 5u) take(v.z >> 16);
            if (left > 6u) take(v.w & 0xffffu);
            if (left > 7u) take(v.w >> 16);
        };
        for (uint32_t u = threadIdx.x; u < n_units; u += kLeanVBlock) {
            const uint32_t wg = u / upb, un = u - wg * upb;
            uint32_t n_all; uint4 v;
            if (u == threadIdx.x) { n_all = hdr0; v = v0; }
            else { n_all = a.f_tab[(size_t)bin * G + wg]; v = reinterpret_cast<const uint4*>(bin_items + (size_t)wg * cap)[un]; }
            const uint32_t n = min(n_all, cap);
            take8(v, n > 8u * un ? n - 8u * un : 0u);
            // the bucket's extension in device memory (a hot bin: a true variant site), its units dealt to the bucket's threads
            const uint32_t g = min(n_all - n, kItemGCap);
            if (g) {
                const uint4* q = reinterpret_cast<const uint4*>(a.f_gext + ((size_t)wg * n_bins + bin) * kItemGCap);
                for (uint32_t i0 = 8u * un; i0 < g; i0 += 8u * upb * 4u) {   // four units in flight
                    uint4 e[4];
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) e[j] = i0 + 8u * upb * j < g ? q[(i0 >> 3) + upb * j] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) take8(e[j], g > i0 + 8u * upb * j ? g - i0 - 8u * upb * j : 0u);
                }
            }
        }
        {   // the overflow list: everything there that names this bin
            const uint32_t n_ov = (uint32_t)(ov_all < (unsigned long long)a.f_ov_cap ? ov_all : (unsigned long long)a.f_ov_cap);
            for (uint32_t i = threadIdx.x; i < n_ov; i += kLeanVBlock) {
                const uint32_t e = a.f_ov[i];
                if ((e >> 16) == bin) take(e & 0xffffu);
            }
        }
    }
    __syncthreads();
    unsigned long long* __restrict__ vc = const_cast<unsigned long long*>(a.counters) + ix.v_off;   // (written only under clear_v)
    const uint32_t lpr = span, gpw = 64u / lpr;                   // lanes per row, rows per wave
    const uint32_t lane64 = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t grp = lane64 / lpr, oo = lane64 - grp * lpr;
    const bool lane_on = grp < gpw;
    unsigned int kept = 0, distinct = 0, variant = 0;
    for (uint32_t g0 = 0; g0 < kLeanVq * kVRowsPerPos; g0 += (kLeanVBlock / 64) * gpw) {   // (wave-uniform)
        const uint32_t g = g0 + wave * gpw + grp;                 // this lane's row among the workgroup's 384
        const uint32_t q = q0 + g / kVRowsPerPos, r6 = g % kVRowsPerPos;
        const bool in_row = lane_on && g < kLeanVq * kVRowsPerPos && q < nq;
        const size_t at = ((size_t)q * kVRowsPerPos + r6) * rl + oo;
        unsigned long long n;
        if constexpr (FUSED) {
            // the items' sum (a difference: sign-extended, the plane's arithmetic wraps modulo 2^64) + what Level 2 left in the plane
            const uint32_t gg = in_row ? g : 0u;
            n = in_row ? (unsigned long long)(long long)(int32_t)acc[gg * rl + oo] : 0ull;
            if (in_row && (tmask[kLeanVq * kVRowsPerPos / 32] || ((tmask[gg >> 5] >> (gg & 31u)) & 1u))) {
                const unsigned long long pv = vc[at];
                if (pv) { n += pv; vc[at] = 0ull; }
            }
        } else {
            n = in_row ? vc[at] : 0ull;
            if (a.clear_v && n) vc[at] = 0ull;                    // (every counter is read by exactly one lane)
        }
        const uint32_t d = r6 & 1u, alt = r6 >> 1;
        const bool inq = in_row && q >= oo && q - oo < ix.n_full;
        const uint32_t p = inq ? q - oo : 0u;
        const uint4 idr = *reinterpret_cast<const uint4*>(ix.id_rec + p);
        const uint64_t kmer_p = (uint64_t)idr.x | ((uint64_t)idr.y << 32);
        const uint32_t ambp = idr.w;
        const int o = (int)oo + ix.v_omin;
        // the row's prefix sums.  The raw counters are differences of counts, small as signed numbers: when every one of the wave's
        // fits 26 bits (rows of at most 32: no sum leaves 32 bits) the shuffles carry one word instead of two
        if (!__ballot(n + (1ull << 26) >= (1ull << 27))) {
            int m = (int)(unsigned int)n;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const int t = __shfl_up(m, off, 64);   // (lane - off is in the same row whenever oo >= off)
                if (oo >= (uint32_t)off) m += t;
            }
            n = (unsigned long long)(long long)m;
        } else {
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const unsigned long long t = __shfl_up(n, off, 64);
                if (oo >= (uint32_t)off) n += t;
            }
        }
        bool act = inq && n != 0;
        const uint32_t rcid = (ambp >> 1) & 1u;
        const int j = rcid ? k - 1 - o : o;
        const int sh = 2 * (k - 1 - (act ? j : 0));
        // the base that stands there instead of the reference k-mer's own: own XOR (alternative + 1), on either strand
        const uint32_t bb = ((uint32_t)(kmer_p >> sh) & 3u) ^ (alt + 1u);
        const uint32_t isrc = d ^ rcid;
        const uint64_t c = (kmer_p & ~(3ull << sh)) | ((uint64_t)bb << sh);
        const uint64_t rc = revcomp_kmer(c, k);
        if (act && !(j >= ix.wstart && j < ix.wstart + ix.W && c < rc)) act = false;   // touches no window bucket
        if (act) { distinct += 1; if (n < a.ci || n > a.cx) act = false; else kept += 1; }   // kmc -ci / -cx act on the true count
        const uint32_t v = (uint32_t)(n > a.cs ? a.cs : n);      // kmc -cs: reported count saturates (cs < 2^32 here)
        // its one bucket is the reference k-mer's own at j unless another reference k-mer is near (dirty: the answer table knows
        // whether it then touches a second bucket); everything that is not "one BucketInfo, known from the record" goes to K2b
        bool defer = false;
        if (act) {
            bool plain = (ambp & kIdSimple) != 0u;
            if (plain && (ambp & 1u)) {
                const uint2 ans = *reinterpret_cast<const uint2*>(ix.dirty_ans + ans_index(p, (uint32_t)o, rcid ? 3u - bb : bb, k));
                plain = !(ans.y & (kAnsNone | kAnsMulti));
            }
            if (!plain) { defer = true; act = false; }
        }
        {   // the wave's deferred k-mers are appended together: one returning atomic on the list's counter per wave
            const unsigned long long dm = __ballot(defer);
            if (dm) {
                unsigned int base = 0;
                if (lane64 == (uint32_t)__builtin_ctzll(dm)) base = atomicAdd(a.n_deferred, (unsigned int)__popcll(dm));
                base = (unsigned int)__shfl((int)base, __builtin_ctzll(dm));
                if (defer) {
                    const unsigned int slot = base + (unsigned int)__popcll(dm & ((1ull << lane64) - 1ull));
                    a.deferred[slot] = (uint32_t)at;
                    if (a.deferred_n) a.deferred_n[slot] = n;
                }
            }
        }
        // the vote: BucketInfo {cell = first cell + j, idx = j, canonical = rcid} (IdRec "simple"); one hit in the genome: "variant" (W > 1)
        if (act) lean_vote(cnt, mxv, kLeanWin, p0, a, idr.z + (uint32_t)j, (uint32_t)j, rcid != 0u, c, isrc, k, v);
        variant += act ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) variant += (unsigned int)__shfl_xor((int)variant, off);
    if (lane64 == 0 && variant) atomicAdd(&lstats[1], variant);
    // ---- K2e for the same region: the reference k-mers [q0, q0 + 64) themselves, their two E counters, their 2 W votes -- into the
    // same table (ids rise with their first cells: these k-mers start where the rows above voted).  16 threads per k-mer share its
    // votes; the first of them keeps its statistics.  (A kernel of its own until round 4's last build: one launch and 0.011 ms.)
    {
        const uint32_t W = (uint32_t)ix.W;
        const uint32_t sub = threadIdx.x & 15u, id = q0 + (threadIdx.x >> 4);
        unsigned int perfect = 0;
        if (id < ix.n_full) {
            const uint4 idr = *reinterpret_cast<const uint4*>(ix.id_rec + id);
            const unsigned long long n0 = a.counters[2 * (size_t)id], n1 = a.counters[2 * (size_t)id + 1];
            if (!(idr.w & kIdSimple)) {
                // Repeats: its buckets hold several BucketInfos each -- a walk of dependent loads that one thread would take a
                // hundred microseconds over while the others wait.  Listed instead; finalize_exact_kernel maps the list (votes and
                // statistics), a thread per (counter, bucket), as it maps the touched k-mers of a large index.
                if (sub == 0u && (n0 | n1)) a.lean_e_list[atomicAdd(a.lean_n_list + 2, 1u)] = id;
            } else {
                const uint64_t km = (uint64_t)idr.x | ((uint64_t)idr.y << 32);
                const uint32_t rcid = (idr.w >> 1) & 1u;
#pragma unroll
                for (uint32_t isrc = 0; isrc < 2u; ++isrc) {
                    const unsigned long long n = isrc ? n1 : n0;
                    if (sub == 0u) distinct += n != 0;
                    if (n == 0 || n < a.ci || n > a.cx) continue;                // kmc -ci / -cx act on the true count
                    if (sub == 0u) { kept += 1; perfect += 1; }                  // (simple: every bucket holds its one occurrence -- perfect in, and unique to, the one genome)
                    const uint32_t v = (uint32_t)(n > a.cs ? a.cs : n);          // kmc -cs: reported count saturates
                    // each of its W buckets holds its own single occurrence: {cell + j, idx = j, canonical = rcid}
                    for (uint32_t t = sub; t < W; t += 16u) {
                        const uint32_t j = (uint32_t)ix.wstart + t;
                        lean_vote(cnt, mxv, kLeanWin, p0, a, idr.z + j, j, rcid != 0u, km, isrc, k, v);
                    }
                }
            }
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) perfect += (unsigned int)__shfl_xor((int)perfect, off);
        if (lane64 == 0 && perfect) { atomicAdd(&lstats[0], perfect); atomicAdd(&lstats[2], perfect); }
    }
    __syncthreads();
    lean_flush(cnt, mxv, kLeanWin, p0, a);
    finalize_epilogue(a, lstats, kept, distinct, lstats + 3, (int)blockIdx.x);
}

bool finalize_lean_ok(const FinalizeArgs& a) {
    return !a.no_lean && a.lean_e_list && a.lean_n_list && !a.ix.slot_files && a.mode == 0 && !a.v_list && !a.p_list && !a.e_list && !a.ktab_keys && a.ix.n_prows == 0 && a.ix.n_u == a.ix.n_full &&
           a.ix.n_full > 0 && a.ix.n_files == 1 && a.ix.dirty_ans && a.ix.W > 1 && a.ix.v_span > 0 && a.ix.v_span <= 32 && a.partials && a.cs < (1ull << 32) &&
           a.elem_lo == 0 && a.elem_hi >= a.ix.v_off + v_plane_len(a.ix.n_full, a.ix.v_span, 0);
}
unsigned launch_finalize_lean_variant(const FinalizeArgs& a, hipStream_t stream) {
    const unsigned grid = (unsigned)((a.ix.n_full + (uint32_t)a.ix.v_span + kLeanVq - 1) / kLeanVq);
    if (a.f_items) {   // (the engine hands the items over only where a V bin is a workgroup's region: vq_log2 = 6, as many bins as workgroups)
        const size_t lds = (size_t)kLeanVq * kVRowsPerPos * ((size_t)a.ix.v_span + 1u) * sizeof(unsigned int);
        (void)raise_lds_limit(reinterpret_cast<const void*>(finalize_vbin_kernel<true>), lds + 20u * 1024u);
        hipLaunchKernelGGL(finalize_vbin_kernel<true>, dim3(grid), dim3(kLeanVBlock), lds, stream, a);
    } else {
        hipLaunchKernelGGL(finalize_vbin_kernel<false>, dim3(grid), dim3(kLeanVBlock), 0, stream, a);
    }
    return grid;
}
}  // namespace bk
