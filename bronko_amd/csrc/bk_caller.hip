// bk_caller.hip -- the stages after the pileup, on the device, for the sample an engine has just finalized:
//   select_genome   pick_best_genome / _paired       /root/reference/src/call.rs:422-502
//   noise           get_baseline_noise               call.rs:799-967
//   call            call_variants                    call.rs:969-1150
// so that a host with many samples in flight never waits between a sample's reads and its variant records.
//
// Exactness.  Everything up to the decisions is IEEE double arithmetic in the reference's order: +, -, *, / and sqrt are
// correctly rounded on the device as on the host, floating-point contraction is off in this file (a fused multiply-add would
// round differently from the reference's separate operations), and the Thompson-tau table is the host's.  The running sums of
// the noise window are updated position by position like upstream (a parallel prefix would round differently), so the window
// is walked by ONE thread per sequence -- the parallelism is across samples (engines / streams) and sequences, and in the
// per-position work before (sorted allele frequencies) and after (calls).  Only ln() and pow() of the call filters come from
// the device's math library; they decide nothing within 1e-15 of a threshold, and the host re-derives SOR for printing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "bk_device.h"
#include "bk_kernels.h"

#pragma clang fp contract(off)

namespace bk {

// Student-t quantile StudentsT(0,1,n-2).inverse_cdf(1 - 0.001/n) for n = 3..300 (call.rs:922-925): the host's table
__device__ const double kTCritDev[298] = {
#include "../host/tcrit_table.inc"
};

__global__ void select_genome_kernel(CallArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int best = -1;
    double best_score = 0.0;
    for (int f = 0; f < a.n_files; ++f) {
        bool present = false;
        unsigned long long perfect = 0;
        for (int m = 0; m < a.n_mates; ++m) {                       // pick_best_genome_paired sums R1 + R2 (call.rs:457-474)
            present = present || a.present[(size_t)m * a.n_files + f] != 0;
            perfect += a.stats[((size_t)m * a.n_files + f) * 3];
        }
        if (!present) continue;
        const double score = __ddiv_rn(__ddiv_rn((double)perfect, (double)a.genome_len[f]), 2.0);   // call.rs:435
        if (score > best_score) { best_score = score; best = f; }                                    // strict >, call.rs:443; ties -> lowest id
    }
    a.out->file_id = best;
    a.out->n_records = 0; a.out->n_major = 0; a.out->n_minor = 0; a.out->covered = 0; a.out->positions = 0; a.out->coverage = 0;
}

// get_baseline_noise for sequence blockIdx.x of the selected genome (call.rs:799-967).  The reference walks the positions with
// a window of the last 100: running sums of the minor-allele frequencies in it (position by position -- any other order of the
// same additions rounds differently) and a table of the ten largest values (a state machine of its own: an evicted value is
// removed, the eleventh largest does not move up).  Both are chains over ~30 k positions, so what counts is the latency of a
// step; a first version that kept the table in LDS and walked with one thread took 66 ms for SARS-CoV-2.  Here:
//   * all threads: the sorted minor-allele frequencies of the next kNoiseChunk positions (call.rs:831-845) into LDS, with the
//     100 positions before them -- the value that leaves the window is read from there, no ring buffer;
//   * wave 0, 64 positions at a time: every lane carries the same running sums, lane q < 10 holds table entry q; removal and
//     insertion are a ballot, a count of trailing zeros and a one-lane DPP shift; the state after each position (sums, table)
//     is left in LDS;
//   * wave 0, one lane per position of those 64: the Thompson-tau stripping (call.rs:917-950: divisions, square roots) from
//     that state -- independent of the other positions.
constexpr int kNoiseWindow = 100, kNoiseTop = kNoiseWindow / 10, kNoiseHalf = kNoiseWindow / 2;   // call.rs:802-804,824
constexpr int kNoiseChunk = 1024;
// the value of lane - 1 / lane + 1 within the row of 16 lanes (0 at the row's ends): the table is 10 lanes of one row
__device__ __forceinline__ double row_from_below(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)u, 0x111, 0xf, 0xf, true);           // row_shr:1
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(u >> 32), 0x111, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double row_from_above(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)u, 0x101, 0xf, 0xf, true);           // row_shl:1
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(u >> 32), 0x101, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__global__ __launch_bounds__(256) void noise_kernel(CallArgs a) {
    __shared__ double tau_s[kNoiseWindow * 3 + 1];
    __shared__ double fr_s[(kNoiseChunk + kNoiseWindow) * 3];   // positions base - 100 .. base + chunk
    __shared__ double st_s[64], st_s2[64], st_top[64 * kNoiseTop];   // the state after each of 64 positions
    __shared__ unsigned int st_n[64];
    const int file = a.out->file_id;
    if (file < 0 || (int)blockIdx.x >= a.n_seqs[file]) return;
    const int sq = a.seq_first[file] + (int)blockIdx.x;
    const uint64_t cell0 = a.seq_cell[sq], len = a.seq_len[sq];
    const unsigned long long* fd = a.pileup + 0 * a.plane + cell0 * 4;
    const unsigned long long* rd = a.pileup + 1 * a.plane + cell0 * 4;
    for (int n = threadIdx.x; n <= kNoiseWindow * 3; n += blockDim.x) {   // thompson_tau(n), call.rs:922-929
        double tau = INFINITY;
        if (n > 2) {
            const double t = kTCritDev[n - 3], dn = (double)n;
            tau = __ddiv_rn(t * (dn - 1.0), __dsqrt_rn(dn) * __dsqrt_rn(dn - 2.0 + t * t));
        }
        tau_s[n] = tau;
    }

    double* out = a.noise + cell0;
    const int lane = threadIdx.x & 63;
    const bool in_table = threadIdx.x < (unsigned)kNoiseTop;
    unsigned int n = 0;          // wave 0: the same in every lane
    double s = 0.0, s2 = 0.0;
    double top = 0.0;            // wave 0, lane q < 10: entry q of the table (descending, zeros behind the values); 0 elsewhere
    for (uint64_t base = 0; base < len + kNoiseHalf; base += kNoiseChunk) {
        __syncthreads();   // (the walk over the previous chunk is done; the tau table is written)
        for (int64_t j = threadIdx.x; j < kNoiseChunk + kNoiseWindow; j += blockDim.x) {
            const int64_t i = (int64_t)base - kNoiseWindow + j;
            double f1 = 0.0, f2 = 0.0, f3 = 0.0;
            if (i >= 0 && (uint64_t)i < len) {
                unsigned long long c0 = fd[i * 4 + 0] + rd[i * 4 + 0], c1 = fd[i * 4 + 1] + rd[i * 4 + 1];
                unsigned long long c2 = fd[i * 4 + 2] + rd[i * 4 + 2], c3 = fd[i * 4 + 3] + rd[i * 4 + 3];
                unsigned long long t;   // descending (a sorting network; equal values are interchangeable)
                if (c0 < c1) { t = c0; c0 = c1; c1 = t; }
                if (c2 < c3) { t = c2; c2 = c3; c3 = t; }
                if (c0 < c2) { t = c0; c0 = c2; c2 = t; }
                if (c1 < c3) { t = c1; c1 = c3; c3 = t; }
                if (c1 < c2) { t = c1; c1 = c2; c2 = t; }
                const unsigned long long depth = c0 + c1 + c2 + c3;
                if (depth) { f1 = __ddiv_rn((double)c1, (double)depth); f2 = __ddiv_rn((double)c2, (double)depth); f3 = __ddiv_rn((double)c3, (double)depth); }
            }
            fr_s[j * 3 + 0] = f1; fr_s[j * 3 + 1] = f2; fr_s[j * 3 + 2] = f3;
        }
        __syncthreads();
        if (threadIdx.x >= 64) continue;
        const uint64_t i_end = min(base + (uint64_t)kNoiseChunk, len + (uint64_t)kNoiseHalf);
        for (uint64_t b0 = base; b0 < i_end; b0 += 64) {
            const int nb = (int)min((uint64_t)64, i_end - b0);
            // (the six values of a position are asked for one position ahead: they depend on nothing the walk computes)
            const double* cur0 = fr_s + (size_t)(b0 - base + kNoiseWindow) * 3;   // position b0; cur - 300: b0 - 100
            double nold[3] = {cur0[0 - kNoiseWindow * 3], cur0[1 - kNoiseWindow * 3], cur0[2 - kNoiseWindow * 3]};
            double nmaf[3] = {cur0[0], cur0[1], cur0[2]};
            for (int t = 0; t < nb; ++t) {
                const double olds[3] = {nold[0], nold[1], nold[2]}, mafs[3] = {nmaf[0], nmaf[1], nmaf[2]};
                if (t + 1 < nb) {
                    const double* nx = cur0 + (size_t)(t + 1) * 3;
#pragma unroll
                    for (int r = 0; r < 3; ++r) { nold[r] = nx[r - kNoiseWindow * 3]; nmaf[r] = nx[r]; }
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {                           // minor ranks 1..3, call.rs:848
                    const double old = olds[r];                         // what position i - 100 put there (0: nothing)
                    if (old > 0.0) {                                    // evict, call.rs:853-870 (a value above 0 is always flagged)
                        n -= 1; s -= old; s2 -= old * old;
                        // the first entry equal to it goes, the entries behind it move up
                        const unsigned long long hit = __ballot(in_table && fabs(top - old) < 1e-12);
                        if (hit) {
                            const double above = row_from_above(top);
                            if (in_table && lane >= __builtin_ctzll(hit)) top = lane + 1 < kNoiseTop ? above : 0.0;
                        }
                    }
                    const double maf = mafs[r];                         // (zeros past the sequence's end)
                    if (maf > 0.0) {                                    // insert, call.rs:873-890
                        n += 1; s += maf; s2 += maf * maf;
                        // from the bottom up while it is larger: the table is sorted, so those entries are its tail
                        const unsigned long long larger = __ballot(in_table && maf > top);
                        // lowest entry of the tail: one above the highest entry that is not smaller
                        const unsigned long long not_larger = ~larger & ((1ull << kNoiseTop) - 1ull);
                        const int p = not_larger ? 64 - (int)__builtin_clzll(not_larger) : 0;
                        if (p < kNoiseTop) {
                            const double below = row_from_below(top);
                            if (in_table && lane > p) top = below;
                            if (lane == p) top = maf;
                        }
                    }
                }
                if (lane == 0) { st_n[t] = n; st_s[t] = s; st_s2[t] = s2; }
                if (in_table) st_top[t * kNoiseTop + lane] = top;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the state is read by other lanes of this wave)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // strip outliers (call.rs:917-950), one lane per position, from the state the walk left after it
            if (lane < nb) {
                const uint64_t i = b0 + (uint64_t)lane;
                const unsigned int n0 = st_n[lane];
                const double s0 = st_s[lane], s20 = st_s2[lane];
                const double* tp = st_top + lane * kNoiseTop;
                double mu = 0.0, var = 0.0;
                if (n0 != 0) { mu = __ddiv_rn(s0, (double)n0); var = __ddiv_rn(s20, (double)n0) - mu * mu; }   // population variance, call.rs:901-907
                int idx = 0;
                unsigned int cn = n0;
                double cs = s0, cs2 = s20;
                while (idx < kNoiseTop && tp[idx] != 0.0) {
                    const double cand = tp[idx];
                    const double tau = cn <= (unsigned int)(kNoiseWindow * 3) ? tau_s[cn] : NAN;
                    if (!(fabs(cand - mu) > tau * __dsqrt_rn(var))) break;
                    cs -= cand;
                    cs2 -= cand;                                        // sic (call.rs:936): the value, not its square
                    cn -= 1;
                    if (cn > 0) { mu = __ddiv_rn(cs, (double)cn); var = __ddiv_rn(cs2, (double)cn) - mu * mu; }
                    else { mu = 0.0; var = 0.0; }
                    idx++;
                }
                if (i >= (uint64_t)kNoiseHalf && i - kNoiseHalf < len)  // call.rs:953-962
                    out[i - kNoiseHalf] = idx < kNoiseTop ? tp[idx] : 0.0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// ---- get_baseline_noise, the walk taken apart (round 5) -------------------------------------------------------------------------
// noise_kernel above is the reference's walk in one wave: sums, table and strip position after position, 14 ms for SARS-CoV-2 --
// a chain of ~30 k steps of ~1000 cycles.  What is serial about it are two INDEPENDENT chains, and each is short once it is alone:
//   * the running sums s, s2 (and the count n): six floating-point additions / subtractions per step in the reference's order (any
//     other order rounds differently) -- nothing else: the squares are taken beforehand, the values come from LDS;
//   * the table of the ten largest values: a state machine over the same stream (an evicted value is removed, nothing moves up
//     from below) that does no arithmetic -- and that nearly every value passes by: a value that is not larger than the full
//     table's smallest entry is not inserted, one that is smaller than it (by 1e-12) is not in it.  Those two tests are on a
//     wave-uniform copy of the smallest entry; only the few values that do change the table take the ballot / shift path.
// noise_walk_kernel runs the two chains in two waves of one workgroup per sequence, side by side, each leaving its state after
// every step; noise_strip_kernel then strips the outliers (call.rs:917-950: divisions, square roots) one thread per position.
// Same operations on the same operands in the same order as the reference wherever order matters: Noise.max bit for bit.
__global__ __launch_bounds__(256) void noise_maf_kernel(CallArgs a) {
    const int file = a.out->file_id;
    if (file < 0) return;
    const uint64_t cell_lo = a.seq_cell[a.seq_first[file]];
    const int sq_hi = a.seq_first[file] + a.n_seqs[file];
    const uint64_t cell_hi = a.n_seqs[file] ? a.seq_cell[sq_hi - 1] + a.seq_len[sq_hi - 1] : cell_lo;
    const unsigned long long* fd = a.pileup + 0 * a.plane;
    const unsigned long long* rd = a.pileup + 1 * a.plane;
    for (uint64_t cell = cell_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < cell_hi; cell += (uint64_t)gridDim.x * blockDim.x) {
        unsigned long long c0 = fd[cell * 4 + 0] + rd[cell * 4 + 0], c1 = fd[cell * 4 + 1] + rd[cell * 4 + 1];
        unsigned long long c2 = fd[cell * 4 + 2] + rd[cell * 4 + 2], c3 = fd[cell * 4 + 3] + rd[cell * 4 + 3];
        unsigned long long t;   // descending (a sorting network; equal values are interchangeable), call.rs:831-845
        if (c0 < c1) { t = c0; c0 = c1; c1 = t; }
        if (c2 < c3) { t = c2; c2 = c3; c3 = t; }
        if (c0 < c2) { t = c0; c0 = c2; c2 = t; }
        if (c1 < c3) { t = c1; c1 = c3; c3 = t; }
        if (c1 < c2) { t = c1; c1 = c2; c2 = t; }
        const unsigned long long depth = c0 + c1 + c2 + c3;
        double f1 = 0.0, f2 = 0.0, f3 = 0.0;
        if (depth) { f1 = __ddiv_rn((double)c1, (double)depth); f2 = __ddiv_rn((double)c2, (double)depth); f3 = __ddiv_rn((double)c3, (double)depth); }
        double* o = a.noise_maf + (cell - cell_lo) * 3;
        o[0] = f1; o[1] = f2; o[2] = f3;
    }
}

// one workgroup of two waves per sequence.  Step i inserts the values of position i (none past the sequence's end) and evicts
// those of position i - 100; the state after step i belongs to output position i - 50.  Steps are taken 64 at a time: lane l loads
// step l's six values, everything that does not depend on the state is done lane-parallel, the chains run over LDS.
__global__ __launch_bounds__(128) void noise_walk_kernel(CallArgs a) {
    __shared__ __attribute__((aligned(16))) double vals[64][6];    // wave 0 (table): the block's values, evicted (3) and inserted (3) per step
    __shared__ __attribute__((aligned(16))) double svals[64][12];  // wave 1 (sums): per step and rank -- old, old^2, new, new^2
    __shared__ __attribute__((aligned(16))) double sout[64][2];    // ... s, s2 after each step
    const int file = a.out->file_id;
    if (file < 0 || (int)blockIdx.x >= a.n_seqs[file]) return;
    const int sq = a.seq_first[file] + (int)blockIdx.x;
    const uint64_t len = a.seq_len[sq], n_steps = len + kNoiseHalf;
    const uint64_t rel = a.seq_cell[sq] - a.seq_cell[a.seq_first[file]];
    const double* maf = a.noise_maf + rel * 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in_table = lane < kNoiseTop;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto load6 = [&](uint64_t i, int nb, double* x) {
#pragma unroll
        for (int r = 0; r < 6; ++r) x[r] = 0.0;
        if (lane < nb) {
            if (i >= (uint64_t)kNoiseWindow && i - kNoiseWindow < len) { const double* p = maf + (i - kNoiseWindow) * 3; x[0] = p[0]; x[1] = p[1]; x[2] = p[2]; }
            if (i < len) { const double* p = maf + i * 3; x[3] = p[0]; x[4] = p[1]; x[5] = p[2]; }
        }
    };
#ifdef BK_TESTING
    if ((a.noise_serial == 2 && wave == 0) || (a.noise_serial == 3 && wave == 1)) return;   // measurement aid (BK_NOISE_SERIAL=2 / 3): one chain alone
#endif
    if (wave == 0) {
        // ---- the table.  Its states are listed (one per step that changed it, state 0 = empty); every position gets the number of the
        // state after its step.  A step none of whose values can touch the table -- judged lane-parallel against the table's smallest
        // entry, again after every change -- is not walked at all.
        double* states = a.noise_tbl + (rel + 64ull * blockIdx.x) * kNoiseTop;
        unsigned int* ids = a.noise_state + rel;
        double top = 0.0, tmin = 0.0;   // lane q < 10: entry q (descending, zeros behind the values); tmin: entry 9, the same in every lane
        // The pass-by tests in INTEGER arithmetic (a double-precision compare or subtraction is a ~32-cycle step of this chain; the values
        // are frequencies in [0, 1]: they compare like their bit patterns).  An evicted value at or below xlo = tmin - 2e-12 surely
        // satisfies "tmin - old >= 1e-12" (the subtraction rounds by far less than 1e-12); what lies above xlo takes the exact test.
        auto bits = [](double v) { return (unsigned long long)__double_as_longlong(v); };
        unsigned long long tmin_b = 0ull, xlo_b = 0ull;
        bool full = false;
        auto new_tmin = [&]() {
            const unsigned long long tb = bits(top);
            tmin_b = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(tb >> 32), kNoiseTop - 1) << 32) |
                     (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)tb, kNoiseTop - 1);
            tmin = __longlong_as_double((long long)tmin_b);
            const double xlo = tmin - 2e-12;
            full = xlo > 0.0;
            xlo_b = bits(xlo);
        };
        unsigned int base_id = 0;
        if (in_table) states[lane] = 0.0;
        for (uint64_t b0 = 0; b0 < n_steps; b0 += 64) {
            const int nb = (int)min((uint64_t)64, n_steps - b0);
            double x[6];
            load6(b0 + (uint64_t)lane, nb, x);
#pragma unroll
            for (int r = 0; r < 6; ++r) vals[lane][r] = x[r];
            wave_sync();
            // evict (call.rs:853-870; a value above 0 is always flagged): the first entry within 1e-12 of it goes -- a full table whose
            // smallest entry lies 1e-12 or more above it holds none.  insert (call.rs:873-890): from the bottom up while it is larger
            // -- not at all when it is not larger than the smallest entry (an entry of a table that is not full is 0).
            unsigned long long xb[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) xb[r] = bits(x[r]);
            auto touches = [&]() {
                bool e = false;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    e |= xb[r] != 0ull && !(full && xb[r] <= xlo_b);
                    e |= xb[3 + r] > tmin_b;
                }
                return e && lane < nb;
            };
            unsigned long long todo = __ballot(touches()), changed_steps = 0ull;
            while (todo) {
                const int t = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                bool changed = false;
#pragma unroll
                for (int r = 0; r < 3; ++r) {                           // minor ranks 1..3, call.rs:848
                    const double old = vals[t][r];
                    const unsigned long long old_b = bits(old);
                    if (old_b != 0ull && !(full && old_b <= xlo_b) && !(tmin != 0.0 && tmin - old >= 1e-12)) {
                        const unsigned long long hit = __ballot(in_table && fabs(top - old) < 1e-12);
                        if (hit) {
                            const double above = row_from_above(top);
                            if (in_table && lane >= __builtin_ctzll(hit)) top = lane + 1 < kNoiseTop ? above : 0.0;
                            new_tmin();
                            changed = true;
                        }
                    }
                    const double mafv = vals[t][3 + r];
                    const unsigned long long maf_b = bits(mafv);
                    if (maf_b > tmin_b) {
                        const unsigned long long larger = __ballot(in_table && maf_b > bits(top));
                        const unsigned long long not_larger = ~larger & ((1ull << kNoiseTop) - 1ull);
                        const int p = not_larger ? 64 - (int)__builtin_clzll(not_larger) : 0;
                        if (p < kNoiseTop) {
                            const double below = row_from_below(top);
                            if (in_table && lane > p) top = below;
                            if (lane == p) top = mafv;
                            new_tmin();
                            changed = true;
                        }
                    }
                }
                if (changed) {
                    changed_steps |= 1ull << t;
                    if (in_table) states[(size_t)(base_id + (unsigned int)__popcll(changed_steps)) * kNoiseTop + lane] = top;
                    todo = __ballot(touches()) & ~((2ull << t) - 1ull);   // (the smallest entry may have moved: who touches the table now?)
                }
            }
            const uint64_t i = b0 + (uint64_t)lane;
            if (lane < nb && i >= (uint64_t)kNoiseHalf) ids[i - kNoiseHalf] = base_id + (unsigned int)__popcll(changed_steps & ((2ull << lane) - 1ull));
            base_id += (unsigned int)__popcll(changed_steps);
            wave_sync();
        }
    } else {
        // ---- the sums: s and s2 take six additions / subtractions per step each, in the reference's order; adding or subtracting the
        // 0.0 of a value that is not there changes nothing (neither sum is ever -0.0), so the chain has no branch.  n is a prefix sum.
        double* sums = a.noise_sums + rel * 2;
        unsigned int* cnts = a.noise_cnt + rel;
        double s = 0.0, s2 = 0.0;
        unsigned int n_base = 0;
        for (uint64_t b0 = 0; b0 < n_steps; b0 += 64) {
            const int nb = (int)min((uint64_t)64, n_steps - b0);
            double x[6];
            load6(b0 + (uint64_t)lane, nb, x);
            int dn = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                svals[lane][4 * r + 0] = x[r]; svals[lane][4 * r + 1] = x[r] * x[r];
                svals[lane][4 * r + 2] = x[3 + r]; svals[lane][4 * r + 3] = x[3 + r] * x[3 + r];
                dn += (x[3 + r] > 0.0 ? 1 : 0) - (x[r] > 0.0 ? 1 : 0);
            }
            int incl = dn;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(incl, off); if (lane >= off) incl += up; }
            wave_sync();
            double c[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) c[q] = svals[0][q];
            for (int t = 0; t < nb; ++t) {
                double nx[12];
                const int tn = t + 1 < 64 ? t + 1 : 63;               // (the next step's values are on their way while this one's are added)
#pragma unroll
                for (int q = 0; q < 12; ++q) nx[q] = svals[tn][q];
#pragma unroll
                for (int r = 0; r < 3; ++r) {                           // call.rs:853-857, :873-877
                    s -= c[4 * r + 0]; s2 -= c[4 * r + 1];
                    s += c[4 * r + 2]; s2 += c[4 * r + 3];
                }
                if (lane == 0) { sout[t][0] = s; sout[t][1] = s2; }
#pragma unroll
                for (int q = 0; q < 12; ++q) c[q] = nx[q];
            }
            wave_sync();
            const uint64_t i = b0 + (uint64_t)lane;
            if (lane < nb && i >= (uint64_t)kNoiseHalf) {
                sums[(i - kNoiseHalf) * 2] = sout[lane][0]; sums[(i - kNoiseHalf) * 2 + 1] = sout[lane][1];
                cnts[i - kNoiseHalf] = n_base + (unsigned int)incl;
            }
            n_base += (unsigned int)__shfl(incl, 63);
            wave_sync();
        }
    }
}

// one thread per position: the strip of call.rs:917-950 from the state the two chains left after that position's step
__global__ __launch_bounds__(256) void noise_strip_kernel(CallArgs a) {
    __shared__ double tau_s[kNoiseWindow * 3 + 1];
    const int file = a.out->file_id;
    if (file < 0) return;
    for (int n = threadIdx.x; n <= kNoiseWindow * 3; n += blockDim.x) {   // thompson_tau(n), call.rs:922-929
        double tau = INFINITY;
        if (n > 2) {
            const double t = kTCritDev[n - 3], dn = (double)n;
            tau = __ddiv_rn(t * (dn - 1.0), __dsqrt_rn(dn) * __dsqrt_rn(dn - 2.0 + t * t));
        }
        tau_s[n] = tau;
    }
    __syncthreads();
    const uint64_t cell_lo = a.seq_cell[a.seq_first[file]];
    const int sq_hi = a.seq_first[file] + a.n_seqs[file];
    const uint64_t cell_hi = a.n_seqs[file] ? a.seq_cell[sq_hi - 1] + a.seq_len[sq_hi - 1] : cell_lo;
    for (uint64_t cell = cell_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < cell_hi; cell += (uint64_t)gridDim.x * blockDim.x) {
        int sq = a.seq_first[file];
        while (sq + 1 < sq_hi && a.seq_cell[sq + 1] <= cell) ++sq;
        // the table after the position's step: state number noise_state[position] of its sequence's list
        const double* tp = a.noise_tbl + ((a.seq_cell[sq] - cell_lo) + 64ull * (uint64_t)(sq - a.seq_first[file]) + a.noise_state[cell - cell_lo]) * kNoiseTop;
        const unsigned int n0 = a.noise_cnt[cell - cell_lo];
        const double s0 = a.noise_sums[(cell - cell_lo) * 2], s20 = a.noise_sums[(cell - cell_lo) * 2 + 1];
        double mu = 0.0, var = 0.0;
        if (n0 != 0) { mu = __ddiv_rn(s0, (double)n0); var = __ddiv_rn(s20, (double)n0) - mu * mu; }   // population variance, call.rs:901-907
        int idx = 0;
        unsigned int cn = n0;
        double cs = s0, cs2 = s20;
        while (idx < kNoiseTop && tp[idx] != 0.0) {
            const double cand = tp[idx];
            const double tau = cn <= (unsigned int)(kNoiseWindow * 3) ? tau_s[cn] : NAN;
            if (!(fabs(cand - mu) > tau * __dsqrt_rn(var))) break;
            cs -= cand;
            cs2 -= cand;                                        // sic (call.rs:936): the value, not its square
            cn -= 1;
            if (cn > 0) { mu = __ddiv_rn(cs, (double)cn); var = __ddiv_rn(cs2, (double)cn) - mu * mu; }
            else { mu = 0.0; var = 0.0; }
            idx++;
        }
        a.noise[cell] = idx < kNoiseTop ? tp[idx] : 0.0;       // call.rs:953-962
    }
}

// call_variants, one thread per position of the selected genome (call.rs:1013-1137).  Records are appended through one
// device counter (the host sorts them by sequence, position, alternative base: the order upstream emits within a sequence).
__global__ __launch_bounds__(256) void call_kernel(CallArgs a) {
    __shared__ unsigned long long red[4];
    if (threadIdx.x < 4) red[threadIdx.x] = 0;
    __syncthreads();
    const int file = a.out->file_id;
    unsigned long long covered = 0, coverage = 0, positions = 0;
    if (file >= 0) {
        const uint64_t cell_lo = a.seq_cell[a.seq_first[file]];
        const int sq_hi = a.seq_first[file] + a.n_seqs[file];
        const uint64_t cell_hi = a.n_seqs[file] ? a.seq_cell[sq_hi - 1] + a.seq_len[sq_hi - 1] : cell_lo;
        const CallParamsDev& p = a.prm;
        for (uint64_t cell = cell_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < cell_hi; cell += (uint64_t)gridDim.x * blockDim.x) {
            // which sequence (a genome has a handful: linear search)
            int sq = a.seq_first[file];
            while (sq + 1 < sq_hi && a.seq_cell[sq + 1] <= cell) ++sq;
            const uint64_t len = a.seq_len[sq];
            const int64_t i = (int64_t)(cell - a.seq_cell[sq]);
            positions += 1;
            int64_t start = 0, end = (int64_t)len;
            if (!p.no_end_filter) { start = p.k; end = (int64_t)len - p.k; }      // call.rs:1013-1016
            if (i < start || i >= end) continue;
            const unsigned long long* row = a.pileup + 0 * a.plane + cell * 4;
            const unsigned long long* rrow = a.pileup + 1 * a.plane + cell * 4;
            const unsigned long long* fk = a.pileup + 2 * a.plane + cell * 4;
            const unsigned long long* rk = a.pileup + 3 * a.plane + cell * 4;
            const unsigned ref = (a.ref_words[cell >> 4] >> (2 * (cell & 15))) & 3u;   // non-ACGT counts as A (lcb.rs:53)
            unsigned long long tot[4], depth = 0;
            for (int b = 0; b < 4; ++b) { tot[b] = row[b] + rrow[b]; depth += tot[b]; }
            if (depth == 0) continue;
            covered += 1;
            coverage += depth;
            for (unsigned alt = 0; alt < 4; ++alt) {
                if (alt == ref || tot[alt] == 0) continue;
                double sor = p.strand_odds_max + 1.0;
                if (!p.no_strand_filter) {                                        // call.rs:1059-1096
                    const double fa = (double)row[ref] + 1.0, fb = (double)rrow[ref] + 1.0;
                    const double fc = (double)row[alt] + 1.0, fdd = (double)rrow[alt] + 1.0;
                    const double min_strand = __ddiv_rn(fmin(fa + fc, fb + fdd), fa + fb + fc + fdd);
                    if (!p.no_strand_balance_filter || min_strand >= p.strand_balance_ratio) {
                        const double r = __ddiv_rn(fa * fdd, fb * fc);
                        sor = log(r + __ddiv_rn(1.0, r)) + log(__ddiv_rn(fmin(fa, fb), fmax(fa, fb))) - log(__ddiv_rn(fmin(fc, fdd), fmax(fc, fdd)));
                        if (sor > p.strand_odds_max) continue;
                        if (fk[alt] < p.n_per_strand && rk[alt] < p.n_per_strand) continue;
                    } else {
                        sor = -1.0;
                    }
                }
                const double af = __ddiv_rn((double)tot[alt], (double)depth);
                const double y0 = p.variant_multiplier;
                const double factor = y0 + 0.5 * pow(0.03, 100.0 * af);           // call.rs:1102-1105
                if (af < p.min_af || af < fmax(factor, y0) * a.noise[cell]) continue;
                if (af >= 0.5) {
                    atomicAdd(&a.out->n_major, 1ull);
                } else {
                    if (depth < p.min_depth) continue;
                    if (tot[alt] < p.min_variant_depth) continue;
                    atomicAdd(&a.out->n_minor, 1ull);
                }
                const unsigned long long at = atomicAdd(&a.out->n_records, 1ull);
                if (at < a.record_cap) {
                    CallRecordDev rec;
                    rec.seq_id = sq - a.seq_first[file]; rec.ref_base = (uint8_t)ref; rec.alt_base = (uint8_t)alt; rec.pad = 0;
                    rec.pos = (uint64_t)i + 1;
                    rec.fwd_ref = row[ref]; rec.rev_ref = rrow[ref]; rec.fwd_alt = row[alt]; rec.rev_alt = rrow[alt]; rec.depth = depth;
                    rec.af = af; rec.sor = sor;
                    a.records[at] = rec;
                }
            }
        }
    }
    atomicAdd(&red[0], covered); atomicAdd(&red[1], coverage); atomicAdd(&red[2], positions);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (red[0]) atomicAdd(&a.out->covered, red[0]);
        if (red[1]) atomicAdd(&a.out->coverage, red[1]);
        if (red[2]) atomicAdd(&a.out->positions, red[2]);
    }
}

void launch_select_genome(const CallArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(select_genome_kernel, dim3(1), dim3(64), 0, stream, a);
}

void launch_call(const CallArgs& a, int max_seqs_per_file, uint64_t max_file_cells, hipStream_t stream) {
    hipLaunchKernelGGL(select_genome_kernel, dim3(1), dim3(64), 0, stream, a);
    const unsigned n_sq = (unsigned)(max_seqs_per_file > 0 ? max_seqs_per_file : 1);
    if (a.noise_tbl && a.noise_serial != 1) {
        // the walk's two chains side by side, then the strip per position (bk_caller.hip "the walk taken apart")
        const unsigned cblocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((max_file_cells + 255) / 256, 4096));
        hipLaunchKernelGGL(noise_maf_kernel, dim3(cblocks), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(noise_walk_kernel, dim3(n_sq), dim3(128), 0, stream, a);
        hipLaunchKernelGGL(noise_strip_kernel, dim3(cblocks), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL(noise_kernel, dim3(n_sq), dim3(256), 0, stream, a);   // (BK_NOISE_SERIAL, testing build: the walk in one wave, as rounds 2-4 ran it)
    }
    const unsigned blocks = (unsigned)((max_file_cells + 255) / 256);
    hipLaunchKernelGGL(call_kernel, dim3(blocks ? (blocks > 4096u ? 4096u : blocks) : 1u), dim3(256), 0, stream, a);
}

}  // namespace bk
