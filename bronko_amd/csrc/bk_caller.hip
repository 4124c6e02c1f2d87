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

// get_baseline_noise for sequence blockIdx.x of the selected genome.  All threads: the sorted minor-allele frequencies of
// every position (call.rs:831-845).  Thread 0: the sliding window (call.rs:848-962), writing Noise.max per position.
constexpr int kNoiseWindow = 100, kNoiseTop = kNoiseWindow / 10, kNoiseHalf = kNoiseWindow / 2;   // call.rs:802-804,824
__global__ __launch_bounds__(256) void noise_kernel(CallArgs a) {
    __shared__ double ring[kNoiseWindow * 3];
    __shared__ unsigned char flagged[kNoiseWindow * 3];
    __shared__ double top[kNoiseTop];
    __shared__ double tau_s[kNoiseWindow * 3 + 1];
    const int file = a.out->file_id;
    if (file < 0 || (int)blockIdx.x >= a.n_seqs[file]) return;
    const int sq = a.seq_first[file] + (int)blockIdx.x;
    const uint64_t cell0 = a.seq_cell[sq], len = a.seq_len[sq];
    const unsigned long long* fd = a.pileup + 0 * a.plane + cell0 * 4;
    const unsigned long long* rd = a.pileup + 1 * a.plane + cell0 * 4;
    double* fr = a.freq + cell0 * 3;
    for (uint64_t i = threadIdx.x; i < len; i += blockDim.x) {
        unsigned long long c0 = fd[i * 4 + 0] + rd[i * 4 + 0], c1 = fd[i * 4 + 1] + rd[i * 4 + 1];
        unsigned long long c2 = fd[i * 4 + 2] + rd[i * 4 + 2], c3 = fd[i * 4 + 3] + rd[i * 4 + 3];
        unsigned long long t;   // descending (a sorting network; equal values are interchangeable)
        if (c0 < c1) { t = c0; c0 = c1; c1 = t; }
        if (c2 < c3) { t = c2; c2 = c3; c3 = t; }
        if (c0 < c2) { t = c0; c0 = c2; c2 = t; }
        if (c1 < c3) { t = c1; c1 = c3; c3 = t; }
        if (c1 < c2) { t = c1; c1 = c2; c2 = t; }
        const unsigned long long depth = c0 + c1 + c2 + c3;
        fr[i * 3 + 0] = depth ? __ddiv_rn((double)c1, (double)depth) : 0.0;
        fr[i * 3 + 1] = depth ? __ddiv_rn((double)c2, (double)depth) : 0.0;
        fr[i * 3 + 2] = depth ? __ddiv_rn((double)c3, (double)depth) : 0.0;
    }
    for (int i = threadIdx.x; i < kNoiseWindow * 3; i += blockDim.x) { ring[i] = 0.0; flagged[i] = 0; }
    for (int i = threadIdx.x; i < kNoiseTop; i += blockDim.x) top[i] = 0.0;
    for (int n = threadIdx.x; n <= kNoiseWindow * 3; n += blockDim.x) {   // thompson_tau(n), call.rs:922-929
        double tau = INFINITY;
        if (n > 2) {
            const double t = kTCritDev[n - 3], dn = (double)n;
            tau = __ddiv_rn(t * (dn - 1.0), __dsqrt_rn(dn) * __dsqrt_rn(dn - 2.0 + t * t));
        }
        tau_s[n] = tau;
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x != 0) return;

    double* out = a.noise + cell0;
    unsigned long long n = 0;
    double s = 0.0, s2 = 0.0;
    for (uint64_t i = 0; i < len + kNoiseHalf; ++i) {
        const size_t slot0 = (size_t)(i % kNoiseWindow) * 3;
        for (int r = 1; r < 4; ++r) {                               // minor ranks 1..3, call.rs:848
            const size_t slot = slot0 + (size_t)(r - 1);
            const double old = ring[slot];
            if (old > 0.0) {                                        // evict, call.rs:853-870
                n -= 1; s -= old; s2 -= old * old;
                if (flagged[slot]) {
                    for (int q = 0; q < kNoiseTop; ++q) {
                        if (fabs(top[q] - old) < 1e-12) {
                            for (int z = q; z + 1 < kNoiseTop; ++z) top[z] = top[z + 1];
                            top[kNoiseTop - 1] = 0.0;
                            break;
                        }
                    }
                    flagged[slot] = 0;
                }
            }
            const double maf = i < len ? fr[i * 3 + (r - 1)] : 0.0;
            if (maf > 0.0) {                                        // insert, call.rs:873-890
                n += 1; s += maf; s2 += maf * maf;
                for (int q = kNoiseTop - 1; q >= 0; --q) {
                    if (!(maf > top[q])) break;
                    if (q + 1 < kNoiseTop) top[q + 1] = top[q];
                    top[q] = maf;
                }
                flagged[slot] = 1;                                  // set whether or not it entered the table
            } else {
                flagged[slot] = 0;
            }
            ring[slot] = maf;
        }
        double mu = 0.0, var = 0.0;
        if (n != 0) { mu = __ddiv_rn(s, (double)n); var = __ddiv_rn(s2, (double)n) - mu * mu; }   // population variance, call.rs:901-907
        int idx = 0;
        unsigned long long cn = n;
        double cs = s, cs2 = s2;
        while (idx < kNoiseTop && top[idx] != 0.0) {                // strip outliers, call.rs:917-950
            const double cand = top[idx];
            const double tau = cn <= (unsigned long long)(kNoiseWindow * 3) ? tau_s[cn] : NAN;
            if (!(fabs(cand - mu) > tau * __dsqrt_rn(var))) break;
            cs -= cand;
            cs2 -= cand;                                            // sic (call.rs:936): the value, not its square
            cn -= 1;
            if (cn > 0) { mu = __ddiv_rn(cs, (double)cn); var = __ddiv_rn(cs2, (double)cn) - mu * mu; }
            else { mu = 0.0; var = 0.0; }
            idx++;
        }
        if (i >= (uint64_t)kNoiseHalf && i - kNoiseHalf < len)      // call.rs:953-962
            out[i - kNoiseHalf] = idx < kNoiseTop ? top[idx] : 0.0;
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
    hipLaunchKernelGGL(noise_kernel, dim3((unsigned)(max_seqs_per_file > 0 ? max_seqs_per_file : 1)), dim3(256), 0, stream, a);
    const unsigned blocks = (unsigned)((max_file_cells + 255) / 256);
    hipLaunchKernelGGL(call_kernel, dim3(blocks ? (blocks > 4096u ? 4096u : blocks) : 1u), dim3(256), 0, stream, a);
}

}  // namespace bk
