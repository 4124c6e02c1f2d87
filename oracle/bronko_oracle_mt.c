/*
 * bronko_oracle_mt.c -- CPU ORACLE, multi-threaded orchestration (test infrastructure, NOT product code).
 *
 * The same two stages as orc_sample_pileup (bronko_oracle.c), run the way the reference runs them on a multi-core host
 * (reference paths relative to /root/reference):
 *
 *   stage 1  exact strand-specific counting of ALL read k-mers = the external `kmc -t{threads} -b` run of
 *            src/call.rs:1166-1181 (KMC3 is not under /root/reference; its contract is restated in bronko_oracle.c).  Here: every
 *            thread scans a slice of the reads and deals the k-mers by hash into per-(thread, shard) buffers; then every thread
 *            counts one shard in a private open-addressing map -- no locks, the result is the multiset the single-threaded
 *            counter gives.
 *   stage 2  map_kmers over chunks of the kept k-mers in parallel = `kmers.par_chunks(chunk_size).for_each` of
 *            src/call.rs:1279-1281: every thread maps its chunks into the shared arrays with atomic += 1 / max (the reference
 *            serialises the same two updates through DashMap entries, call.rs:1341-1357); per-thread statistics are summed
 *            (call.rs:1420-1430).
 *
 * bench.py's cpu_baseline leg times this on the GPU box's host cores and reports the two stage times separately; the tests
 * check it against the single-threaded orc_sample_pileup bit for bit.
 */
#include "bronko_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static inline uint64_t mix64mt(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

typedef struct { uint64_t* v; uint64_t n, cap; } vec64;
static void vec_push(vec64* b, uint64_t x) {
    if (b->n == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 4096;
        b->v = (uint64_t*)realloc(b->v, b->cap * 8);
        if (!b->v) { fprintf(stderr, "oracle_mt: out of memory\n"); abort(); }
    }
    b->v[b->n++] = x;
}

typedef struct job_s {
    /* shared, read-only */
    const orc_index* ix;
    const orc_map_params* mp;
    const uint8_t* bases;
    const uint64_t* offsets;
    uint64_t r_lo, r_hi;        /* reads of the current mate file */
    int k, T, tid;
    /* stage 1a out: buckets[shard] of this thread; stage 1b in: all threads' buckets of shard tid */
    vec64* buckets;             /* [T] */
    vec64** all_buckets;        /* [T] -> each thread's buckets */
    uint64_t n_reads, n_kmers;  /* of this thread's slice */
    /* stage 1b out */
    uint64_t* kept_kmers; uint64_t* kept_counts; uint64_t n_kept, n_distinct;
    /* stage 2 */
    const uint64_t* kmers; const uint64_t* counts; uint64_t n_all;
    uint64_t* stats; uint8_t* present;   /* per thread, summed afterwards */
    uint64_t** merge_out;                /* the four shared output arrays */
} job_t;

static void* stage1a(void* arg) {
    job_t* j = (job_t*)arg;
    const int k = j->k;
    const uint64_t mask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    const uint64_t n = j->r_hi - j->r_lo;
    const uint64_t lo = j->r_lo + n * (uint64_t)j->tid / (uint64_t)j->T, hi = j->r_lo + n * (uint64_t)(j->tid + 1) / (uint64_t)j->T;
    {   /* reserve: the slice's k-mers spread evenly over the shards */
        const uint64_t bases = j->offsets[hi] - j->offsets[lo];
        const uint64_t per = bases / (uint64_t)j->T + bases / (uint64_t)j->T / 8 + 64;
        for (int s = 0; s < j->T; s++) { j->buckets[s].cap = per; j->buckets[s].v = (uint64_t*)malloc(per * 8); j->buckets[s].n = 0; }
    }
    for (uint64_t r = lo; r < hi; r++) {
        const uint8_t* seq = j->bases + j->offsets[r];
        const uint64_t len = j->offsets[r + 1] - j->offsets[r];
        uint64_t cur = 0; int valid = 0;
        j->n_reads++;
        for (uint64_t i = 0; i < len; i++) {   /* same k-mer enumeration as orc_counter_add_read */
            int b;
            switch (seq[i]) {
                case 'A': case 'a': b = 0; break;
                case 'C': case 'c': b = 1; break;
                case 'G': case 'g': b = 2; break;
                case 'T': case 't': b = 3; break;
                default: b = -1;
            }
            if (b < 0) { valid = 0; cur = 0; continue; }
            cur = ((cur << 2) | (uint64_t)b) & mask;
            if (++valid >= k) { vec_push(&j->buckets[mix64mt(cur) % (uint64_t)j->T], cur); j->n_kmers++; }
        }
    }
    return NULL;
}

static void* stage1b(void* arg) {
    job_t* j = (job_t*)arg;
    uint64_t total = 0;
    for (int t = 0; t < j->T; t++) total += j->all_buckets[t][j->tid].n;
    (void)total;
    uint64_t cap = 1 << 16;   /* grows with the distinct keys, load <= 0.6 (like orc_kmer_counter) */
    uint64_t* keys = (uint64_t*)malloc(cap * 8);
    uint64_t* vals = (uint64_t*)calloc(cap, 8);
    if (!keys || !vals) { fprintf(stderr, "oracle_mt: out of memory\n"); abort(); }
    memset(keys, 0xff, cap * 8);
    uint64_t m = cap - 1;
    uint64_t distinct = 0;
    for (int t = 0; t < j->T; t++) {
        const vec64* b = &j->all_buckets[t][j->tid];
        for (uint64_t i = 0; i < b->n; i++) {
            const uint64_t key = b->v[i];
            if ((distinct + 1) * 10 > cap * 6) {
                const uint64_t ocap = cap; uint64_t* ok = keys; uint64_t* ov = vals;
                cap *= 2; m = cap - 1;
                keys = (uint64_t*)malloc(cap * 8);
                vals = (uint64_t*)calloc(cap, 8);
                if (!keys || !vals) { fprintf(stderr, "oracle_mt: out of memory\n"); abort(); }
                memset(keys, 0xff, cap * 8);
                for (uint64_t q = 0; q < ocap; q++) {
                    if (ok[q] == ~0ull) continue;
                    uint64_t h2 = (mix64mt(ok[q]) >> 20) & m;
                    while (keys[h2] != ~0ull) h2 = (h2 + 1) & m;
                    keys[h2] = ok[q]; vals[h2] = ov[q];
                }
                free(ok); free(ov);
            }
            uint64_t h = (mix64mt(key) >> 20) & m;
            for (;;) {
                if (keys[h] == key) { vals[h]++; break; }
                if (keys[h] == ~0ull) { keys[h] = key; vals[h] = 1; distinct++; break; }
                h = (h + 1) & m;
            }
        }
    }
    j->kept_kmers = (uint64_t*)malloc((distinct ? distinct : 1) * 8);
    j->kept_counts = (uint64_t*)malloc((distinct ? distinct : 1) * 8);
    uint64_t o = 0;
    for (uint64_t i = 0; i < cap; i++) {
        if (keys[i] == ~0ull) continue;
        const uint64_t cnt = vals[i];
        if (cnt >= j->mp->ci && cnt <= j->mp->cx) {          /* -ci / -cx on the true count (orc_counter_finish) */
            j->kept_kmers[o] = keys[i];
            j->kept_counts[o] = cnt > j->mp->cs ? j->mp->cs : cnt;   /* -cs saturates */
            o++;
        }
    }
    j->n_kept = o; j->n_distinct = distinct;
    free(keys); free(vals);
    return NULL;
}

/* map_kmers over this thread's chunks (call.rs:1279-1281 par_chunks), voting straight into the shared arrays: the loop is
 * orc_map_kmers' (bronko_oracle.c, cited line by line there) with the two updates made atomic -- upstream serialises them
 * through DashMap entry locks (call.rs:1341-1357, :1366-1383): #k-mers += 1, depth = max(depth, n).  Statistics are
 * per-thread and summed afterwards (call.rs:1420-1430). */
static void* stage2(void* arg) {
    job_t* j = (job_t*)arg;
    const orc_index* ix = j->ix;
    const int k = j->k, n_files = orc_index_n_files(ix);
    const orc_bucket_info* entries = orc_index_entries(ix);
    uint64_t* hits = (uint64_t*)calloc((size_t)n_files + 1, 8);
    int* touched = (int*)malloc(sizeof(int) * ((size_t)n_files + 1));
    int w0, w1;                                                      /* window slice call.rs:1291-1300 */
    if (j->mp->use_full_kmer) { w0 = 0; w1 = k; }
    else if (j->mp->n_fixed * 2 + 1 >= k) { w0 = 0; w1 = 0; }
    else { w0 = j->mp->n_fixed; w1 = k - j->mp->n_fixed - 1; }
    const uint64_t num_buckets_perfect = (uint64_t)(w1 - w0);       /* call.rs:1302 */
    uint64_t chunk = j->n_all / (uint64_t)j->T;                      /* call.rs:1277 */
    if (chunk > 10000) chunk = 10000;
    if (chunk == 0) chunk = 1;
    uint64_t buckets[32];
    for (uint64_t c0 = (uint64_t)j->tid * chunk; c0 < j->n_all; c0 += (uint64_t)j->T * chunk) {
        const uint64_t c1 = c0 + chunk <= j->n_all ? c0 + chunk : j->n_all;
        for (uint64_t t = c0; t < c1; t++) {
            const uint64_t fwd = j->kmers[t], n = j->counts[t];
            const uint64_t rev = orc_reverse_complement_u64(fwd, k);
            uint64_t kmer_bin; int rc;
            if (fwd < rev) { kmer_bin = fwd; rc = 0; } else { kmer_bin = rev; rc = 1; }   /* lcb.rs:90-94 */
            orc_assign_buckets(kmer_bin, k, buckets);                                     /* call.rs:1289 */
            int n_touched = 0;
            for (int b = w0; b < w1; b++) {                                               /* call.rs:1305 */
                uint64_t first;
                const uint64_t cnt = orc_index_lookup(ix, buckets[b], &first);            /* call.rs:1307 */
                for (uint64_t e = 0; e < cnt; e++) {                                      /* call.rs:1309 */
                    const orc_bucket_info* info = &entries[first + e];
                    if (hits[info->file_id]++ == 0) touched[n_touched++] = info->file_id; /* call.rs:1316-1318 */
                    if ((int)info->file_id >= n_files || (int)info->seq_id >= orc_index_n_seqs(ix, info->file_id)) continue;
                    const uint64_t nuc_x = info->idx, idx = (uint64_t)info->location + nuc_x;   /* call.rs:1328-1334 */
                    if (idx >= orc_index_seq_len(ix, info->file_id, info->seq_id)) continue;
                    const uint64_t cell = (orc_index_cell_offset(ix, info->file_id, info->seq_id) + idx) * 4;
                    uint64_t bit_idx; int forward;
                    if (info->canonical) {
                        const uint64_t pos = (uint64_t)k - nuc_x - 1;                     /* call.rs:1332 */
                        bit_idx = ((kmer_bin >> (2 * ((uint64_t)k - pos - 1))) & 3u) ^ 3u; /* call.rs:1333 */
                        forward = rc ? 1 : 0;                                             /* call.rs:1336-1357 */
                    } else {
                        bit_idx = (kmer_bin >> (2 * ((uint64_t)k - nuc_x - 1))) & 3u;     /* call.rs:1360 */
                        forward = rc ? 0 : 1;                                             /* call.rs:1363-1383 */
                    }
                    uint64_t* nk = (forward ? j->merge_out[2] : j->merge_out[3]) + cell + bit_idx;
                    uint64_t* dp = (forward ? j->merge_out[0] : j->merge_out[1]) + cell + bit_idx;
                    __atomic_fetch_add(nk, 1, __ATOMIC_RELAXED);
                    uint64_t cur = __atomic_load_n(dp, __ATOMIC_RELAXED);
                    while (cur < n && !__atomic_compare_exchange_n(dp, &cur, n, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
                }
            }
            int n_perfect = 0, uniq = -1;                                                 /* call.rs:1390-1418 */
            for (int q = 0; q < n_touched; q++)
                if (hits[touched[q]] == num_buckets_perfect) { n_perfect++; uniq = touched[q]; }
            for (int q = 0; q < n_touched; q++) {
                const int f = touched[q];
                j->present[f] = 1;
                if (hits[f] == num_buckets_perfect) j->stats[f * 3 + 0] += 1;
                else if (hits[f] > 0) j->stats[f * 3 + 1] += 1;
                hits[f] = 0;
            }
            if (n_perfect == 1) { j->stats[uniq * 3 + 2] += 1; j->present[uniq] = 1; }
        }
    }
    free(hits); free(touched);
    return NULL;
}

static void run_all(job_t* jobs, int T, void* (*fn)(void*)) {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)T);
    for (int t = 1; t < T; t++) pthread_create(&th[t], NULL, fn, &jobs[t]);
    fn(&jobs[0]);
    for (int t = 1; t < T; t++) pthread_join(th[t], NULL);
    free(th);
}

void orc_sample_pileup_mt(const orc_index* ix, const orc_map_params* mp, int n_mates, const uint8_t* bases,
                          const uint64_t* offsets, const uint64_t* mate_off, int n_threads,
                          uint64_t* fwd_depth, uint64_t* rev_depth, uint64_t* fwd_nk, uint64_t* rev_nk,
                          uint64_t* stats, uint8_t* present, uint64_t* kmc_stats, double* stage_seconds) {
    const int T = n_threads < 1 ? 1 : n_threads;
    const int k = orc_index_k(ix);
    const int n_files = orc_index_n_files(ix);
    uint64_t* out[4] = {fwd_depth, rev_depth, fwd_nk, rev_nk};
    if (stage_seconds) stage_seconds[0] = stage_seconds[1] = 0.0;
    job_t* jobs = (job_t*)calloc((size_t)T, sizeof(job_t));
    vec64** all = (vec64**)calloc((size_t)T, sizeof(vec64*));
    for (int m = 0; m < n_mates; m++) {   /* one KMC run per mate file (call.rs:301-307), then map R1, R2 (call.rs:316-317) */
        const double t0 = now_s();
        for (int t = 0; t < T; t++) {
            memset(&jobs[t], 0, sizeof(job_t));
            jobs[t].ix = ix; jobs[t].mp = mp; jobs[t].bases = bases; jobs[t].offsets = offsets;
            jobs[t].r_lo = mate_off[m]; jobs[t].r_hi = mate_off[m + 1]; jobs[t].k = k; jobs[t].T = T; jobs[t].tid = t;
            jobs[t].buckets = (vec64*)calloc((size_t)T, sizeof(vec64));
            all[t] = jobs[t].buckets;
            jobs[t].all_buckets = all;
        }
        run_all(jobs, T, stage1a);
        run_all(jobs, T, stage1b);
        uint64_t n_all = 0, n_distinct = 0, n_reads = 0, n_kmers = 0;
        for (int t = 0; t < T; t++) { n_all += jobs[t].n_kept; n_distinct += jobs[t].n_distinct; n_reads += jobs[t].n_reads; n_kmers += jobs[t].n_kmers; }
        uint64_t* kmers = (uint64_t*)malloc((n_all ? n_all : 1) * 8);
        uint64_t* counts = (uint64_t*)malloc((n_all ? n_all : 1) * 8);
        uint64_t o = 0;
        for (int t = 0; t < T; t++) {
            memcpy(kmers + o, jobs[t].kept_kmers, jobs[t].n_kept * 8);
            memcpy(counts + o, jobs[t].kept_counts, jobs[t].n_kept * 8);
            o += jobs[t].n_kept;
            free(jobs[t].kept_kmers); free(jobs[t].kept_counts);
            for (int s = 0; s < T; s++) free(jobs[t].buckets[s].v);
            free(jobs[t].buckets);
        }
        if (kmc_stats) { kmc_stats[4 * m + 0] = n_reads; kmc_stats[4 * m + 1] = n_kmers; kmc_stats[4 * m + 2] = n_distinct; kmc_stats[4 * m + 3] = n_all; }
        const double t1 = now_s();
        for (int t = 0; t < T; t++) {
            jobs[t].kmers = kmers; jobs[t].counts = counts; jobs[t].n_all = n_all; jobs[t].merge_out = out;
            jobs[t].stats = (uint64_t*)calloc((size_t)n_files * 3 + 1, 8);
            jobs[t].present = (uint8_t*)calloc((size_t)n_files + 1, 1);
        }
        run_all(jobs, T, stage2);
        for (int t = 0; t < T; t++) {
            for (int f = 0; f < n_files; f++) {
                for (int q = 0; q < 3; q++) stats[((size_t)m * n_files + f) * 3 + q] += jobs[t].stats[f * 3 + q];
                if (jobs[t].present[f]) present[(size_t)m * n_files + f] = 1;
            }
            free(jobs[t].stats); free(jobs[t].present);
        }
        free(kmers); free(counts);
        const double t2 = now_s();
        if (stage_seconds) { stage_seconds[0] += t1 - t0; stage_seconds[1] += t2 - t1; }
    }
    free(jobs); free(all);
}
