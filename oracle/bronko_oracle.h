/*
 * bronko_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A literal, single-threaded C restatement of treangenlab/bronko v0.1.0's `call` k-mer -> pileup path
 * (and the host stages around it), written from the reference's behaviour.  Every function cites the
 * reference file:line it follows (paths relative to /root/reference).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product (bronko_amd/) never links or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   pinned by upstream artefacts : assign_buckets (src/lcb.rs:146-154 known answers), build_indexes + .bkdb
 *                                  codec (test_data/hpv.bkdb <=> test_data/HPV16.fa at k=21).
 *   PARITY UNPINNED              : everything on the `call` side (k-mer counting = external KMC3 binary,
 *                                  map_kmers, selection, noise (statrs Student-t), variant calls, writers):
 *                                  the reference has no test, golden output or runnable binary for it here
 *                                  (no cargo/rustc, no kmc in this environment).
 */
#ifndef BRONKO_ORACLE_H
#define BRONKO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- src/lcb.rs */
uint8_t  orc_nt_to_bits(uint8_t nt);                                   /* lcb.rs:47-55  */
uint64_t orc_kmer_to_u64(const uint8_t* kmer, int k);                  /* lcb.rs:67-74  */
uint64_t orc_reverse_complement_u64(uint64_t v, int k);                /* lcb.rs:76-85  */
uint64_t orc_canonical_kmer(const uint8_t* kmer, int k, int* was_rc);  /* lcb.rs:87-95  */
void     orc_assign_buckets(uint64_t kmer, int k, uint64_t* out_k);    /* lcb.rs:1-45   */

/* ---------------------------------------------------------------- src/build.rs types */
/* build.rs:52-60  #[repr(C)] BucketInfo: u16 @0, u8 @2, u32 @4, u8 @8, bool @9, size 12 */
typedef struct {
    uint16_t file_id;
    uint8_t  seq_id;
    uint32_t location;
    uint8_t  idx;
    uint8_t  canonical;
} orc_bucket_info;

typedef struct orc_index orc_index; /* BronkoIndex (build.rs:23-50): k + bucket map + ViralMetadata */

/* build.rs:145-231 build_indexes (files in the given order; needletail-style FASTA(.gz) parsing) */
orc_index* orc_index_build(int k, const char* const* fasta_paths, int n_files);
/* same, from in-memory sequences: one file per entry, n_seqs[f] sequences each (names / seqs flattened) */
orc_index* orc_index_build_mem(int k, int n_files, const char* const* file_names, const int* n_seqs,
                               const char* const* seq_names, const uint8_t* const* seqs, const uint64_t* seq_lens);
/* build.rs:122-143 save_index / call.rs:179-200 decode (bincode 2 standard config, varint) */
orc_index* orc_bkdb_load(const char* path);
int        orc_bkdb_save(const orc_index* ix, const char* path);
void       orc_index_free(orc_index* ix);
const char* orc_last_error(void);

int       orc_index_k(const orc_index* ix);
int       orc_index_meta_k(const orc_index* ix);
uint64_t  orc_index_n_buckets(const orc_index* ix);
uint64_t  orc_index_n_entries(const orc_index* ix);
/* buckets are exposed sorted by id; entries of bucket b are entries[off[b] .. off[b+1]) in insertion order */
const uint64_t*        orc_index_bucket_ids(const orc_index* ix);
const uint64_t*        orc_index_bucket_off(const orc_index* ix);
const orc_bucket_info* orc_index_entries(const orc_index* ix);
int       orc_index_n_files(const orc_index* ix);
const char* orc_index_file_name(const orc_index* ix, int f);
int       orc_index_n_seqs(const orc_index* ix, int f);
const char* orc_index_seq_name(const orc_index* ix, int f, int s);
uint64_t  orc_index_seq_len(const orc_index* ix, int f, int s);
const uint8_t* orc_index_seq(const orc_index* ix, int f, int s);
uint64_t  orc_index_total_cells(const orc_index* ix);            /* sum of all sequence lengths      */
uint64_t  orc_index_cell_offset(const orc_index* ix, int f, int s); /* first cell of (file, seq)       */
/* lookup: returns entry count, *first = index of first entry; 0 if the bucket id is absent */
uint64_t  orc_index_lookup(const orc_index* ix, uint64_t bucket_id, uint64_t* first);

/* ---------------------------------------------------------------- KMC3 contract (call.rs:1152-1255) */
/* Exact strand-specific k-mer counting of ASCII reads, "believed" KMC3 semantics for
 * `kmc -k{k} -b -ci{ci} -cs{cs}` (cx default 1e9): reads split at non-ACGT symbols, lower-case accepted,
 * counts are exact; a k-mer is kept iff ci <= count <= cx and reported as min(count, cs). */
typedef struct orc_kmer_counter orc_kmer_counter;
orc_kmer_counter* orc_counter_new(int k);
void     orc_counter_add_read(orc_kmer_counter* c, const uint8_t* seq, uint64_t len);
/* add a FASTQ(.gz) file; returns number of reads or (uint64_t)-1 on error */
uint64_t orc_counter_add_fastq(orc_kmer_counter* c, const char* path);
/* stats[4] = total_reads, total_kmers, unique_kmers, unique_counted_kmers (call.rs:1190-1199) */
uint64_t orc_counter_finish(orc_kmer_counter* c, uint64_t ci, uint64_t cs, uint64_t cx, uint64_t* stats4);
/* after finish: kept k-mers as 2-bit MSB-first values in read orientation (= kmer_to_u64 of the dumped
 * string, call.rs:1288 via lcb.rs:67) and their reported counts */
const uint64_t* orc_counter_kmers(const orc_kmer_counter* c);
const uint64_t* orc_counter_counts(const orc_kmer_counter* c);
void     orc_counter_free(orc_kmer_counter* c);

/* ---------------------------------------------------------------- map_kmers (call.rs:1257-1434) */
/* Pileups: four arrays of total_cells*4 u64 in (file, seq, pos, base) order:
 *   fwd_depth, rev_depth, fwd_nk, rev_nk  (initialize_output_maps call.rs:1437-1480; caller zeroes them).
 * stats: n_files*3 u64 (perfect, variant, unique) ADDED to; present: n_files bytes set to 1 when the file
 * has an entry in the returned map (call.rs:1405-1418). */
void orc_map_kmers(const orc_index* ix, const uint64_t* kmers, const uint64_t* counts, uint64_t n_kmers,
                   int n_fixed, int use_full_kmer,
                   uint64_t* fwd_depth, uint64_t* rev_depth, uint64_t* fwd_nk, uint64_t* rev_nk,
                   uint64_t* stats, uint8_t* present);

/* pick_best_genome (call.rs:422-450) / _paired (call.rs:452-502): stats already summed for paired.
 * Deterministic tie-break: ascending file id (upstream order is hash-map order).  Returns -1 for None. */
int orc_pick_best_genome(const orc_index* ix, const uint64_t* stats, const uint8_t* present);

/* ---------------------------------------------------------------- noise + calling (call.rs:792-1150) */
typedef struct {
    int     k;
    double  min_af;              /* cli.rs:88  default 0.03 */
    int     no_end_filter;
    int     no_strand_filter;
    int     no_strand_balance_filter;
    double  strand_balance_ratio;/* 0.1  */
    uint64_t n_per_strand;       /* 2    */
    double  strand_odds_max;     /* 6.0  */
    uint64_t min_depth;          /* 300  */
    uint64_t min_variant_depth;  /* 3    */
    double  variant_multiplier;  /* 1.5  */
} orc_call_params;
void orc_call_params_default(orc_call_params* p);

typedef struct {
    int      seq_id;
    uint64_t pos;            /* 1-based */
    uint8_t  ref_base, alt_base;
    uint64_t fwd_ref, rev_ref, fwd_alt, rev_alt, depth;
    double   af, sor;
} orc_vcf_record;

/* get_baseline_noise (call.rs:799-967): writes Noise.max / mean / std per position (len doubles each) */
void orc_baseline_noise(const uint64_t* fwd_depth4, const uint64_t* rev_depth4, uint64_t len,
                        double* nmax, double* nmean, double* nstd);

/* call_variants (call.rs:969-1150) over the selected file; sequences in metadata order.
 * Returns number of records (malloc'ed array in *out, free with orc_free); summary4 = n_major, n_minor,
 * then breadth and depth are returned through the double pointers. */
uint64_t orc_call_variants(const orc_index* ix, int file_id, const orc_call_params* p,
                           const uint64_t* fwd_depth, const uint64_t* rev_depth,
                           const uint64_t* fwd_nk, const uint64_t* rev_nk,
                           orc_vcf_record** out, uint64_t* n_major, uint64_t* n_minor,
                           double* breadth, double* depth_cov);
void orc_free(void* p);

/* ---------------------------------------------------------------- writers + names */
/* util.rs:30-50 clean_sample_id -> buf */
void orc_clean_sample_id(const char* path, char* buf, size_t buflen);
/* call.rs:735-774 print_output; call.rs:648-695 print_pileup.  Return 0 on success. */
int orc_write_vcf(const char* out_path, const char* reads_path_as_given, const orc_index* ix, int file_id,
                  const orc_vcf_record* recs, uint64_t n);
int orc_write_pileup(const char* out_path, const orc_index* ix, int file_id,
                     const uint64_t* fwd_depth, const uint64_t* rev_depth);

/* ---------------------------------------------------------------- orchestration (call.rs:212-387) */
typedef struct {
    int      n_fixed;        /* 2 */
    int      use_full_kmer;  /* 0 */
    uint64_t ci;             /* --min-kmers, 3 */
    uint64_t cs;             /* 1000000 (call.rs:1173) */
    uint64_t cx;             /* KMC default 1e9 */
} orc_map_params;
void orc_map_params_default(orc_map_params* p);

/* One sample through count -> map (per mate file, shared pileups) ; reads given as in-memory ASCII.
 * mate_off[m]..mate_off[m+1] index the reads of mate file m (n_mates = 1 single-end, 2 paired).
 * pileups (4 x total_cells*4) must be zeroed by the caller; stats = n_mates*n_files*3, present =
 * n_mates*n_files, kmc_stats = n_mates*4. */
void orc_sample_pileup(const orc_index* ix, const orc_map_params* mp, int n_mates,
                       const uint8_t* const* reads, const uint64_t* read_lens, const uint64_t* mate_off,
                       uint64_t* fwd_depth, uint64_t* rev_depth, uint64_t* fwd_nk, uint64_t* rev_nk,
                       uint64_t* stats, uint8_t* present, uint64_t* kmc_stats);

/* The same result computed on n_threads host threads the way the reference uses its cores (bronko_oracle_mt.c): stage 1 =
 * sharded exact counting (kmc -t, call.rs:1166-1181), stage 2 = map_kmers over chunks in parallel (call.rs:1279-1281).
 * Reads are given back to back: read r = bases[offsets[r] .. offsets[r+1]).  stage_seconds (optional) receives the
 * wall-clock of the two stages, summed over the mate files. */
void orc_sample_pileup_mt(const orc_index* ix, const orc_map_params* mp, int n_mates, const uint8_t* bases,
                          const uint64_t* offsets, const uint64_t* mate_off, int n_threads,
                          uint64_t* fwd_depth, uint64_t* rev_depth, uint64_t* fwd_nk, uint64_t* rev_nk,
                          uint64_t* stats, uint8_t* present, uint64_t* kmc_stats, double* stage_seconds);

#ifdef __cplusplus
}
#endif
#endif
