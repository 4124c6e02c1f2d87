/*
 * bronko_hip.h -- C ABI of the MI355X (gfx950) k-mer -> pileup engine.
 *
 * This is the seam a host program binds (Rust `extern "C"` block, C++, ctypes ...).  It replaces, inside
 * bronko's `call()` per-sample loop, the stages
 *
 *     get_kmers        /root/reference/src/call.rs:630-646   (external KMC3: count_kmers_kmc :1152-1233,
 *                                                             load_kmers :1241-1255)
 *     initialize_output_maps                  call.rs:1437-1480
 *     map_kmers                               call.rs:1257-1434
 *
 * i.e. everything between "a FASTQ record was parsed" and "four pileup arrays + per-genome
 * (perfect, variant, unique) statistics exist" (call sites: call.rs:217-226 single-end, :301-317 paired).
 * The reference has no FFI of its own for this path; INTEGRATION.md shows the Rust binding a maintainer
 * would add.  Plain pointers and sizes only; every function returns 0 on success and a negative bk_status on
 * failure (message via bk_last_error(), thread-local).  The library never calls exit(); the host maps a
 * non-zero status to the reference's `error!(..); std::process::exit(1)` convention.  An engine is
 * thread-compatible (one engine per host thread / per GPU); there is no global state.
 *
 * There is NO CPU fallback: if no gfx950 device is visible every entry point that needs one fails with
 * BK_ERR_NO_DEVICE.
 */
#ifndef BRONKO_HIP_H
#define BRONKO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BK_ABI_VERSION 8

typedef enum {
    BK_OK = 0,
    BK_ERR_INVALID = -1,   /* bad argument / inconsistent index            */
    BK_ERR_NO_DEVICE = -2, /* no HIP device / not gfx950                   */
    BK_ERR_HIP = -3,       /* a HIP runtime call failed                    */
    BK_ERR_UNSUPPORTED = -4,
    BK_ERR_STATE = -5,     /* call order violated (e.g. push before begin) */
    BK_ERR_RANGE = -6      /* a counter did not fit the width its plane was exchanged at (bk_shard_transport) */
} bk_status;

/* build.rs:52-60  `#[repr(C)] struct BucketInfo` -- same field order, same layout (12 bytes) */
typedef struct {
    uint16_t file_id;   /* index of the genome file in ViralMetadata.files             */
    uint8_t  seq_id;    /* index of the sequence inside that file                      */
    uint32_t location;  /* 0-based start of the reference k-mer in the forward sequence */
    uint8_t  idx;       /* wildcard position inside the canonical k-mer                */
    uint8_t  canonical; /* 1 = the reference k-mer was reverse-complemented            */
} bk_bucket_info;

/* A decoded BronkoIndex (build.rs:23-50), flattened.  All pointers are host memory, borrowed for the
 * duration of bk_engine_create() only.
 *   global_index: FxHashMap<u64, Vec<BucketInfo>>  ->  bucket_ids[n_buckets] (any order, unique),
 *                 bucket_off[n_buckets + 1] into entries[n_entries]
 *   metadata    : files -> n_seqs[n_files]; sequences of all files concatenated in (file, seq) order:
 *                 seq_lens[total], seqs[total] (raw FASTA bytes as stored in SeqMeta.seq)               */
typedef struct {
    int32_t  k;
    uint64_t n_buckets;
    const uint64_t* bucket_ids;
    const uint64_t* bucket_off;
    const bk_bucket_info* entries;
    uint64_t n_entries;
    int32_t  n_files;
    const int32_t*  n_seqs;
    const uint64_t* seq_lens;
    const uint8_t* const* seqs;
} bk_index_desc;

/* Parameters that reach the hot path from CallArgs (cli.rs:61-166) and the KMC command line (call.rs:1166-1177) */
typedef struct {
    int32_t  n_fixed;        /* --n-fixed        (consts.rs:17, default 2)                                  */
    int32_t  use_full_kmer;  /* --use-full-kmer  (consts.rs:18, default 0)                                  */
    uint64_t ci;             /* --min-kmers -> kmc -ci (consts.rs:5, default 3): keep count >= ci           */
    uint64_t cs;             /* kmc -cs1000000 (call.rs:1173): reported count saturates at cs               */
    uint64_t cx;             /* kmc -cx default 1e9: drop k-mers whose true count exceeds cx                */
    int32_t  device;         /* HIP device ordinal                                                          */
    int32_t  full_kmer_stats;/* 1: also count every k-mer that does NOT touch the index in a device hash table, so
                                that KMC's "No. of unique k-mers" / "No. of unique counted k-mers" (call.rs:1190-1199)
                                are exact; 0 (default): only index-touching k-mers are counted (pileups identical) */
    uint32_t kmer_table_log2;/* initial capacity of that table = 2^kmer_table_log2 slots (default 26); it is rehashed into a
                                larger one whenever a batch might take its load above one half (up to 2^31 slots)          */
    uint32_t pileup_selected_only;/* 1: with several genome files, map_kmers' votes are cast for the SELECTED genome only (two passes:
                                per-genome statistics of all genomes, pick_best_genome on the device, then the votes): the
                                statistics and the selected genome's pileup rows are what the reference computes, the rows of
                                the other genomes stay zero -- nothing downstream of call.rs:229-235 reads them.  0 (default):
                                every genome's rows, as call.rs:1305-1384 fills them */
} bk_params;

typedef struct bk_engine bk_engine;

int         bk_abi_version(void);
/* Number of visible HIP devices (0 when there is none or the runtime cannot be initialised).  A host with several samples
 * creates one engine per device and deals whole samples to them (call.rs:212 / :297: samples are independent). */
int         bk_device_count(void);
/* Free and total memory of a device in bytes (v6): a host that runs several engines per device (bk_engine_fork: every fork holds
 * a sample's counter planes and outputs) sizes their number by it. */
int         bk_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes);
const char* bk_last_error(void);
void        bk_params_default(bk_params* p);

/* Uploads the index as a device-resident bucket table (see DESIGN.md "HBM layout") and allocates the
 * counter planes (one per mate file) and the four pileup arrays. */
int  bk_engine_create(const bk_index_desc* index, const bk_params* params, bk_engine** out);
void bk_engine_destroy(bk_engine* e);

/* A second engine on the same index: it reads the parent's device-resident tables (immutable after
 * bk_engine_create) and owns everything a sample writes -- counter planes, scan scratch, pileup / statistics
 * outputs, stream.  Samples are independent (call.rs:212 / :390 handle them one after the other), so a host
 * with many samples alternates them over two engines: sample i+1's scan overlaps sample i's finalize on the
 * device.  Same parameters as the parent.  Destroy the forks before the parent. */
int  bk_engine_fork(const bk_engine* parent, bk_engine** out);
/* The same with parameters of its own (v7): ci / cs / cx, pileup_selected_only and kmer_table_log2 belong to a sample's state and
 * may differ from the parent's; n_fixed, use_full_kmer, full_kmer_stats and device shape the shared tables and must not
 * (BK_ERR_INVALID).  E.g. one index, `--min-kmers 3` and `--min-kmers 5` samples side by side. */
int  bk_engine_fork_params(const bk_engine* parent, const bk_params* params, bk_engine** out);

/* Launch all work of this engine on an existing HIP stream (hipStream_t passed as void*); NULL restores the
 * engine's own stream.  Lets a host that already owns a stream (e.g. PyTorch's current stream) order and
 * time the engine's kernels. */
int bk_engine_set_stream(bk_engine* e, void* hip_stream);
/* The stream the engine launches on (hipStream_t as void*): its own, created with the engine, unless
 * bk_engine_set_stream replaced it.  A host orders its own work on the engine's buffers (e.g. RCCL collectives on
 * the counter plane) by enqueueing it on this stream. */
void* bk_engine_get_stream(const bk_engine* e);

/* Geometry of the outputs */
uint64_t bk_total_cells(const bk_engine* e);     /* sum of all sequence lengths = rows of each pileup array */
int32_t  bk_n_files(const bk_engine* e);
uint64_t bk_n_slots(const bk_engine* e);         /* distinct window buckets on the device                   */
uint64_t bk_counter_len(const bk_engine* e);     /* u64 elements in one counter plane                       */
int      bk_can_shard(const bk_engine* e);       /* v8: 1 = one sample's reads may be sharded over engines (bk_counters_device_ptr,
                                                  * bk_shard_*, bk_sample_finalize_shard); 0 = the index is so large that its planes
                                                  * are kept sparse: shard whole samples instead                                      */

/* ---- per-sample protocol (mirrors one iteration of call.rs:213-293 / :298-386) ---------------------------
 * bk_sample_begin      = initialize_output_maps (call.rs:224,314): zero pileups, stats and counter planes.
 * bk_push_reads_*      = the reads of mate file `mate` (0 = -r file or R1, 1 = R2); callable repeatedly.
 * bk_sample_finish     = KMC thresholds (ci/cs/cx, per mate file) + map_kmers for mate 0 then mate 1 into the
 *                        shared arrays (call.rs:316-317), then copy-out.
 *
 * Read batches are 2-bit packed fixed-stride records (produced by bk_pack_reads or by the host itself):
 *   record r = words[r*stride_words .. +stride_words), base i in word i/16 at bits [2*(i%16), 2*(i%16)+2),
 *   A=0 C=1 G=2 T=3; lens[r] = number of valid bases (<= 16*stride_words).  A record holds one maximal
 *   ACGT run (KMC splits reads at any other symbol) or an overlapping chunk of one.
 * The host buffer may be reused as soon as the call returns. */
int bk_sample_begin(bk_engine* e);
int bk_push_reads_packed(bk_engine* e, int mate, const uint32_t* words, uint32_t stride_words,
                         const uint16_t* lens, uint64_t n_records);
/* (bk_push_reads_packed stages the batch through one of two device buffers: the copy of a batch overlaps the scan of the
 * previous one; the call blocks only when both are still in use.) */
/* K0 from device memory (v7): the sequence lines are already resident (d_bases: bytes back to back, d_offsets: u64[n_reads + 1]);
 * the engine packs them into 2-bit records on its stream and scans them -- no copy, no host work beyond the launches.
 * total_bases = offsets[n_reads] - offsets[0] and longest_read (bases; sizes the record stride) are the host's to know.
 * d_bases may have any alignment (a sub-buffer of a larger device allocation is fine); d_offsets is 8-byte aligned. */
int bk_push_reads_ascii_device(bk_engine* e, int mate, const void* d_bases, const void* d_offsets, uint64_t n_reads,
                               uint64_t total_bases, uint32_t longest_read);
/* Same, for a batch that is already resident in device memory (no copy; asynchronous on the engine stream). */
int bk_push_reads_packed_device(bk_engine* e, int mate, const void* d_words, uint32_t stride_words,
                                const void* d_lens, uint64_t n_records);

/* K0 on the device + asynchronous ingest.  `buf` holds the sequence lines of n_reads reads back to back (read i =
 * buf[offsets[i] .. offsets[i+1]), any symbols); the engine copies them into a pinned staging slot, uploads them,
 * packs them into 2-bit records on the GPU (same splitting / chunking rules as bk_pack_reads) and scans them.  The
 * call returns as soon as the staging copy is made -- `buf` may be reused immediately -- and up to three batches
 * are in flight, so FASTQ parsing overlaps the copy, the packing and the scan.  bk_sample_finish waits for all. */
int bk_push_reads_ascii(bk_engine* e, int mate, const uint8_t* buf, const uint64_t* offsets, uint64_t n_reads);

/* Multi-GPU hook: the only additive quantity is the per-k-mer occurrence counter plane.
 * (Its u64 elements are difference arrays and counters whose sums wrap modulo 2^64 -- opaque to the host, linear.)
 * A host that shards one sample's reads over several GPUs all-reduces (sum, u64) each plane in place between
 * the last push and bk_sample_finish.  The pointer is device memory of bk_counter_len() u64; call this after the
 * sample's bk_sample_begin (a plane nothing was pushed to yet is zeroed here, not at begin). */
int bk_counters_device_ptr(bk_engine* e, int mate, void** d_ptr);

/* Runs the threshold + map_kmers kernels for mates [0, n_mates) on the device (asynchronous). */
int bk_sample_finalize(bk_engine* e, int n_mates);

/* Multi-GPU, cheaper form: instead of all-reducing the planes and finalizing everything on every rank, REDUCE-SCATTER each
 * plane over the n ranks (n divides 64; bk_counter_len() is a multiple of 64 and a part never cuts a counter row), then
 *   bk_sample_finalize_shard(e, n_mates, rank, n)   maps only the rank-th of n equal parts of each plane -- the part the
 *                                                   reduce-scatter left summed in place on this rank -- into this rank's
 *                                                   pileup arrays and statistics;
 *   all-reduce MAX the two depth planes and SUM the two #k-mer planes (bk_pileup_device_ptr: 4 planes of total_cells * 4
 *   u64, depth fwd, depth rev, #k-mers fwd, #k-mers rev), and SUM the bk_shard_sums_device_ptr vector;
 *   bk_sample_merge_shards(e)                       installs the summed statistics; bk_sample_download then returns the
 *                                                   same results as the all-reduce form on every rank.
 * (map_kmers' votes are max / += per k-mer, so maps of disjoint sets of k-mers combine by max / sum.) */
int bk_sample_finalize_shard(bk_engine* e, int n_mates, int shard, int n_shards);
int bk_shard_sums_device_ptr(bk_engine* e, void** d_ptr, uint64_t* len);
int bk_sample_merge_shards(bk_engine* e);
/* Transport of the planes for the sharded finalize (v7).  The u64 elements of a plane are counts and differences of counts; as
 * signed numbers they are small, and a reduce-scatter of a narrower copy moves a half or a quarter of the bytes over xGMI.  The
 * engine packs and widens on its own stream -- no host-side temporaries -- and leaves the plane itself untouched:
 *   bk_shard_measure(e, mate, &d_max)        optional.  d_max -> two u64 on the device: the largest E count and the largest |V
 *                                            element| of this rank's plane (asynchronous).  The host all-reduces them (MAX) and
 *                                            picks the width: 16 needs max|V| * n <= 32767 and max E < 2^32; 32 needs
 *                                            max(E, |V|) * n <= 2^31 - 1; 64 always fits.
 *   bk_shard_transport(e, mate, n, width, &d_send, &part_bytes, &d_recv)
 *                                            packs the plane (asynchronous) into n parts of part_bytes bytes at d_send: int32
 *                                            words that hold two 16-bit lanes each (width 16: a V element v as v + 32767 / n,
 *                                            an E count as four 8-bit digits -- unsigned lanes whose sums stay below 2^16 add
 *                                            up inside 32-bit additions; RCCL has no 16-bit integer type), int32 elements
 *                                            (32), or the plane itself (64: d_send is the plane).  The host reduce-scatters(sum)
 *                                            d_send -- element type int32 for widths 16 and 32, int64 for 64 -- leaving this
 *                                            rank's part at d_recv.  A packer that meets an element beyond (lane maximum) / n
 *                                            raises a flag that travels with bk_shard_sums_device_ptr to every rank.
 *   bk_shard_received(e, mate, shard, n, width)  widens the received part (asynchronous); bk_sample_finalize_shard(.., shard, n)
 *                                            then maps it instead of the plane's own elements.
 *   bk_transport_overflow(e, &flag)          synchronises; flag = some sample since the last call met such an element (its results are
 *                                            garbage: repeat it wider).  bk_sample_download reports the same as BK_ERR_RANGE.
 * The transport and `reduced` buffers are allocated once, at the first call, for the largest shard count and width there is:
 * pointers stay valid for the engine's lifetime.  Width 16 is refused (BK_ERR_INVALID) where it would send no fewer bytes than
 * width 32 -- every E count travels as four 16-bit lanes, so with many shards, or a plane that is mostly E counts, it does not pay. */
int bk_shard_measure(bk_engine* e, int mate, void** d_max);
int bk_shard_transport(bk_engine* e, int mate, int n_shards, int width, void** d_send, uint64_t* part_bytes, void** d_recv);
int bk_shard_received(bk_engine* e, int mate, int shard, int n_shards, int width);
int bk_transport_overflow(bk_engine* e, int* overflowed);
/* full_kmer_stats with a sharded finalize (v6): a k-mer that touches no window bucket sits in the statistics table of every
 * rank whose reads held it, and KMC's "unique (counted) k-mers" (call.rs:1190-1199) want it once, with its total count.  Between
 * the last push and bk_sample_finalize_shard every rank
 *   bk_kmer_table_partition(e, n, &keys, &counts, off)   lists its table's entries grouped by owner rank (a hash of the key):
 *                                                        device arrays of off[n] keys (u64) / counts (u32), group r = entries
 *                                                        [off[r], off[r + 1]); synchronises (the host sizes the exchange);
 *   exchanges the groups (all-to-all: group r goes to rank r);
 *   bk_kmer_table_replace(e, keys, counts, n_received)   rebuilds its table from what it received (equal keys add up).
 * bk_sample_finalize_shard refuses n_shards > 1 with full_kmer_stats unless the table was replaced in this sample.  The totals
 * then travel in the bk_shard_sums_device_ptr vector like the other statistics. */
int bk_kmer_table_partition(bk_engine* e, int n_parts, void** d_keys, void** d_counts, uint64_t* part_off /* [n_parts + 1] */);
int bk_kmer_table_replace(bk_engine* e, const void* d_keys, const void* d_counts, uint64_t n);
/* Device pointers of the finalized arrays: 4 planes (fwd depth, rev depth, fwd #kmers, rev #kmers) of
 * total_cells*4 u64 each, contiguous, in (file, seq, pos, base) order. */
int bk_pileup_device_ptr(bk_engine* e, void** d_ptr);
/* Synchronises and copies results to host memory.  Any pointer may be NULL (skipped).
 *   fwd_depth/rev_depth/fwd_nk/rev_nk : total_cells*4 u64 each    (OutputData.counts, call.rs:1235-1239)
 *   stats   : n_mates * n_files * 3 u64  (perfect, variant, unique) per mate file (call.rs:1272)
 *   present : n_mates * n_files bytes, 1 iff the file has a key in map_kmers' returned map
 *   kmer_stats : n_mates * 4 u64 = [0] records pushed, [1] k-mer occurrences scanned (KMC "Total no. of k-mers"),
 *                [2] distinct k-mers ("No. of unique k-mers"), [3] distinct k-mers kept by -ci/-cx ("No. of unique
 *                counted k-mers").  With full_kmer_stats = 0, [2] = 0 and [3] counts index-touching k-mers only;
 *                if the k-mer table could not grow any further and overflowed, [2] = [3] = UINT64_MAX.                */
int bk_sample_download(bk_engine* e, int n_mates, uint64_t* fwd_depth, uint64_t* rev_depth, uint64_t* fwd_nk,
                       uint64_t* rev_nk, uint64_t* stats, uint8_t* present, uint64_t* kmer_stats);
/* bk_sample_finalize + bk_sample_download */
int bk_sample_finish(bk_engine* e, int n_mates, uint64_t* fwd_depth, uint64_t* rev_depth, uint64_t* fwd_nk,
                     uint64_t* rev_nk, uint64_t* stats, uint8_t* present, uint64_t* kmer_stats);

/* ---- after the pileup, on the device (optional) ----------------------------------------------
 * For the sample just finalized, asynchronously on the engine's stream:
 *     pick_best_genome / pick_best_genome_paired   call.rs:422-502  (ties -> lowest file id; statistics summed over mates)
 *     get_baseline_noise                           call.rs:799-967  (IEEE doubles in upstream's order; one thread per
 *                                                                    sequence walks the window, see bk_caller.hip)
 *     call_variants                                call.rs:969-1150 (one thread per position)
 * A host with many samples in flight (one engine / fork each) never waits between a sample's reads and its records:
 *     bk_sample_begin .. bk_push_reads_* .. bk_sample_finalize .. bk_sample_call   (all asynchronous)
 *     bk_sample_download_calls                                                        (synchronises, copies the records)
 * The records come back sorted by (sequence, position, alternative base) = upstream's order within a sequence;
 * breadth = covered / positions, depth = coverage / covered (call.rs:1144-1145).  af is exact (one division); sor is the
 * device's ln() of exact ratios -- a host that prints it may re-derive it from the DP4 counts. */
typedef struct {            /* CallArgs fields that reach calling (cli.rs:92-135; defaults consts.rs:2-21) */
    int32_t  k;
    int32_t  no_end_filter, no_strand_filter, no_strand_balance_filter;
    double   min_af;                /* 0.03 */
    double   strand_balance_ratio;  /* 0.1  */
    double   strand_odds_max;       /* 6.0  */
    double   variant_multiplier;    /* 1.5  */
    uint64_t n_per_strand;          /* 2    */
    uint64_t min_depth;             /* 300  */
    uint64_t min_variant_depth;     /* 3    */
} bk_call_params;
typedef struct {            /* call.rs:776-789 VcfRecord */
    int32_t  seq_id;                /* index of the sequence inside the selected genome file */
    uint8_t  ref_base, alt_base;    /* 2-bit codes (A C G T) */
    uint16_t pad;
    uint64_t pos;                   /* 1-based */
    uint64_t fwd_ref, rev_ref, fwd_alt, rev_alt, depth;
    double   af, sor;
} bk_call_record;
typedef struct {
    int32_t  file_id;               /* selected genome file, -1 = none (call.rs:229-235: the host reports and exits) */
    uint32_t pad;
    uint64_t n_records;             /* records produced (all of them were copied iff <= cap) */
    uint64_t n_major, n_minor;      /* AF >= 0.5 / below */
    uint64_t covered, positions, coverage;
} bk_call_summary;
void bk_call_params_default(bk_call_params* p);
int  bk_sample_call(bk_engine* e, int n_mates, const bk_call_params* p);
int  bk_sample_download_calls(bk_engine* e, bk_call_summary* summary, bk_call_record* records, uint64_t cap);
/* Diagnostic: Noise.max of get_baseline_noise (call.rs:953-962) for every position of the genome bk_sample_call selected, in
 * (sequence, position) order -- what call_variants' AF filter compared with (call.rs:1107).  `cap` doubles at `out`; returns
 * the number of positions through *n (0 when no genome was selected).  Synchronises. */
int  bk_sample_download_noise(bk_engine* e, double* out, uint64_t cap, uint64_t* n);

/* ---- build_indexes on the device (optional) -----------------------------------------------------
 * build.rs:145-231 for the metadata sequences given like bk_index_desc gives them: one thread per k-mer writes its k
 * (bucket id, BucketInfo) pairs in generation order, a stable device radix sort groups them by bucket id (inside a bucket the
 * reference's order -- file, sequence, location -- survives), the host cuts the run into buckets.  The result is what
 * bk_index_desc takes: bucket ids ascending, bucket_off[n_buckets + 1], entries.  Arrays are malloc'ed; release them with
 * bk_built_index_free.  Message of a failure: bk_build_last_error(). */
typedef struct {
    uint64_t n_buckets, n_entries;
    uint64_t* bucket_ids;
    uint64_t* bucket_off;
    bk_bucket_info* entries;
} bk_built_index;
int  bk_build_index(int32_t k, int32_t n_files, const int32_t* n_seqs, const uint64_t* seq_lens, const uint8_t* const* seqs,
                    int32_t device, bk_built_index* out);
void bk_built_index_free(bk_built_index* ix);
const char* bk_build_last_error(void);

/* ---- K0: host-side read packer (the step KMC's FASTQ reader performs before counting) ---------------------
 * Splits each ASCII read at every non-ACGT/acgt symbol, drops runs shorter than k, cuts runs longer than
 * 16*stride_words into chunks overlapping by k-1 bases (so every k-mer occurrence is kept exactly once), and
 * writes fixed-stride 2-bit records.  Returns the number of records that the input produces; writes at most
 * `cap_records` of them (call with cap_records = 0 to size the buffers). */
uint64_t bk_pack_reads(const uint8_t* const* reads, const uint64_t* read_lens, uint64_t n_reads, int32_t k,
                       uint32_t stride_words, uint32_t* out_words, uint16_t* out_lens, uint64_t cap_records);
/* Same for reads stored back to back in one buffer: read i = buf[offsets[i] .. offsets[i+1]) */
uint64_t bk_pack_reads_flat(const uint8_t* buf, const uint64_t* offsets, uint64_t n_reads, int32_t k,
                            uint32_t stride_words, uint32_t* out_words, uint16_t* out_lens, uint64_t cap_records);

/* ---- measurement ------------------------------------------------------------------------------------------
 * When enabled, every kernel launch is bracketed by HIP events on the launch stream.  bk_timing_read
 * synchronises and returns accumulated milliseconds and launch counts since the last reset:
 *   [0] scan_count kernel, [1] finalize kernels, [2] memsets + H2D/D2H copies, [3] level2 + fold kernels.
 * on = 1 brackets all four kinds; on = 2 << kind (or-able) only the selected ones, e.g. 2 = the scan kernel alone
 * (two event records per launch instead of ten per sample).  Bits 8..15 of `on`, when not 0: only every N-th launch of a
 * kind is bracketed -- an event record makes the stream wait for the kernel before it and costs the GPU ~10 us of idle
 * time on that stream (rocprofv3 kernel trace), so a measurement that must not disturb what it measures samples. */
int bk_timing_enable(bk_engine* e, int on);
int bk_timing_read(bk_engine* e, double ms[4], uint64_t n[4], int reset);

#ifdef __cplusplus
}
#endif
#endif
