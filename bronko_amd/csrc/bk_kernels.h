// bk_kernels.h -- launch interface between the engine (host) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/bronko_hip.h"
#include "bk_device.h"

namespace bk {

// The binned scan (bk_scan_items.hip).  scan_items_kernel settles the reads exactly as scan_count_kernel does, but instead of adding
// to a whole-genome difference array in LDS (slabs, fold) and to the V plane with global atomics it EMITS what it would have
// added as 16-bit items, grouped by region of the reference:
//   E item  a run of n <= 255 consecutive cells each seen once more, in the bin of 128 window cells its first cell lies in:
//           bits 0-6 first cell & 127, bits 7-14 n - 1 (never 255: 0xffff is the null item), bit 15 the read runs against the reference
//   V item  +1 / -1 at one counter of the V part of the plane, in the bin of (1 << vq_log2) row positions q it lies in:
//           bits 0-14 offset of the counter from the bin's first, bit 15 = -1
// A workgroup collects its items per bin in LDS buckets (cap_e / cap_v items: what 1/256 of a million reads at the benchmark's
// error rate puts there, and three to four standard deviations) and writes the bucket area out as it is when it ends; a bin that
// outgrows its bucket -- the V rows of a sample's true variants, a coverage spike, every bin of a sample with ten times the
// benchmark's error rate -- continues in its EXTENSION in device memory (gext[wg][bin][kItemGCap]: a plain 2-byte store, the
// slot is the bin's LDS count minus the bucket's capacity).  tab[wg][bin] = items of the bin in that workgroup, bucket and extension
// together.  bin_count_kernel, one workgroup per bin, adds them up in LDS (a 2 x 384-cell difference array or the bin's V rows)
// and adds what is not zero to the plane: no slab, no fold, no same-address global atomic.  An item that finds no room even in the
// extension goes to a device-wide overflow list (bin << 16 | item) every bin's workgroup looks through, and past that list's end
// to the plane directly.  (Until round 4's last build a bin that outgrew its bucket took one of 96 extension buckets in LDS and
// everything else went to the list: with 700 bins overflowing -- 5 % sequencing errors -- the list was a million entries that
// every workgroup of bin_count read through.)
struct ItemGeom {
    uint32_t n_ebins, n_vbins;      // bins [0, n_ebins) are E bins, [n_ebins, n_ebins + n_vbins) V bins; at most 2 * 1024 in all
    uint32_t cap_e, cap_v;          // multiples of 8, <= 64
    uint32_t vq_log2;               // V bin = rows of 1 << vq_log2 positions q: (6 << vq_log2) * (v_span + 1) <= 32767 counters
    uint32_t wg_items;              // = n_ebins * cap_e + n_vbins * cap_v: bucket slots of one workgroup
    uint32_t wg_stride;             // = wg_items: u16 items of one workgroup's bucket area in LDS
    uint32_t grid_max;              // scan workgroups the buffers are laid out for: items[bin][grid_max][cap] -- a bin's buckets of all workgroups are one stretch
                                    // (bin_count reads it front to back) --, tab[bin][grid_max]
};
constexpr uint32_t kEBinLog2 = 7;   // E bins of 128 cells
constexpr uint32_t kERunMax = 255;  // cells one E item covers at most
constexpr uint32_t kEBinSpan = (1u << kEBinLog2) + kERunMax;   // cells an E bin's items reach from its first: 383 (the difference array has one more)
constexpr uint32_t kItemGCap = 256;   // slots of a (scan workgroup, bin)'s extension in device memory (ScanArgs::gext): what its LDS bucket has no room for

struct ScanArgs {
    const IndexView* ixp;           // device copy of the index view: only the rare paths of scan_count read it
    // what the hot path needs (kept in kernel-argument registers)
    int32_t k, wstart, W, v_omin, v_span;   // v_*: IndexView::v_omin / v_span / v_off
    uint64_t v_off;
    uint32_t total_cells, n_u;
    const uint32_t* ref_words;      // IndexView::ref_words / cell_codes (both with scan_ref_pad_words() words of front padding)
    const uint32_t* cell_codes;
    const uint32_t* cell_has;       // IndexView::cell_has / cell_clean (scan_bit_pad_words() words of front padding)
    const uint32_t* cell_clean;
    const uint32_t* cell_clean3;    // IndexView::cell_clean3 (same padding)
    const uint32_t* cell_yf;        // IndexView::cell_yf / cell_yr (same padding) / id_at
    const uint32_t* cell_yr;
    const uint32_t* id_at;
    const uint32_t* cell_fast;      // IndexView::cell_fast (1 bit per cell, padded like cell_has) / cell_blk (one entry per 64 cells)
    const uint2* cell_blk;
    const uint32_t* cell_nat;       // IndexView::cell_nat (or null)
    const uint32_t* cell_natrow;    // IndexView::cell_natrow (or null)
    const uint2* seed_tab;          // IndexView::seed_tab / seed_log2
    uint32_t seed_log2;
    const uint2* seed_tab2;         // scan_items_kernel: [n_files << seed2_log2] seed tables keyed by the k-mer as a read shows it (bases in reading
    uint32_t seed2_log2;            //   order from bit 0), both strands of every reference k-mer: entry = cell | strand << 27 | tag << 28
    const uint32_t* rc_words;       // ... and the reference reverse-complemented: symbol J = complement of symbol total_cells - 1 - J (paddings of ref_words)
    const uint32_t* words;          // [n_records][stride_words] 2-bit packed, 16 bases per word, LSB first
    const uint16_t* lens;           // [n_records] valid bases
    uint64_t rec_base;              // this launch covers records [rec_base, rec_base + n_records)
    uint64_t n_records;             // upper bound when n_records_dev is set (sizes nothing but the tile loop)
    const unsigned long long* n_records_dev;   // optional: actual record count written by pack_reads_kernel
    uint32_t stride_words;
    unsigned long long* counters;   // u64 plane [E | V] (bk_device.h)
    unsigned int* slabs;            // [grid][n_lds_bins] packed (rc<<16 | fwd) per-cell bins of each workgroup
    // Level 2's work lists, per launch: one bit per k-mer of each record and one bit per record that has any -- all zero between
    // launches (launch_level2 clears what it takes) -- and the diagonal of every marked record.
    //   n_bits / n_any   set by the scan: the k-mers of the N runs it does not settle itself (several mismatches in reach of each
    //                    other, cells that are not "fast", reads off the LDS window or without a diagonal); level2's first pass
    //                    resolves them run by run (S runs, dead pairs) and marks what is left in
    //   l2_bits / l2_any the k-mers that are looked at one by one (level2's second pass).
    unsigned int* n_bits;           // [n_records][l2_words]
    unsigned int* n_any;            // [ceil(n_records / 32)]
    unsigned int* l2_bits;          // [n_records][l2_words]
    unsigned int* l2_any;           // [ceil(n_records / 32)]
    unsigned int* l2_plan;          // null, or one word launch_level2 fills before level2_kernel: how many of its workgroups work
    uint32_t l2_min_grid;           // (l2_plan_kernel: marked records / 256, at least this many) -- with four or more samples in flight
    uint2* l2_diag;                 // [n_records] {cell of the reference k-mer aligned with read k-mer 0, bit 0 same strand | bit 1 known}
    uint32_t l2_words;              // = scan_l2_words(stride_words, k)
    uint32_t n_lds_bins;            // exact hits at cells [win_lo, win_lo + n_lds_bins) are counted in LDS
    uint32_t win_lo;                // first cell of that window (a multiple of 32): the genome the sample looks like
    const uint32_t* win_dev;        // multi-genome index: {win_file, win_lo} chosen on the device per sample (overrides the two fields)
    const uint32_t* occ;            // [n_full][n_files] cell | rc << 31 of the k-mer's first occurrence in each genome file
                                    // (0xffffffff: none), or null; with win_file, seeds land on that genome's copy of a k-mer
    int32_t win_file, n_files;
    int ref_in_lds;                 // stage the packed reference + cell codes in LDS (they fit next to the bins)
    unsigned long long* kmer_total; // optional: += k-mer occurrences scanned
    // full_kmer_stats: k-mers that do not touch the index are counted in this open-addressing table (null = off)
    unsigned long long* ktab_keys;  // [1 << ktab_log2] canonical k-mer | orientation << 63 | mate << 62; ~0 = free
    unsigned int* ktab_cnt;
    uint32_t ktab_log2;
    unsigned long long* ktab_overflow;
    uint32_t mate;
    // Large indexes ("sparse finalize"): whoever adds to a counter also sets the bit of its V row / pseudo row / reference k-mer, so
    // that finalize walks the touched ones instead of a plane of which a thousandth is non-zero, and clears what it read (no
    // plane memset).  Null for small indexes (dense finalize).
    unsigned int* touch_v;          // [v_real_rows / 32 + 1] bit per V row of the reference k-mers
    unsigned int* touch_b;          // [cells / 64 / 32 + 2] scan_count_kernel: bit per block of 64 cells whose rows it counted into (expanded into touch_v)
    unsigned int* touch_p;          // [n_prows / 32 + 1]     bit per pseudo k-mer row (8 counters)
    unsigned int* touch_e;          // [n_u / 32 + 1]         bit per id (its two E counters)
    unsigned long long rl_recip;    // ceil(2^64 / (v_span + 1)): counter index -> row by __umul64hi
    // the binned scan (launch_scan_items): where the items go
    ItemGeom ig;
    unsigned short* items;          // [bin][ig.grid_max][cap_e or cap_v] (E bins first)
    unsigned short* tab;            // [n_bins][ig.grid_max] items of the bin in that workgroup: the first cap_e / cap_v in its bucket, the rest in its extension
    unsigned short* gext;           // [grid][n_bins][kItemGCap] the extensions
    unsigned int* ov;               // [ov_cap] overflow list
    unsigned long long* ov_n;       // [2] entries appended by the launches of even / odd parity (may run past ov_cap: those went to the
    uint32_t ov_cap;                //     plane); bin_count_kernel zeroes the other parity's for the next launch
    uint32_t ov_par;
    int n_direct;                   // the binned scan on one single-sequence genome that fits the window: a read it cannot settle has no usable diagonal
                                    // (none found, or taken away: kMaxChunkMismatches) -- nbatch_kernel would only hand its k-mers on, so the scan
                                    // marks them for level2_kernel itself and nbatch_kernel is not launched
    uint32_t stage_off;             // set by launch_scan_items: byte offset of the waves' record buffers in LDS (0: records are read from memory)
    int ablate;                     // measurement aid (-DBK_TESTING build only): 1 = Level 1 only, 2 = no V atomics, 3 = no slow path
    unsigned long long* dbg;        // -DBK_TESTING build, BK_L2_STATS=1: [32] tallies of what is left to Level 2 and why; null otherwise
};

struct FinalizeArgs {
    IndexView ix;
    const unsigned long long* counters;
    uint64_t elem_lo, elem_hi;      // finalize the counters [elem_lo, elem_hi) of the plane only (a whole number of V rows)
    unsigned long long ci, cs, cx;
    unsigned long long* pileup;     // 4 planes of `plane` u64: fwd depth, rev depth, fwd #kmers, rev #kmers
    size_t plane;                   // total_cells * 4
    unsigned long long* stats;      // [n_files][3]
    unsigned char* present;         // [n_files]
    unsigned long long* kept_total; // optional: += distinct k-mers that passed the thresholds
    unsigned long long* distinct_total; // optional: += distinct k-mers seen at all (non-zero counters)
    unsigned int* partials;         // [finalize_partial_rows()][n_files*3 + 2] per-workgroup tallies, or null: use atomics
    int row_exact, row_general;     // first partials row of K2e / K2b (set by launch_finalize)
    const uint32_t* file_cell_lo;   // [n_files] first cell of each genome file, and the cells of the largest one: pileup_selected_only's
    uint32_t max_file_cells;        // voting pass walks the selected genome's cells (finalize_exact_own_kernel); null / 0 = it does not
    unsigned int* deferred;         // [v_plane_len] V counter indices K2a hands to K2b
    unsigned long long* deferred_n; // ... and their counts (clear_v: K2b cannot read them from the plane any more); may be null
    int clear_v;                    // K2a zeroes every V counter it reads (dense planes, the whole plane in this call, last pass):
                                    // the plane needs no memset before the next sample
    unsigned int* deferred_mask;    // ... pileup_selected_only with file bitmaps: per deferred k-mer, the window positions at which the statistics
                                    // pass found a bucket (what the voting pass probes); may be null
    unsigned int* n_deferred;       // [1], zeroed before each finalize
    // full_kmer_stats: k-mers recorded in V rows that cannot touch the index join the statistics table (null = off)
    unsigned long long* ktab_keys;
    unsigned int* ktab_cnt;
    uint32_t ktab_log2;
    unsigned long long* ktab_overflow;
    uint32_t mate;
    // bk_params.pileup_selected_only: 0 = statistics and votes in one pass; 1 = statistics only (first pass); 2 = votes only, and
    // only for BucketInfos of genome file *sel (second pass, after the genome was selected from the statistics)
    // sparse finalize (see ScanArgs::touch_v): the touched rows / ids, listed by launch_compact_touched; null = walk the plane
    const unsigned int* v_list;     // V rows of the reference k-mers
    const unsigned int* p_list;     // pseudo rows
    const unsigned int* e_list;     // ids
    const unsigned int* n_list;     // their lengths: [0] v_list, [2] reference k-mer ids (front of e_list), [3] pseudo k-mer ids (its tail), [4] p_list
    int mode;
    const int* sel;
    int sel_file;                   // = *sel, read by each kernel of the second pass
    unsigned int* lean_e_list;      // bk_finalize_lean.hip: [n_full] the reference k-mers finalize_ecell_kernel leaves to finalize_exact_kernel (repeats), and
    unsigned int* lean_n_list;      //   [8] their number at [2] (the layout of n_list); null: no regional kernels
    int no_lean;                    // testing aid (BK_NO_LEAN_FINALIZE): the general K2a / K2e even where the regional kernels of bk_finalize_lean.hip apply
    const unsigned int* tail_e_list;   // set by launch_finalize for finalize_general_kernel: the regional finalize's lean_e_list / lean_n_list, whose
    const unsigned int* tail_n_list;   //   k-mers' E counters it maps behind its deferred k-mers (null otherwise)
    unsigned long long* zero_e;     // launch_finalize's last kernel also zeroes zero_e[0, zero_e_n): the plane's E part behind the sample (or null)
    size_t zero_e_n;
    // The regional finalize fed from the scan's V items (a mate file whose reads were one launch: the V part of its plane is never
    // written).  A V bin of the binned scan is the 64 row positions one finalize_vbin workgroup owns: the workgroup adds the bin's
    // items up in LDS -- what bin_count_kernel does before it stores the bin -- and takes its counts from there.  What Level 2 added
    // to the plane meanwhile is found through f_touch (a bit per V row it wrote: the SPARSE form of level2_kernel); the rows read
    // are zeroed, the bits cleared.  f_items null: the counts are the plane's.
    // Gathered votes (bk_gather.hip; many genomes, sparse planes): mode 1 is the statistics pass as ever, except that
    // finalize_general_kernel notes the deferred k-mers' alias hits; the voting pass (mode 2: the selected genome, mode 3: every
    // genome) is gather_votes_kernel plus, with `gather` set, the general kernels restricted to what it cannot see: votes through
    // alias keys (IndexView::slot_alias) from the pseudo rows, the pseudo k-mers' E counters and the noted hits.
    int gather;
    int gather_ablate;              // measurement aid (-DBK_TESTING build, BK_GATHER_ABLATE): 1 no LDS atomics, 2 no counter loads, 3 no answers either
    unsigned long long* alias_hits; // [alias_cap][3]: canonical k-mer | isrc << 62, count, slot
    unsigned int* n_alias_hits;     // [1], zeroed before the statistics pass
    unsigned int alias_cap;
    const uint64_t* merged_slots;   // [n_merged_slots][2] the window slots of buckets that hold several keys (k = 31: reference buckets whose ids wrapped
    uint32_t n_merged_slots;        //   onto each other): slot | window position << 32, key.  What votes through one key also votes for the other keys' BucketInfos
    const unsigned short* f_items;  // BinArgs::items / tab / gext / ov / ov_n of the launch
    const unsigned short* f_tab;
    const unsigned short* f_gext;
    const unsigned int* f_ov;
    const unsigned long long* f_ov_n;
    uint32_t f_ov_cap, f_ov_par, f_n_wg;
    ItemGeom f_ig;
    unsigned int* f_touch;          // [v_real_rows / 32 + 1]
};

struct FoldArgs {
    const unsigned int* slabs;
    uint32_t n_slabs;               // = grid of the scan launch
    uint32_t n_lds_bins;
    const uint32_t* id_at;          // cell -> id (bins are per cell)
    uint32_t win_lo;                // slab cell i is cell win_lo + i
    const uint32_t* win_dev;        // ... or {win_file, win_lo} chosen on the device
    unsigned int* touch_e;          // sparse finalize: bit per id that received a count (null: dense)
    const uint32_t* cell_codes;     // IndexView::cell_codes + its front padding: symbol 0 = cell 0
    unsigned long long* counters;
};

struct PackArgs {
    const uint8_t* bases;           // sequence lines back to back (16-byte aligned: the caller's pointer rounded down)
    uint32_t shift;                 // ... and by how many bytes: line i starts at bases + shift + offsets[i]
    const unsigned long long* offsets;   // [n_reads + 1]
    uint64_t n_reads;
    int32_t k;
    uint32_t stride_words;
    uint32_t* words;                // out: [cap][stride_words]
    uint16_t* lens;                 // out: [cap]
    uint64_t cap;
    unsigned long long* n_records;  // out (device) [3]: record slots in use (read i's record in slot i, everything else appended behind
                                    // the n_reads slots), records that hold a run, reads on the work list; all set by the launcher
    unsigned long long* n_real;     // the tally of records that hold a run (set by the launcher: its `tally` argument)
    uint32_t* work;                 // [n_reads] scratch: the reads that are not one clean run that fits a record (or null: no word-per-thread kernel)
};
void launch_pack_reads(const PackArgs& a, unsigned long long* tally, hipStream_t stream);   // tally: the sample's count of records that hold a run (+= this batch's)
// votes[f] += number of the first records' middle k-mers that occur in genome file f (which genome does the sample look like?)
void launch_pick_window(const ScanArgs& a, uint64_t n_probe, unsigned int* votes, const uint32_t* file_cell_lo, int forced, uint32_t* win,
                        hipStream_t stream);
void launch_count_kmers(const ScanArgs& a, hipStream_t stream);   // empty window: kmer_total += k-mer occurrences, nothing else
void launch_add_u64(unsigned long long* dst, const unsigned long long* src, hipStream_t stream);   // *dst += *src
void launch_add_const_u64(unsigned long long* dst, unsigned long long v, hipStream_t stream);        // *dst += v
uint32_t scan_grid(uint64_t n_records, int n_cus);
uint32_t scan_l2_words(uint32_t stride_words, int k);   // bitmap words per record (ScanArgs::l2_words)
uint64_t scan_max_records(uint32_t grid);   // most records one launch_scan_count may be given
int scan_ref_pad_words();
int scan_ref_back_words();
int scan_bit_pad_words();
int scan_bit_back_words();
size_t scan_lds_budget();   // bytes available for histogram bins (4 B each) + the staged reference
size_t scan_ref_lds_bytes(uint32_t total_cells);
size_t scan_lds_bytes(uint32_t n_lds_bins, bool ref_in_lds, uint32_t total_cells);
hipError_t launch_level2(const ScanArgs& a, int n_cus, hipStream_t stream);   // after launch_scan_count, same arguments
hipError_t launch_scan_count(const ScanArgs& a, uint32_t grid, hipStream_t stream);
void launch_fold(const FoldArgs& f, hipStream_t stream);
// ---- the binned scan (bk_scan_items.hip) ----
struct BinArgs {
    ItemGeom ig;
    const unsigned short* items;    // as ScanArgs
    const unsigned short* tab;
    const unsigned short* gext;
    uint32_t n_wg;                  // grid of the scan_items launch
    const unsigned int* ov;
    unsigned long long* ov_n;
    uint32_t ov_cap, ov_par;
    const uint32_t* id_at;          // cell -> id
    const uint32_t* cell_codes;     // IndexView::cell_codes past its front padding: symbol 0 = cell 0
    uint32_t win_lo;                // window cell i is cell win_lo + i
    const uint32_t* win_dev;        // ... or {win_file, win_lo} chosen on the device
    uint32_t total_cells;
    unsigned long long* counters;   // the plane
    uint64_t v_off, v_real_len;     // its V part (the reference k-mers' rows)
    uint32_t rl;                    // v_span + 1
    int v_mode;                     // how a V bin reaches the plane: 0 atomics, 1 plain read-modify-write, 2 plain stores (the V part is all zero)
    int part;                       // 0: every bin; 1: the E bins only (the V bins' items wait for the regional finalize, FinalizeArgs::f_items, or
                                    // for a launch with part = 2); 2: the V bins only
    int ablate;                     // measurement aid (-DBK_TESTING build only): 1 no items read, 2 nothing written to the plane, 3 no E atomics, 4 no V writes
};
// can this index take the binned scan (window of win_cells cells in LDS, dense planes)?  Fills g.
bool item_geometry(uint32_t win_cells, uint32_t n_full, int v_span, ItemGeom* g);
uint32_t items_grid(uint64_t n_records, int n_cus);
uint32_t items_max_grid(int n_cus);
size_t items_lds_bytes(const ItemGeom& g, uint32_t win_cells);
hipError_t launch_scan_items(const ScanArgs& a, uint32_t grid, hipStream_t stream);
hipError_t launch_bin_count(const BinArgs& b, hipStream_t stream);
void launch_ktab_stats(const unsigned long long* keys, const unsigned int* cnt, uint32_t log2n, unsigned long long ci,
                       unsigned long long cx, unsigned long long* out, hipStream_t stream);
void launch_ktab_rehash(const unsigned long long* okeys, const unsigned int* ocnt, uint32_t olog2, unsigned long long* nkeys, unsigned int* ncnt,
                        uint32_t nlog2, unsigned long long* overflow, hipStream_t stream);
void launch_ktab_count_parts(const unsigned long long* keys, uint32_t log2n, uint32_t n_parts, unsigned long long* counts, hipStream_t stream);
void launch_ktab_scatter_parts(const unsigned long long* keys, const unsigned int* cnt, uint32_t log2n, uint32_t n_parts, unsigned long long* cursors,
                               unsigned long long* out_keys, unsigned int* out_cnt, hipStream_t stream);
void launch_ktab_import(const unsigned long long* in_keys, const unsigned int* in_cnt, uint64_t n, unsigned long long* keys, unsigned int* cnt, uint32_t log2n,
                        unsigned long long* overflow, hipStream_t stream);
void launch_ktab_totals_to_kstats(unsigned long long* ktab_out, unsigned long long* kstats, int n_mates, hipStream_t stream);
uint32_t ktab_fill_words();   // tallies of new keys behind the overflow word: ktab_out[8 ..]
void launch_finalize(const FinalizeArgs& a, hipStream_t stream);
// K2a / K2e organised by region of the reference (bk_finalize_lean.hip): one genome file, dense planes, one pass
bool finalize_lean_ok(const FinalizeArgs& a);
unsigned launch_finalize_lean_variant(const FinalizeArgs& a, hipStream_t stream);   // returns its grid (rows of FinalizeArgs::partials it writes)
// sparse finalize: touch bitmaps -> lists (the bitmaps are cleared on the way); lists -> their counters zeroed again
void launch_expand_touched_blocks(unsigned int* touch_b, uint32_t n_blocks, const uint2* cell_blk, unsigned int* touch_v, uint32_t span, uint64_t n_q, hipStream_t stream);
void launch_compact_touched(unsigned int* touch_v, uint64_t n_rows, unsigned int* touch_p, uint64_t n_prows, unsigned int* touch_e, uint64_t n_ids,
                            uint64_t n_full, unsigned int* v_list, unsigned int* p_list, unsigned int* e_list, unsigned int* n_list, hipStream_t stream);
void launch_clear_touched(unsigned long long* counters, uint64_t v_off, uint64_t v_real_len, uint32_t rl, const unsigned int* v_list,
                          const unsigned int* p_list, const unsigned int* e_list, const unsigned int* n_list, uint64_t n_ids, hipStream_t stream);
// one launch zeroes what a sample starts from: the engine's small per-sample buffers and the pileup arrays (`big`)
void launch_zero_small(unsigned long long* a, size_t na, unsigned long long* b, size_t nb, unsigned long long* c, size_t nc,
                       unsigned char* d, size_t nd, unsigned int* e, size_t ne, unsigned long long* big, size_t nbig, hipStream_t stream);
// shard_sums <-> {stats, present, kstats}: the small additive results of a sharded finalize as one u64 vector
void launch_pack_sums(unsigned long long* sums, const unsigned long long* stats, const unsigned char* present, const unsigned long long* kstats,
                      int n_files, unsigned long long* xflag, hipStream_t stream);
void launch_unpack_sums(const unsigned long long* sums, unsigned long long* stats, unsigned char* present, unsigned long long* kstats,
                        int n_files, unsigned long long* xflag, hipStream_t stream);
// sharded finalize: a counter plane packed for the reduce-scatter (16- or 32-bit elements, n_parts parts) and a received part
// widened again; xflag[0] is raised when an element does not fit (bk_kernels.hip, xport_pack_kernel)
uint64_t xport_part_bytes(uint64_t plane_len, uint64_t v_off, uint32_t n_parts, int width);
void launch_xport_measure(const unsigned long long* plane, uint64_t plane_len, uint64_t v_off, unsigned long long* out, hipStream_t stream);
void launch_xport_pack(const unsigned long long* plane, uint64_t plane_len, uint64_t v_off, uint32_t n_parts, int width, void* buf, unsigned long long* flag,
                       hipStream_t stream);
void launch_xport_unpack(const void* recv, uint64_t plane_len, uint64_t v_off, uint32_t n_parts, uint32_t shard, int width, unsigned long long* reduced,
                         hipStream_t stream);
// ---- after the pileup (bk_caller.hip) ----
typedef bk_call_params CallParamsDev;
typedef bk_call_record CallRecordDev;
typedef bk_call_summary CallSummaryDev;
struct CallArgs {
    CallParamsDev prm;
    int32_t n_files, n_mates;
    const unsigned long long* stats;     // [n_mates][n_files][3]
    const unsigned char* present;        // [n_mates][n_files]
    const uint64_t* genome_len;          // [n_files] sum of the file's sequence lengths
    const int32_t* seq_first;            // [n_files] index of the file's first sequence
    const int32_t* n_seqs;               // [n_files]
    const uint64_t* seq_cell;            // [sequences] first cell
    const uint64_t* seq_len;             // [sequences]
    const uint32_t* ref_words;           // IndexView::ref_words past its front padding: symbol 0 = cell 0
    const unsigned long long* pileup;    // 4 planes of `plane` u64
    size_t plane;
    double* noise;                       // [total_cells] Noise.max per position
    // get_baseline_noise with its two chains apart (bk_caller.hip): scratch of the selected genome -- [cells][3] sorted minor-allele
    // frequencies, [cells][10] the table / [cells][2] s, s2 / [cells] n after each position's step; null: the walk in one wave
    double* noise_maf;
    double* noise_tbl;                   // [cells + 64 sequences][10]: per sequence the list of the table's states (state 0: empty; one more per step that changed it)
    unsigned int* noise_state;           // [cells] the state after each position's step
    double* noise_sums;
    unsigned int* noise_cnt;
    int noise_serial;                    // testing aid (BK_NOISE_SERIAL): the serial kernel for every sequence
    CallRecordDev* records;
    uint64_t record_cap;
    CallSummaryDev* out;
};
void launch_call(const CallArgs& a, int max_seqs_per_file, uint64_t max_file_cells, hipStream_t stream);
void launch_select_genome(const CallArgs& a, hipStream_t stream);   // the first kernel of launch_call alone: a.out->file_id
size_t finalize_lds_bytes(int n_files);
size_t finalize_partial_rows();
void launch_prefix_rows(unsigned long long* counters, const IndexView& ix, const unsigned int* v_list, const unsigned int* n_list, unsigned int* row_bits, hipStream_t stream);   // bk_gather.hip
void launch_gather_votes(const FinalizeArgs& a, const unsigned long long* counters1, hipStream_t stream);
bool vote_table_fits(const FinalizeArgs& a);        // every genome's rows and -cs small enough for the table of voters (bk_gather.hip)
size_t vote_table_words(const IndexView& ix);
void launch_gather_votes_table(const FinalizeArgs& a, const unsigned long long* counters1, uint32_t* tab, const unsigned int* row_bits, hipStream_t stream);
void launch_alias_votes(const FinalizeArgs& a, hipStream_t stream);
void launch_merged_votes(const FinalizeArgs& a, hipStream_t stream);
void launch_zero_genome_rows(unsigned long long* pileup, size_t plane, const uint32_t* file_cell_lo, int n_files, uint32_t total_cells, const int* last_sel,
                             hipStream_t stream);
void launch_copy_int(int* dst, const int* src, hipStream_t stream);
// bk_build.hip: order[i] = index of the i-th smallest of n u64 keys (radix sort on the device over bits [0, end_bit), stable); h_sorted_keys may be null
hipError_t device_sort_order(const unsigned long long* h_keys, size_t n, int end_bit, unsigned int* h_order, unsigned long long* h_sorted_keys);
// bk_build.hip (bk_engine_create): out[i * W + t] = window bucket (slot) of keys[i] at window position t, `empty` where there is none (device buffer),
// valid[i] = the positions found (host); dst[id_of[i] * W + t] = src[i * W + t] (device buffers)
hipError_t device_lookup_slots(const TableSlot* d_table, uint32_t log2s, const unsigned long long* h_keys, size_t n, int W, int wstart, int k, uint32_t empty,
                               uint32_t* d_out, uint32_t* h_valid);
hipError_t device_build_table(TableSlot* d_table, size_t n_table, uint32_t log2s, const unsigned long long* h_keys, const unsigned char* h_ts, size_t n, bool* dup);
hipError_t device_permute_rows(const uint32_t* d_src, const uint32_t* h_id_of, size_t n, int W, uint32_t* d_dst);
bool finalize_runs_by_region(const FinalizeArgs& a);   // launch_finalize will take the regional kernel of bk_finalize_lean.hip (one genome file, whole dense planes, ...)
hipError_t raise_lds_limit(const void* fn, size_t lds);   // the dynamic-LDS limit of a kernel, raised once per process and device

}  // namespace bk
