// bk_device.h -- structures shared by the host-side table builder and the gfx950 kernels.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define BK_HD __host__ __device__ __forceinline__
#else
#define BK_HD inline
#endif

namespace bk {

constexpr uint64_t kEmptyKey = ~0ull;
constexpr int kMaxK = 31;            // consts.rs:4 MAX_KMER_SIZE
constexpr int kCountersPerSlot = 8;  // 4 bases x 2 read orientations

// One position of a window sub-table (open addressing, linear probing).  16 B so that a probe is one
// global_load_dwordx4.  key = canonical k-mer with the sub-table's wildcard position zeroed.
struct alignas(16) TableSlot {
    uint64_t key;
    uint32_t slot;   // dense id of the window bucket: indexes the V counters, slot_key, slot_t, ent_off/ent_len
    uint32_t pad;
};

// One BucketInfo (build.rs:52-60) prepared for the vote of call.rs:1327-1384.
struct alignas(8) DevEntry {
    uint32_t cell;       // cell_offset(file, seq) + location + idx   (row of the pileup arrays)
    uint16_t file;
    uint8_t  idx;        // nuc_x
    uint8_t  canonical;
};

// Entry list of a window bucket + a copy of its first entry: 16 B, one global_load_dwordx4.
struct alignas(16) SlotRec {
    uint32_t off, len;   // entries[off .. off + len)
    DevEntry first;
};

// Per reference k-mer (id), everything finalize needs to replay map_kmers on a k-mer named by that id, in one 16-byte load.
// "simple": every one of its W window buckets holds exactly one BucketInfo -- its own single occurrence -- so the bucket at
// canonical position j is {cell = cell + j, file, idx = j, canonical = rc} and no table is read at all (every k-mer of a
// single genome without repeats).
struct alignas(16) IdRec {
    uint64_t kmer;     // canonical reference k-mer (= kmer_of[id])
    uint32_t cell;     // cell at which its first occurrence starts
    uint32_t flags;    // bit 0 dirty, bit 1 first occurrence reverse-complemented (= amb[id]); bit 2 simple; bits 16..31 file (simple only)
};
constexpr uint32_t kIdDirty = 1u, kIdRc = 2u, kIdSimple = 4u, kIdAllOwn = 8u, kIdOwnMirror = 16u;   // kIdAllOwn: every BucketInfo of its W buckets is one of id_own_files' (IndexView); kIdOwnMirror: those run against the reference

// Precomputed outcome of "reference k-mer `id` with base bb at position j (both in the orientation of the canonical k-mer)"
// for reference k-mers that are not clean (another reference k-mer form within Hamming distance 2, or occurrences in both
// orientations): is it another reference k-mer, a neighbour of which one (smallest (window position, NbEntry::p), the rule
// of the neighbour search), or nothing.  A function of the index alone, so Level 2 settles a read k-mer that differs from
// the reference in one base at such a cell with two loads (dirty_ix[id], then the answer) instead of a hash-table search.
// The read's orientation enters at run time: isrc = the canonical form is the reverse complement of the k-mer as read.
struct alignas(8) DirtyAns {
    uint32_t idx;    // kind 1: E counter 2 * id' (+ isrc); kind 2: V counter of direction 0 relative to the V part (+ row length
                     // when isrc ^ rcu); kind 3: pseudo k-mer counter relative to the V part (+ isrc)
    uint32_t meta;   // bits 0-1 kind (0 = touches nothing); bit 2 rcu (kind 2); bit 3 (kind 2): -1 on the next counter as well;
                     // bit 4: no answer was worked out for this reference k-mer (search instead); bit 5: its neighbours in the
                     // window sit at more than one position -- it touches several window buckets (finalize: the general path)
};
constexpr uint32_t kAnsNone = 16u, kAnsMulti = 32u;
// per-cell flags byte (Level 2): bits 0-1 = cell_codes symbol (0 no k-mer of U starts here, 1 canonical as written, 2 reverse-
// complemented), bit 2 = clean (cell_yf bit 0), bit 3 = cell_clean3, bit 4 = the cell stands in the orientation of its k-mer's
// first occurrence (the coordinates the answer table is laid out in)
constexpr uint32_t kCellClean = 4u, kCellClean3 = 8u, kCellFirstOri = 16u;
// bit 5 (round 5): no other reference k-mer form at Hamming distance exactly 2 or 3 from the cell's k-mer (forms one base away are
// allowed -- the other strains' variants of a few-genome index).  A read k-mer x with two differences from such a u can only equal or
// neighbour a form within distance 3 of u, i.e. one base from u, i.e. "u with one of x's two differences": two answer-table loads
// settle it (neither is a reference k-mer: x touches nothing) where the slow path took a membership test and two directory walks.
constexpr uint32_t kCellIso23 = 32u;
// The answer table (DirtyAns) in reference coordinates of the k-mer's first occurrence: offset o of the changed base from the
// k-mer's start along the reference, base b on the forward strand.  Laid out by diagonal, [id + o][b][o]: the k-mers of a read
// that cover one sequencing error have consecutive ids and falling offsets -- their answers are neighbours in memory (8 bytes
// apart) instead of k * 32 bytes apart, which is what Level 2's one random load per k-mer costs on a many-genome index.
BK_HD size_t ans_index(uint32_t id, uint32_t o, uint32_t b, int k) { return (((size_t)id + o) * 4u + b) * (size_t)k + o; }
BK_HD size_t ans_table_len(uint32_t n_full, int k) { return ((size_t)n_full + (size_t)k) * 4u * (size_t)k; }

// Seed tables of the scan (one per genome file; IndexView::seed_tab): 2^seed_log2 buckets of two entries,
//   entry = cell (27 bits) | rc << 27 | tag << 28      (~0 = free)
// for the reference k-mers of that file -- cell where one starts, rc = it was reverse-complemented to become canonical, tag =
// the low 4 bits of seed_hash(canonical k-mer), bucket = its top seed_log2 bits.  No key is stored: a candidate is verified
// against the reference itself (the scan has it in LDS).  A k-mer whose bucket was full when it came is not in the table; the
// scan then falls back on the perfect hash of U.
BK_HD uint32_t seed_hash(uint64_t x) {
    uint32_t h = (uint32_t)x * 0x9E3779B1u + (uint32_t)(x >> 32) * 0x85EBCA6Bu;
    return h ^ (h >> 15);
}
constexpr uint32_t kSeedCellBits = 27;

// Multiplicative hash into a table of 2^log2s positions.
BK_HD uint32_t hash_key(uint64_t key, uint32_t log2s) {
    return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - log2s));
}

// Perfect hash of the reference k-mer set U ("hash and displace" with per-bucket pilots): position of k-mer x
// in kmer_pos[m] = phf_pos(x, pilots[phf_bucket(x)], m).  The host searches, bucket by bucket (largest first),
// the smallest pilot that sends all keys of the bucket to free positions, so every u in U has a private
// position and a lookup is exactly two loads (pilot, key) with no probe chain -- no lane of a wave waits for
// another lane's collisions.
BK_HD uint32_t phf_bucket(uint64_t x, uint32_t log2nb) { return hash_key(x, log2nb); }
// The table is 2^log2p independent sub-tables of msub positions each (the host builds them on as many threads: 9 M keys of a
// 100-strain index took 8 s on one); the buckets of sub-table s are the ones whose index starts with s.
BK_HD uint32_t phf_pos(uint64_t x, uint32_t pilot, uint32_t msub, uint32_t log2nb, uint32_t log2p) {
    // (the pilot enters before the multiplication: two keys whose products agree in the upper half for one pilot part for another)
    uint32_t h = (uint32_t)(((x ^ ((uint64_t)pilot * 0x9E3779B97F4A7C15ull)) * 0xD6E8FEB86659FD93ull) >> 32);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    const uint32_t sub = log2p ? phf_bucket(x, log2nb) >> (log2nb - log2p) : 0u;
    return sub * msub + (uint32_t)(((uint64_t)h * msub) >> 32);
}

// Perfect-hash table entry of a reference k-mer: key, where it first occurs in the reference, and its id.
// Reference k-mers are numbered ("id") in order of first occurrence in reference order, so that the k-mers
// along a stretch of reference have consecutive ids; the id names the k-mer's E and V counters.
struct alignas(16) KmerPos {
    uint64_t key;      // canonical reference k-mer (kEmptyKey = free position)
    uint32_t refcell;  // cell of its first occurrence (cells = positions of all sequences concatenated in
                       // (file, seq) order = the pileup row order)
    uint32_t idflags;  // id | (1u << 31 if that occurrence was reverse-complemented to become canonical)
};
constexpr uint32_t kIdMask = 0x7fffffffu;

// A reference k-mer in a neighbour list: 16 B so that one candidate is one global_load_dwordx4.
struct alignas(16) NbEntry {
    uint64_t u;      // canonical reference k-mer
    uint32_t p;      // reference k-mer: its id; pseudo k-mer: n_full + its first pseudo V row.  Unique per k-mer, so it also
                     // orders neighbours
    uint32_t valid;  // bit t (t < 31): the k-mer owns a bucket at window position wstart+t (all of them for a reference k-mer;
                     // k = 31 "pseudo" k-mers own only the positions where a wrapped bucket id aliases, see bk_engine.cpp);
                     // bit 31: the reference k-mer's first occurrence was reverse-complemented to become canonical
};

// Directory entry of a half-key: the reference k-mers sharing that half are cand[off .. off + cnt).
struct alignas(16) HalfDir {
    uint32_t key;    // the half-key itself (<= 16 bases = 32 bits)
    uint32_t off;
    uint32_t cnt;    // 0 = free position
    uint32_t pad;
};

// One half (low or high) of the neighbour search: perfect hash over the distinct half-keys -> HalfDir -> list.
struct HalfView {
    const uint16_t* pilots;   // [1 << log2nb]
    const HalfDir*  dir;      // [m]
    const NbEntry*  cand;     // reference k-mers sorted by this half
    uint32_t m;               // positions per sub-table (phf_pos)
    uint32_t log2nb;
    uint32_t log2p;           // 2^log2p sub-tables
    // Presence of a half-key among the reference k-mers' (round 6): one bit per possible half where the half is at most 24 bits
    // (k <= 25: exact), else one bit at hash_key(half) of 2^bits_log2 (never a false "absent").  A k-mer can be a reference k-mer or
    // one base from one only if one of its halves IS a reference k-mer's half (pigeonhole): two bit tests, both "absent", settle
    // "touches nothing" before the six loads of the membership test and the two directory walks (Level 2's slow path).
    const uint32_t* bits;     // null: no filter
    uint32_t bits_log2;
    uint32_t bits_exact;
};
BK_HD uint32_t half_bit_index(uint64_t half, uint32_t bits_log2, uint32_t exact) { return exact ? (uint32_t)half : hash_key(half, bits_log2); }

// Everything the kernels need to know about the index; passed by value.
//
// Views of the reference k-mer set U (distinct canonical reference k-mers that own a window bucket):
//  * kmer_pos / pilots : perfect hash, "is this read k-mer a reference k-mer, and which one" in two loads.
//  * lo / hi           : every u in U listed under its low half (lo_bases bases) resp. high half, the half-keys
//                        perfect-hashed.  Two k-mers at Hamming distance 1 agree on one half (pigeonhole), so
//                        the two lists of a k-mer contain every reference k-mer it can vote for; a lookup is
//                        pilot -> directory entry -> candidates, three dependent loads whatever the table load.
//  * table (+ slot_key, slot_t, ent_off, ent_len, entries): the index itself, window buckets keyed by
//    (wildcard position, masked k-mer) with their BucketInfo lists; used by finalize to replay map_kmers.
struct IndexView {
    const KmerPos*   kmer_pos; // [m] perfect-hash table of U (membership test, diagonal seeding)
    const uint64_t*  kmer_of;  // [n_u] id -> canonical k-mer
    const IdRec*     id_rec;   // [n_u] id -> k-mer, first cell, flags (see IdRec)
    const DirtyAns*  dirty_ans;// [n_full + k][4][k] (see DirtyAns, ans_index); entries of k-mers all of whose cells are clean are
                               // never read; null when the table was not built (index too large)
    const uint8_t*   cell_flags;// [total_cells] per-cell flags byte (see kCellClean)
    // the reference in reference order, for the diagonal walk of scan_count (staged in LDS when it fits):
    const uint32_t*  ref_words;   // 2-bit packed bases of all cells (nt_to_bits, 16 per word, LSB first), padded in front
                                  // (bk_kernels.h scan_ref_pad_words) and behind
    const uint32_t*  cell_codes;  // 2 bits per cell, same layout and padding: 0 = no k-mer of U starts here, 1 = one does and
                                  // it is canonical as written, 2 = it was reverse-complemented to become canonical
    const uint32_t*  cell_has;    // 1 bit per cell (32 per word, 2 words of front padding, 3 behind): a k-mer of U starts here
    const uint32_t*  cell_clean;  // same layout: ... and it is "clean" (bit 0 of cell_yf)
    const uint32_t*  cell_clean3; // same layout: ... and no other reference k-mer, on either strand, lies within Hamming distance 3
                                  // of it, nor its own reverse complement (all zero when the index is too large to work that out)
    const uint32_t*  cell_yf;     // 2 bits per cell q, same layout and padding, for a walk along the reference: bit 0 = the k-mer
                                  // that starts at q is in U and "clean" (see amb); bit 1 = id(q) == id(q-1) + 1
    const uint32_t*  cell_yr;     // ... for a walk against it: bit 0 the same; bit 1 = id(q) == id(q+1) - 1
    const uint32_t*  id_at;       // [total_cells] id of the k-mer starting at cell q (0xffffffff: none)
    const uint32_t*  cell_fast;   // 1 bit per cell, layout of cell_has: the k-mer that starts at q is in U, "clean", and its id is
                                  // q + cell_blk[q >> 6].x (modulo 2^32) -- what the scan needs to count an isolated mismatch on the spot
    const uint32_t*  cell_nat;    // [(total_cells + k) * 3] per reference position q and alternative a (read base XOR reference base, - 1):
                                  // bit o = the k-mer that starts at q - o is in U, stands in its first occurrence's orientation, has the
                                  // id q - o + cell_blk[..].x, and with that other base at q it takes its OWN V row (clean, or its DirtyAns
                                  // says so) -- cell_fast per (position, offset, base) for cells that are not clean; the k-mers that cover
                                  // one mismatch all ask the same word.  Null when no answer table was built.
    const uint32_t*  cell_natrow; // [total_cells + k] without touch lists (dense planes): the id + o that cell_nat's bits of position q stand for -- the
                                  // k-mers over q whose id is NOT q - o + that constant are left out instead of those that leave cell_blk's
                                  // sequence; null with touch lists (there cell_nat promises id = cell + cell_blk[..].x)
    const uint2*     seed_tab;    // [n_files << seed_log2] the scan's seed tables (see seed_hash), or null (index too large)
    uint32_t         seed_log2;
    const uint2*     cell_blk;    // per block of 64 cells (two entries of padding behind): x = the most common id_at[q] - q among the block's cells
                                  // that stand in their k-mer's first orientation; y = the first cell >= 64 * block at which no k-mer of U starts (sequence tails)
    uint32_t total_cells;
    uint32_t n_u;                 // |U| = number of ids
    uint32_t n_full;              // ids < n_full are reference k-mers (W V rows each); the rest are k = 31 pseudo k-mers
    uint64_t n_prows;             // V rows of the pseudo k-mers (8 counters each)
    const uint32_t*  prow_id;     // [n_prows] id of the pseudo k-mer that owns pseudo V row i
    const uint8_t*   prow_t;      //   ... and the window position t of that row
    int32_t  v_omin, v_span;      // layout of the reference k-mers' V counters (see v_row_base)
    uint64_t v_off;               // element of the counter plane at which the V part starts (see counter_plane_layout)
    const uint16_t*  pilots;   // [1 << log2nb]
    HalfView         lo, hi;
    const uint32_t*  slot_of;  // [n_u][W] window bucket (slot) of reference k-mer id at window position t
    const SlotRec*   slot_rec; // [n_full][W] the same bucket's entry list, with its first BucketInfo inline (one load for the
                               //             common single-entry bucket)
    const uint4*     ent_files;  // [n_slots] / slot_files [n_full][W]: bit f = the bucket holds a BucketInfo of genome file f -- exactly one, the
    const uint4*     slot_files; //   entries sorted by file, so file f's is entry number popcount(bits below f); all zero = not such a bucket
                                 //   (or more than 128 files, or W == 1: both null): look at the entries.  bk_params.pileup_selected_only.
    const uint4*     id_own_files; // [n_full] bit f = in each of the k-mer's W buckets genome f's one BucketInfo is the k-mer's own occurrence in f:
                                   //   that of bucket t = that of bucket 0 with cell + t, idx + t (kIdOwnMirror: - t); null with slot_files
    const uint32_t*  id_rest_off;  // [n_full + 1] / id_rest: per k-mer with own files, the BucketInfos of its buckets that are not its own occurrences
    const uint32_t*  id_rest;      //   (indices into entries); null with slot_files
    const uint16_t*  cell_file;    // [total_cells] genome file of each cell; null with slot_files
    const uint4*     estat_files;  // [n_full][2] estat as bitmaps: genomes in which the k-mer is perfect / a variant; null with slot_files
    const uint32_t*  slot_alias;   // [n_slots / 32 + 1] bit s = slot s is reached through an alias key (k = 31: the other exact rank that wraps onto its
                                   //   bucket's id), i.e. by k-mers that match a pseudo k-mer; null: there is none
    uint32_t         gather_ok;    // the votes of a genome's BucketInfos can be gathered cell by cell (bk_gather.hip): every window bucket has one
                                   //   key and holds each occurrence of its k-mers exactly once, every reference k-mer that needs one has an answer row
    const uint8_t*   amb;      // [n_u] bit 1 = the k-mer's first occurrence was reverse-complemented to become canonical; bit 0 = "dirty": another reference k-mer (either strand) lies within Hamming
                               //       distance 2 of it, or it is within distance 2 of its own reverse complement
    const uint32_t*  estat_off;// [n_u + 1] per reference k-mer: its genomes, precomputed from the index alone
    const uint32_t*  estat;    //   (file << 1) | 1 if hits == W ("perfect"), | 0 otherwise ("variant")
    const TableSlot* table;    // [W][S]
    const uint32_t*  ent_off;  // [n_slots]
    const uint32_t*  ent_len;  // [n_slots]
    const DevEntry*  entries;  // [n_entries in window]
    uint64_t n_slots;
    uint32_t m;                // positions of kmer_pos per sub-table (phf_pos)
    uint32_t log2nb;           // pilots has 1 << log2nb buckets
    uint32_t log2p;            // kmer_pos is 2^log2p sub-tables
    uint32_t log2s;            // S = 1 << log2s positions per window sub-table
    int32_t  lo_bases;         // bases in the low half (k / 2)
    int32_t  k;
    int32_t  wstart;           // first wildcard position of the window (call.rs:1291-1300)
    int32_t  W;                // number of window buckets per k-mer = num_buckets_perfect (call.rs:1302)
    int32_t  n_files;
};

// Counter plane of one mate file, u64 (all arithmetic wraps modulo 2^64, so planes of read shards simply add):
//   [ E : 2 * n_u ][ V of the reference k-mers : v_real_len ][ V of the pseudo k-mers : n_prows * 8 ]
//   E[2id + rc]      occurrences of the reference k-mer u = kmer_of[id] read as-is (rc=0) / as its reverse complement (rc=1)
//   V                occurrences of the non-reference k-mers "u with another base at one position".  A non-reference k-mer may
//                    neighbour several reference k-mers; it is always counted under the smallest (window position, NbEntry::p)
//                    -- a function of the k-mer alone, so all its occurrences share one counter and it owns no other.
//
// V of the reference k-mers.  A sequencing error at one reference base turns the k reference k-mers that cover it --
// consecutive ids along the reference -- into non-reference k-mers, each differing from "its" reference k-mer at a different
// offset o from the k-mer's start along the reference (o = j for a k-mer that is canonical as written, k-1-j for one that
// was reverse-complemented; j = position in the canonical k-mer).  id + o is the same for all of them.  So the plane is a set
// of rows (q = id + o - omin, alternative a = which of the three OTHER bases stands there -- (base XOR reference base) - 1, the
// same on either strand (complementing both leaves the XOR): the reference base itself can never be counted here, so a
// position has 3 x 2 rows, not 4 x 2 --, direction d of the read: 0 along / 1 against the reference), each a DIFFERENCE ARRAY over o: the number of occurrences of "(q - o, o, b) read in direction d" is the prefix
// sum row[0] + ... + row[o - omin] (b = reference base of k-mer q - o at offset o, XOR (a + 1)).  A read with one error adds +1 at the first offset its k-mers cover and -1 after the last
// -- two atomics for up to k k-mers; a single k-mer is +1 at o, -1 at o + 1.  finalize takes the prefix sums and derives each
// k-mer (its canonical form and orientation included) from the row coordinates.
BK_HD uint64_t e_plane_len(uint32_t n_u) { return 2ull * n_u; }
BK_HD int v_layout_omin(int k, int wstart, int W, bool all_offsets) { if (all_offsets) return 0; const int m = k - wstart - W; return wstart < m ? wstart : m; }
BK_HD int v_layout_span(int k, int wstart, int W, bool all_offsets) {
    if (W <= 0) return 0;
    if (all_offsets) return k;
    const int hi1 = wstart + W - 1, hi2 = k - 1 - wstart;
    return (hi1 > hi2 ? hi1 : hi2) - v_layout_omin(k, wstart, W, false) + 1;
}
constexpr uint32_t kVRowsPerPos = 6;   // 3 alternative bases x 2 read directions
BK_HD uint64_t v_real_rows(uint32_t n_full, int span) { return span > 0 ? ((uint64_t)n_full + (uint32_t)span) * kVRowsPerPos : 0ull; }
BK_HD uint64_t v_real_len(uint32_t n_full, int span) { return v_real_rows(n_full, span) * (uint64_t)(span + 1); }
// which alternative: base b where the reference has r (both on the same strand, either one); 0..2 (b == r is never counted)
BK_HD uint32_t v_alt(uint32_t b, uint32_t r) { return ((b ^ r) & 3u) - 1u; }
// file bitmaps (IndexView::ent_files): is file f there, and which entry of the bucket is its
BK_HD bool files_any(const uint4& b) { return (b.x | b.y | b.z | b.w) != 0u; }
BK_HD bool files_has(const uint4& b, uint32_t f) { const uint32_t w = f < 32u ? b.x : f < 64u ? b.y : f < 96u ? b.z : b.w; return (w >> (f & 31u)) & 1u; }
BK_HD uint32_t files_rank(const uint4& b, uint32_t f) {
    const uint32_t m = (1u << (f & 31u)) - 1u;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t c0 = __popc(b.x), c1 = __popc(b.y), c2 = __popc(b.z);
    return f < 32u ? __popc(b.x & m) : f < 64u ? c0 + __popc(b.y & m) : f < 96u ? c0 + c1 + __popc(b.z & m) : c0 + c1 + c2 + __popc(b.w & m);
#else
    const uint32_t c0 = (uint32_t)__builtin_popcount(b.x), c1 = (uint32_t)__builtin_popcount(b.y), c2 = (uint32_t)__builtin_popcount(b.z);
    return f < 32u ? (uint32_t)__builtin_popcount(b.x & m) : f < 64u ? c0 + (uint32_t)__builtin_popcount(b.y & m)
         : f < 96u ? c0 + c1 + (uint32_t)__builtin_popcount(b.z & m) : c0 + c1 + c2 + (uint32_t)__builtin_popcount(b.w & m);
#endif
}
BK_HD uint32_t v_row_index(uint32_t q, uint32_t alt, uint32_t d) { return (q * 3u + alt) * 2u + d; }
// first counter of row (q, alt, d); the row has span + 1 counters (the last only ever receives a -1)
BK_HD uint64_t v_row_base(uint32_t q, uint32_t alt, uint32_t d, int span) { return (((uint64_t)q * 3ull + alt) * 2ull + d) * (uint64_t)(span + 1); }
BK_HD uint64_t v_plane_len(uint32_t n_full, int span, uint64_t n_prows) { return v_real_len(n_full, span) + n_prows * 8ull; }
// The whole plane: [ E | padding | V rows | pseudo counters | padding ].  The V part starts at a multiple of the row length
// and the total is a multiple of kMaxShards row lengths, so that cutting the plane into n equal parts (n dividing kMaxShards:
// what a reduce-scatter over n ranks leaves on each) never cuts a row.
constexpr uint32_t kMaxShards = 64;
BK_HD void counter_plane_layout(uint32_t n_u, uint32_t n_full, int span, uint64_t n_prows, uint64_t& v_off, uint64_t& total) {
    const uint64_t rl = (uint64_t)(span > 0 ? span + 1 : 1);
    v_off = (e_plane_len(n_u) + rl - 1) / rl * rl;
    const uint64_t unit = rl * kMaxShards;
    total = (v_off + v_plane_len(n_full, span, n_prows) + unit - 1) / unit * unit;
}

}  // namespace bk
