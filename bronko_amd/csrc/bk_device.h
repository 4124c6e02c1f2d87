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
constexpr int kXcdPlanes = 8;        // MI355X: 8 XCDs, each with a private L2

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
BK_HD uint32_t phf_pos(uint64_t x, uint32_t pilot, uint32_t m) {
    uint32_t h = (uint32_t)((x * 0xD6E8FEB86659FD93ull) >> 32) ^ (pilot * 0x9E3779B1u);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return (uint32_t)(((uint64_t)h * m) >> 32);
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
    uint32_t p;      // first V row of the k-mer (bk_device.h "V rows"); unique per k-mer, so it also orders neighbours
    uint32_t valid;  // bit t: the k-mer owns a bucket at window position wstart+t (all ones for a reference k-mer;
                     // k = 31 "pseudo" k-mers own only the positions where a wrapped bucket id aliases, see bk_engine.cpp)
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
    uint32_t m;
    uint32_t log2nb;
};

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
    // the reference in reference order, for the diagonal walk of scan_count (staged in LDS when it fits):
    const uint32_t*  ref_words;   // 2-bit packed bases of all cells (nt_to_bits, 16 per word, LSB first), padded in front
                                  // (bk_kernels.h scan_ref_pad_words) and behind
    const uint32_t*  cell_codes;  // 2 bits per cell, same layout and padding: 0 = no k-mer of U starts here, 1 = one does and
                                  // it is canonical as written, 2 = it was reverse-complemented to become canonical
    const uint32_t*  cell_flags;  // 4 bits per cell q (8 cells per word): bit 0 = a k-mer starts at q, it is in U and it
                                  // is "clean" (see amb); bit 1 = it was reverse-complemented to become canonical;
                                  // bit 2 = id(q) == id(q-1) + 1; bit 3 = id(q+1) == id(q) + 1
    const uint32_t*  id_at;       // [total_cells] id of the k-mer starting at cell q (0xffffffff: none)
    uint32_t total_cells;
    uint32_t n_u;                 // |U| = number of ids
    uint32_t n_full;              // ids < n_full are reference k-mers (W V rows each); the rest are k = 31 pseudo k-mers
    uint64_t n_rows;              // V rows in all (8 counters each)
    const uint32_t*  prow_id;     // [n_rows - n_full*W] id of the pseudo k-mer that owns V row n_full*W + i
    const uint8_t*   prow_t;      //   ... and the window position t of that row
    const uint16_t*  pilots;   // [1 << log2nb]
    HalfView         lo, hi;
    const uint32_t*  slot_of;  // [n_u][W] window bucket (slot) of reference k-mer id at window position t
    const uint8_t*   amb;      // [n_u] 1 = "dirty": another reference k-mer (either strand) lies within Hamming
                               //       distance 2 of it, or it is within distance 2 of its own reverse complement
    const uint32_t*  estat_off;// [n_u + 1] per reference k-mer: its genomes, precomputed from the index alone
    const uint32_t*  estat;    //   (file << 1) | 1 if hits == W ("perfect"), | 0 otherwise ("variant")
    const TableSlot* table;    // [W][S]
    const uint32_t*  ent_off;  // [n_slots]
    const uint32_t*  ent_len;  // [n_slots]
    const DevEntry*  entries;  // [n_entries in window]
    uint64_t n_slots;
    uint32_t m;                // positions of kmer_pos (>= |U|)
    uint32_t log2nb;           // pilots has 1 << log2nb buckets
    uint32_t log2s;            // S = 1 << log2s positions per window sub-table
    int32_t  lo_bases;         // bases in the low half (k / 2)
    int32_t  k;
    int32_t  wstart;           // first wildcard position of the window (call.rs:1291-1300)
    int32_t  W;                // number of window buckets per k-mer = num_buckets_perfect (call.rs:1302)
    int32_t  n_files;
};

// Counter plane of one mate file, u64: [ E : 2 * n_u ][ V : n_rows * 8 ]
//   E[2id + rc]                       occurrences of the reference k-mer u = kmer_of[id] read as-is (rc=0) / as
//                                     its reverse complement (rc=1)
//   V[(row(id,t)*4 + b)*2 + rc]       occurrences of the non-reference k-mer "u with base b at window position
//                                     wstart+t".  A non-reference k-mer may neighbour several reference k-mers;
//                                     it is always counted under the smallest (t, row) -- a function of the k-mer
//                                     alone, so all its occurrences share one counter and it owns no other.
// V rows: a reference k-mer (id < n_full) has rows id*W + t; a pseudo k-mer has one row per window position at which
// it owns a bucket, numbered after those (NbEntry::p = its first row; row = p + popcount(valid below t)).
BK_HD uint64_t e_plane_len(uint32_t n_u) { return 2ull * n_u; }
BK_HD uint64_t v_plane_len(uint64_t n_rows) { return n_rows * 8ull; }
// (id, t) of V row `row`
BK_HD void row_owner(const struct IndexView& ix, uint64_t row, uint32_t& id, uint32_t& t);

BK_HD void row_owner(const IndexView& ix, uint64_t row, uint32_t& id, uint32_t& t) {
    const uint64_t full_rows = (uint64_t)ix.n_full * (uint32_t)ix.W;
    if (row < full_rows) { id = (uint32_t)(row / (uint32_t)ix.W); t = (uint32_t)(row % (uint32_t)ix.W); }
    else { id = ix.prow_id[row - full_rows]; t = ix.prow_t[row - full_rows]; }
}

}  // namespace bk
