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
constexpr int kMaxK = 31;          // consts.rs:4 MAX_KMER_SIZE
constexpr int kCountersPerSlot = 8; // 4 bases x 2 read orientations

// One position of a window sub-table (open addressing, linear probing).  16 B so that a probe is one
// global_load_dwordx4.  key = canonical k-mer with the sub-table's wildcard position zeroed.
struct alignas(16) TableSlot {
    uint64_t key;
    uint32_t slot;   // dense id of the window bucket: indexes counters, slot_key, slot_t, ent_off/ent_len
    uint32_t pad;
};

// One BucketInfo (build.rs:52-60) prepared for the vote of call.rs:1327-1384.
struct alignas(8) DevEntry {
    uint32_t cell;       // cell_offset(file, seq) + location + idx   (row of the pileup arrays)
    uint16_t file;
    uint8_t  idx;        // nuc_x
    uint8_t  canonical;
};

// Multiplicative hash of a masked canonical k-mer into a sub-table of 2^log2s positions.
BK_HD uint32_t hash_key(uint64_t key, uint32_t log2s) {
    return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - log2s));
}

// Everything the kernels need to know about the index; passed by value.
struct IndexView {
    const TableSlot* table;   // [W][S]
    const uint64_t*  slot_key; // [n_slots] masked canonical k-mer of the slot
    const uint8_t*   slot_t;   // [n_slots] window-relative wildcard position t (absolute = wstart + t)
    const uint32_t*  ent_off;  // [n_slots]
    const uint32_t*  ent_len;  // [n_slots]
    const DevEntry*  entries;  // [n_entries in window]
    uint64_t n_slots;
    uint32_t log2s;            // S = 1 << log2s positions per sub-table
    int32_t  k;
    int32_t  wstart;           // first wildcard position of the window (call.rs:1291-1300)
    int32_t  W;                // number of window buckets per k-mer = num_buckets_perfect (call.rs:1302)
    int32_t  n_files;
};

}  // namespace bk
