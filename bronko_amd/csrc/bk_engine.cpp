// bk_engine.cpp -- host side of the C ABI declared in include/bronko_hip.h.
//
// Builds the device-resident window-bucket table from a decoded BronkoIndex, owns the HBM buffers
// (table, counter planes, pileups) and sequences the kernels of bk_kernels.hip on one HIP stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <atomic>
#include <unordered_map>
#include <vector>

#include "../../include/bronko_hip.h"
#include "../host/lcb.hpp"
#include "bk_device.h"
#include "bk_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define BK_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return fail(BK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool owned = true;
    ~DevBuf() { if (p && owned) (void)hipFree(p); }
    void alias(const DevBuf& o) { if (p && owned) (void)hipFree(p); p = o.p; n = o.n; owned = false; }   // a view of another engine's table
    hipError_t alloc(size_t count) {
        if (p && owned) (void)hipFree(p);
        p = nullptr; owned = true;
        n = count;
        return hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T));
    }
    template <class A>
    hipError_t upload(const std::vector<T, A>& h) {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

struct TimedSpan { hipEvent_t a, b; int kind; };

// fn(begin, end) over [0, n) on up to hardware_concurrency() threads (capped at 256): host-side table construction only
template <typename F>
void parallel_for(size_t n, F&& fn) {
    unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 256u);
    if (n < (size_t)nt * 1024) { fn((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (unsigned i = 0; i < nt; i++) {
        const size_t b = std::min(n, i * per), en = std::min(n, b + per);
        if (b < en) th.emplace_back([&fn, b, en] { fn(b, en); });
    }
    for (auto& t : th) t.join();
}

// A host array whose elements are not initialised by its constructor (the GB-sized tables of a many-genome index: a serial
// value-initialisation was 0.3 s each; they are filled by parallel_for)
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(p)) U;
        else ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
    }
};
template <class T> using HostVec = std::vector<T, NoInitAlloc<T>>;
template <class T>
HostVec<T> filled(size_t n, const T& v) {
    HostVec<T> a(n);
    parallel_for(n, [&](size_t i0, size_t i1) { std::fill(a.begin() + (ptrdiff_t)i0, a.begin() + (ptrdiff_t)i1, v); });
    return a;
}

// Testing / measurement aids exist only in the -DBK_TESTING build (libbronko_hip_testing.so, loaded by the tests that force a
// path and by the profiling tools); the release library reads no environment variable.
#ifdef BK_TESTING
const char* test_env(const char* name) { return getenv(name); }
#else
const char* test_env(const char*) { return nullptr; }
#endif

// std::sort on `nt` host threads: sorted chunks, then pairwise merges level by level
template <typename T, typename Cmp>
void parallel_sort(std::vector<T>& v, Cmp cmp, unsigned nt) {
    if (nt < 2 || v.size() < (size_t)nt * 65536) { std::sort(v.begin(), v.end(), cmp); return; }
    std::vector<size_t> cut(nt + 1);
    for (unsigned i = 0; i <= nt; i++) cut[i] = v.size() * i / nt;
    {
        std::vector<std::thread> th;
        for (unsigned i = 0; i < nt; i++) th.emplace_back([&, i] { std::sort(v.begin() + cut[i], v.begin() + cut[i + 1], cmp); });
        for (auto& t : th) t.join();
    }
    for (unsigned step = 1; step < nt; step *= 2) {
        std::vector<std::thread> th;
        for (unsigned i = 0; i + step < nt; i += 2 * step)
            th.emplace_back([&, i, step] { std::inplace_merge(v.begin() + cut[i], v.begin() + cut[i + step], v.begin() + cut[std::min(i + 2 * step, nt)], cmp); });
        for (auto& t : th) t.join();
    }
}

// BK_CREATE_TIMING=1 (testing build): wall-clock of the phases of bk_engine_create on stderr (host-side table construction)
struct PhaseClock {
    bool on = test_env("BK_CREATE_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[bk_engine_create] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

}  // namespace

namespace {

// assign_buckets (lcb.rs:1-45) is the 1-based lexicographic rank of (V, position) where V is the k-mer with the
// wildcard position set to A: ranks are ordered by V, then by position, and V contributes one rank per A it contains
// (verified exhaustively for small k against the reference's known answers).  In exact arithmetic the rank of a k = 31
// bucket reaches 31 * 4^30 ~ 1.94 * 2^64, and the reference keeps it modulo 2^64: two different (position, k-mer)
// pairs whose ranks differ by 2^64 share a bucket.  rank128 / unrank128 are the exact map and its inverse.
using u128 = unsigned __int128;

u128 rank128(uint64_t v /* wildcard position already A */, int pos, int k) {
    u128 cum = 0;   // sum of (number of A digits) over all k-digit strings < v
    int a_pre = 0;
    for (int i = 0; i < k; i++) {
        const int d = (int)((v >> (2 * (k - 1 - i))) & 3);
        const int rest = k - 1 - i;
        const u128 pw = (u128)1 << (2 * rest);                  // 4^rest strings below each smaller digit
        const u128 free_a = rest ? (u128)rest * (pw >> 2) : 0;  // A digits inside the free suffix, summed over them
        for (int x = 0; x < d; x++) cum += (u128)(a_pre + (x == 0)) * pw + free_a;
        a_pre += d == 0;
    }
    int before = 0;
    for (int i = 0; i < pos; i++) before += ((v >> (2 * (k - 1 - i))) & 3) == 0;
    return cum + (u128)before + 1;
}

bool unrank128(u128 r1, int k, uint64_t* v_out, int* pos_out) {
    if (r1 == 0) return false;
    u128 r = r1 - 1;
    uint64_t v = 0;
    int a_pre = 0;
    for (int i = 0; i < k; i++) {
        const int rest = k - 1 - i;
        const u128 pw = (u128)1 << (2 * rest);
        const u128 free_a = rest ? (u128)rest * (pw >> 2) : 0;
        int x = 0;
        for (; x < 4; x++) {
            const u128 c = (u128)(a_pre + (x == 0)) * pw + free_a;
            if (r < c) break;
            r -= c;
        }
        if (x == 4) return false;   // rank beyond k * 4^(k-1)
        v |= (uint64_t)x << (2 * rest);
        a_pre += x == 0;
    }
    // r-th A position of v
    for (int i = 0; i < k; i++)
        if (((v >> (2 * (k - 1 - i))) & 3) == 0) { if (r == 0) { *v_out = v; *pos_out = i; return true; } r -= 1; }
    return false;
}

// Perfect hash of distinct keys (bk_device.h phf_*): buckets of ~4 keys, largest first, smallest free pilot.  Large key sets are
// cut into 2^log2p sub-tables by the leading bits of the bucket index and built on as many host threads; every sub-table has
// msub positions.  On success pos[i] is the position of keys[i] in a table of (msub << log2p) positions.
bool build_phf(const std::vector<uint64_t>& keys, std::vector<uint16_t>& pilots, uint32_t& log2nb, uint32_t& msub_out, uint32_t& log2p_out,
               std::vector<uint32_t>& pos) {
    const size_t n = keys.size();
    uint32_t log2nb0 = 0;
    while ((4ull << log2nb0) < n) log2nb0++;
    pos.assign(n, 0);
    // A construction can fail only when a bucket finds no pilot among 65536: first the tables grow (msub), then the buckets
    // shrink (twice as many, half the keys each) -- the device reads all sizes from the view, so any outcome is a valid
    // table; an index is never refused because of its hash.
    for (uint32_t extra = 0; extra <= 6; extra++) {
        log2nb = log2nb0 + extra;
        if (log2nb > 30) break;
        const uint32_t log2p = n >= (1u << 20) && log2nb >= 10 ? 5u : 0u;
        const size_t P = (size_t)1 << log2p;
        const size_t nb = (size_t)1 << log2nb, nb_sub = nb >> log2p;
        // keys by bucket (counting sort), buckets by sub-table
        std::vector<uint32_t> b_of(n), b_cnt(nb + 1, 0u), by_bucket(n);
        for (size_t i = 0; i < n; i++) { b_of[i] = bk::phf_bucket(keys[i], log2nb); b_cnt[b_of[i] + 1]++; }
        for (size_t x = 0; x < nb; x++) b_cnt[x + 1] += b_cnt[x];
        { std::vector<uint32_t> at(b_cnt.begin(), b_cnt.end() - 1); for (size_t i = 0; i < n; i++) by_bucket[at[b_of[i]]++] = (uint32_t)i; }
        uint64_t max_sub = 0;
        for (size_t sp = 0; sp < P; sp++) max_sub = std::max<uint64_t>(max_sub, b_cnt[(sp + 1) * nb_sub] - b_cnt[sp * nb_sub]);
        uint64_t msub = std::max<uint64_t>(64, (uint64_t)((double)max_sub / 0.97) + 1);
        for (int attempt = 0; attempt <= 8 && (msub << log2p) < (1ull << 31); attempt++, msub += msub / 8 + 1) {
            pilots.assign(nb, 0);
            std::atomic<bool> ok{true};
            auto build_sub = [&](size_t sp) {
                std::vector<uint32_t> order(nb_sub);
                for (size_t x = 0; x < nb_sub; x++) order[x] = (uint32_t)(sp * nb_sub + x);
                std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return b_cnt[x + 1] - b_cnt[x] > b_cnt[y + 1] - b_cnt[y]; });
                std::vector<uint8_t> used(msub, 0);
                std::vector<uint32_t> trial;
                for (uint32_t bkt : order) {
                    const uint32_t m0 = b_cnt[bkt], m1 = b_cnt[bkt + 1];
                    if (m0 == m1) break;
                    uint32_t pilot = 0;
                    for (; pilot < 65536; pilot++) {
                        trial.clear();
                        bool good = true;
                        for (uint32_t q = m0; q < m1; q++) {
                            const uint32_t p = bk::phf_pos(keys[by_bucket[q]], pilot, (uint32_t)msub, log2nb, 0u);   // position inside the sub-table
                            if (used[p] || std::find(trial.begin(), trial.end(), p) != trial.end()) { good = false; break; }
                            trial.push_back(p);
                        }
                        if (good) break;
                    }
                    if (pilot == 65536) { ok = false; return; }
                    pilots[bkt] = (uint16_t)pilot;
                    for (uint32_t q = m0; q < m1; q++) { pos[by_bucket[q]] = (uint32_t)(sp * msub) + trial[q - m0]; used[trial[q - m0]] = 1; }
                }
            };
            if (P == 1) build_sub(0);
            else {
                std::vector<std::thread> th;
                for (size_t sp = 0; sp < P; sp++) th.emplace_back(build_sub, sp);
                for (auto& t : th) t.join();
            }
            if (ok) { msub_out = (uint32_t)msub; log2p_out = log2p; return true; }
        }
    }
    return false;
}

}  // namespace

struct bk_engine {
    bk_params params{};
    // engines that share this one's index tables (itself and its forks, alive): with samples in flight next to each other the
    // binned scan leaves a quarter of the CUs to the siblings' small kernels (push_device)
    std::shared_ptr<std::atomic<int>> family = std::make_shared<std::atomic<int>>(1);
    int k = 0, wstart = 0, W = 0, n_files = 0;
    uint64_t total_cells = 0, n_slots = 0;
    uint32_t log2s = 4, log2nb = 0, log2p = 0, m = 1, n_u = 0, n_full = 0, n_lds_bins = 0;
    uint64_t n_prows = 0;  // V rows of the pseudo k-mers (bk_device.h)
    int v_omin = 0, v_span = 0;
    uint64_t v_off = 0, plane_len = 0;      // counter_plane_layout (bk_device.h)
    DevBuf<unsigned long long> shard_sums;  // sharded finalize: [stats 2*n_files*3 | present 2*n_files | kstats 8 | transport flag]
    // sharded finalize, transport of the planes (bk_shard_transport / bk_shard_received): the packed plane of the mate file being
    // exchanged, the part the reduce-scatter leaves here, and per mate file the received part widened to u64 again -- what
    // bk_sample_finalize_shard maps (the plane itself stays as the scans left it)
    DevBuf<unsigned char> xport_send, xport_recv;
    DevBuf<unsigned long long> reduced[2];
    DevBuf<unsigned long long> xport_flag;  // [0] a packer of this sample met a counter too large for its width, [1] sticky copy after
                                            // the ranks' sums were merged, [2..3] bk_shard_measure: max E count, max |V element|
    bool xport_ever = false;                // some sample of this engine went through bk_shard_transport (bk_sample_download then looks at the flag)
    int reduced_shards[2] = {0, 0};         // > 0: reduced[m] holds part `reduced_shard[m]` of that many for the current sample
    int reduced_shard[2] = {0, 0};
    DevBuf<uint32_t> prow_id;
    DevBuf<uint8_t> prow_t;
    bool ref_in_lds = false;
    int lo_bases = 0, n_cus = 256;
    int device = 0;

    DevBuf<bk::KmerPos> kmer_pos;
    DevBuf<bk::IndexView> d_view;   // device copy of view()
    DevBuf<uint64_t> kmer_of;
    DevBuf<bk::IdRec> id_rec;
    DevBuf<bk::DirtyAns> dirty_ans;
    DevBuf<uint8_t> cell_flags;
    DevBuf<uint32_t> ref_words, cell_codes, cell_has, cell_clean, cell_clean3, cell_yf, cell_yr, id_at, cell_fast, cell_nat, cell_natrow;
    DevBuf<uint2> cell_blk, seed_tab, seed_tab2;
    uint32_t seed_log2 = 0, seed2_log2 = 0;
    DevBuf<uint32_t> rc_words;              // the reference read backwards and complemented (scan_items_kernel: reads against the reference)
    struct HalfBufs { DevBuf<uint16_t> pilots; DevBuf<bk::HalfDir> dir; DevBuf<bk::NbEntry> cand; uint32_t m = 1, log2nb = 0, log2p = 0; DevBuf<uint32_t> bits; uint32_t bits_log2 = 0, bits_exact = 0; } half_lo, half_hi;
    DevBuf<unsigned int> deferred, n_deferred, deferred_mask;
    DevBuf<unsigned long long> deferred_n;   // dense planes: the deferred k-mers' counts (K2a zeroes the counters it reads)
    DevBuf<unsigned int> fin_partials;      // per-workgroup finalize tallies (small genome sets only)
    DevBuf<unsigned long long> ktab_keys;   // full_kmer_stats: open-addressing table of non-index-touching k-mers
    DevBuf<unsigned int> ktab_cnt;
    DevBuf<unsigned long long> ktab_out;    // [2 mates][2] distinct, kept  + [4] overflow flag + [8 ..] tallies of new keys
    uint32_t ktab_log2 = 0;                 // current capacity (grows with the sample: ensure_ktab_room)
    unsigned long long* h_fill = nullptr;   // pinned copy of the tallies, refreshed after every push
    hipEvent_t fill_ev = nullptr;
    bool fill_pending = false;              // a copy of the tallies is in flight / unread
    uint64_t fill_known = 0, fill_unknown_upper = 0;   // keys in the table at the last reading; k-mers pushed since (upper bound on new keys)
    std::vector<std::pair<unsigned long long*, unsigned int*>> ktab_old;   // outgrown tables, freed at the next sample / destroy
    DevBuf<unsigned long long> xchg_keys, xchg_cursors;   // bk_kmer_table_partition: the table's entries grouped by owner rank
    DevBuf<unsigned int> xchg_cnt;
    bool ktab_exchanged = false;            // bk_kmer_table_replace was called in this sample
    DevBuf<uint32_t> slot_of, estat_off, estat;
    DevBuf<bk::SlotRec> slot_rec;
    DevBuf<uint4> ent_files, slot_files, id_own_files, estat_files;
    DevBuf<uint16_t> cell_file;
    DevBuf<uint32_t> slot_alias;
    DevBuf<uint64_t> merged_slots;          // [n_merged_slots][2]: slot | window position << 32, the slot's key
    uint32_t n_merged_slots = 0;
    bool gather_ok = false;                 // IndexView::gather_ok
    // gathered votes (bk_gather.hip): this engine's voting pass is gather_votes_kernel (sparse planes of a many-genome index)
    bool gather_mode = false;
    DevBuf<unsigned int> row_bits;              // one bit per V row of the reference k-mers: touched by the sample (set by prefix_rows_kernel for voter_table_kernel)
    DevBuf<uint32_t> vote_tab;                  // [n_full][W][8] the voters of every (reference k-mer, window position) of the sample (bk_gather.hip; every genome's rows)
    DevBuf<unsigned long long> alias_hits[2];   // per mate file: the deferred k-mers that reach a bucket through an alias key
    DevBuf<unsigned int> n_alias_hits;          // [2]
    static constexpr unsigned int kAliasCap = 1u << 20;
    DevBuf<int> last_sel;                   // pileup_selected_only with gathered votes: the genome whose rows the previous sample wrote (-1: none) -- all that
                                            // the next sample has to zero
    DevBuf<uint32_t> id_rest_off, id_rest;
    DevBuf<uint8_t> amb;
    DevBuf<uint16_t> pilots;
    DevBuf<unsigned int> slabs;             // [n_cus][n_lds_bins] workgroup histograms of the last scan launch (scan_count_kernel only)
    // the binned scan (bk_scan_items.hip; dense planes with the window's reference in LDS): the scan workgroups' items and where
    // each bin's segment starts, the overflow list and its fill
    bool use_items = false;
    bk::ItemGeom ig{};
    DevBuf<unsigned short> items, item_tab;
    DevBuf<unsigned short> item_gext;       // [items_max_grid][bins][kItemGCap] the bins' extensions in device memory
    DevBuf<unsigned int> ov;
    DevBuf<unsigned long long> ov_n;
    uint32_t ov_par = 0;                    // parity of the next scan_items launch (which of the two overflow counts it appends to)
    DevBuf<unsigned int> lean_e_list, lean_n_list;   // bk_finalize_lean.hip: the reference k-mers finalize_ecell_kernel leaves to finalize_exact_kernel
    bool v_clean[2] = {false, false};       // the V part of the mate file's plane is known to be all zero (dense planes between samples)
    // The V items of a mate file's first scan launch are not added to the plane: they wait (`pending`) for the regional finalize,
    // which takes its counts from them (FinalizeArgs::f_items) -- the whole story for a mate file whose reads are one launch.  A
    // second launch into the engine's item buffers first sends them to the plane after all (flush_pending_items: bin_count_kernel,
    // V bins only), and the mate file's later launches go straight there as before.
    bool fuse_ok = false;                   // this engine's index, planes and parameters admit it (alloc_sample_state)
    struct PendingItems { bool on = false; int mate = 0; bk::BinArgs b{}; } pending;
    bool fuse_off[2] = {false, false};      // this sample's mate file has had a second launch: no more waiting
    bool touch_used[2] = {false, false};    // Level 2 set bits in fuse_touch[m] that no regional finalize has cleared
    DevBuf<unsigned int> fuse_touch[2];     // a bit per V row Level 2 wrote to while the launch's items wait
    int item_v_mode = -1;                   // testing aid (BK_ITEM_V_MODE): force BinArgs::v_mode
    DevBuf<unsigned int> n_bits, n_any;     // scan -> Level 2: one bit per k-mer of each record of a launch / per record (bk_kernels.h ScanArgs): the N runs; all zero between launches
    DevBuf<unsigned int> l2_bits;           // Level 2's first pass -> its second: the k-mers looked at one by one, same layout
    DevBuf<unsigned int> l2_any;            // ... one bit per record: its row has bits
    DevBuf<unsigned int> l2_plan;           // one word: the workgroups of level2_kernel that work (ScanArgs::l2_plan)
    DevBuf<uint2> l2_diag;                  // ... and each record's diagonal
    uint64_t kmers_since_fold = 0;
    DevBuf<bk::TableSlot> table;
    DevBuf<uint32_t> ent_off, ent_len;
    DevBuf<bk::DevEntry> entries;
    DevBuf<unsigned long long> counters[2];
    // sparse finalize (large indexes): per mate file the touch bitmaps the counter writers set and the lists finalize walks
    bool sparse = false;
    DevBuf<unsigned int> touch_v[2], touch_b[2], touch_p[2], touch_e[2], v_list[2], p_list[2], e_list[2], n_list[2];
    bool plane_used[2] = {false, false};   // counters were added to since the planes were last known to be all zero
    DevBuf<unsigned long long> pileup;      // 4 planes
    DevBuf<unsigned long long> stats;       // [2][n_files][3]
    DevBuf<unsigned char> present;          // [2][n_files]
    DevBuf<unsigned long long> kstats;      // [2][4]
    // bk_push_reads_packed: two staging slots, so that the copy of a batch overlaps the scan of the previous one
    struct StageSlot {
        DevBuf<uint32_t> words; DevBuf<uint16_t> lens;
        uint8_t* h = nullptr; size_t h_cap = 0;     // pinned host copy of the caller's batch (words, then lens)
        hipEvent_t done = nullptr; bool busy = false;
    } stage[2];
    int next_stage = 0;

    // asynchronous ASCII ingest (bk_push_reads_ascii): pinned staging + device buffers per slot
    struct IngestSlot {
        uint8_t* h_bases = nullptr; size_t h_bases_cap = 0;
        unsigned long long* h_off = nullptr; size_t h_off_cap = 0;
        DevBuf<uint8_t> d_bases;
        DevBuf<unsigned long long> d_off, d_nrec;
        DevBuf<uint32_t> d_work;           // pack_words_kernel's work list
        DevBuf<uint32_t> d_words;
        DevBuf<uint16_t> d_lens;
        hipEvent_t uploaded = nullptr, done = nullptr;
        bool busy = false;
    };
    IngestSlot slots[3];
    IngestSlot dev_ascii;                   // bk_push_reads_ascii_device: the packed records of the batch being scanned (device buffers only)
    int next_slot = 0;
    hipStream_t copy_stream = nullptr;

    hipStream_t own_stream = nullptr, stream = nullptr;
    bool in_sample = false;
    int finalized_mates = 0;                // mate files of the sample whose finalize was enqueued last (0: none since bk_sample_begin / create)
    uint64_t pushed_records[2] = {0, 0};
    // multi-genome indexes: the LDS window (difference array + Level 1's arrays) sits on the genome the sample looks like
    DevBuf<uint32_t> occ;                   // [n_full][n_files] first occurrence of each reference k-mer in each genome file
    DevBuf<unsigned int> win_votes;         // [n_files]
    DevBuf<uint32_t> win_sel;               // {win_file, win_lo} of the current sample, chosen on the device
    DevBuf<uint32_t> file_cell_lo_d;        // [n_files] (shared by forks)
    std::vector<uint32_t> file_cell_lo;     // first cell of each genome file
    uint32_t win_lo = 0;
    int win_file = 0;
    bool win_chosen = false;                // for the current sample
    bool plane_stale[2] = {true, true};   // the mate's counter plane still holds an earlier sample (zeroed at its first push / at finalize)

    // after the pileup (bk_sample_call): sequence geometry (shared by forks), per-engine scratch and results
    DevBuf<uint64_t> genome_len, seq_cell, seq_len_d;
    DevBuf<int32_t> seq_first, n_seqs_d;
    int max_seqs_per_file = 0;
    uint64_t max_file_cells = 0;
    uint64_t max_file_cells_idx = 0;        // cells of the genome file with the most (pileup rows)
    DevBuf<double> call_noise, noise_maf, noise_tbl, noise_sums;   // (get_baseline_noise, the walk taken apart: CallArgs)
    DevBuf<unsigned int> noise_cnt, noise_state;
    DevBuf<bk_call_record> call_records;
    DevBuf<bk_call_summary> call_out;
    DevBuf<bk_call_summary> sel_out;        // pileup_selected_only: the genome selected between the two finalize passes
    DevBuf<unsigned long long> dbg;   // BK_L2_STATS (testing build): tallies of what the scan leaves to Level 2
    int ablate = 0;   // BK_SCAN_ABLATE (measurement aid): see scan_count_kernel
    uint64_t max_launch_records = 0;   // BK_MAX_LAUNCH_RECORDS (testing aid): split pushes into launches of at most this many records
    bool timing = false;
    unsigned timing_kinds = 0xfu, timing_every = 1, timing_seen[4] = {0, 0, 0, 0};
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> free_events;

    bk::IndexView view() const {
        bk::IndexView v{};
        v.kmer_pos = kmer_pos.p; v.pilots = pilots.p; v.m = m; v.log2nb = log2nb; v.log2p = log2p;
        v.kmer_of = kmer_of.p; v.id_rec = id_rec.p; v.dirty_ans = dirty_ans.p; v.cell_flags = cell_flags.p; v.ref_words = ref_words.p; v.cell_codes = cell_codes.p; v.cell_has = cell_has.p; v.cell_clean = cell_clean.p; v.cell_clean3 = cell_clean3.p; v.cell_yf = cell_yf.p; v.cell_yr = cell_yr.p; v.id_at = id_at.p; v.cell_fast = cell_fast.p; v.cell_nat = cell_nat.p; v.cell_natrow = cell_natrow.p; v.cell_blk = cell_blk.p; v.seed_tab = seed_tab.p; v.seed_log2 = seed_log2; v.total_cells = (uint32_t)total_cells; v.n_u = n_u;
        v.n_full = n_full; v.n_prows = n_prows; v.prow_id = prow_id.p; v.prow_t = prow_t.p; v.v_omin = v_omin; v.v_span = v_span; v.v_off = v_off;
        v.lo = bk::HalfView{half_lo.pilots.p, half_lo.dir.p, half_lo.cand.p, half_lo.m, half_lo.log2nb, half_lo.log2p, half_lo.bits.p, half_lo.bits_log2, half_lo.bits_exact};
        v.hi = bk::HalfView{half_hi.pilots.p, half_hi.dir.p, half_hi.cand.p, half_hi.m, half_hi.log2nb, half_hi.log2p, half_hi.bits.p, half_hi.bits_log2, half_hi.bits_exact};
        v.lo_bases = lo_bases; v.slot_of = slot_of.p; v.slot_rec = slot_rec.p; v.ent_files = ent_files.p; v.slot_files = slot_files.p; v.slot_alias = slot_alias.p; v.gather_ok = gather_ok ? 1u : 0u; v.id_own_files = id_own_files.p; v.cell_file = cell_file.p; v.id_rest_off = id_rest_off.p; v.id_rest = id_rest.p; v.estat_files = estat_files.p; v.amb = amb.p; v.estat_off = estat_off.p; v.estat = estat.p;
        v.table = table.p; v.ent_off = ent_off.p; v.ent_len = ent_len.p;
        v.entries = entries.p; v.n_slots = n_slots; v.log2s = log2s; v.k = k; v.wstart = wstart; v.W = W; v.n_files = n_files;
        return v;
    }

    hipEvent_t get_event() {
        if (!free_events.empty()) { hipEvent_t e = free_events.back(); free_events.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    struct Span {
        bk_engine* e; int kind; hipEvent_t a = nullptr;
        Span(bk_engine* eng, int k) : e(eng), kind(k) {
            if (e->timing && (e->timing_kinds >> k) & 1 && e->timing_seen[k]++ % e->timing_every == 0) { a = e->get_event(); (void)hipEventRecord(a, e->stream); }
        }
        ~Span() {
            if (a) { hipEvent_t b = e->get_event(); (void)hipEventRecord(b, e->stream); e->spans.push_back({a, b, kind}); }
        }
    };
};

// everything a sample writes (counter planes, scan scratch, outputs) and the engine's stream: per engine, never shared by forks
static int alloc_sample_state(bk_engine* e) {
    const bk_params* prm = &e->params;
    for (int m = 0; m < 2; m++) BK_HIP(e->counters[m].alloc(e->plane_len));
    // A plane of a large index is a thousandth full after a sample: above 16 M counters (128 MB) the writers note what they touch
    // and finalize walks lists and clears what it read instead of scanning and zeroing planes (BK_SPARSE_FINALIZE forces it in
    // the testing build)
    e->sparse = e->W > 0 && (e->plane_len >= (16ull << 20) || test_env("BK_SPARSE_FINALIZE") != nullptr);
    if (e->sparse) {
        const uint64_t n_rows = bk::v_real_rows(e->n_full, e->v_span);
        for (int m = 0; m < 2; m++) {
            BK_HIP(hipMemset(e->counters[m].p, 0, e->counters[m].n * sizeof(unsigned long long)));
            BK_HIP(e->touch_v[m].alloc(n_rows / 32 + 1)); BK_HIP(e->touch_p[m].alloc(e->n_prows / 32 + 1)); BK_HIP(e->touch_e[m].alloc((size_t)e->n_u / 32 + 1));
            BK_HIP(hipMemset(e->touch_v[m].p, 0, e->touch_v[m].n * 4)); BK_HIP(hipMemset(e->touch_p[m].p, 0, e->touch_p[m].n * 4));
            BK_HIP(hipMemset(e->touch_e[m].p, 0, e->touch_e[m].n * 4));
            BK_HIP(e->touch_b[m].alloc((size_t)e->total_cells / 64 / 32 + 2)); BK_HIP(hipMemset(e->touch_b[m].p, 0, e->touch_b[m].n * 4));
            BK_HIP(e->v_list[m].alloc(n_rows)); BK_HIP(e->p_list[m].alloc(e->n_prows)); BK_HIP(e->e_list[m].alloc(e->n_u));
            BK_HIP(e->n_list[m].alloc(8));
        }
    }
    BK_HIP(e->shard_sums.alloc((size_t)2 * e->n_files * 5 + 9));
    BK_HIP(e->xport_flag.alloc(4));
    BK_HIP(hipMemset(e->xport_flag.p, 0, 4 * sizeof(unsigned long long)));
    if (prm->full_kmer_stats) {
        if (prm->kmer_table_log2 < 10 || prm->kmer_table_log2 > 31) return fail(BK_ERR_INVALID, "kmer_table_log2 out of range");
        BK_HIP(e->ktab_keys.alloc((size_t)1 << prm->kmer_table_log2));
        BK_HIP(e->ktab_cnt.alloc((size_t)1 << prm->kmer_table_log2));
        e->ktab_log2 = prm->kmer_table_log2;
        BK_HIP(hipHostMalloc(reinterpret_cast<void**>(&e->h_fill), bk::ktab_fill_words() * sizeof(unsigned long long), hipHostMallocDefault));
        BK_HIP(hipEventCreateWithFlags(&e->fill_ev, hipEventDisableTiming));
    }
    BK_HIP(e->ktab_out.alloc(8 + bk::ktab_fill_words()));
    // one row of per-genome tallies per finalize workgroup (8192 rows: 10 MB at 100 genomes); without it every workgroup adds its
    // tallies to the same few cache lines of `stats` with global atomics -- 2 ms per kernel at 100 genomes
    if (e->n_files <= 2048) BK_HIP(e->fin_partials.alloc(bk::finalize_partial_rows() * ((size_t)e->n_files * 3 + 2)));
    BK_HIP(e->deferred.alloc(bk::v_plane_len(e->n_full, e->v_span, e->n_prows) * (prm->pileup_selected_only ? 2 : 1)));   // (one list per mate file when it is kept between two passes)
    BK_HIP(e->n_deferred.alloc(2));   // one per mate file
    if (prm->pileup_selected_only && e->ent_files.p) BK_HIP(e->deferred_mask.alloc(e->deferred.n));
    if (!e->sparse) {
        // dense planes: K2a zeroes the V counters as it reads them and the (small) E part is zeroed behind K2e, so a plane is
        // clean again when its sample is finalized -- no 37 MB memset per sample (config 2); the deferred k-mers' counts
        // travel with their indices
        BK_HIP(e->deferred_n.alloc(e->deferred.n));
        for (int m = 0; m < 2; m++) BK_HIP(hipMemset(e->counters[m].p, 0, std::max<size_t>(e->counters[m].n, 1) * sizeof(unsigned long long)));
    }
    if (e->n_files == 1 && !e->sparse && e->n_full > 0) { BK_HIP(e->lean_e_list.alloc((size_t)e->n_full)); BK_HIP(e->lean_n_list.alloc(8)); BK_HIP(hipMemset(e->lean_n_list.p, 0, 8 * sizeof(unsigned int))); }
    BK_HIP(e->pileup.alloc(e->total_cells * 4 * 4));
    e->gather_mode = e->gather_ok && e->sparse && e->cell_file.p && e->dirty_ans.p && e->W > 1 && e->file_cell_lo_d.p && !test_env("BK_NO_GATHER");
    if (e->gather_mode) {
        for (int m = 0; m < 2; m++) BK_HIP(e->alias_hits[m].alloc((size_t)bk_engine::kAliasCap * 3));
        BK_HIP(e->n_alias_hits.alloc(2));
        BK_HIP(e->last_sel.upload(std::vector<int>(1, -1)));
        BK_HIP(hipMemset(e->pileup.p, 0, std::max<size_t>(e->pileup.n, 1) * sizeof(unsigned long long)));   // (selected-only: the rows of genomes never selected stay zero)
        // every genome's rows by the table of voters (bk_gather.hip): the table and the touched-row bits are this engine's for its
        // lifetime -- allocated here, never inside a sample (a hipMalloc synchronises the device: siblings in flight would stall)
        const bool two_pass = prm->pileup_selected_only != 0 && e->n_files > 1;
        if (!two_pass && prm->cs < (1ull << 28) && e->total_cells >= 2 * (uint64_t)e->n_full && !test_env("BK_NO_VOTE_TABLE")) {
            BK_HIP(e->row_bits.alloc((size_t)((bk::v_real_rows(e->n_full, e->v_span) + 31) / 32) + 1));
            BK_HIP(e->vote_tab.alloc(bk::vote_table_words(e->view())));
        }
    }
    BK_HIP(e->stats.alloc((size_t)2 * e->n_files * 3));
    BK_HIP(e->present.alloc((size_t)2 * e->n_files));
    if (prm->pileup_selected_only != 0 && e->n_files > 1) BK_HIP(e->sel_out.alloc(1));   // (the genome selected between the two finalize passes)
    BK_HIP(e->l2_plan.alloc(4));
    BK_HIP(e->kstats.alloc(8));
    // the scan: binned (items) when the planes are dense, the window's reference is staged in LDS and the bins are few enough;
    // else the whole-window difference array of scan_count_kernel with its slabs
    e->use_items = !e->sparse && e->ref_in_lds && e->W > 0 && e->n_lds_bins > 0 && !test_env("BK_NO_ITEMS") && e->seed_tab2.p && e->rc_words.p &&
                   bk::item_geometry(std::min<uint32_t>(e->n_lds_bins, (uint32_t)e->total_cells), e->n_full, e->v_span, &e->ig) &&
                   bk::items_lds_bytes(e->ig, std::min<uint32_t>(e->n_lds_bins, (uint32_t)e->total_cells)) <= 128u * 1024u;
    if (e->use_items) {
        if (const char* cp = test_env("BK_ITEM_CAPS")) {   // measurement aid: "cap_e,cap_v" (multiples of 8, at most 64)
            unsigned ce = 0, cv = 0;
            if (sscanf(cp, "%u,%u", &ce, &cv) == 2 && ce >= 8 && cv >= 8 && ce <= 64 && cv <= 64 && ce % 8 == 0 && cv % 8 == 0) {
                bk::ItemGeom g2 = e->ig;
                g2.cap_e = ce; g2.cap_v = cv; g2.wg_items = g2.n_ebins * ce + g2.n_vbins * cv; g2.wg_stride = g2.wg_items;
                if (bk::items_lds_bytes(g2, std::min<uint32_t>(e->n_lds_bins, (uint32_t)e->total_cells)) <= 118u * 1024u) e->ig = g2;
            }
        }
        const size_t g = bk::items_max_grid(e->n_cus);
        e->ig.grid_max = (uint32_t)g;
        BK_HIP(e->items.alloc(g * e->ig.wg_stride + 64));   // (+ 64: bin_count reads whole 16-byte units)
        BK_HIP(e->item_tab.alloc(g * ((size_t)e->ig.n_ebins + e->ig.n_vbins)));
        BK_HIP(e->item_gext.alloc(g * ((size_t)e->ig.n_ebins + e->ig.n_vbins) * bk::kItemGCap));   // (92 MB for one SARS-CoV-2 genome: 2 bytes x 256 slots x 701 bins x 256 workgroups)
        BK_HIP(e->ov.alloc((size_t)1 << 20));
        BK_HIP(e->ov_n.alloc(4));   // [2] overflow counts by launch parity, behind them (as 32-bit words) the scan's two chunk counters
        BK_HIP(hipMemset(e->ov_n.p, 0, 4 * sizeof(unsigned long long)));
    } else {
        BK_HIP(e->slabs.alloc((size_t)e->n_cus * std::max<uint32_t>(e->n_lds_bins, 1)));
    }
    if (e->n_files > 1) { BK_HIP(e->win_votes.alloc((size_t)e->n_files)); BK_HIP(e->win_sel.upload(std::vector<uint32_t>(2, 0u))); }
    // the scan's V items straight into the regional finalize (bk_finalize_lean.hip): where that kernel runs (one genome file, dense
    // planes, no statistics table, no pseudo k-mers, an answer table; FinalizeArgs are checked again at finalize), a V bin is the 64
    // row positions of one of its workgroups, and Level 2 is the only other writer of the V part (ScanArgs::n_direct: no nbatch_kernel)
    e->fuse_ok = e->use_items && e->n_files == 1 && e->max_seqs_per_file == 1 && (uint64_t)e->n_lds_bins >= e->total_cells && !e->ktab_keys.p && e->n_prows == 0 &&
                 e->n_u == e->n_full && e->n_full > 0 && e->dirty_ans.p && e->W > 1 && e->v_span > 0 && e->v_span <= 32 && e->fin_partials.p && e->lean_e_list.p &&
                 prm->cs < (1ull << 32) && e->ig.vq_log2 == 6 && !test_env("BK_NO_LEAN_FINALIZE") && !test_env("BK_NO_FUSE") && !test_env("BK_NO_N_DIRECT");
    if (e->fuse_ok)
        for (int m = 0; m < 2; m++) {
            BK_HIP(e->fuse_touch[m].alloc(((size_t)e->n_full + (size_t)e->v_span + 63) / 64 * 12 + 16));
            BK_HIP(hipMemset(e->fuse_touch[m].p, 0, e->fuse_touch[m].n * sizeof(unsigned int)));
        }
    BK_HIP(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    e->stream = e->own_stream;
    return BK_OK;
}

extern "C" {

int bk_abi_version(void) { return BK_ABI_VERSION; }
int bk_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}
int bk_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
    if (!free_bytes || !total_bytes) return fail(BK_ERR_INVALID, "null argument");
    BK_HIP(hipSetDevice(device));
    size_t f = 0, t = 0;
    BK_HIP(hipMemGetInfo(&f, &t));
    *free_bytes = f; *total_bytes = t;
    return BK_OK;
}
const char* bk_last_error(void) { return g_err.c_str(); }

void bk_params_default(bk_params* p) {
    if (!p) return;
    p->n_fixed = 2;           // consts.rs:17
    p->use_full_kmer = 0;     // consts.rs:18
    p->ci = 3;                // consts.rs:5
    p->cs = 1000000;          // call.rs:1173
    p->cx = 1000000000ull;    // KMC default -cx
    p->device = 0;
    p->full_kmer_stats = 0;
    p->kmer_table_log2 = 26;
    p->pileup_selected_only = 0;
}

int bk_engine_create(const bk_index_desc* ix, const bk_params* prm, bk_engine** out) {
    if (!ix || !prm || !out) return fail(BK_ERR_INVALID, "null argument");
    *out = nullptr;
    const int k = ix->k;
    if (k < 3 || k > bk::kMaxK || (k & 1) == 0) return fail(BK_ERR_INVALID, "Invalid kmer size %d, must be odd and <= %d", k, bk::kMaxK);
    if (prm->n_fixed < 0) return fail(BK_ERR_INVALID, "n_fixed must be >= 0");
    if (ix->n_files <= 0 || ix->n_files > 65536) return fail(BK_ERR_INVALID, "n_files out of range");
    if (ix->n_buckets && (!ix->bucket_ids || !ix->bucket_off || !ix->entries)) return fail(BK_ERR_INVALID, "null index arrays");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(BK_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (prm->device < 0 || prm->device >= ndev) return fail(BK_ERR_NO_DEVICE, "device %d not present (%d visible)", prm->device, ndev);
    BK_HIP(hipSetDevice(prm->device));
    {
        hipDeviceProp_t prop;
        BK_HIP(hipGetDeviceProperties(&prop, prm->device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)   // the kernels are built for gfx950 only
            return fail(BK_ERR_NO_DEVICE, "device %d is %s, not gfx950 (MI355X); this library has no other code path", prm->device, prop.gcnArchName);
    }

    std::unique_ptr<bk_engine> e(new bk_engine());
    e->params = *prm;
    e->k = k;
    e->device = prm->device;
    e->n_files = ix->n_files;
    // window slice of call.rs:1291-1300
    if (prm->use_full_kmer) { e->wstart = 0; e->W = k; }
    else if (prm->n_fixed * 2 + 1 >= k) { e->wstart = 0; e->W = 0; }
    else { e->wstart = prm->n_fixed; e->W = k - 2 * prm->n_fixed - 1; }

    // cell offsets in (file, seq) order = layout of initialize_output_maps (call.rs:1437-1480)
    std::vector<std::vector<uint64_t>> cell_off(ix->n_files);
    std::vector<size_t> seq_base(ix->n_files);
    uint64_t cells = 0;
    size_t q = 0;
    for (int f = 0; f < ix->n_files; f++) {
        if (ix->n_seqs[f] < 0 || ix->n_seqs[f] > 256) return fail(BK_ERR_INVALID, "file %d: n_seqs out of range (seq_id is u8)", f);
        seq_base[f] = q;
        cell_off[f].resize(ix->n_seqs[f]);
        for (int s = 0; s < ix->n_seqs[f]; s++, q++) { cell_off[f][s] = cells; cells += ix->seq_lens[q]; }
    }
    if (cells >= (1ull << 32)) return fail(BK_ERR_UNSUPPORTED, "more than 2^32 reference positions");
    e->total_cells = cells;
    {   // sequence geometry for the device caller
        std::vector<uint64_t> g_len(ix->n_files, 0), s_cell, s_len;
        std::vector<int32_t> s_first(ix->n_files, 0), n_s(ix->n_files, 0);
        size_t sq = 0;
        for (int f = 0; f < ix->n_files; f++) {
            s_first[f] = (int32_t)sq; n_s[f] = ix->n_seqs[f];
            e->max_seqs_per_file = std::max(e->max_seqs_per_file, (int)ix->n_seqs[f]);
            for (int s2 = 0; s2 < ix->n_seqs[f]; s2++, sq++) { s_cell.push_back(cell_off[f][s2]); s_len.push_back(ix->seq_lens[sq]); g_len[f] += ix->seq_lens[sq]; }
            e->max_file_cells = std::max(e->max_file_cells, g_len[f]);
        }
        if (s_cell.empty()) { s_cell.push_back(0); s_len.push_back(0); }
        BK_HIP(e->genome_len.upload(g_len)); BK_HIP(e->seq_cell.upload(s_cell)); BK_HIP(e->seq_len_d.upload(s_len));
        BK_HIP(e->seq_first.upload(s_first)); BK_HIP(e->n_seqs_d.upload(n_s));
    }

    PhaseClock pc;
    // ---- window buckets -> device slots ------------------------------------------------------------------
    // Device key of a bucket = (wildcard position j, canonical reference k-mer with position j zeroed).  It is
    // recomputed from the metadata sequence at (file, seq, location) and checked against the stored bucket id
    // with assign_buckets, so an index that disagrees with its own metadata is rejected instead of miscounted.
    std::vector<uint64_t> h_slot_key;
    std::vector<uint8_t> h_slot_t, h_slot_alias;   // h_slot_alias: the slot's key is the OTHER exact rank that wraps onto its bucket's id (k = 31)
    uint64_t n_merged_buckets = 0, n_dup_entries = 0, n_window_entries = 0;
    std::vector<uint32_t> h_merged_slots;   // the window slots of buckets that hold more than one key
    std::vector<uint32_t> h_off, h_len;
    std::vector<bk::DevEntry> h_ent;
    std::vector<uint64_t> per_t(e->W > 0 ? e->W : 1, 0);
    std::vector<uint64_t> h_u;   // canonical reference k-mers that own at least one window bucket (with repeats)
    // k = 31 only: "pseudo" k-mers u*.  A read k-mer equal to u* except possibly at one window position can reach an
    // index bucket through the u64 wrap of its bucket id (see rank128): u* stands for the alias key (j', V') of a
    // real bucket (j, V) with the base of the real k-mer at j' filled in.  The wrap is structured (changing a few
    // leading bases shifts every rank of a k-mer by exactly 2^64), so the W buckets of a reference k-mer usually
    // share one pseudo k-mer.  Which window positions of u* really lead to a bucket is read back from the table.
    std::vector<uint64_t> pseudo;
    // The buckets are taken in contiguous chunks by host threads, each filling its own output; the chunks are then joined in
    // order, so the result is the one a single pass over all buckets gives.
    struct ChunkOut {
        std::vector<uint64_t> h_slot_key, h_u, pseudo, per_t;
        std::vector<uint8_t> h_slot_t, h_slot_alias;
        std::vector<uint32_t> h_off, h_len;   // h_off: relative to this chunk's h_ent
        std::vector<bk::DevEntry> h_ent;
        std::vector<uint32_t> merged;   // slots (relative to this chunk's) of buckets that hold more than one key (k = 31: two reference buckets whose ids wrapped onto each other)
        uint64_t n_merged = 0;    // ... the number of such buckets
        uint64_t n_dup = 0;       // buckets that hold one BucketInfo twice
        uint64_t n_real_ent = 0;  // BucketInfos of the window's buckets (each once)
        int code = BK_OK;
        std::string err;
        bool fail(int c, const char* fmt, ...) {
            char buf[512];
            va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
            code = c; err = buf;
            return false;
        }
    };
    auto process_buckets = [&](uint64_t b0, uint64_t b1, ChunkOut& o) -> bool {
        uint64_t ids[32];
        o.per_t.assign(e->W > 0 ? e->W : 1, 0);
        std::vector<std::pair<int, uint64_t>> keys;   // (one allocation per chunk, not per bucket: 37 M mallocs from 256 threads with a hundred strains)
        for (uint64_t b = b0; b < b1; b++) {
            const uint64_t lo = ix->bucket_off[b], hi = ix->bucket_off[b + 1];
            if (hi <= lo) continue;
            if (hi > ix->n_entries) return o.fail(BK_ERR_INVALID, "bucket_off out of range");
            // distinct (j, masked) keys present in this bucket: exactly one unless k = 31 ids wrapped onto each other
            keys.clear();
            uint64_t first_kmer = 0;
            bool any_in_window = false;
            for (uint64_t i = lo; i < hi; i++) {
                const bk_bucket_info& bi = ix->entries[i];
                if (bi.file_id >= ix->n_files || bi.seq_id >= ix->n_seqs[bi.file_id]) return o.fail(BK_ERR_INVALID, "entry %llu references a missing sequence", (unsigned long long)i);
                const size_t sq = seq_base[bi.file_id] + bi.seq_id;
                if ((uint64_t)bi.location + k > ix->seq_lens[sq] || bi.idx >= k) return o.fail(BK_ERR_INVALID, "entry %llu lies outside its sequence", (unsigned long long)i);
                const bronko::Canon cn = bronko::canonical_kmer(ix->seqs[sq] + bi.location, k);
                if (cn.rc != (bi.canonical != 0)) return o.fail(BK_ERR_INVALID, "entry %llu: canonical flag disagrees with the metadata sequence", (unsigned long long)i);
                const int j = bi.idx;
                const uint64_t masked = cn.kmer & ~(3ull << (2 * (k - 1 - j)));
                if (std::find(keys.begin(), keys.end(), std::make_pair(j, masked)) == keys.end()) {
                    bronko::assign_buckets(cn.kmer, k, ids);
                    if (ids[j] != ix->bucket_ids[b]) return o.fail(BK_ERR_INVALID, "bucket %llu: id does not match assign_buckets of its entries", (unsigned long long)ix->bucket_ids[b]);
                    if (keys.empty()) first_kmer = cn.kmer;
                    keys.emplace_back(j, masked);
                }
                if (j >= e->wstart && j < e->wstart + e->W) {
                    any_in_window = true;
                    if (o.h_u.empty() || o.h_u.back() != cn.kmer) o.h_u.push_back(cn.kmer);   // (a bucket of a many-genome index names one k-mer again and again)
                }
            }
            // the other exact rank that wraps onto this bucket's id, if the reference did not already put a k-mer there
            int alias_j = -1;
            uint64_t alias_masked = 0;
            if (k == 31 && keys.size() == 1) {
                const u128 own = rank128(keys[0].second, keys[0].first, k);
                if ((uint64_t)own != ix->bucket_ids[b]) return o.fail(BK_ERR_INVALID, "internal: exact bucket rank disagrees with assign_buckets");
                const u128 two64 = (u128)1 << 64;
                const u128 other = own >= two64 ? own - two64 : own + two64;
                uint64_t av; int aj;
                if (unrank128(other, k, &av, &aj) && aj >= e->wstart && aj < e->wstart + e->W) { alias_j = aj; alias_masked = av; }
            }
            if (!any_in_window && alias_j < 0) continue;
            // every entry of the bucket is voted for by a probe of any of its keys (call.rs:1307-1309 iterates the
            // whole Vec<BucketInfo>), using each entry's own idx (call.rs:1329)
            const uint32_t off = (uint32_t)o.h_ent.size();
            for (uint64_t i = lo; i < hi; i++) {
                const bk_bucket_info& bi = ix->entries[i];
                bk::DevEntry de;
                de.cell = (uint32_t)(cell_off[bi.file_id][bi.seq_id] + bi.location + bi.idx);
                de.file = bi.file_id; de.idx = bi.idx; de.canonical = bi.canonical ? 1 : 0;
                o.h_ent.push_back(de);
            }
            // finalize_variant counts hits per file as run lengths: keep each bucket grouped by file (build_indexes
            // already appends file by file, build.rs:223-228; votes are order-independent)
            std::stable_sort(o.h_ent.begin() + off, o.h_ent.end(), [](const bk::DevEntry& x, const bk::DevEntry& y) { return x.file < y.file; });
            // (what the gathered votes of bk_gather.hip rest on: one key per bucket, every BucketInfo once)
            if (keys.size() > 1) o.n_merged++;
            for (uint64_t i = lo; i < hi; i++) o.n_real_ent += ix->entries[i].idx >= e->wstart && ix->entries[i].idx < e->wstart + e->W;
            if (any_in_window) {
                for (size_t x = off; x < o.h_ent.size(); x++)
                    for (size_t y = x + 1; y < o.h_ent.size() && o.h_ent[y].file == o.h_ent[x].file; y++)
                        if (o.h_ent[y].cell == o.h_ent[x].cell && o.h_ent[y].idx == o.h_ent[x].idx) o.n_dup++;
            }
            for (auto& kv : keys) {
                if (kv.first < e->wstart || kv.first >= e->wstart + e->W) continue;
                if (keys.size() > 1) o.merged.push_back((uint32_t)o.h_slot_key.size());
                o.h_slot_key.push_back(kv.second);
                o.h_slot_alias.push_back(0);
                o.h_slot_t.push_back((uint8_t)(kv.first - e->wstart));
                o.h_off.push_back(off);
                o.h_len.push_back((uint32_t)(hi - lo));
                o.per_t[kv.first - e->wstart]++;
            }
            if (alias_j >= 0) {
                o.h_slot_key.push_back(alias_masked);
                o.h_slot_alias.push_back(1);
                o.h_slot_t.push_back((uint8_t)(alias_j - e->wstart));
                o.h_off.push_back(off);
                o.h_len.push_back((uint32_t)(hi - lo));
                o.per_t[alias_j - e->wstart]++;
                o.pseudo.push_back(alias_masked | (first_kmer & (3ull << (2 * (k - 1 - alias_j)))));
            }
            if (o.h_ent.size() >= (1ull << 32)) return o.fail(BK_ERR_UNSUPPORTED, "more than 2^32 index entries in the window");
        }
        // a reference k-mer is named by every window bucket it owns: each chunk hands over its own distinct ones
        std::sort(o.h_u.begin(), o.h_u.end());
        o.h_u.erase(std::unique(o.h_u.begin(), o.h_u.end()), o.h_u.end());
        return true;
    };
    {
        const uint64_t nbk = e->W > 0 ? ix->n_buckets : 0;
        const unsigned nt = nbk < 65536 ? 1u : std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 256u);
        std::vector<ChunkOut> outs(nt);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) {
            const uint64_t b0 = nbk * t / nt, b1 = nbk * (t + 1) / nt;
            if (nt == 1) process_buckets(b0, b1, outs[0]);
            else th.emplace_back([&, t, b0, b1] { process_buckets(b0, b1, outs[t]); });
        }
        for (auto& t : th) t.join();
        for (auto& o : outs) if (o.code != BK_OK) return fail(o.code, "%s", o.err.c_str());
        pc.lap("  buckets: chunks");
        uint64_t n_ent = 0;
        for (auto& o : outs) n_ent += o.h_ent.size();
        if (n_ent >= (1ull << 32)) return fail(BK_ERR_UNSUPPORTED, "more than 2^32 index entries in the window");
        // the chunks' lists back to back, in chunk order: where each goes is a prefix sum, the copies run side by side
        std::vector<size_t> e0(nt + 1, h_ent.size()), s0(nt + 1, h_slot_key.size()), u0(nt + 1, h_u.size()), p0(nt + 1, pseudo.size());
        for (unsigned t = 0; t < nt; t++) {
            e0[t + 1] = e0[t] + outs[t].h_ent.size(); s0[t + 1] = s0[t] + outs[t].h_slot_key.size();
            u0[t + 1] = u0[t] + outs[t].h_u.size(); p0[t + 1] = p0[t] + outs[t].pseudo.size();
            for (size_t w = 0; w < per_t.size() && w < outs[t].per_t.size(); w++) per_t[w] += outs[t].per_t[w];
        }
        for (auto& o : outs) { n_merged_buckets += o.n_merged; n_dup_entries += o.n_dup; n_window_entries += o.n_real_ent; }
        for (unsigned t = 0; t < nt; t++) for (uint32_t rel : outs[t].merged) h_merged_slots.push_back((uint32_t)(s0[t] + rel));
        h_ent.resize(e0[nt]); h_slot_key.resize(s0[nt]); h_slot_t.resize(s0[nt]); h_slot_alias.resize(s0[nt]); h_len.resize(s0[nt]); h_off.resize(s0[nt]);
        h_u.resize(u0[nt]); pseudo.resize(p0[nt]);
        {
            std::vector<std::thread> cp;
            for (unsigned t = 0; t < nt; t++) cp.emplace_back([&, t] {
                ChunkOut& o = outs[t];
                std::copy(o.h_ent.begin(), o.h_ent.end(), h_ent.begin() + (ptrdiff_t)e0[t]);
                std::copy(o.h_slot_key.begin(), o.h_slot_key.end(), h_slot_key.begin() + (ptrdiff_t)s0[t]);
                std::copy(o.h_slot_t.begin(), o.h_slot_t.end(), h_slot_t.begin() + (ptrdiff_t)s0[t]);
                std::copy(o.h_slot_alias.begin(), o.h_slot_alias.end(), h_slot_alias.begin() + (ptrdiff_t)s0[t]);
                std::copy(o.h_len.begin(), o.h_len.end(), h_len.begin() + (ptrdiff_t)s0[t]);
                for (size_t i = 0; i < o.h_off.size(); i++) h_off[s0[t] + i] = (uint32_t)e0[t] + o.h_off[i];
                std::copy(o.h_u.begin(), o.h_u.end(), h_u.begin() + (ptrdiff_t)u0[t]);
                std::copy(o.pseudo.begin(), o.pseudo.end(), pseudo.begin() + (ptrdiff_t)p0[t]);
                o = ChunkOut();   // free
            });
            for (auto& t : cp) t.join();
        }
    }
    pc.lap("buckets -> slots (+aliases)");
    e->n_slots = h_slot_key.size();
    if (e->n_slots >= (1ull << 31)) return fail(BK_ERR_UNSUPPORTED, "too many window buckets");

    uint64_t max_t = 1;
    for (uint64_t c : per_t) max_t = std::max(max_t, c);
    e->log2s = 4;
    while ((1ull << e->log2s) < 2 * max_t) e->log2s++;   // load factor <= 0.5
    const size_t S = (size_t)1 << e->log2s;
    // (a large index: built on the device, where the probes of U below run too -- the host never holds it)
    const bool table_on_device = e->n_slots >= (1u << 18) && e->W > 0;
    HostVec<bk::TableSlot> h_table;
    if (table_on_device) {
        bool dup = false;
        BK_HIP(e->table.alloc((size_t)e->W * S));
        static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "");
        BK_HIP(bk::device_build_table(e->table.p, (size_t)e->W * S, e->log2s, reinterpret_cast<const unsigned long long*>(h_slot_key.data()), h_slot_t.data(), e->n_slots, &dup));
        if (dup && k != 31) return fail(BK_ERR_INVALID, "duplicate window bucket in the index");
    } else {
    h_table = filled((size_t)std::max(e->W, 1) * S, bk::TableSlot{bk::kEmptyKey, 0u, 0u});
    pc.lap("  window tables: allocation");
    {
        // one sub-table per window position: each is filled by its own host thread, in slot order (the first of equal keys stays)
        std::atomic<bool> dup{false};
        auto fill = [&](int t0, int t1) {
            for (uint64_t s = 0; s < e->n_slots; s++) {
                const int t = h_slot_t[s];
                if (t < t0 || t >= t1) continue;
                bk::TableSlot* sub = h_table.data() + (size_t)t * S;
                uint32_t h = bk::hash_key(h_slot_key[s], e->log2s);
                while (sub[h].key != bk::kEmptyKey) {
                    if (sub[h].key == h_slot_key[s]) {
                        if (k != 31) dup = true;   // (k = 31: an alias key that coincides with a real key -- same wrapped id, same bucket: keep the first)
                        break;
                    }
                    h = (h + 1) & (uint32_t)(S - 1);
                }
                if (sub[h].key == bk::kEmptyKey) { sub[h].key = h_slot_key[s]; sub[h].slot = (uint32_t)s; }
            }
        };
        const int nth = e->n_slots < 262144 ? 1 : std::max(1, std::min<int>(e->W, (int)std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        for (int q = 1; q < nth; q++) th.emplace_back(fill, e->W * q / nth, e->W * (q + 1) / nth);
        fill(0, e->W / nth > 0 ? e->W / nth : e->W);
        for (auto& t : th) t.join();
        if (dup) return fail(BK_ERR_INVALID, "duplicate window bucket in the index");
    }
    }
    pc.lap("window tables");
    // a slot with no entries: "this k-mer has no bucket at that window position" (pseudo k-mers)
    const uint32_t empty_slot = (uint32_t)h_off.size();
    h_off.push_back(0);
    h_len.push_back(0);

    // ---- reference k-mer set U ------------------------------------------------------------------------------
    // ids in order of first occurrence in reference order; perfect hash (membership + diagonal seeding);
    // half-key directories (neighbour search); the reference in reference order (diagonal walk); per-id tables.
    const unsigned sort_threads = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u);
    parallel_sort(h_u, std::less<uint64_t>(), sort_threads);
    h_u.erase(std::unique(h_u.begin(), h_u.end()), h_u.end());
    pc.lap("  U: sort");
    // pseudo k-mers join U (so that the membership / neighbour machinery finds the read k-mers that alias), but they
    // own only the window positions at which the table holds a key for them.  A pseudo value that is a real
    // reference k-mer needs nothing: its alias key is that k-mer's own bucket key.
    std::vector<uint64_t> extra;
    parallel_sort(pseudo, std::less<uint64_t>(), sort_threads);
    pseudo.erase(std::unique(pseudo.begin(), pseudo.end()), pseudo.end());
    {
        std::vector<uint8_t> keep(pseudo.size(), 0);
        parallel_for(pseudo.size(), [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) keep[i] = std::binary_search(h_u.begin(), h_u.end(), pseudo[i]) ? 0 : 1;
        });
        for (size_t i = 0; i < pseudo.size(); i++) if (keep[i]) extra.push_back(pseudo[i]);   // (sorted, like pseudo)
    }
    {
        const size_t mid = h_u.size();
        h_u.insert(h_u.end(), extra.begin(), extra.end());
        std::inplace_merge(h_u.begin(), h_u.begin() + (ptrdiff_t)mid, h_u.end());   // two sorted, disjoint runs
    }
    pc.lap("  U: pseudo k-mers sorted, merged");
    // bucket (slot) of every k-mer of U at every window position, by table lookup; h_valid = positions with a bucket
    std::vector<uint32_t> h_valid(h_u.size(), 0u);
    std::vector<uint8_t> h_is_pseudo(h_u.size(), 0);
    // (a large index: the probes run on the device, against the tables where they will stay -- 400 M of them with a hundred strains
    // at k = 31, DRAM latency on the host; the slots stay on the device until they are laid out by id, slot_of below)
    const bool slots_on_device = (h_u.size() >= (1u << 18) || table_on_device) && e->W > 0;
    DevBuf<uint32_t> d_slot_by_index;
    HostVec<uint32_t> slot_by_index;
    std::atomic<bool> lacks{false};
    if (slots_on_device) {
        if (!table_on_device) BK_HIP(e->table.upload(h_table));
        BK_HIP(d_slot_by_index.alloc(h_u.size() * (size_t)e->W));
        static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "");
        BK_HIP(bk::device_lookup_slots(e->table.p, e->log2s, reinterpret_cast<const unsigned long long*>(h_u.data()), h_u.size(), e->W, e->wstart, k, empty_slot,
                                       d_slot_by_index.p, h_valid.data()));
        const uint32_t all = e->W >= 32 ? 0xffffffffu : (1u << e->W) - 1u;
        parallel_for(h_u.size(), [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) {
                h_is_pseudo[i] = std::binary_search(extra.begin(), extra.end(), h_u[i]) ? 1 : 0;
                if (!h_is_pseudo[i] && h_valid[i] != all) lacks = true;
            }
        });
    } else {
    slot_by_index = filled((size_t)std::max<size_t>(h_u.size(), 1) * std::max(e->W, 1), empty_slot);
    pc.lap("  U: slot_by_index allocation");
    parallel_for(h_u.size(), [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            h_is_pseudo[i] = std::binary_search(extra.begin(), extra.end(), h_u[i]) ? 1 : 0;
            for (int t = 0; t < e->W; t++) {
                const uint64_t key = h_u[i] & ~(3ull << (2 * (k - 1 - (e->wstart + t))));
                const bk::TableSlot* sub = h_table.data() + (size_t)t * S;
                uint32_t h = bk::hash_key(key, e->log2s);
                while (sub[h].key != key && sub[h].key != bk::kEmptyKey) h = (h + 1) & (uint32_t)(S - 1);
                if (sub[h].key == key) { slot_by_index[i * e->W + t] = sub[h].slot; h_valid[i] |= 1u << t; }
                else if (!h_is_pseudo[i]) lacks = true;
            }
        }
    });
    }
    if (lacks) return fail(BK_ERR_INVALID, "index lacks a window bucket of one of its own reference k-mers");
    pc.lap("U + slot lookup");
    e->lo_bases = k / 2;
    if (h_u.size() >= (1ull << 31)) return fail(BK_ERR_UNSUPPORTED, "too many distinct reference k-mers");
    e->n_u = (uint32_t)h_u.size();
    {
        // walk the metadata sequences: ids, first occurrences, packed bases and the per-cell flag bits
        const uint32_t kNone = 0xffffffffu;
        std::vector<uint32_t> id_of(h_u.size(), kNone), first_cell(h_u.size(), kNone);
        std::vector<uint8_t> first_rc(h_u.size(), 0);
        const uint64_t cells = e->total_cells;
        std::vector<uint32_t> h_id_at(std::max<uint64_t>(cells, 1), kNone);
        const size_t pad_w = (size_t)bk::scan_ref_pad_words();   // front padding of the two 2-bit arrays
        std::vector<uint32_t> h_refw(pad_w + (cells + 15) / 16 + (size_t)bk::scan_ref_back_words(), 0u), h_brc((cells + 31) / 32 + 1, 0u);
        uint32_t next_id = 0;
        // first every cell's k-mer is looked up in U (the sequences side by side on host threads; h_id_at holds the index into
        // h_u for the moment), then the ids are handed out in reference order
        std::vector<uint8_t> cell_rc(std::max<uint64_t>(cells, 1), 0);
        {
            struct SeqJob { const uint8_t* seq; uint64_t len, c0; };
            std::vector<SeqJob> jobs;
            size_t sq = 0;
            for (int f = 0; f < ix->n_files; f++)
                for (int sidx = 0; sidx < ix->n_seqs[f]; sidx++, sq++) {
                    const uint64_t len = ix->seq_lens[sq], c0 = cell_off[f][sidx];
                    const uint8_t* seq = ix->seqs[sq];
                    for (uint64_t i = 0; i < len; i++) h_refw[pad_w + ((c0 + i) >> 4)] |= (uint32_t)bronko::nt_to_bits(seq[i]) << (2 * ((c0 + i) & 15));
                    // long sequences in pieces (every piece re-reads the k - 1 bases before it)
                    for (uint64_t a = 0; a + k <= len; a += 8192) jobs.push_back(SeqJob{seq + a, std::min<uint64_t>(len - a, 8192 + (uint64_t)k - 1), c0 + a});
                }
            std::atomic<size_t> next_job{0};
            auto work = [&] {
                const uint64_t mask = bronko::kmer_mask(k);
                for (size_t j = next_job++; j < jobs.size(); j = next_job++) {
                    const SeqJob& jb = jobs[j];
                    uint64_t fwd = 0;
                    for (int i = 0; i < k - 1; i++) fwd = (fwd << 2) | bronko::nt_to_bits(jb.seq[i]);
                    for (uint64_t i = 0; i + k <= jb.len; i++) {
                        fwd = ((fwd << 2) | bronko::nt_to_bits(jb.seq[i + k - 1])) & mask;
                        const bronko::Canon cn = bronko::canonical_u64(fwd, k);
                        const auto it = std::lower_bound(h_u.begin(), h_u.end(), cn.kmer);   // h_u is sorted
                        if (it == h_u.end() || *it != cn.kmer) continue;   // not in the index: never predicted, never counted
                        h_id_at[jb.c0 + i] = (uint32_t)(it - h_u.begin());
                        cell_rc[jb.c0 + i] = cn.rc ? 1 : 0;
                    }
                }
            };
            const unsigned nt = jobs.size() < 4 ? 1u : std::min<unsigned>(sort_threads, (unsigned)jobs.size());
            std::vector<std::thread> th;
            for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
            work();
            for (auto& t : th) t.join();
        }
        uint64_t n_occurrences = 0;   // cells at which a k-mer of U starts
        for (uint64_t cell = 0; cell < cells; cell++) {
            const uint32_t ui = h_id_at[cell];
            if (ui == kNone) continue;
            ++n_occurrences;
            if (id_of[ui] == kNone) { id_of[ui] = next_id++; first_cell[ui] = (uint32_t)cell; first_rc[ui] = cell_rc[cell]; }
            h_id_at[cell] = id_of[ui];
            if (cell_rc[cell]) h_brc[cell >> 5] |= 1u << (cell & 31);
        }
        for (size_t i = 0; i < h_u.size(); i++)   // k-mers known only through index entries: ids after the others
            if (id_of[i] == kNone && !h_is_pseudo[i]) id_of[i] = next_id++;
        e->n_full = next_id;
        for (size_t i = 0; i < h_u.size(); i++)   // pseudo k-mers last: they own V rows only where they own a bucket
            if (id_of[i] == kNone) id_of[i] = next_id++;
        // NbEntry::p (bk_device.h): the id of a reference k-mer; n_full + first pseudo V row of a pseudo k-mer (rows in id order)
        std::vector<uint32_t> row_base(h_u.size(), 0u);
        // with full_kmer_stats the rows keep every offset, so that k-mers differing outside the window are not lost to the statistics
        e->v_omin = bk::v_layout_omin(k, e->wstart, e->W, prm->full_kmer_stats != 0);
        e->v_span = bk::v_layout_span(k, e->wstart, e->W, prm->full_kmer_stats != 0);
        std::vector<uint32_t> h_nat, h_natrow;
        {
            std::vector<uint32_t> idx_by_id(h_u.size());
            for (size_t i = 0; i < h_u.size(); i++) idx_by_id[id_of[i]] = (uint32_t)i;
            uint64_t rows = 0;
            std::vector<uint32_t> h_prow_id;
            std::vector<uint8_t> h_prow_t;
            for (size_t id = 0; id < h_u.size(); id++) {
                const uint32_t i = idx_by_id[id];
                if (id < e->n_full) { row_base[i] = (uint32_t)id; continue; }
                row_base[i] = (uint32_t)(e->n_full + rows);
                for (int t = 0; t < e->W; t++)
                    if ((h_valid[i] >> t) & 1u) { h_prow_id.push_back((uint32_t)id); h_prow_t.push_back((uint8_t)t); rows++; }
                if (e->n_full + rows >= (1ull << 31)) return fail(BK_ERR_UNSUPPORTED, "index too large: too many pseudo k-mer buckets");
            }
            e->n_prows = rows;
            if (bk::v_plane_len(e->n_full, e->v_span, rows) >= (1ull << 32)) return fail(BK_ERR_UNSUPPORTED, "index too large: variant counter plane exceeds 2^32 counters");
            bk::counter_plane_layout(e->n_u, e->n_full, e->v_span, rows, e->v_off, e->plane_len);
            if (h_prow_id.empty()) { h_prow_id.push_back(0); h_prow_t.push_back(0); }
            BK_HIP(e->prow_id.upload(h_prow_id));
            BK_HIP(e->prow_t.upload(h_prow_t));
        }

        pc.lap("reference walk + ids");
        // dirty flags (bk_device.h amb): another reference k-mer, on either strand, within Hamming distance 2, or
        // the k-mer within distance 2 of its own reverse complement.  Any two 2k-bit words at distance <= 2 agree
        // on at least one of three parts, so group all forms (u and rc(u)) by each part and compare inside groups.
        std::vector<uint8_t> h_amb(h_u.size(), 0);   // by id
        // amb3: the same with distance 3 (four parts); lets Level 2 discard k-mers with two differences on the spot.  The
        // groups grow with |U| (a quarter of a k-mer distinguishes little): above kAmb3MaxKmers everything is flagged.
        std::vector<uint8_t> h_amb3(h_u.size(), 0);
        // far23: another reference k-mer form at distance 2 or 3 (forms one base away do not count): where there is none, a read k-mer
        // two bases from u can only equal or neighbour the two k-mers "u with one of its two differences" (Level 2, kCellIso23)
        std::vector<uint8_t> h_far23(h_u.size(), 0);
        constexpr size_t kAmb3MaxKmers = 300000;
        std::vector<uint64_t> h_near;                     // (canonical form's index << 32 | near form), sorted: the near lists
        std::vector<uint8_t> h_no_list(h_u.size(), 0);    // by index: in a group too large to enumerate -- no near list
        {
            struct Form { uint64_t w; uint32_t id; uint32_t fi; };   // fi = 2 * (index into h_u) + (1: the reverse complement)
            std::vector<Form> forms(h_u.size() * 2);
            parallel_for(h_u.size(), [&](size_t i0, size_t i1) {
                for (size_t i = i0; i < i1; i++) {
                    forms[2 * i] = Form{h_u[i], id_of[i], (uint32_t)(2 * i)};
                    forms[2 * i + 1] = Form{bronko::reverse_complement_u64(h_u[i], k), id_of[i], (uint32_t)(2 * i + 1)};
                }
            });
            // collect: for every canonical form, the forms within `dist` of it (the near lists the dirty answers are worked out from)
            // out_far (optional): the same for pairs at distance 2 or more only -- what kCellIso23 is made of (bk_device.h)
            std::vector<uint8_t>* out_far = nullptr;
            auto flag_within = [&](int dist, std::vector<uint8_t>& out, std::vector<std::vector<uint64_t>>* collect) {
                const int parts = dist + 1;   // words at distance <= dist agree on at least one of dist + 1 parts
                if (collect) collect->assign(parts, {});
                std::vector<std::thread> th;
                const unsigned per_part = std::max(1u, std::min(64u, std::thread::hardware_concurrency() / (unsigned)parts));
                for (int part = 0; part < parts; part++) th.emplace_back([&, part] {   // (flags are only ever set to 1: benign races)
                    std::vector<uint64_t>* near = collect ? &(*collect)[part] : nullptr;
                    const int c0 = (part * k) / parts, c1 = ((part + 1) * k) / parts;
                    const uint64_t mask = (((1ull << (2 * (c1 - c0))) - 1ull) << (2 * c0));
                    // the forms grouped by this part: a radix sort of the part's bits on the device, the forms gathered in that order
                    // (std::sort of 30 M forms on 24 host threads per part was 1.4 s of a 100-strain create)
                    std::vector<Form> fs;
                    bool on_device = false;
                    if (forms.size() >= (1u << 16) && hipSetDevice(e->device) == hipSuccess) {
                        std::vector<unsigned long long> keys(forms.size());
                        std::vector<unsigned int> order(forms.size());
                        parallel_for(forms.size(), [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) keys[i] = (forms[i].w & mask) >> (2 * c0); });
                        if (bk::device_sort_order(keys.data(), keys.size(), 2 * (c1 - c0), order.data(), nullptr) == hipSuccess) {
                            fs.resize(forms.size());
                            parallel_for(forms.size(), [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) fs[i] = forms[order[i]]; });
                            on_device = true;
                        }
                    }
                    if (!on_device) {
                        fs = forms;
                        parallel_sort(fs, [&](const Form& x, const Form& y) { return (x.w & mask) < (y.w & mask); }, per_part);
                    }
                    // the groups (equal parts), dealt to threads in runs of whole groups; every thread collects its own near pairs
                    const unsigned nt = fs.size() < 262144 ? 1u : per_part;
                    std::vector<size_t> cut(nt + 1, fs.size());
                    cut[0] = 0;
                    for (unsigned t = 1; t < nt; t++) {
                        size_t a = std::max(fs.size() * t / nt, cut[t - 1]);
                        while (a < fs.size() && a > 0 && (fs[a].w & mask) == (fs[a - 1].w & mask)) a++;
                        cut[t] = a;
                    }
                    std::vector<std::vector<uint64_t>> mine(nt);
                    auto scan_groups = [&](size_t lo, size_t hi, std::vector<uint64_t>& out_near) {
                        for (size_t a0 = lo; a0 < hi;) {
                            size_t a1 = a0 + 1;
                            while (a1 < hi && (fs[a1].w & mask) == (fs[a0].w & mask)) a1++;
                            if (a1 - a0 > 4096) {   // pathological low-complexity group: flag all, skip the quadratic pass
                                for (size_t x = a0; x < a1; x++) { out[fs[x].id] = 1; if (out_far) (*out_far)[fs[x].id] = 1; if (near) h_no_list[fs[x].fi >> 1] = 1; }
                            } else {
                                // (pseudo k-mers -- 95 % of U with a hundred strains at k = 31 -- are flagged dirty whatever their neighbours
                                // and own no near list: a pair of two of them says nothing, and only a reference k-mer's list is kept)
                                for (size_t x = a0; x < a1; x++) {
                                    const bool px = h_is_pseudo[fs[x].fi >> 1] != 0;
                                    for (size_t y = x + 1; y < a1; y++) {
                                        const bool py = h_is_pseudo[fs[y].fi >> 1] != 0;
                                        if (px && py) continue;
                                        const uint64_t d = fs[x].w ^ fs[y].w;
                                        const int nd = __builtin_popcountll((d | (d >> 1)) & 0x5555555555555555ull);
                                        if (nd <= dist) {
                                            out[fs[x].id] = out[fs[y].id] = 1;   // also catches u vs rc(u) (same id)
                                            if (out_far && nd >= 2) (*out_far)[fs[x].id] = (*out_far)[fs[y].id] = 1;
                                            if (near) {   // (owner canonical form << 32) | the other form
                                                if (!(fs[x].fi & 1u) && !px) out_near.push_back(((uint64_t)(fs[x].fi >> 1) << 32) | fs[y].fi);
                                                if (!(fs[y].fi & 1u) && !py) out_near.push_back(((uint64_t)(fs[y].fi >> 1) << 32) | fs[x].fi);
                                            }
                                        }
                                    }
                                }
                            }
                            a0 = a1;
                        }
                    };
                    {
                        std::vector<std::thread> gt;
                        for (unsigned t = 1; t < nt; t++) gt.emplace_back([&, t] { scan_groups(cut[t], cut[t + 1], mine[t]); });
                        scan_groups(cut[0], cut[1], mine[0]);
                        for (auto& t : gt) t.join();
                    }
                    if (near) for (auto& v : mine) { near->insert(near->end(), v.begin(), v.end()); std::vector<uint64_t>().swap(v); }
                });
                for (auto& t : th) t.join();
            };
            std::vector<std::vector<uint64_t>> near_parts;
            flag_within(2, h_amb, &near_parts);
            pc.lap("  dirty: distance 2");
            if (h_u.size() <= kAmb3MaxKmers) { out_far = &h_far23; flag_within(3, h_amb3, nullptr); out_far = nullptr; }
            else { std::fill(h_amb3.begin(), h_amb3.end(), (uint8_t)1); std::fill(h_far23.begin(), h_far23.end(), (uint8_t)1); }
            pc.lap("  dirty: distance 3");
            size_t tot = 0;
            for (auto& v : near_parts) tot += v.size();
            h_near.reserve(tot);
            for (auto& v : near_parts) { h_near.insert(h_near.end(), v.begin(), v.end()); std::vector<uint64_t>().swap(v); }
            {
                bool on_device = false;
                if (h_near.size() >= (1u << 20) && h_near.size() < (1ull << 32) && hipSetDevice(e->device) == hipSuccess) {
                    std::vector<unsigned int> order(h_near.size());
                    std::vector<unsigned long long> sorted(h_near.size());
                    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "");
                    if (bk::device_sort_order(reinterpret_cast<const unsigned long long*>(h_near.data()), h_near.size(), 64, order.data(), sorted.data()) == hipSuccess) {
                        parallel_for(h_near.size(), [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) h_near[i] = sorted[i]; });
                        on_device = true;
                    }
                }
                if (!on_device) parallel_sort(h_near, std::less<uint64_t>(), sort_threads);
            }
            pc.lap("  dirty: near lists sorted");
            h_near.erase(std::unique(h_near.begin(), h_near.end()), h_near.end());
        }
        for (size_t i = 0; i < h_u.size(); i++) if (h_is_pseudo[i]) { h_amb3[id_of[i]] = 1; h_far23[id_of[i]] = 1; }
        if (test_env("BK_NO_ISO23")) std::fill(h_far23.begin(), h_far23.end(), (uint8_t)1);
        for (size_t i = 0; i < h_u.size(); i++) if (h_is_pseudo[i]) h_amb[id_of[i]] = 1;
        pc.lap("dirty flags (dist 2, 3)");
        std::vector<uint8_t> rc_of_id(h_u.size(), 0);   // the k-mer's first occurrence was reverse-complemented to become canonical
        for (size_t i = 0; i < h_u.size(); i++) rc_of_id[id_of[i]] = first_rc[i];
        std::vector<uint32_t> h_codes(h_refw.size(), 0u), h_yf(h_refw.size(), 0u), h_yr(h_refw.size(), 0u);   // bk_device.h
        std::vector<uint8_t> h_needs_ans(h_u.size(), 0);   // by id: some cell of this reference k-mer is not clean
        std::vector<uint8_t> h_cflags(std::max<uint64_t>(cells, 1), 0);   // bk_device.h kCellClean
        const size_t bpad_w = (size_t)bk::scan_bit_pad_words();
        std::vector<uint32_t> h_has(bpad_w + (cells + 31) / 32 + (size_t)bk::scan_bit_back_words(), 0u), h_clean(h_has.size(), 0u), h_clean3(h_has.size(), 0u);
        for (uint64_t c = 0; c < cells; c++) {
            if (h_id_at[c] == kNone) continue;
            const size_t wi = pad_w + (c >> 4);
            const int sh = 2 * (int)(c & 15);
            h_codes[wi] |= (((h_brc[c >> 5] >> (c & 31)) & 1u) ? 2u : 1u) << sh;
            // clean also promises the orientation of the k-mer's first occurrence (the V layout is built on it): an
            // occurrence on the other strand of a reverse-complement repeat is resolved by the general path
            const uint32_t rc_here = (h_brc[c >> 5] >> (c & 31)) & 1u;
            const uint32_t clean = (h_amb[h_id_at[c]] || rc_here != rc_of_id[h_id_at[c]]) ? 0u : 1u;
            const bool from_prev = c > 0 && h_id_at[c - 1] != kNone && h_id_at[c] == h_id_at[c - 1] + 1;
            const bool to_next = c + 1 < cells && h_id_at[c + 1] != kNone && h_id_at[c + 1] == h_id_at[c] + 1;
            h_has[bpad_w + (c >> 5)] |= 1u << (c & 31);
            if (clean) h_clean[bpad_w + (c >> 5)] |= 1u << (c & 31);
            else h_needs_ans[h_id_at[c]] = 1;
            h_cflags[c] = (uint8_t)((rc_here ? 2u : 1u) | (clean ? bk::kCellClean : 0u) | (h_amb3[h_id_at[c]] ? 0u : bk::kCellClean3) |
                                    (rc_here == rc_of_id[h_id_at[c]] ? bk::kCellFirstOri : 0u) | (h_far23[h_id_at[c]] ? 0u : bk::kCellIso23));
            if (!h_amb3[h_id_at[c]]) h_clean3[bpad_w + (c >> 5)] |= 1u << (c & 31);
            h_yf[wi] |= (clean | (from_prev ? 2u : 0u)) << sh;
            h_yr[wi] |= (clean | (to_next ? 2u : 0u)) << sh;
        }

        // what the scan needs to count an isolated mismatch on the spot (bk_device.h cell_fast / cell_blk): per block of 64 cells the
        // constant id - cell of its clean cells and the end of the stretch of cells that carry a reference k-mer
        std::vector<uint32_t> h_fast(h_has.size(), 0u);
        std::vector<uint2> h_blk((cells + 63) / 64 + 3, make_uint2(0u, (uint32_t)cells));
        {
            uint32_t next_none = (uint32_t)cells;
            for (uint64_t c = cells; c-- > 0;) {
                if (h_id_at[c] == kNone) next_none = (uint32_t)c;
                if ((c & 63) == 0) h_blk[c >> 6].y = next_none;
            }
            // x: the most common id - cell among the block's cells that stand in their k-mer's first orientation (clean or not:
            // with many related genomes no cell is clean, and cell_nat below still wants the constant)
            for (uint64_t b0 = 0; b0 < cells; b0 += 64) {
                uint32_t best = 0u, best_n = 0u;
                const uint64_t b1 = std::min<uint64_t>(b0 + 64, cells);
                for (uint64_t c = b0; c < b1; c++) {
                    if (!(h_cflags[c] & bk::kCellFirstOri)) continue;
                    const uint32_t delta = h_id_at[c] - (uint32_t)c;
                    if (best_n && delta == best) continue;
                    uint32_t n = 0;
                    for (uint64_t q = c; q < b1; q++) n += (h_cflags[q] & bk::kCellFirstOri) && h_id_at[q] - (uint32_t)q == delta;
                    if (n > best_n) { best_n = n; best = delta; }
                }
                h_blk[b0 >> 6].x = best;
            }
            for (uint64_t c = 0; c < cells; c++)
                if ((h_cflags[c] & bk::kCellClean) && h_id_at[c] - (uint32_t)c == h_blk[c >> 6].x) h_fast[bpad_w + (c >> 5)] |= 1u << (c & 31);
        }
        pc.lap("per-cell arrays");
        // ---- dirty answers (bk_device.h DirtyAns): for every reference k-mer with a cell that is not clean, what "this k-mer with
        // base bb at position j" is -- worked out from its near list (every reference k-mer form within Hamming distance 2: a
        // k-mer one base away from u can only equal, or neighbour, forms within distance 2 of u).  Same rule as the neighbour
        // search of Level 2's slow pipeline: a reference k-mer if it equals one, else the smallest (window position, NbEntry::p)
        // among the reference k-mers one base away in the window that own a bucket there, else nothing.
        {
            std::vector<uint32_t> idx_by_id(h_u.size());
            for (size_t i = 0; i < h_u.size(); i++) idx_by_id[id_of[i]] = (uint32_t)i;
            // rows are indexed by id (no indirection: Level 2 reads an answer with one load); only the rows of k-mers with a
            // cell that is not clean are filled in -- the others are never read
            std::vector<uint32_t> owners;   // index into h_u of each filled row
            for (size_t id = 0; id < e->n_full; id++)
                if (h_needs_ans[id] || h_amb[id]) owners.push_back(idx_by_id[id]);   // (a dirty k-mer without a cell: finalize still asks)
            const bool build = e->W > 0 && bk::ans_table_len(e->n_full, k) * sizeof(bk::DirtyAns) <= ((size_t)16 << 30);
            if (build && !owners.empty()) {
                std::vector<bk::DirtyAns> h_ans(bk::ans_table_len(e->n_full, k), bk::DirtyAns{0u, 0u});
                // entry of "reference k-mer id (index i into h_u) with base bb at position j of its canonical form" (bk_device.h ans_index)
                auto ans_at = [&](uint32_t i, int j, uint32_t bb) -> bk::DirtyAns& {
                    const bool rc1 = first_rc[i] != 0;
                    return h_ans[bk::ans_index(id_of[i], (uint32_t)(rc1 ? k - 1 - j : j), rc1 ? 3u - bb : bb, k)];
                };
                const uint64_t vreal = bk::v_real_len(e->n_full, e->v_span);
                auto diff1 = [&](uint64_t a, uint64_t b) -> int {   // position (from the left) of the single differing base, or -1
                    const uint64_t x = a ^ b, y = (x | (x >> 1)) & 0x5555555555555555ull;
                    if (y == 0 || (y & (y - 1)) != 0) return -1;
                    return k - 1 - (__builtin_ctzll(y) >> 1);
                };
                parallel_for(owners.size(), [&](size_t r0, size_t r1) {
                    std::vector<std::pair<uint64_t, uint32_t>> fl;   // (form word, form index) of u itself and its near forms
                    for (size_t r = r0; r < r1; r++) {
                        const uint32_t i = owners[r];
                        const uint64_t u = h_u[i];
                        if (h_no_list[i]) {   // a low-complexity group too large to enumerate: no near list, no answers
                            for (int j = 0; j < k; j++) for (uint32_t bb = 0; bb < 4; bb++) ans_at(i, j, bb) = bk::DirtyAns{0u, bk::kAnsNone};
                            continue;
                        }
                        fl.clear();
                        fl.emplace_back(u, (uint32_t)(2 * i));
                        for (auto it = std::lower_bound(h_near.begin(), h_near.end(), (uint64_t)i << 32); it != h_near.end() && (*it >> 32) == i; ++it) {
                            const uint32_t fi = (uint32_t)*it;
                            fl.emplace_back((fi & 1u) ? bronko::reverse_complement_u64(h_u[fi >> 1], k) : h_u[fi >> 1], fi);
                        }
                        for (int j = 0; j < k; j++) {
                            const int sh = 2 * (k - 1 - j);
                            for (uint32_t bb = 0; bb < 4; bb++) {
                                if (((u >> sh) & 3ull) == bb) continue;
                                bk::DirtyAns& A = ans_at(i, j, bb);
                                const uint64_t z = (u & ~(3ull << sh)) | ((uint64_t)bb << sh);
                                const uint64_t zr = bronko::reverse_complement_u64(z, k);
                                const bool flip = zr < z;              // the canonical form of z is its reverse complement
                                const uint64_t c = flip ? zr : z;
                                bool member = false;
                                uint64_t best = ~0ull; uint32_t best_fi = 0, jmask = 0;
                                for (auto& f : fl) {
                                    // a form says something about c (the canonical form of z) only in c's orientation: c vs u' is z vs u', or
                                    // rc(z) vs u' = z vs rc(u').  (k = 31 pseudo k-mers are not canonical values: the other pairing does occur.)
                                    if (((f.second & 1u) != 0) != flip) continue;
                                    if (f.first == z) { A.idx = 2u * id_of[f.second >> 1]; A.meta = 1u; member = true; break; }
                                    const int pp = diff1(f.first, z);
                                    if (pp < 0) continue;
                                    const int jn = flip ? k - 1 - pp : pp;            // position in c (= in the neighbour's canonical form)
                                    if (jn < e->wstart || jn >= e->wstart + e->W || !((h_valid[f.second >> 1] >> (jn - e->wstart)) & 1u)) continue;
                                    const uint64_t key = ((uint64_t)jn << 32) | row_base[f.second >> 1];
                                    jmask |= 1u << (jn - e->wstart);
                                    if (key < best) { best = key; best_fi = f.second; }
                                }
                                if (member || best == ~0ull) continue;
                                const uint32_t multi = (jmask & (jmask - 1u)) ? bk::kAnsMulti : 0u;
                                const int jn = (int)(best >> 32);
                                const uint32_t pnb = (uint32_t)best, ni = best_fi >> 1;
                                const uint32_t bc = (uint32_t)(c >> (2 * (k - 1 - jn))) & 3u;
                                if (pnb < e->n_full) {
                                    const uint32_t rcu = first_rc[ni] ? 1u : 0u;
                                    const int oo = (rcu ? k - 1 - jn : jn) - e->v_omin;
                                    if (oo < 0 || oo >= e->v_span) continue;            // (v_point's guard)
                                    const uint32_t nbc = (uint32_t)(h_u[ni] >> (2 * (k - 1 - jn))) & 3u;   // the neighbour's own base there
                                    A.idx = (uint32_t)(bk::v_row_base(pnb + (uint32_t)oo, bk::v_alt(bc, nbc), 0u, e->v_span) + (uint32_t)oo);
                                    A.meta = 2u | (rcu << 2) | ((oo + 1 < e->v_span) ? 8u : 0u) | multi;
                                } else {
                                    const uint32_t row = pnb - e->n_full + (uint32_t)__builtin_popcount(h_valid[ni] & ((1u << (jn - e->wstart)) - 1u));
                                    A.idx = (uint32_t)(vreal + ((uint64_t)row * 4 + bc) * 2);
                                    A.meta = 3u | multi;
                                }
                            }
                        }
                    }
                });
                if (test_env("BK_VERIFY_ANSWERS")) {
                    // testing build: every answer against the definition -- membership in U and the neighbour search spelled out
                    // (all 3 W substitutions inside the window), no near lists involved
                    std::atomic<uint64_t> bad{0};
                    parallel_for(owners.size(), [&](size_t r0, size_t r1) {
                        for (size_t r = r0; r < r1; r++) {
                            const uint64_t u = h_u[owners[r]];
                            if (h_no_list[owners[r]]) continue;
                            for (int j = 0; j < k; j++) for (uint32_t bb = 0; bb < 4; bb++) {
                                const int sh = 2 * (k - 1 - j);
                                if (((u >> sh) & 3ull) == bb) continue;
                                const uint64_t z = (u & ~(3ull << sh)) | ((uint64_t)bb << sh), zr = bronko::reverse_complement_u64(z, k), c = zr < z ? zr : z;
                                bk::DirtyAns want{0u, 0u};
                                const auto it = std::lower_bound(h_u.begin(), h_u.end(), c);
                                if (it != h_u.end() && *it == c) { want.idx = 2u * id_of[it - h_u.begin()]; want.meta = 1u; }
                                else {
                                    uint64_t best = ~0ull; size_t bi = 0; uint32_t jm = 0;
                                    for (int jn = e->wstart; jn < e->wstart + e->W; jn++) for (uint64_t alt = 0; alt < 4; alt++) {
                                        const int s2 = 2 * (k - 1 - jn);
                                        if (((c >> s2) & 3ull) == alt) continue;
                                        const uint64_t cand = (c & ~(3ull << s2)) | (alt << s2);
                                        const auto ct = std::lower_bound(h_u.begin(), h_u.end(), cand);
                                        if (ct == h_u.end() || *ct != cand) continue;
                                        const size_t ci = ct - h_u.begin();
                                        if (!((h_valid[ci] >> (jn - e->wstart)) & 1u)) continue;
                                        const uint64_t key = ((uint64_t)jn << 32) | row_base[ci];
                                        jm |= 1u << (jn - e->wstart);
                                        if (key < best) { best = key; bi = ci; }
                                    }
                                    const uint32_t multi = (jm & (jm - 1u)) ? bk::kAnsMulti : 0u;
                                    if (best != ~0ull) {
                                        const int jn = (int)(best >> 32);
                                        const uint32_t pnb = (uint32_t)best, bc = (uint32_t)(c >> (2 * (k - 1 - jn))) & 3u;
                                        if (pnb < e->n_full) {
                                            const uint32_t rcu = first_rc[bi] ? 1u : 0u;
                                            const int oo = (rcu ? k - 1 - jn : jn) - e->v_omin;
                                            if (oo >= 0 && oo < e->v_span) {
                                                const uint32_t nbc = (uint32_t)(h_u[bi] >> (2 * (k - 1 - jn))) & 3u;
                                                want.idx = (uint32_t)(bk::v_row_base(pnb + (uint32_t)oo, bk::v_alt(bc, nbc), 0u, e->v_span) + (uint32_t)oo);
                                                want.meta = 2u | (rcu << 2) | ((oo + 1 < e->v_span) ? 8u : 0u) | multi;
                                            }
                                        } else {
                                            want.idx = (uint32_t)(vreal + ((uint64_t)(pnb - e->n_full + (uint32_t)__builtin_popcount(h_valid[bi] & ((1u << (jn - e->wstart)) - 1u))) * 4 + bc) * 2);
                                            want.meta = 3u | multi;
                                        }
                                    }
                                }
                                const bk::DirtyAns& got = ans_at(owners[r], j, bb);
                                if (got.idx != want.idx || got.meta != want.meta) {
                                    if (bad++ < 5) fprintf(stderr, "[bk] dirty answer differs: id %u j %d bb %u: table (%u, %u) definition (%u, %u)\n", id_of[owners[r]], j, bb, got.idx, got.meta, want.idx, want.meta);
                                }
                            }
                        }
                    });
                    if (bad) return fail(BK_ERR_INVALID, "internal: %llu dirty answers disagree with their definition", (unsigned long long)bad.load());
                }
                // cell_nat (bk_device.h): per reference position q and alternative a, bit o = "the k-mer that starts at q - o, with
                // that other base at q, takes its own V row" -- its cell is clean, or its answer says exactly that (or says that it
                // touches nothing, which is what finalize makes of the own row's count then: position outside the window or
                // canonical form on the other strand)
                // Which id a bit promises: with touch lists (large planes: the scan notes touched rows per block of cells) the one
                // cell_blk gives, id = cell + block constant; otherwise whatever row most of the k-mers over q agree on --
                // cell_natrow[q] = id + o -- which also covers the cells whose ids leave the block's sequence (a later genome's own
                // k-mers around its differences from an earlier one)
                const bool lists = e->W > 0 && (e->plane_len >= (16ull << 20) || test_env("BK_SPARSE_FINALIZE") != nullptr);   // (= bk_engine::sparse, set later)
                h_nat.assign(((size_t)cells + (size_t)k) * 3u, 0u);
                if (!lists) h_natrow.assign((size_t)cells + (size_t)k, 0u);
                parallel_for((size_t)cells + (size_t)k, [&](size_t q0, size_t q1) {
                    for (size_t q = q0; q < q1; q++) {
                        const uint32_t rb = q < cells ? (h_refw[pad_w + (q >> 4)] >> (2 * (q & 15))) & 3u : 0u;
                        uint32_t row = 0u;
                        if (!lists) {   // the most common id + o among the first-orientation cells q - o
                            uint32_t best_n = 0u;
                            for (int o = 0; o < k; o++) {
                                if (q < (size_t)o || q - (size_t)o >= cells) continue;
                                const size_t c = q - (size_t)o;
                                if (h_id_at[c] == kNone || !(h_cflags[c] & bk::kCellFirstOri)) continue;
                                const uint32_t r = h_id_at[c] + (uint32_t)o;
                                if (best_n && r == row) continue;
                                uint32_t n = 0u;
                                for (int o2 = o; o2 < k; o2++) {
                                    if (q < (size_t)o2 || q - (size_t)o2 >= cells) continue;
                                    const size_t c2 = q - (size_t)o2;
                                    n += h_id_at[c2] != kNone && (h_cflags[c2] & bk::kCellFirstOri) && h_id_at[c2] + (uint32_t)o2 == r;
                                }
                                if (n > best_n) { best_n = n; row = r; }
                            }
                            h_natrow[q] = row;
                        }
                        for (int o = 0; o < k; o++) {
                            if (q < (size_t)o || q - (size_t)o >= cells) continue;
                            const size_t c = q - (size_t)o;
                            const uint32_t id = h_id_at[c];
                            if (id == kNone || !(h_cflags[c] & bk::kCellFirstOri)) continue;
                            if (lists ? id - (uint32_t)c != h_blk[c >> 6].x : id + (uint32_t)o != row) continue;
                            for (uint32_t al = 0; al < 3; al++) {
                                bool nat = (h_cflags[c] & bk::kCellClean) != 0;
                                if (!nat) {
                                    const bk::DirtyAns& A = h_ans[bk::ans_index(id, (uint32_t)o, rb ^ (al + 1u), k)];
                                    const uint32_t kind = A.meta & 3u;
                                    const int oo = o - e->v_omin;
                                    if (A.meta & bk::kAnsNone) nat = false;
                                    else if (kind == 0u) nat = true;
                                    else if (kind == 2u && oo >= 0 && oo < e->v_span)
                                        nat = A.idx == (uint32_t)(bk::v_row_base(id + (uint32_t)oo, al, 0u, e->v_span) + (uint32_t)oo) && ((A.meta >> 2) & 1u) == rc_of_id[id];
                                }
                                if (nat) h_nat[q * 3u + al] |= 1u << o;
                            }
                        }
                    }
                });
                BK_HIP(e->dirty_ans.upload(h_ans));
                // Votes gathered cell by cell (bk_gather.hip) replace the walk over BucketInfo lists when a genome's BucketInfos ARE the
                // occurrences of its k-mers: every window bucket under one key (no two reference buckets merged by the k = 31 wrap),
                // holding each occurrence once and nothing else (an index built by `bronko build` does; a .bkdb from elsewhere might
                // not), every reference k-mer with a cell, and an answer for every dirty one
                bool all_listed = true, all_cells = true;
                for (uint32_t i : owners) if (h_no_list[i]) { all_listed = false; break; }
                for (size_t i = 0; i < h_u.size() && all_cells; i++) if (!h_is_pseudo[i] && first_cell[i] == kNone) all_cells = false;
                if (test_env("BK_L2_STATS") || test_env("BK_CREATE_TIMING"))
                    fprintf(stderr, "[bk] gathered votes: answers for all %d, cells for all %d, merged buckets %llu, doubled BucketInfos %llu, window BucketInfos %llu for %llu occurrences x %d\n",
                            (int)all_listed, (int)all_cells, (unsigned long long)n_merged_buckets, (unsigned long long)n_dup_entries, (unsigned long long)n_window_entries,
                            (unsigned long long)n_occurrences, e->W);
                e->gather_ok = all_listed && all_cells && n_dup_entries == 0 && n_window_entries == n_occurrences * (uint64_t)e->W &&
                               !test_env("BK_NO_GATHER");
            }
            std::vector<uint64_t>().swap(h_near);
        }
        pc.lap("dirty answers");
        // perfect hash over U
        std::vector<uint16_t> h_pilots;
        std::vector<uint32_t> u_pos;
        if (!build_phf(h_u, h_pilots, e->log2nb, e->m, e->log2p, u_pos)) return fail(BK_ERR_HIP, "internal error: perfect hash construction failed after every fallback");
        pc.lap("  perfect hash of U");
        std::vector<bk::KmerPos> t_pos((size_t)e->m << e->log2p, bk::KmerPos{bk::kEmptyKey, kNone, 0u});
        std::vector<uint64_t> h_kmer_of(std::max<size_t>(h_u.size(), 1), bk::kEmptyKey);
        for (size_t i = 0; i < h_u.size(); i++) {
            t_pos[u_pos[i]] = bk::KmerPos{h_u[i], first_cell[i], id_of[i] | (first_rc[i] ? 0x80000000u : 0u)};
            h_kmer_of[id_of[i]] = h_u[i];
        }
        BK_HIP(e->pilots.upload(h_pilots));
        BK_HIP(e->kmer_pos.upload(t_pos));
        BK_HIP(e->kmer_of.upload(h_kmer_of));
        BK_HIP(e->ref_words.upload(h_refw));
        BK_HIP(e->cell_has.upload(h_has));
        BK_HIP(e->cell_clean.upload(h_clean));
        BK_HIP(e->cell_clean3.upload(h_clean3));
        BK_HIP(e->cell_yf.upload(h_yf));
        BK_HIP(e->cell_yr.upload(h_yr));
        BK_HIP(e->cell_fast.upload(h_fast));
        BK_HIP(e->cell_blk.upload(h_blk));
        if (!h_nat.empty()) BK_HIP(e->cell_nat.upload(h_nat));
        if (!h_natrow.empty()) BK_HIP(e->cell_natrow.upload(h_natrow));
        BK_HIP(e->cell_codes.upload(h_codes));
        BK_HIP(e->cell_flags.upload(h_cflags));
        BK_HIP(e->id_at.upload(h_id_at));
        pc.lap("  tables of U filled, uploaded");
        e->file_cell_lo.assign((size_t)ix->n_files, 0u);
        for (int f = 0; f < ix->n_files; f++) e->file_cell_lo[f] = ix->n_seqs[f] ? (uint32_t)cell_off[f][0] : (uint32_t)cells;
        // the scan's seed tables (bk_device.h seed_hash): per genome file, where each of its reference k-mers starts
        if (cells > 0 && cells < (1ull << bk::kSeedCellBits) && e->n_full > 0) {
            uint64_t max_file_cells = 1;
            for (int f = 0; f < ix->n_files; f++) e->max_file_cells_idx = std::max<uint64_t>(e->max_file_cells_idx, (f + 1 < ix->n_files ? e->file_cell_lo[f + 1] : cells) - e->file_cell_lo[f]);
            for (int f = 0; f < ix->n_files; f++)
                max_file_cells = std::max<uint64_t>(max_file_cells, (f + 1 < ix->n_files ? e->file_cell_lo[f + 1] : cells) - e->file_cell_lo[f]);
            uint32_t L = 6;
            while ((1ull << L) < max_file_cells) L++;
            if (((uint64_t)ix->n_files << L) * sizeof(uint2) <= (8ull << 30)) {
                e->seed_log2 = L;
                std::vector<uint2> h_seed((size_t)ix->n_files << L, make_uint2(0xffffffffu, 0xffffffffu));
                parallel_for((size_t)ix->n_files, [&](size_t f0, size_t f1) {
                    for (size_t f = f0; f < f1; f++) {
                        const uint64_t c_lo = e->file_cell_lo[f], c_hi = f + 1 < (size_t)ix->n_files ? e->file_cell_lo[f + 1] : cells;
                        for (uint64_t c = c_lo; c < c_hi; c++) {
                            const uint32_t id = h_id_at[c];
                            if (id == kNone || id >= e->n_full) continue;
                            const uint32_t h = bk::seed_hash(h_kmer_of[id]);
                            const uint32_t ent = (uint32_t)c | (((h_brc[c >> 5] >> (c & 31)) & 1u) << bk::kSeedCellBits) | ((h & 15u) << 28);
                            uint2& b = h_seed[(f << L) + (h >> (32 - L))];
                            auto same = [&](uint32_t o) { return o != 0xffffffffu && h_id_at[o & ((1u << bk::kSeedCellBits) - 1u)] == id; };   // (a repeat: one entry does)
                            if (same(b.x) || same(b.y)) continue;
                            if (b.x == 0xffffffffu) b.x = ent; else if (b.y == 0xffffffffu) b.y = ent;   // (else: not in the table)
                        }
                    }
                });
                BK_HIP(e->seed_tab.upload(h_seed));
            }
            // ... and, for the binned scan, the reference reverse-complemented (symbol J = complement of symbol cells - 1 - J, same
            // paddings) with seed tables keyed by the k-mer AS A READ SHOWS IT -- bases in reading order, 2 bits each from bit 0 -- on
            // either strand: two entries per reference k-mer (along the reference: strand 0; against it: strand 1), four times the
            // buckets (a k-mer that finds its bucket full is no seed: 9% of them at twice the buckets, 3% at four times -- every
            // lost seed is a second round of seeds for its tile).  A read's k-mer is hashed as it stands -- no reverse complement, no canonical form -- and verified against the
            // reference (strand 0) or its reverse complement (strand 1) with one comparison.
            if (((uint64_t)ix->n_files << (L + 2)) * sizeof(uint2) <= (8ull << 30) && cells >= (uint64_t)k) {
                std::vector<uint32_t> h_rcw(h_refw.size() + 1, 0u);   // (+ 1: a window's slice starts inside a word, scan_items_kernel stages one word more)
                parallel_for((size_t)((cells + 15) / 16), [&](size_t w0, size_t w1) {
                    for (size_t w = w0; w < w1; w++) {
                        uint32_t acc = 0;
                        for (uint64_t J = (uint64_t)w * 16; J < std::min<uint64_t>((uint64_t)w * 16 + 16, cells); J++) {
                            const uint64_t c = cells - 1 - J;
                            acc |= (3u - ((h_refw[pad_w + (c >> 4)] >> (2 * (c & 15))) & 3u)) << (2 * (J & 15));
                        }
                        h_rcw[pad_w + w] = acc;
                    }
                });
                auto syms = [&](const std::vector<uint32_t>& a, uint64_t pos) -> uint64_t {   // k symbols from symbol `pos`, the first at bit 0
                    uint64_t g = 0;
                    for (int t = 0; t < k; t++) g |= (uint64_t)((a[pad_w + ((pos + t) >> 4)] >> (2 * ((pos + t) & 15))) & 3u) << (2 * t);
                    return g;
                };
                const uint32_t L2 = L + 2;
                e->seed2_log2 = L2;
                std::vector<uint2> h_seed2((size_t)ix->n_files << L2, make_uint2(0xffffffffu, 0xffffffffu));
                // Only the k-mers that start at ONE cell of their genome file are seeds: a repeat's entry would name one of its cells
                // for a read from any of them -- a diagonal that passes the verification (the k-mer is there) and is wrong; the scan
                // would then see a read of mismatches, all of them Level 2's to sort out.  Reads in repeats have other seeds.
                parallel_for((size_t)ix->n_files, [&](size_t f0, size_t f1) {
                    std::vector<uint8_t> seen(e->n_full, 0);   // per worker: occurrences of each id in the file at hand (saturating at 2)
                    for (size_t f = f0; f < f1; f++) {
                        const uint64_t c_lo = e->file_cell_lo[f], c_hi = f + 1 < (size_t)ix->n_files ? e->file_cell_lo[f + 1] : cells;
                        for (uint64_t c = c_lo; c < c_hi; c++) {
                            const uint32_t id = h_id_at[c];
                            if (id != kNone && id < e->n_full && seen[id] < 2) seen[id]++;
                        }
                        for (uint64_t c = c_lo; c < c_hi; c++) {
                            const uint32_t id = h_id_at[c];
                            if (id == kNone || id >= e->n_full || seen[id] != 1) continue;
                            for (uint32_t strand = 0; strand < 2u; strand++) {
                                const uint64_t g = strand ? syms(h_rcw, cells - (uint64_t)k - c) : syms(h_refw, c);
                                const uint32_t h = bk::seed_hash(g);
                                const uint32_t ent = (uint32_t)c | (strand << bk::kSeedCellBits) | ((h & 15u) << 28);
                                uint2& b = h_seed2[(f << L2) + (h >> (32 - L2))];
                                if (b.x == 0xffffffffu) b.x = ent; else if (b.y == 0xffffffffu) b.y = ent;   // (else: not in the table)
                            }
                        }
                        for (uint64_t c = c_lo; c < c_hi; c++) {   // (back to zero for the worker's next file: the cells, not the whole array)
                            const uint32_t id = h_id_at[c];
                            if (id != kNone && id < e->n_full) seen[id] = 0;
                        }
                    }
                });
                BK_HIP(e->rc_words.upload(h_rcw));
                BK_HIP(e->seed_tab2.upload(h_seed2));
            }
        }
        // (round 6: up to 2^31 entries -- 8 GB of the 288 --: 250 strains are 0.47 G; at 2^28 the window stayed on the first genome
        // and every strain difference of a sample went to Level 2)
        if (ix->n_files > 1 && (uint64_t)e->n_full * (uint64_t)ix->n_files <= (1ull << 31)) {
            std::vector<uint32_t> h_occ((size_t)e->n_full * ix->n_files, 0xffffffffu);
            for (int f = 0; f < ix->n_files; f++) {
                const uint64_t c_lo = e->file_cell_lo[f], c_hi = f + 1 < ix->n_files ? e->file_cell_lo[f + 1] : cells;
                for (uint64_t c = c_lo; c < c_hi; c++) {
                    const uint32_t id = h_id_at[c];
                    if (id == kNone || id >= e->n_full) continue;
                    uint32_t& o = h_occ[(size_t)id * ix->n_files + f];
                    if (o == 0xffffffffu) o = (uint32_t)c | (((h_brc[c >> 5] >> (c & 31)) & 1u) << 31);
                }
            }
            BK_HIP(e->occ.upload(h_occ));
            BK_HIP(e->file_cell_lo_d.upload(e->file_cell_lo));
        }
        {
            std::vector<uint8_t> h_amb2(h_amb);
            for (size_t id = 0; id < h_amb2.size(); id++) h_amb2[id] = (h_amb[id] ? 1 : 0) | (rc_of_id[id] ? 2 : 0);
            BK_HIP(e->amb.upload(h_amb2));
        }

        pc.lap("perfect hash of U + uploads");
        // half-key directories (neighbour search)
        const int lo_bits = 2 * e->lo_bases;
        const uint64_t lo_mask = (1ull << lo_bits) - 1ull;
        {
            // both halves at once (host threads); the low half needs a sort of its own, the high half is h_u's order
            struct HalfHost { std::vector<uint16_t> hp; std::vector<bk::HalfDir> dir; std::vector<bk::NbEntry> cand; std::vector<uint32_t> bits; bool ok = true; };
            HalfHost hh[2];
            auto build_half = [&](int which) {
                auto half_of = [&](uint64_t u) { return which == 0 ? (u & lo_mask) : (u >> lo_bits); };
                PhaseClock hc;
                hc.on = hc.on && which == 0;
                std::vector<uint32_t> order(h_u.size());
                for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
                if (which == 0) {
                    // by low half, then by value (= by high half): one radix sort on the device of the k-mers with their halves swapped
                    // (an indirect std::sort on 32 host threads was 1.5 s of a 100-strain create); which == 1: h_u is sorted by value,
                    // hence by its high half, then by value
                    bool on_device = false;
                    if (order.size() >= (1u << 16) && hipSetDevice(e->device) == hipSuccess) {
                        const int hi_bits = 2 * k - lo_bits;
                        std::vector<unsigned long long> keys(h_u.size());
                        parallel_for(h_u.size(), [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) keys[i] = ((h_u[i] & lo_mask) << hi_bits) | (h_u[i] >> lo_bits); });
                        on_device = bk::device_sort_order(keys.data(), keys.size(), 2 * k, order.data(), nullptr) == hipSuccess;
                    }
                    if (!on_device)
                        parallel_sort(order, [&](uint32_t x, uint32_t y) {
                            const uint64_t hx = half_of(h_u[x]), hy = half_of(h_u[y]);
                            return hx != hy ? hx < hy : h_u[x] < h_u[y];
                        }, sort_threads);
                }
                hc.lap("  half 0: order sorted");
                std::vector<bk::NbEntry>& cand = hh[which].cand;
                cand.resize(order.size());
                parallel_for(order.size(), [&](size_t i0, size_t i1) {   // (the gather through `order` is what costs: host threads)
                    for (size_t i = i0; i < i1; i++)
                        cand[i] = bk::NbEntry{h_u[order[i]], row_base[order[i]], (h_valid[order[i]] & 0x7fffffffu) | (first_rc[order[i]] ? 0x80000000u : 0u)};
                });
                std::vector<uint64_t> halves;
                std::vector<uint32_t> first, count;
                for (size_t i = 0; i < order.size(); i++) {
                    const uint64_t hf = half_of(cand[i].u);
                    if (halves.empty() || halves.back() != hf) { halves.push_back(hf); first.push_back((uint32_t)i); count.push_back(0); }
                    count.back()++;
                }
                hc.lap("  half 0: candidates gathered, halves listed");
                bk_engine::HalfBufs& hb = which == 0 ? e->half_lo : e->half_hi;
                {   // the presence filter of this half (bk_device.h HalfView::bits): exact up to 24 bits, hashed above (16 bits per half-key: 6 % false "present")
                    const int half_bits = which == 0 ? lo_bits : 2 * k - lo_bits;
                    hb.bits_exact = half_bits <= 24 ? 1u : 0u;
                    uint32_t l2 = (uint32_t)half_bits;
                    if (!hb.bits_exact) { l2 = 16; while (l2 < 28 && (1ull << l2) < 16ull * halves.size()) l2++; }
                    hb.bits_log2 = std::max<uint32_t>(l2, 5);
                    hh[which].bits.assign((size_t)1 << (hb.bits_log2 - 5), 0u);
                    for (uint64_t hf : halves) { const uint32_t b = bk::half_bit_index(hf, hb.bits_log2, hb.bits_exact); hh[which].bits[b >> 5] |= 1u << (b & 31u); }
                }
                std::vector<uint32_t> hpos;
                if (!build_phf(halves, hh[which].hp, hb.log2nb, hb.m, hb.log2p, hpos)) { hh[which].ok = false; return; }
                hh[which].dir.assign((size_t)hb.m << hb.log2p, bk::HalfDir{0u, 0u, 0u, 0u});
                hc.lap("  half 0: perfect hash");
                for (size_t i = 0; i < halves.size(); i++) hh[which].dir[hpos[i]] = bk::HalfDir{(uint32_t)halves[i], first[i], count[i], 0u};
                hc.lap("  half 0: directory");
            };
            std::thread t0(build_half, 0);
            build_half(1);
            t0.join();
            pc.lap("  halves built");
            for (int which = 0; which < 2; which++) {
                if (!hh[which].ok) return fail(BK_ERR_HIP, "internal error: perfect hash construction failed after every fallback");
                bk_engine::HalfBufs& hb = which == 0 ? e->half_lo : e->half_hi;
                BK_HIP(hb.pilots.upload(hh[which].hp));
                BK_HIP(hb.dir.upload(hh[which].dir));
                BK_HIP(hb.cand.upload(hh[which].cand));
                BK_HIP(hb.bits.upload(hh[which].bits));
            }
        }

        pc.lap("half-key directories");
        // slot_of[id*W + t]: the window bucket (wstart+t, u masked) of reference k-mer id -- every reference k-mer
        // owns all of its buckets, so finalize needs no table probe for them (pseudo k-mers: empty_slot where none).
        HostVec<uint32_t> h_slot_of;
        if (slots_on_device) {   // laid out by id on the device, where the table stays; the host phases below read a copy
            BK_HIP(e->slot_of.alloc(h_u.size() * (size_t)e->W));
            BK_HIP(bk::device_permute_rows(d_slot_by_index.p, id_of.data(), h_u.size(), e->W, e->slot_of.p));
            BK_HIP(d_slot_by_index.alloc(0));   // (freed)
            h_slot_of = HostVec<uint32_t>(h_u.size() * (size_t)e->W);
            BK_HIP(hipMemcpy(h_slot_of.data(), e->slot_of.p, h_slot_of.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        } else {
        h_slot_of = filled((size_t)std::max<size_t>(h_u.size(), 1) * std::max(e->W, 1), empty_slot);
        parallel_for(h_u.size(), [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++)
                for (int t = 0; t < e->W; t++) h_slot_of[(size_t)id_of[i] * e->W + t] = slot_by_index[i * e->W + t];
        });
        HostVec<uint32_t>().swap(slot_by_index);
        BK_HIP(e->slot_of.upload(h_slot_of));
        }
        pc.lap("  slot_of filled, uploaded");
        {
            std::vector<uint8_t> h_all_own, h_own_mirror;   // by id (file bitmaps only)
            std::vector<bk::SlotRec> h_rec((size_t)std::max<size_t>(e->n_full, 1) * std::max(e->W, 1));
            parallel_for(e->n_full, [&](size_t id0, size_t id1) {
                for (size_t id = id0; id < id1; id++)
                    for (int t = 0; t < e->W; t++) {
                        const uint32_t sl = h_slot_of[id * e->W + t];
                        bk::SlotRec r{};
                        r.off = h_off[sl]; r.len = h_len[sl];
                        if (r.len) r.first = h_ent[r.off];
                        h_rec[id * e->W + t] = r;
                    }
            });
            BK_HIP(e->slot_rec.upload(h_rec));
            pc.lap("  slot_rec");
            // Which genome files a bucket holds, as a bitmap (IndexView::ent_files / slot_files): with up to 128 files, for buckets
            // that hold at most one BucketInfo per file -- the rule with many related genomes.  The statistics pass of
            // pileup_selected_only then tallies a k-mer's genomes without reading its ~100 entries, and the voting pass finds the
            // selected genome's entry by a popcount instead of a bisection.  All zero = look at the entries.
            if (ix->n_files > 1 && ix->n_files <= 128 && e->W > 1 && !h_len.empty()) {
                std::vector<uint4> h_ef(h_len.size(), make_uint4(0u, 0u, 0u, 0u));
                parallel_for(h_len.size(), [&](size_t s0, size_t s1) {
                    for (size_t sl = s0; sl < s1; sl++) {
                        uint32_t w[4] = {0u, 0u, 0u, 0u};
                        bool ok = h_len[sl] > 0;
                        for (uint32_t q = 0; q < h_len[sl] && ok; q++) {
                            const uint32_t f = h_ent[h_off[sl] + q].file;
                            ok = f < 128u && (q == 0 || f > h_ent[h_off[sl] + q - 1].file);   // sorted by file, one entry each
                            w[(f >> 5) & 3u] |= 1u << (f & 31u);
                        }
                        if (ok) h_ef[sl] = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                });
                std::vector<uint4> h_sf((size_t)std::max<size_t>(e->n_full, 1) * e->W, make_uint4(0u, 0u, 0u, 0u));
                parallel_for(e->n_full, [&](size_t id0, size_t id1) {
                    for (size_t id = id0; id < id1; id++)
                        for (int t = 0; t < e->W; t++) {
                            const uint32_t sl = h_slot_of[id * e->W + t];
                            if (sl != empty_slot) h_sf[id * e->W + t] = h_ef[sl];
                        }
                });
                BK_HIP(e->ent_files.upload(h_ef));
                BK_HIP(e->slot_files.upload(h_sf));
                pc.lap("  file bitmaps");
                // id_own_files (IndexView): bit f of id = in every one of the k-mer's W buckets genome f's only BucketInfo is the k-mer's
                // own occurrence in f -- cell, idx and orientation of bucket t are those of bucket 0, t further on.  The voting pass
                // for the selected genome then needs bucket 0 alone (one load shared by the W lanes of a counter).
                std::vector<uint4> h_own(std::max<size_t>(e->n_full, 1), make_uint4(0u, 0u, 0u, 0u));
                h_own_mirror.assign(e->n_full, 0);
                parallel_for(e->n_full, [&](size_t id0, size_t id1) {
                    for (size_t id = id0; id < id1; id++) {
                        const uint32_t s0 = h_slot_of[id * e->W];
                        if (s0 == empty_slot || !bk::files_any(h_ef[s0])) continue;
                        // along the reference or against it (an occurrence that was reverse-complemented to become canonical): bucket
                        // t's BucketInfo is bucket 0's with cell and idx t further on, or t further back -- one direction per k-mer,
                        // that of its first genome's entries
                        const int dir = h_ent[h_off[s0]].idx == (uint8_t)e->wstart ? 1 : -1;
                        const int idx0 = dir > 0 ? e->wstart : k - 1 - e->wstart;
                        uint32_t w[4] = {h_ef[s0].x, h_ef[s0].y, h_ef[s0].z, h_ef[s0].w};
                        for (uint32_t q0 = 0; q0 < h_len[s0]; q0++) {   // bucket 0's BucketInfo of f is the occurrence of this very k-mer at cell - idx
                            const bk::DevEntry& a0 = h_ent[h_off[s0] + q0];
                            const bool ok = a0.idx == (uint8_t)idx0 && a0.cell >= (uint32_t)idx0 && a0.cell - (uint32_t)idx0 < cells &&
                                            h_id_at[a0.cell - (uint32_t)idx0] == (uint32_t)id;
                            if (!ok) w[(a0.file >> 5) & 3u] &= ~(1u << (a0.file & 31u));
                        }
                        for (int t = 1; t < e->W; t++) {
                            const uint32_t st = h_slot_of[id * e->W + t];
                            if (st == empty_slot || !bk::files_any(h_ef[st])) { w[0] = w[1] = w[2] = w[3] = 0u; break; }
                            w[0] &= h_ef[st].x; w[1] &= h_ef[st].y; w[2] &= h_ef[st].z; w[3] &= h_ef[st].w;
                            // both lists hold one entry per file, sorted: walk them together
                            uint32_t q0 = 0, qt = 0;
                            const uint32_t n0 = h_len[s0], nt_ = h_len[st];
                            while (q0 < n0 && qt < nt_) {
                                const bk::DevEntry& a0 = h_ent[h_off[s0] + q0];
                                const bk::DevEntry& at = h_ent[h_off[st] + qt];
                                if (a0.file < at.file) { ++q0; continue; }
                                if (at.file < a0.file) { ++qt; continue; }
                                if (at.cell != a0.cell + (uint32_t)(dir * t) || at.idx != (uint8_t)(a0.idx + dir * t) || at.canonical != a0.canonical)
                                    w[(a0.file >> 5) & 3u] &= ~(1u << (a0.file & 31u));
                                ++q0; ++qt;
                            }
                        }
                        h_own[id] = make_uint4(w[0], w[1], w[2], w[3]);
                        h_own_mirror[id] = dir < 0 ? 1 : 0;
                    }
                });
                BK_HIP(e->id_own_files.upload(h_own));
                pc.lap("  own files");
                if (test_env("BK_L2_STATS")) {
                    uint64_t n_own = 0, n_b0 = 0, n_mir = 0, n_any = 0;
                    auto pc4 = [](const uint4& b) { return (uint64_t)(__builtin_popcount(b.x) + __builtin_popcount(b.y) + __builtin_popcount(b.z) + __builtin_popcount(b.w)); };
                    for (size_t id = 0; id < e->n_full; id++) {
                        n_own += pc4(h_own[id]); n_any += bk::files_any(h_own[id]); n_mir += h_own_mirror[id];
                        const uint32_t s0 = h_slot_of[id * e->W];
                        if (s0 != empty_slot) n_b0 += h_len[s0];
                    }
                    fprintf(stderr, "[bk] own files: %llu (k-mer, genome) pairs of %llu in bucket 0; %llu of %llu k-mers with any, %llu against the reference\n",
                            (unsigned long long)n_own, (unsigned long long)n_b0, (unsigned long long)n_any, (unsigned long long)e->n_full, (unsigned long long)n_mir);
                }
                // id_rest (IndexView): per k-mer, the BucketInfos of its W buckets that are NOT its own occurrences -- other k-mers of
                // other genomes that differ at the bucket's position -- as indices into `entries`; what every-genome votes have
                // left to do for a k-mer after finalize_exact_own_kernel.  (k-mers without file bitmaps: no list, own is all zero.)
                {
                    std::vector<uint32_t> h_roff((size_t)e->n_full + 1, 0u);
                    parallel_for(e->n_full, [&](size_t id0, size_t id1) {
                        for (size_t id = id0; id < id1; id++) {
                            uint32_t n = 0;
                            if (bk::files_any(h_own[id]))
                                for (int t = 0; t < e->W; t++) {
                                    const uint4& f = h_sf[id * e->W + t];
                                    n += (uint32_t)(__builtin_popcount(f.x & ~h_own[id].x) + __builtin_popcount(f.y & ~h_own[id].y) + __builtin_popcount(f.z & ~h_own[id].z) + __builtin_popcount(f.w & ~h_own[id].w));
                                }
                            h_roff[id + 1] = n;
                        }
                    });
                    for (size_t id = 0; id < e->n_full; id++) h_roff[id + 1] += h_roff[id];
                    std::vector<uint32_t> h_rest(std::max<size_t>(h_roff[e->n_full], 1), 0u);
                    parallel_for(e->n_full, [&](size_t id0, size_t id1) {
                        for (size_t id = id0; id < id1; id++) {
                            if (!bk::files_any(h_own[id])) continue;
                            uint32_t at = h_roff[id];
                            for (int t = 0; t < e->W; t++) {
                                const uint32_t sl = h_slot_of[id * e->W + t];
                                for (uint32_t q = 0; q < h_len[sl]; q++)
                                    if (!bk::files_has(h_own[id], h_ent[h_off[sl] + q].file)) h_rest[at++] = h_off[sl] + q;
                            }
                        }
                    });
                    BK_HIP(e->id_rest_off.upload(h_roff));
                    BK_HIP(e->id_rest.upload(h_rest));
                    pc.lap("  rest lists");
                }
                // kIdAllOwn: nothing else in any of the k-mer's buckets
                h_all_own.assign(e->n_full, 0);
                parallel_for(e->n_full, [&](size_t id0, size_t id1) {
                    for (size_t id = id0; id < id1; id++) {
                        bool all = bk::files_any(h_own[id]);
                        for (int t = 0; t < e->W && all; t++) {
                            const uint4& f = h_sf[id * e->W + t];
                            all = f.x == h_own[id].x && f.y == h_own[id].y && f.z == h_own[id].z && f.w == h_own[id].w;
                        }
                        h_all_own[id] = all ? 1 : 0;
                    }
                });
            }
            // (which genome file a cell belongs to.  Round 6: for any number of genome files -- the gathered votes of bk_gather.hip need
            // no file bitmap, and with more than 128 files, where there is none, they are what keeps every genome's rows affordable:
            // 250 strains, 68 ms a sample through the BucketInfo lists)
            if (ix->n_files > 1 && ix->n_files <= 65535 && e->W > 1 && !h_len.empty()) {
                std::vector<uint16_t> h_cf(std::max<uint64_t>(cells, 1), 0);
                size_t sq2 = 0;
                for (int f = 0; f < ix->n_files; f++)
                    for (int s2 = 0; s2 < ix->n_seqs[f]; s2++, sq2++) {
                        const uint64_t lo = cell_off[f][s2], hi = lo + ix->seq_lens[sq2];
                        for (uint64_t c = lo; c < hi && c < cells; c++) h_cf[c] = (uint16_t)f;
                    }
                BK_HIP(e->cell_file.upload(h_cf));
            }
            // IdRec: k-mer, first cell, flags; "simple" = each of the W buckets holds the k-mer's own single occurrence and nothing else
            HostVec<bk::IdRec> h_idrec = filled(std::max<size_t>(h_u.size(), 1), bk::IdRec{bk::kEmptyKey, 0u, 0u});
            parallel_for(h_u.size(), [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) {
                const uint32_t id = id_of[i];
                bk::IdRec r{h_u[i], first_cell[i] == kNone ? 0u : first_cell[i], (h_amb[id] ? bk::kIdDirty : 0u) | (first_rc[i] ? bk::kIdRc : 0u)};
                if (id < e->n_full && first_cell[i] != kNone && e->W > 0) {
                    bool simple = true;
                    for (int t = 0; t < e->W && simple; t++) {
                        const bk::SlotRec& sr = h_rec[(size_t)id * e->W + t];
                        simple = sr.len == 1 && sr.first.cell == first_cell[i] + (uint32_t)(e->wstart + t) && sr.first.idx == (uint8_t)(e->wstart + t) &&
                                 sr.first.canonical == (first_rc[i] ? 1 : 0);
                    }
                    if (simple) r.flags |= bk::kIdSimple | ((uint32_t)h_rec[(size_t)id * e->W].first.file << 16);
                }
                if (id < h_all_own.size() && h_all_own[id]) r.flags |= bk::kIdAllOwn;
                if (id < h_own_mirror.size() && h_own_mirror[id]) r.flags |= bk::kIdOwnMirror;
                h_idrec[id] = r;
            }
            });
            BK_HIP(e->id_rec.upload(h_idrec));
        }

        pc.lap("slot_of + slot_rec");
        // estat: per reference k-mer, its per-genome hit totals over its W window buckets (call.rs:1316-1318) and
        // hence perfect (== W) / variant -- a property of the index alone
        std::vector<uint32_t> h_estat_off(h_u.size() + 1, 0u), h_estat;
        {
            // per id, independently: chunks on host threads, each with its own list, joined in id order
            const size_t n_ids = h_u.size();
            const unsigned nt = n_ids < 65536 ? 1u : std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 256u);
            // (the reference k-mers -- ids below n_full, each with W buckets of ~70 BucketInfos -- and the pseudo k-mers -- twenty times as
            // many, nearly nothing each -- are cut into nt chunks EACH: a thread takes one of either kind.  Cut as one range, the first
            // twentieth of the threads did all the work.)
            const size_t n_real = std::min<size_t>(e->n_full, n_ids);
            auto cut = [&](unsigned c) -> size_t { return c <= nt ? n_real * c / nt : n_real + (n_ids - n_real) * (c - nt) / nt; };   // chunk c = [cut(c), cut(c + 1)), c < 2 nt
            std::vector<std::vector<uint32_t>> part(2 * nt);
            std::vector<uint32_t> n_of(n_ids, 0u);
            auto work = [&](unsigned t) {
                std::vector<uint32_t> hits(e->n_files, 0u), touched;
                for (unsigned c : {t, nt + t})
                for (size_t id = cut(c); id < cut(c + 1); id++) {
                    touched.clear();
                    for (int w = 0; w < e->W; w++) {
                        const uint32_t sl = h_slot_of[id * e->W + w];
                        for (uint32_t q = 0; q < h_len[sl]; q++) {
                            const uint32_t file = h_ent[h_off[sl] + q].file;
                            if (hits[file]++ == 0) touched.push_back(file);
                        }
                    }
                    for (uint32_t file : touched) {
                        part[c].push_back((file << 1) | (hits[file] == (uint32_t)e->W ? 1u : 0u));
                        hits[file] = 0;
                    }
                    n_of[id] = (uint32_t)touched.size();
                }
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
            work(0);
            for (auto& t : th) t.join();
            for (size_t id = 0; id < n_ids; id++) h_estat_off[id + 1] = h_estat_off[id] + n_of[id];
            h_estat.resize(h_estat_off[n_ids]);
            {   // the chunks' lists back to back (chunk t starts where its first id's list starts), copied side by side
                std::vector<std::thread> cp;
                for (unsigned t = 0; t < nt; t++) cp.emplace_back([&, t] {
                    for (unsigned c : {t, nt + t}) { if (!part[c].empty()) std::copy(part[c].begin(), part[c].end(), h_estat.begin() + (ptrdiff_t)h_estat_off[cut(c)]); std::vector<uint32_t>().swap(part[c]); }
                });
                for (auto& t : cp) t.join();
            }
        }
        BK_HIP(e->estat_off.upload(h_estat_off));
        BK_HIP(e->estat.upload(h_estat));
        if (e->n_files > 1 && e->n_files <= 128 && e->W > 1 && e->n_full > 0) {
            // the same as two bitmaps per reference k-mer (IndexView::estat_files): genomes in which it is perfect, ... a variant
            std::vector<uint4> h_esf((size_t)e->n_full * 2, make_uint4(0u, 0u, 0u, 0u));
            parallel_for(e->n_full, [&](size_t id0, size_t id1) {
                for (size_t id = id0; id < id1; id++)
                    for (uint32_t q = h_estat_off[id]; q < h_estat_off[id + 1]; q++) {
                        const uint32_t f = h_estat[q] >> 1;
                        uint4& b = h_esf[id * 2 + ((h_estat[q] & 1u) ? 0 : 1)];
                        (f < 32u ? b.x : f < 64u ? b.y : f < 96u ? b.z : b.w) |= 1u << (f & 31u);
                    }
            });
            BK_HIP(e->estat_files.upload(h_esf));
        }
    }
    {
        hipDeviceProp_t prop;
        BK_HIP(hipGetDeviceProperties(&prop, prm->device));
        e->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        const size_t budget = bk::scan_lds_budget();
        // LDS holds, for the first n_lds_bins cells (the first genome(s) of the index): the difference array (4 B per cell) and
        // Level 1's copies of the per-cell arrays (2 + 1 bits per cell).  As many cells as fit.
        e->ref_in_lds = true;
        if (const char* rl = test_env("BK_REF_IN_LDS")) e->ref_in_lds = atoi(rl) != 0;
        uint64_t nb = std::min<uint64_t>(e->total_cells, budget / sizeof(unsigned int));
        if (e->ref_in_lds) {
            nb = std::min<uint64_t>(e->total_cells, budget * 32 / 141);   // 4 + 1/4 + 1/8 + 1/32 bytes per cell ...
            while (nb > 0 && nb * sizeof(unsigned int) + bk::scan_ref_lds_bytes((uint32_t)nb) > budget) nb -= std::min<uint64_t>(nb, 64);   // ... and the paddings
        }
        e->n_lds_bins = (uint32_t)nb;
        if (const char* nl = test_env("BK_LDS_BINS")) e->n_lds_bins = std::min<uint32_t>(e->n_lds_bins, (uint32_t)atol(nl));
        if (e->n_lds_bins < e->total_cells) e->n_lds_bins &= ~63u;   // (a window is a whole number of 64-cell blocks unless it holds everything)
    }
    pc.lap("estat + LDS policy");

    if (bk::finalize_lds_bytes(e->n_files) > 160 * 1024) return fail(BK_ERR_UNSUPPORTED, "more than ~8000 genome files are not supported by the finalize kernel");

    if (!e->table.p) BK_HIP(e->table.upload(h_table));
    {
        bool any = false;
        std::vector<uint32_t> bits(h_slot_alias.size() / 32 + 2, 0u);
        for (size_t sl = 0; sl < h_slot_alias.size(); sl++) if (h_slot_alias[sl]) { bits[sl >> 5] |= 1u << (sl & 31); any = true; }
        if (any) BK_HIP(e->slot_alias.upload(bits));
        // the merged buckets' window slots: slot, window position and key of each (bk_gather.hip merged_votes_kernel)
        std::vector<uint64_t> mg;
        for (uint32_t sl : h_merged_slots) { mg.push_back(((uint64_t)h_slot_t[sl] << 32) | sl); mg.push_back(h_slot_key[sl]); }
        e->n_merged_slots = (uint32_t)h_merged_slots.size();
        if (!mg.empty()) BK_HIP(e->merged_slots.upload(mg));
    }
    BK_HIP(e->ent_off.upload(h_off));
    BK_HIP(e->ent_len.upload(h_len));
    BK_HIP(e->entries.upload(h_ent));
    if (int rc = alloc_sample_state(e.get())) return rc;
    BK_HIP(e->d_view.upload(std::vector<bk::IndexView>(1, e->view())));
    if (const char* ab = test_env("BK_SCAN_ABLATE")) e->ablate = atoi(ab);
    if (const char* vm = test_env("BK_ITEM_V_MODE")) e->item_v_mode = atoi(vm);
    if (const char* ml = test_env("BK_MAX_LAUNCH_RECORDS")) e->max_launch_records = strtoull(ml, nullptr, 10);
    pc.lap("uploads + buffers");
    *out = e.release();
    return BK_OK;
}

int bk_engine_fork(const bk_engine* parent, bk_engine** out) { return bk_engine_fork_params(parent, nullptr, out); }

int bk_engine_fork_params(const bk_engine* parent, const bk_params* prm, bk_engine** out) {
    if (!parent || !out) return fail(BK_ERR_INVALID, "null argument");
    if (prm && (prm->n_fixed != parent->params.n_fixed || (prm->use_full_kmer != 0) != (parent->params.use_full_kmer != 0) ||
                prm->device != parent->params.device || (prm->full_kmer_stats != 0) != (parent->params.full_kmer_stats != 0)))
        return fail(BK_ERR_INVALID, "bk_engine_fork_params: n_fixed, use_full_kmer, full_kmer_stats and device shape the shared tables and must equal the parent's");
    if (prm && prm->cs == 0) return fail(BK_ERR_INVALID, "cs must be >= 1");
    BK_HIP(hipSetDevice(parent->device));
    std::unique_ptr<bk_engine> e(new bk_engine());
    const bk_engine* p = parent;
    e->family = p->family; e->family->fetch_add(1);
    e->params = prm ? *prm : p->params; e->k = p->k; e->wstart = p->wstart; e->W = p->W; e->n_files = p->n_files;
    e->total_cells = p->total_cells; e->n_slots = p->n_slots; e->log2s = p->log2s; e->log2nb = p->log2nb; e->log2p = p->log2p; e->m = p->m;
    e->n_u = p->n_u; e->n_full = p->n_full; e->n_lds_bins = p->n_lds_bins; e->n_prows = p->n_prows;
    e->v_omin = p->v_omin; e->v_span = p->v_span; e->v_off = p->v_off; e->plane_len = p->plane_len;
    e->ref_in_lds = p->ref_in_lds; e->lo_bases = p->lo_bases; e->n_cus = p->n_cus; e->device = p->device;
    e->file_cell_lo = p->file_cell_lo; e->max_file_cells_idx = p->max_file_cells_idx; e->ablate = p->ablate; e->item_v_mode = p->item_v_mode; e->max_launch_records = p->max_launch_records;
    e->half_lo.m = p->half_lo.m; e->half_lo.log2nb = p->half_lo.log2nb; e->half_lo.log2p = p->half_lo.log2p;
    e->half_hi.m = p->half_hi.m; e->half_hi.log2nb = p->half_hi.log2nb; e->half_hi.log2p = p->half_hi.log2p;
    e->half_lo.bits_log2 = p->half_lo.bits_log2; e->half_lo.bits_exact = p->half_lo.bits_exact; e->half_lo.bits.alias(p->half_lo.bits);
    e->half_hi.bits_log2 = p->half_hi.bits_log2; e->half_hi.bits_exact = p->half_hi.bits_exact; e->half_hi.bits.alias(p->half_hi.bits);
    // the index tables are immutable after bk_engine_create: the fork reads the parent's
    e->prow_id.alias(p->prow_id); e->prow_t.alias(p->prow_t); e->kmer_pos.alias(p->kmer_pos); e->d_view.alias(p->d_view); e->kmer_of.alias(p->kmer_of); e->id_rec.alias(p->id_rec); e->dirty_ans.alias(p->dirty_ans); e->cell_flags.alias(p->cell_flags);
    e->ref_words.alias(p->ref_words); e->cell_codes.alias(p->cell_codes); e->cell_has.alias(p->cell_has); e->cell_clean.alias(p->cell_clean);
    e->cell_clean3.alias(p->cell_clean3); e->cell_yf.alias(p->cell_yf); e->cell_yr.alias(p->cell_yr); e->id_at.alias(p->id_at);
    e->cell_fast.alias(p->cell_fast); e->cell_nat.alias(p->cell_nat); e->cell_natrow.alias(p->cell_natrow); e->cell_blk.alias(p->cell_blk); e->seed_tab.alias(p->seed_tab); e->seed_log2 = p->seed_log2; e->seed_tab2.alias(p->seed_tab2); e->seed2_log2 = p->seed2_log2; e->rc_words.alias(p->rc_words);
    e->half_lo.pilots.alias(p->half_lo.pilots); e->half_lo.dir.alias(p->half_lo.dir); e->half_lo.cand.alias(p->half_lo.cand);
    e->half_hi.pilots.alias(p->half_hi.pilots); e->half_hi.dir.alias(p->half_hi.dir); e->half_hi.cand.alias(p->half_hi.cand);
    e->slot_of.alias(p->slot_of); e->estat_off.alias(p->estat_off); e->estat.alias(p->estat); e->slot_rec.alias(p->slot_rec); e->ent_files.alias(p->ent_files); e->slot_files.alias(p->slot_files); e->id_own_files.alias(p->id_own_files); e->cell_file.alias(p->cell_file); e->slot_alias.alias(p->slot_alias); e->gather_ok = p->gather_ok; e->merged_slots.alias(p->merged_slots); e->n_merged_slots = p->n_merged_slots; e->id_rest_off.alias(p->id_rest_off); e->id_rest.alias(p->id_rest); e->estat_files.alias(p->estat_files); e->amb.alias(p->amb);
    e->pilots.alias(p->pilots); e->table.alias(p->table); e->ent_off.alias(p->ent_off); e->ent_len.alias(p->ent_len); e->entries.alias(p->entries);
    e->occ.alias(p->occ); e->file_cell_lo_d.alias(p->file_cell_lo_d);
    e->genome_len.alias(p->genome_len); e->seq_cell.alias(p->seq_cell); e->seq_len_d.alias(p->seq_len_d); e->seq_first.alias(p->seq_first);
    e->n_seqs_d.alias(p->n_seqs_d); e->max_seqs_per_file = p->max_seqs_per_file; e->max_file_cells = p->max_file_cells;
    if (int rc = alloc_sample_state(e.get())) return rc;
    *out = e.release();
    return BK_OK;
}

void bk_engine_destroy(bk_engine* e) {
    if (!e) return;
    e->family->fetch_sub(1);
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (auto& s : e->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto ev : e->free_events) (void)hipEventDestroy(ev);
    for (auto& sl : e->slots) {
        if (sl.h_bases) (void)hipHostFree(sl.h_bases);
        if (sl.h_off) (void)hipHostFree(sl.h_off);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    for (auto& o : e->ktab_old) { (void)hipFree(o.first); (void)hipFree(o.second); }
    for (auto& st : e->stage) { if (st.done) (void)hipEventDestroy(st.done); if (st.h) (void)hipHostFree(st.h); }
    if (e->h_fill) (void)hipHostFree(e->h_fill);
    if (e->fill_ev) (void)hipEventDestroy(e->fill_ev);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

int bk_engine_set_stream(bk_engine* e, void* hip_stream) {
    if (!e) return fail(BK_ERR_INVALID, "null engine");
    BK_HIP(hipStreamSynchronize(e->stream));
    e->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : e->own_stream;
    return BK_OK;
}

void* bk_engine_get_stream(const bk_engine* e) { return e ? reinterpret_cast<void*>(e->stream) : nullptr; }

uint64_t bk_total_cells(const bk_engine* e) { return e ? e->total_cells : 0; }
int32_t bk_n_files(const bk_engine* e) { return e ? e->n_files : 0; }
uint64_t bk_n_slots(const bk_engine* e) { return e ? e->n_slots : 0; }
uint64_t bk_counter_len(const bk_engine* e) { return e ? e->plane_len : 0; }
int bk_can_shard(const bk_engine* e) { return e && !e->sparse ? 1 : 0; }

int bk_sample_begin(bk_engine* e) {
    if (!e) return fail(BK_ERR_INVALID, "null engine");
    BK_HIP(hipSetDevice(e->device));
    bk_engine::Span sp(e, 2);
    e->plane_stale[0] = e->plane_stale[1] = true;   // a plane is zeroed when its mate file is first pushed (or finalized unpushed)
    e->win_chosen = false;
    // gathered votes (bk_gather.hip) store the rows they own: every genome's rows -- nothing to zero; the selected genome's -- the rows
    // the previous sample wrote are all that is not zero
    const bool sel_rows = e->gather_mode && e->params.pileup_selected_only != 0 && e->n_files > 1;
    bk::launch_zero_small(e->stats.p, e->stats.n, e->kstats.p, e->kstats.n, e->ktab_out.p, e->ktab_out.n, e->present.p, e->present.n,
                          e->n_deferred.p, e->n_deferred.n, e->pileup.p, e->gather_mode ? 0 : e->pileup.n, e->stream);
    if (sel_rows) bk::launch_zero_genome_rows(e->pileup.p, (size_t)e->total_cells * 4, e->file_cell_lo_d.p, e->n_files, (uint32_t)e->total_cells, e->last_sel.p, e->stream);
    if (e->ktab_keys.p) {
        if (!e->ktab_old.empty()) {   // tables the previous sample outgrew
            BK_HIP(hipStreamSynchronize(e->stream));
            for (auto& o : e->ktab_old) { (void)hipFree(o.first); (void)hipFree(o.second); }
            e->ktab_old.clear();
        }
        BK_HIP(hipMemsetAsync(e->ktab_keys.p, 0xff, e->ktab_keys.n * sizeof(unsigned long long), e->stream));
        BK_HIP(hipMemsetAsync(e->ktab_cnt.p, 0, e->ktab_cnt.n * sizeof(unsigned int), e->stream));
        e->fill_known = 0; e->fill_unknown_upper = 0; e->fill_pending = false;
    }
    e->ktab_exchanged = false;
    e->reduced_shards[0] = e->reduced_shards[1] = 0;
    e->pushed_records[0] = e->pushed_records[1] = 0;
    e->in_sample = true;
    e->finalized_mates = 0;
    // items of a sample that was begun and never finalized are nobody's any more; neither are the rows Level 2 noted for them
    e->pending.on = false;
    for (int m = 0; m < 2; m++) {
        e->fuse_off[m] = false;
        if (e->touch_used[m]) { BK_HIP(hipMemsetAsync(e->fuse_touch[m].p, 0, e->fuse_touch[m].n * sizeof(unsigned int), e->stream)); e->touch_used[m] = false; }
    }
    return BK_OK;
}

// the waiting V items of an earlier launch go to their plane after all: bin_count_kernel over the V bins, adding (Level 2 of that
// launch may have written to the plane since)
static int flush_pending_items(bk_engine* e) {
    if (!e->pending.on) return BK_OK;
    bk_engine::Span sp(e, 3);
    bk::BinArgs b = e->pending.b;
    b.part = 2;
    b.v_mode = e->item_v_mode >= 0 && e->item_v_mode != 2 ? e->item_v_mode : 1;
    e->pending.on = false;
    e->fuse_off[e->pending.mate] = true;
    BK_HIP(bk::launch_bin_count(b, e->stream));
    return BK_OK;
}

static int zero_plane_if_stale(bk_engine* e, int mate) {
    if (e->sparse) {
        // the planes are all zero between samples (finalize clears what it read); only a sample that was begun and never
        // finalized leaves something behind
        if (e->plane_stale[mate] && e->plane_used[mate]) {
            bk_engine::Span sp(e, 2);
            BK_HIP(hipMemsetAsync(e->counters[mate].p, 0, std::max<size_t>(e->counters[mate].n, 1) * sizeof(unsigned long long), e->stream));
            BK_HIP(hipMemsetAsync(e->touch_v[mate].p, 0, e->touch_v[mate].n * 4, e->stream));
            BK_HIP(hipMemsetAsync(e->touch_p[mate].p, 0, e->touch_p[mate].n * 4, e->stream));
            BK_HIP(hipMemsetAsync(e->touch_b[mate].p, 0, e->touch_b[mate].n * 4, e->stream));
            BK_HIP(hipMemsetAsync(e->touch_e[mate].p, 0, e->touch_e[mate].n * 4, e->stream));
            e->plane_used[mate] = false;
        }
        e->plane_stale[mate] = false;
        return BK_OK;
    }
    if (e->plane_stale[mate]) {
        if (e->plane_used[mate]) {   // (a sample that was abandoned, or finalized in shards: whole samples leave their planes clean)
            bk_engine::Span sp(e, 2);
            BK_HIP(hipMemsetAsync(e->counters[mate].p, 0, std::max<size_t>(e->counters[mate].n, 1) * sizeof(unsigned long long), e->stream));
            e->plane_used[mate] = false;
        }
        e->plane_stale[mate] = false;
        e->v_clean[mate] = true;   // all zero now: the sample's first bin_count launch stores where it would add
    }
    return BK_OK;
}

// full_kmer_stats: the statistics table holds every distinct k-mer that touches no window bucket -- as many as the sample has
// sequencing errors, unknown in advance.  Its load stays below one half: before a batch of at most `upper` k-mers is pushed,
// the keys it holds (read back from the device tallies after every push; the engine only waits for that reading when the
// bound says the batch might not fit) plus `upper` must fit, else the table is rehashed into one four times larger.
static int ensure_ktab_room(bk_engine* e, uint64_t upper) {
    if (!e->ktab_keys.p) return BK_OK;
    const uint64_t cap = 1ull << e->ktab_log2;
    if (e->fill_known + e->fill_unknown_upper + upper > cap / 2) {
        if (e->fill_pending) {
            BK_HIP(hipEventSynchronize(e->fill_ev));
            uint64_t f = 0;
            for (uint32_t i = 0; i < bk::ktab_fill_words(); i++) f += e->h_fill[i];
            e->fill_known = f; e->fill_unknown_upper = 0; e->fill_pending = false;
        }
        uint32_t nl = e->ktab_log2;
        while (nl < 31 && e->fill_known + e->fill_unknown_upper + upper > (1ull << nl) / 2) nl += 2;
        if (nl > 31) nl = 31;
        if (nl != e->ktab_log2) {
            unsigned long long* nk = nullptr; unsigned int* nc = nullptr;
            BK_HIP(hipMalloc(reinterpret_cast<void**>(&nk), ((size_t)1 << nl) * sizeof(unsigned long long)));
            BK_HIP(hipMalloc(reinterpret_cast<void**>(&nc), ((size_t)1 << nl) * sizeof(unsigned int)));
            BK_HIP(hipMemsetAsync(nk, 0xff, ((size_t)1 << nl) * sizeof(unsigned long long), e->stream));
            BK_HIP(hipMemsetAsync(nc, 0, ((size_t)1 << nl) * sizeof(unsigned int), e->stream));
            bk::launch_ktab_rehash(e->ktab_keys.p, e->ktab_cnt.p, e->ktab_log2, nk, nc, nl, e->ktab_out.p + 4, e->stream);
            e->ktab_old.emplace_back(e->ktab_keys.p, e->ktab_cnt.p);   // still read by the rehash in flight
            e->ktab_keys.p = nk; e->ktab_keys.n = (size_t)1 << nl;
            e->ktab_cnt.p = nc; e->ktab_cnt.n = (size_t)1 << nl;
            e->ktab_log2 = nl;
        }
    }
    e->fill_unknown_upper += upper;
    return BK_OK;
}
static int note_ktab_fill(bk_engine* e) {   // after a push: a fresh copy of the tallies
    if (!e->ktab_keys.p) return BK_OK;
    BK_HIP(hipMemcpyAsync(e->h_fill, e->ktab_out.p + 8, bk::ktab_fill_words() * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    BK_HIP(hipEventRecord(e->fill_ev, e->stream));
    e->fill_pending = true;
    return BK_OK;
}

static int push_device(bk_engine* e, int mate, const uint32_t* d_words, uint32_t stride_words, const uint16_t* d_lens, uint64_t n,
                       const unsigned long long* n_records_dev = nullptr, uint64_t kmers_upper = 0) {
    if (int rc = zero_plane_if_stale(e, mate)) return rc;
    if (int rc = ensure_ktab_room(e, kmers_upper ? kmers_upper : n * (uint64_t)stride_words * 16)) return rc;
    bk::ScanArgs a{};
    a.n_records_dev = n_records_dev;
    a.ixp = e->d_view.p;
    a.k = e->k; a.wstart = e->wstart; a.W = e->W; a.v_omin = e->v_omin; a.v_span = e->v_span; a.v_off = e->v_off; a.total_cells = (uint32_t)e->total_cells; a.n_u = e->n_u;
    a.ref_words = e->ref_words.p; a.cell_codes = e->cell_codes.p; a.cell_has = e->cell_has.p; a.cell_clean = e->cell_clean.p; a.cell_clean3 = e->cell_clean3.p; a.cell_yf = e->cell_yf.p; a.cell_yr = e->cell_yr.p; a.id_at = e->id_at.p; a.cell_fast = e->cell_fast.p; a.cell_nat = e->cell_nat.p; a.cell_natrow = e->cell_natrow.p; a.cell_blk = e->cell_blk.p; a.seed_tab = e->seed_tab.p; a.seed_log2 = e->seed_log2;
    a.seed_tab2 = e->seed_tab2.p; a.seed2_log2 = e->seed2_log2; a.rc_words = e->rc_words.p;
    a.n_direct = e->use_items && e->n_files == 1 && e->max_seqs_per_file == 1 && (uint64_t)e->n_lds_bins >= e->total_cells && !test_env("BK_NO_N_DIRECT");
    a.words = d_words; a.lens = d_lens; a.n_records = n; a.stride_words = stride_words;
    a.counters = e->counters[mate].p;
    a.kmer_total = e->kstats.p + mate * 4 + 1;
    a.ablate = e->ablate;
    a.slabs = e->slabs.p;
    a.n_lds_bins = e->n_lds_bins;
    a.ref_in_lds = e->ref_in_lds ? 1 : 0;
    a.ktab_keys = e->ktab_keys.p; a.ktab_cnt = e->ktab_cnt.p; a.ktab_log2 = e->ktab_log2;
    a.ktab_overflow = e->ktab_out.p + 4; a.mate = (uint32_t)mate;
    a.occ = e->occ.p; a.n_files = e->n_files;
    e->plane_used[mate] = true;
    if (e->sparse) {
        a.touch_v = e->touch_v[mate].p; a.touch_b = e->touch_b[mate].p; a.touch_p = e->touch_p[mate].p; a.touch_e = e->touch_e[mate].p;
        a.rl_recip = ~0ull / (unsigned long long)(e->v_span + 1) + 1ull;   // ceil(2^64 / row length): exact quotients for 32-bit counter indices
    }
    if (test_env("BK_L2_STATS") && !e->dbg.p) { BK_HIP(e->dbg.alloc(32 + 4 * 1024)); BK_HIP(hipMemsetAsync(e->dbg.p, 0, (32 + 4 * 1024) * sizeof(unsigned long long), e->stream)); }
    a.dbg = e->dbg.p;
    if (e->W <= 0) {
        // empty window: nothing can touch the index (map_kmers finds no bucket, call.rs:1291-1307); KMC's total k-mer count is all
        bk::launch_count_kmers(a, e->stream);
        BK_HIP(hipGetLastError());
        if (!n_records_dev) e->pushed_records[mate] += n;
        return note_ktab_fill(e);
    }
    if (e->occ.p && !e->win_chosen && n > 0) {
        // first records of the sample vote for the genome they look like; the LDS window goes on that genome and stays there for
        // the sample.  Vote and choice are made on the device (the scan reads the window from device memory): no host round trip
        // between a sample's first push and its scan.  Any choice gives the same counts -- this is about speed.
        a.win_file = 0; a.win_lo = 0; a.win_dev = nullptr;
        BK_HIP(hipMemsetAsync(e->win_votes.p, 0, e->win_votes.n * sizeof(unsigned int), e->stream));
        int forced = -1;
        if (const char* wf = test_env("BK_WINDOW_FILE")) forced = std::max(0, atoi(wf));   // testing aid
        bk::launch_pick_window(a, 16384, e->win_votes.p, e->file_cell_lo_d.p, forced, e->win_sel.p, e->stream);
        e->win_chosen = true;
    }
    a.win_file = 0; a.win_lo = 0;
    a.win_dev = e->occ.p ? e->win_sel.p : nullptr;
    // a launch takes at most scan_max_records records (bound on what one workgroup's 16-bit LDS bins can receive), and no
    // more than keeps Level 2's bitmap below 1 GiB
    a.l2_words = bk::scan_l2_words(stride_words, e->k);
    const uint64_t l2_cap = std::max<uint64_t>(64, ((1ull << 30) / sizeof(unsigned int)) / a.l2_words);
    if (e->use_items) {
        a.ig = e->ig; a.items = e->items.p; a.tab = e->item_tab.p; a.gext = e->item_gext.p; a.ov = e->ov.p; a.ov_n = e->ov_n.p; a.ov_cap = (uint32_t)e->ov.n;
    }
    for (uint64_t base = 0; base < n;) {
        uint32_t grid = bk::scan_grid(n - base, e->n_cus);
        uint64_t take = std::min<uint64_t>(std::min<uint64_t>(n - base, bk::scan_max_records(grid)), l2_cap);
        if (e->use_items) take = std::min<uint64_t>(n - base, l2_cap);   // (no 16-bit LDS bins to keep from wrapping)
        if (e->max_launch_records) take = std::min<uint64_t>(take, e->max_launch_records);
        if (e->l2_bits.n < take * a.l2_words || e->l2_diag.n < take) {
            BK_HIP(hipStreamSynchronize(e->stream));
            const uint64_t recs = std::min<uint64_t>(std::max<uint64_t>(take + take / 4, 1 << 16), l2_cap);
            BK_HIP(e->l2_bits.alloc((size_t)recs * a.l2_words));
            BK_HIP(e->n_bits.alloc((size_t)recs * a.l2_words));
            BK_HIP(e->l2_diag.alloc((size_t)recs));
            BK_HIP(e->l2_any.alloc((size_t)(recs + 31) / 32));
            BK_HIP(e->n_any.alloc((size_t)(recs + 31) / 32));
            BK_HIP(hipMemsetAsync(e->l2_bits.p, 0, e->l2_bits.n * sizeof(unsigned int), e->stream));
            BK_HIP(hipMemsetAsync(e->l2_any.p, 0, e->l2_any.n * sizeof(unsigned int), e->stream));
            BK_HIP(hipMemsetAsync(e->n_bits.p, 0, e->n_bits.n * sizeof(unsigned int), e->stream));
            BK_HIP(hipMemsetAsync(e->n_any.p, 0, e->n_any.n * sizeof(unsigned int), e->stream));
        }
        a.l2_bits = e->l2_bits.p; a.l2_diag = e->l2_diag.p; a.l2_any = e->l2_any.p; a.n_bits = e->n_bits.p; a.n_any = e->n_any.p;
        // (one genome file, the binned scan, four or more samples in flight: Level 2 on as many workgroups as its marks are worth)
        a.l2_plan = e->use_items && e->n_files == 1 && e->family->load() >= 4 && !test_env("BK_NO_L2_PLAN") ? e->l2_plan.p : nullptr;
        a.l2_min_grid = (uint32_t)std::max(1, e->n_cus / 2);
        a.rec_base = base; a.n_records = take;
        if (int rc = flush_pending_items(e)) return rc;   // (the scan below overwrites the item buffers)
        const bool wait_v = e->fuse_ok && a.n_direct && !e->fuse_off[mate] && e->item_v_mode < 0;
        if (e->use_items) {
            // A scan workgroup fills its CU (16 waves of 128 registers, 127 KB of LDS): on every CU it shuts out the other samples'
            // finalize / Level 2 kernels, which are chains of short launches that wait for latency, not for CUs.  With siblings in
            // flight three quarters of the CUs scan and the rest keep those chains moving (config 2, three samples in flight: 8.35
            // -> 9.06 G reads/s; one sample alone is 4% slower that way and keeps the whole chip).
            // (round 6, tiles dealt as the workgroups come: with four or more samples in flight -- a host that raises HIP's hardware
            // queues from their default of four -- five eighths; 12.1 against 11.3 G reads/s, flat from 8 to 11 sixteenths)
            int share = e->family->load() >= 4 ? e->n_cus * 5 / 8 : e->n_cus - e->n_cus / 4;
            if (const char* fr = test_env("BK_ITEM_SHARE")) share = std::max(1, std::min(e->n_cus, e->n_cus * atoi(fr) / 16));   // measurement aid: sixteenths of the CUs
            grid = bk::items_grid(take, e->family->load() > 1 ? share : e->n_cus);
            if (const char* gr = test_env("BK_ITEM_GRID")) grid = std::max<uint32_t>(1, std::min<uint32_t>(grid, (uint32_t)atoi(gr)));
        }
        {
            bk_engine::Span sp(e, 0);
            if (e->use_items) { a.ov_par = e->ov_par; BK_HIP(bk::launch_scan_items(a, grid, e->stream)); }
            else BK_HIP(bk::launch_scan_count(a, grid, e->stream));
        }
        if (e->use_items) {
            bk_engine::Span sp(e, 3);
            // the scan's items, bin by bin -> u64 plane (before nbatch / level2 add to it: a sample's first launch finds the V part all zero)
            bk::BinArgs b{};
            b.ig = e->ig; b.items = e->items.p; b.tab = e->item_tab.p; b.gext = e->item_gext.p; b.n_wg = grid; b.ov = e->ov.p; b.ov_n = e->ov_n.p; b.ov_cap = (uint32_t)e->ov.n;
            b.ov_par = e->ov_par; e->ov_par ^= 1u;
            b.id_at = e->id_at.p; b.cell_codes = e->cell_codes.p + bk::scan_ref_pad_words(); b.win_lo = a.win_lo; b.win_dev = a.win_dev;
            b.total_cells = (uint32_t)e->total_cells; b.counters = e->counters[mate].p; b.v_off = e->v_off;
            b.v_real_len = bk::v_real_len(e->n_full, e->v_span); b.rl = (uint32_t)e->v_span + 1u;
            b.v_mode = e->item_v_mode >= 0 ? e->item_v_mode : (e->v_clean[mate] ? 2 : 1);
            e->v_clean[mate] = false;
            if (const char* ba = test_env("BK_BIN_ABLATE")) b.ablate = atoi(ba);
            if (wait_v) {
                // the mate file's first launch: its V items wait for the regional finalize (or for the next launch, which sends them
                // to the plane); Level 2 below notes the V rows it writes to
                e->pending.on = true; e->pending.mate = mate; e->pending.b = b;
                b.part = 1;
                a.touch_v = e->fuse_touch[mate].p; e->touch_used[mate] = true;
                a.rl_recip = ~0ull / (unsigned long long)(e->v_span + 1) + 1ull;
            } else {
                e->fuse_off[mate] = true;
                if (e->fuse_ok) a.touch_v = nullptr;   // (a push of several launches: set by the first)
            }
            BK_HIP(bk::launch_bin_count(b, e->stream));
        }
        if (e->W > 0) {
            bk_engine::Span sp(e, 3);
            // the k-mers the scan left marked (it clears the marks it takes)
            if (test_env("BK_L2_COUNT")) {   // debugging aid: how much is left to Level 2
                std::vector<unsigned int> hb((size_t)take * a.l2_words), ha((size_t)(take + 31) / 32);
                BK_HIP(hipMemcpyAsync(hb.data(), e->n_bits.p, hb.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, e->stream));
                BK_HIP(hipMemcpyAsync(ha.data(), e->n_any.p, ha.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, e->stream));
                BK_HIP(hipStreamSynchronize(e->stream));
                uint64_t nk = 0, nr = 0, runs = 0;
                for (size_t i = 0; i < hb.size(); i++) { nk += (uint64_t)__builtin_popcount(hb[i]); runs += (uint64_t)__builtin_popcount(hb[i] & ~(hb[i] << 1)); }
                for (unsigned int w : ha) nr += (uint64_t)__builtin_popcount(w);
                fprintf(stderr, "[bk] left to level 2 by the scan: %llu of %llu records marked, %llu k-mers in %llu N runs (per 32-bit word)\n", (unsigned long long)nr,
                        (unsigned long long)take, (unsigned long long)nk, (unsigned long long)runs);
            }
            if (e->ablate == 1 || e->ablate == 4) {   // measurement aids: without Level 2
                BK_HIP(hipMemsetAsync(e->n_bits.p, 0, (size_t)take * a.l2_words * sizeof(unsigned int), e->stream));
                BK_HIP(hipMemsetAsync(e->n_any.p, 0, e->n_any.n * sizeof(unsigned int), e->stream));
            }
            else BK_HIP(bk::launch_level2(a, e->n_cus, e->stream));
            if (!e->use_items) {
                // per-cell bin slabs -> u64 plane
                bk::FoldArgs f{};
                f.slabs = e->slabs.p; f.n_slabs = grid; f.n_lds_bins = e->n_lds_bins; f.id_at = e->id_at.p; f.cell_codes = e->cell_codes.p + bk::scan_ref_pad_words(); f.win_lo = a.win_lo; f.win_dev = a.win_dev; f.touch_e = a.touch_e;
                f.counters = e->counters[mate].p;
                bk::launch_fold(f, e->stream);
            }
        }
        base += take;
    }
    BK_HIP(hipGetLastError());
    if (!n_records_dev) e->pushed_records[mate] += n;
    return note_ktab_fill(e);
}

int bk_push_reads_ascii(bk_engine* e, int mate, const uint8_t* buf, const uint64_t* offsets, uint64_t n_reads) {
    if (!e) return fail(BK_ERR_INVALID, "null engine");
    if (!e->in_sample) return fail(BK_ERR_STATE, "bk_push_reads_* called before bk_sample_begin");
    if (mate < 0 || mate > 1) return fail(BK_ERR_INVALID, "mate must be 0 or 1");
    if (n_reads == 0) return BK_OK;
    if (!buf || !offsets) return fail(BK_ERR_INVALID, "bad read batch");
    const uint64_t base0 = offsets[0], total = offsets[n_reads] - base0;
    if (total >= (1ull << 32)) return fail(BK_ERR_INVALID, "batch too large: push at most 2^32 bases per call");
    BK_HIP(hipSetDevice(e->device));
    if (!e->copy_stream) BK_HIP(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    bk_engine::IngestSlot& sl = e->slots[e->next_slot];
    e->next_slot = (e->next_slot + 1) % 3;
    if (!sl.done) { BK_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming)); BK_HIP(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming)); }
    if (sl.busy) { BK_HIP(hipEventSynchronize(sl.done)); sl.busy = false; }   // slot still owned by an earlier batch

    // staging copy (the caller's buffer is free as soon as we return) + longest read of the batch
    if (sl.h_bases_cap < total + 1) {
        if (sl.h_bases) BK_HIP(hipHostFree(sl.h_bases));
        sl.h_bases_cap = total + total / 4 + 4096;
        BK_HIP(hipHostMalloc(reinterpret_cast<void**>(&sl.h_bases), sl.h_bases_cap, hipHostMallocDefault));
    }
    if (sl.h_off_cap < n_reads + 1) {
        if (sl.h_off) BK_HIP(hipHostFree(sl.h_off));
        sl.h_off_cap = n_reads + n_reads / 4 + 1024;
        BK_HIP(hipHostMalloc(reinterpret_cast<void**>(&sl.h_off), sl.h_off_cap * sizeof(unsigned long long), hipHostMallocDefault));
    }
    std::memcpy(sl.h_bases, buf + base0, total);
    uint64_t longest = (uint64_t)e->k;
    for (uint64_t i = 0; i <= n_reads; i++) {
        sl.h_off[i] = offsets[i] - base0;
        if (i) longest = std::max(longest, offsets[i] - offsets[i - 1]);
    }
    const uint32_t stride = (uint32_t)std::min<uint64_t>((longest + 15) / 16, 4095);
    const uint64_t maxb = std::min<uint64_t>((uint64_t)stride * 16, 65535);
    const uint64_t cap = n_reads + total / (uint64_t)e->k + total / (maxb - (uint64_t)(e->k - 1)) + 16;   // bound on the records

    if (sl.d_bases.n < total + 1) BK_HIP(sl.d_bases.alloc(total + total / 4 + 4096));
    if (sl.d_off.n < n_reads + 1) BK_HIP(sl.d_off.alloc(n_reads + n_reads / 4 + 1024));
    if (sl.d_words.n < cap * stride) BK_HIP(sl.d_words.alloc(cap * stride + cap * stride / 4));
    if (sl.d_lens.n < cap) BK_HIP(sl.d_lens.alloc(cap + cap / 4));
    if (!sl.d_nrec.p) BK_HIP(sl.d_nrec.alloc(4));
    if (sl.d_work.n < n_reads) BK_HIP(sl.d_work.alloc(n_reads + n_reads / 4 + 1024));

    BK_HIP(hipMemcpyAsync(sl.d_bases.p, sl.h_bases, total, hipMemcpyHostToDevice, e->copy_stream));
    BK_HIP(hipMemcpyAsync(sl.d_off.p, sl.h_off, (n_reads + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, e->copy_stream));
    BK_HIP(hipEventRecord(sl.uploaded, e->copy_stream));
    BK_HIP(hipStreamWaitEvent(e->stream, sl.uploaded, 0));
    {
        bk::PackArgs pa{};
        pa.bases = sl.d_bases.p; pa.offsets = sl.d_off.p; pa.n_reads = n_reads; pa.k = e->k; pa.stride_words = stride;
        pa.words = sl.d_words.p; pa.lens = sl.d_lens.p; pa.cap = cap; pa.n_records = sl.d_nrec.p; pa.work = sl.d_work.p;
        bk_engine::Span sp(e, 2);
        bk::launch_pack_reads(pa, e->kstats.p + mate * 4 + 0, e->stream);   // (records pushed: tallied on the device)
    }
    int rc = push_device(e, mate, sl.d_words.p, stride, sl.d_lens.p, cap, sl.d_nrec.p, total);   // (a batch holds fewer k-mers than bases)
    if (rc != BK_OK) return rc;
    BK_HIP(hipEventRecord(sl.done, e->stream));
    sl.busy = true;
    return BK_OK;
}

int bk_push_reads_ascii_device(bk_engine* e, int mate, const void* d_bases, const void* d_offsets, uint64_t n_reads, uint64_t total_bases,
                               uint32_t longest_read) {
    if (!e) return fail(BK_ERR_INVALID, "null engine");
    if (!e->in_sample) return fail(BK_ERR_STATE, "bk_push_reads_* called before bk_sample_begin");
    if (mate < 0 || mate > 1) return fail(BK_ERR_INVALID, "mate must be 0 or 1");
    if (n_reads == 0) return BK_OK;
    if (!d_bases || !d_offsets) return fail(BK_ERR_INVALID, "bad read batch");
    if (total_bases >= (1ull << 32)) return fail(BK_ERR_INVALID, "batch too large: push at most 2^32 bases per call");
    BK_HIP(hipSetDevice(e->device));
    // (everything is ordered by the engine's stream: the records of the previous batch were consumed by its scan before this
    // batch's packer starts, so one set of record buffers does)
    bk_engine::IngestSlot& sl = e->dev_ascii;
    const uint64_t longest = std::max<uint64_t>(longest_read, (uint64_t)e->k);
    const uint32_t stride = (uint32_t)std::min<uint64_t>((longest + 15) / 16, 4095);
    const uint64_t maxb = std::min<uint64_t>((uint64_t)stride * 16, 65535);
    const uint64_t cap = n_reads + total_bases / (uint64_t)e->k + total_bases / (maxb - (uint64_t)(e->k - 1)) + 16;   // bound on the records
    if (sl.d_words.n < cap * stride || sl.d_lens.n < cap || sl.d_work.n < n_reads) {
        BK_HIP(hipStreamSynchronize(e->stream));
        BK_HIP(sl.d_words.alloc(cap * stride + cap * stride / 4));
        BK_HIP(sl.d_lens.alloc(cap + cap / 4));
        BK_HIP(sl.d_work.alloc(n_reads + n_reads / 4 + 1024));
    }
    if (!sl.d_nrec.p) BK_HIP(sl.d_nrec.alloc(4));
    {
        bk::PackArgs pa{};
        // the packer stages the lines with 16-byte loads from a 16-byte boundary: a pointer into the middle of an allocation (any
        // alignment) is rounded down and the offsets carry the difference (a device allocation starts on a 256-byte boundary, so the
        // bytes in front belong to the same allocation)
        pa.shift = (uint32_t)(reinterpret_cast<uintptr_t>(d_bases) & 15u);
        pa.bases = static_cast<const uint8_t*>(d_bases) - pa.shift; pa.offsets = static_cast<const unsigned long long*>(d_offsets); pa.n_reads = n_reads;
        pa.k = e->k; pa.stride_words = stride;
        pa.words = sl.d_words.p; pa.lens = sl.d_lens.p; pa.cap = cap; pa.n_records = sl.d_nrec.p; pa.work = sl.d_work.p;
        bk_engine::Span sp(e, 2);
        bk::launch_pack_reads(pa, e->kstats.p + mate * 4 + 0, e->stream);   // (records pushed: tallied on the device)
    }
    return push_device(e, mate, sl.d_words.p, stride, sl.d_lens.p, cap, sl.d_nrec.p, total_bases);
}

int bk_push_reads_packed_device(bk_engine* e, int mate, const void* d_words, uint32_t stride_words, const void* d_lens, uint64_t n) {
    if (!e) return fail(BK_ERR_INVALID, "null engine");
    if (!e->in_sample) return fail(BK_ERR_STATE, "bk_push_reads_* called before bk_sample_begin");
    if (mate < 0 || mate > 1) return fail(BK_ERR_INVALID, "mate must be 0 or 1");
    if (n == 0) return BK_OK;
    if (!d_words || !d_lens || stride_words == 0 || stride_words > 4096) return fail(BK_ERR_INVALID, "bad record batch");
    if (n > (1ull << 32) / ((uint64_t)stride_words * 16)) return fail(BK_ERR_INVALID, "batch too large: push at most 2^32 bases per call");
    BK_HIP(hipSetDevice(e->device));
    return push_device(e, mate, static_cast<const uint32_t*>(d_words), stride_words, static_cast<const uint16_t*>(d_lens), n);
}

int bk_push_reads_packed(bk_engine* e, int mate, const uint32_t* words, uint32_t stride_words, const uint16_t* lens, uint64_t n) {
    if (!e) return fail(BK_ERR_INVALID, "null engine");
    if (!e->in_sample) return fail(BK_ERR_STATE, "bk_push_reads_* called before bk_sample_begin");
    if (mate < 0 || mate > 1) return fail(BK_ERR_INVALID, "mate must be 0 or 1");
    if (n == 0) return BK_OK;
    if (!words || !lens || stride_words == 0 || stride_words > 4096) return fail(BK_ERR_INVALID, "bad record batch");
    if (n > (1ull << 32) / ((uint64_t)stride_words * 16)) return fail(BK_ERR_INVALID, "batch too large: push at most 2^32 bases per call");
    BK_HIP(hipSetDevice(e->device));
    const size_t nw = (size_t)n * stride_words;
    bk_engine::StageSlot& sl = e->stage[e->next_stage];
    e->next_stage ^= 1;
    if (!sl.done) BK_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (sl.busy) { BK_HIP(hipEventSynchronize(sl.done)); sl.busy = false; }   // the scan that read this slot two pushes ago
    if (sl.words.n < nw) BK_HIP(sl.words.alloc(nw + nw / 4));
    if (sl.lens.n < n) BK_HIP(sl.lens.alloc(n + n / 4));
    {
        // the caller's buffer is free when this call returns: the batch is copied into the slot's pinned host buffer, from where
        // it travels asynchronously (an asynchronous copy straight from pageable memory would still be reading the caller's pages)
        const size_t bytes_w = nw * sizeof(uint32_t), bytes_l = (size_t)n * sizeof(uint16_t);
        if (sl.h_cap < bytes_w + bytes_l) {
            if (sl.h) BK_HIP(hipHostFree(sl.h));
            sl.h = nullptr;
            sl.h_cap = bytes_w + bytes_l + (bytes_w + bytes_l) / 4;
            BK_HIP(hipHostMalloc(reinterpret_cast<void**>(&sl.h), sl.h_cap, hipHostMallocDefault));
        }
        std::memcpy(sl.h, words, bytes_w);
        std::memcpy(sl.h + bytes_w, lens, bytes_l);
        bk_engine::Span sp(e, 2);
        BK_HIP(hipMemcpyAsync(sl.words.p, sl.h, bytes_w, hipMemcpyHostToDevice, e->stream));
        BK_HIP(hipMemcpyAsync(sl.lens.p, sl.h + bytes_w, bytes_l, hipMemcpyHostToDevice, e->stream));
    }
    int rc = push_device(e, mate, sl.words.p, stride_words, sl.lens.p, n);
    if (rc != BK_OK) return rc;
    BK_HIP(hipEventRecord(sl.done, e->stream));
    sl.busy = true;
    if (test_env("BK_SYNC_PUSH")) BK_HIP(hipStreamSynchronize(e->stream));
    return BK_OK;
}

int bk_counters_device_ptr(bk_engine* e, int mate, void** d_ptr) {
    if (!e || !d_ptr || mate < 0 || mate > 1) return fail(BK_ERR_INVALID, "bad argument");
    if (e->sparse) return fail(BK_ERR_UNSUPPORTED, "an index this large keeps its counter planes sparse: shard whole samples over GPUs, not one sample's reads");
    if (e->pending.on) { BK_HIP(hipSetDevice(e->device)); if (int rc = flush_pending_items(e)) return rc; }
    if (e->in_sample) {   // a mate file nothing was pushed for yet: its plane is zeroed lazily -- now, before the caller reduces it
        BK_HIP(hipSetDevice(e->device));
        if (int rc = zero_plane_if_stale(e, mate)) return rc;
    }
    e->plane_used[mate] = true;   // (the caller may write it: collectives)
    e->v_clean[mate] = false;     // ... so the next bin_count launch adds to the V part instead of storing over it
    *d_ptr = e->counters[mate].p;
    return BK_OK;
}

int bk_pileup_device_ptr(bk_engine* e, void** d_ptr) {
    if (!e || !d_ptr) return fail(BK_ERR_INVALID, "bad argument");
    *d_ptr = e->pileup.p;
    return BK_OK;
}

static int finalize_part(bk_engine* e, int n_mates, uint64_t elem_lo, uint64_t elem_hi) {
    if (!e->in_sample) return fail(BK_ERR_STATE, "bk_sample_finalize called before bk_sample_begin");
    if (n_mates < 1 || n_mates > 2) return fail(BK_ERR_INVALID, "n_mates must be 1 or 2");
    BK_HIP(hipSetDevice(e->device));
    // bk_params.pileup_selected_only (several genome files): first the statistics of every genome without a single vote, then the
    // genome is selected on the device (call.rs:422-502), then the votes -- only the BucketInfos of that genome
    const bool two_pass = e->params.pileup_selected_only != 0 && e->n_files > 1;
    if (two_pass && (elem_lo != 0 || elem_hi != e->plane_len)) return fail(BK_ERR_UNSUPPORTED, "pileup_selected_only cannot be combined with a sharded finalize");
    if (e->sparse && (elem_lo != 0 || elem_hi != e->plane_len)) return fail(BK_ERR_UNSUPPORTED, "an index this large cannot be finalized in shards");
    const bool via_reduced = e->reduced_shards[0] > 0 || e->reduced_shards[1] > 0;   // (the planes themselves are not what is mapped: they are zeroed at the next push)
    const bool clean_dense = !e->sparse && elem_lo == 0 && elem_hi == e->plane_len && !via_reduced;   // this call maps whole planes: it leaves them zeroed
    if (e->sparse) {
        for (int m = 0; m < n_mates; m++) {
            bk_engine::Span sp(e, 1);
            BK_HIP(hipMemsetAsync(e->n_list[m].p, 0, 8 * sizeof(unsigned int), e->stream));
#ifdef BK_TESTING
            if (e->ablate == 15) BK_HIP(hipMemsetAsync(e->touch_b[m].p, 0xff, e->touch_b[m].n * 4, e->stream));   // (15: every block counts as touched)
#endif
            bk::launch_expand_touched_blocks(e->touch_b[m].p, (uint32_t)((e->total_cells + 63) / 64), e->cell_blk.p, e->touch_v[m].p, (uint32_t)e->k,
                                             (uint64_t)e->n_full + (uint64_t)e->v_span, e->stream);
            bk::launch_compact_touched(e->touch_v[m].p, bk::v_real_rows(e->n_full, e->v_span), e->touch_p[m].p, e->n_prows, e->touch_e[m].p, e->n_u,
                                       e->n_full, e->v_list[m].p, e->p_list[m].p, e->e_list[m].p, e->n_list[m].p, e->stream);
        }
    }
    // gathered votes (bk_gather.hip): the statistics pass as ever, then gather_votes_kernel for the selected genome's cells or for all
    const bool gather = e->gather_mode && elem_lo == 0 && elem_hi == e->plane_len && !via_reduced;
    if (gather) BK_HIP(hipMemsetAsync(e->n_alias_hits.p, 0, 2 * sizeof(unsigned int), e->stream));
    for (int pass = 0; pass < ((two_pass || gather) ? 2 : 1); pass++) {
        auto dbg_sync = [&](const char* what) -> int {   // testing build, BK_SYNC_DEBUG: which launch of the gathered voting pass faults
            if (test_env("BK_SYNC_DEBUG")) { fprintf(stderr, "[bk] %s ...", what); BK_HIP(hipStreamSynchronize(e->stream)); fprintf(stderr, " ok\n"); }
            return BK_OK;
        };
        if (gather && pass == 1) {   // difference arrays -> counts (the rows are zeroed behind the sample)
            bk_engine::Span sp(e, 1);
            if (int rc = dbg_sync("statistics pass")) return rc;
            // (every genome's rows by the table of voters: which V rows the sample's mate files touched, as bits)
            unsigned int* row_bits = nullptr;
            if (e->row_bits.p) {   // (allocated with the engine: alloc_sample_state)
                BK_HIP(hipMemsetAsync(e->row_bits.p, 0, e->row_bits.n * sizeof(unsigned int), e->stream));
                row_bits = e->row_bits.p;
            }
            for (int m = 0; m < n_mates; m++) bk::launch_prefix_rows(e->counters[m].p, e->view(), e->v_list[m].p, e->n_list[m].p, row_bits, e->stream);
            if (int rc = dbg_sync("prefix_rows")) return rc;
        }
        for (int m = 0; m < n_mates; m++) {   // R1 then R2 into the same arrays (call.rs:316-317)
            bk::FinalizeArgs a{};
            a.ix = e->view();
            a.counters = e->counters[m].p;
            // (a part that came through bk_shard_received lives in its own buffer: element i of the plane is reduced[i - elem_lo])
            if (e->reduced_shards[m] > 0) a.counters = e->reduced[m].p - elem_lo;
            a.elem_lo = elem_lo; a.elem_hi = elem_hi;
            a.ci = e->params.ci; a.cs = e->params.cs; a.cx = e->params.cx;
            a.pileup = e->pileup.p;
            a.plane = (size_t)e->total_cells * 4;
            a.stats = e->stats.p + (size_t)m * e->n_files * 3;
            a.present = e->present.p + (size_t)m * e->n_files;
            a.kept_total = e->kstats.p + m * 4 + 3;
            a.distinct_total = e->kstats.p + m * 4 + 2;
            a.partials = e->fin_partials.p;
            a.deferred = e->deferred.p + (two_pass ? (size_t)m * (e->deferred.n / 2) : 0);   // (kept from the first pass to the second)
            a.file_cell_lo = e->file_cell_lo_d.p; a.max_file_cells = (uint32_t)e->max_file_cells_idx;
            a.n_deferred = e->n_deferred.p + m;
            a.deferred_mask = two_pass && e->deferred_mask.p ? e->deferred_mask.p + (size_t)m * (e->deferred_mask.n / 2) : nullptr;
            a.ktab_keys = e->ktab_keys.p; a.ktab_cnt = e->ktab_cnt.p; a.ktab_log2 = e->ktab_log2;
            a.ktab_overflow = e->ktab_out.p + 4; a.mate = (uint32_t)m;
            if (e->sparse) { a.v_list = e->v_list[m].p; a.p_list = e->p_list[m].p; a.e_list = e->e_list[m].p; a.n_list = e->n_list[m].p; }
            a.deferred_n = e->deferred_n.p ? e->deferred_n.p + (two_pass ? (size_t)m * (e->deferred_n.n / 2) : 0) : nullptr;
            a.clear_v = clean_dense && pass == (two_pass ? 1 : 0);
            a.mode = two_pass ? pass + 1 : gather ? (pass == 0 ? 1 : 3) : 0;
            if (gather) if (const char* ga = test_env("BK_GATHER_ABLATE")) a.gather_ablate = atoi(ga);
            if (gather) { a.merged_slots = e->merged_slots.p; a.n_merged_slots = e->n_merged_slots; }
            if (gather) { a.gather = pass == 1 ? 1 : 0; a.alias_hits = e->alias_hits[m].p; a.n_alias_hits = e->n_alias_hits.p + m; a.alias_cap = bk_engine::kAliasCap; }
            a.sel = two_pass ? &e->sel_out.p->file_id : nullptr;
            a.sel_file = -1;
            // dense planes mapped whole: K2a zeroes the V counters it reads; the E part (two counters per reference k-mer) is zeroed by
            // the reduce kernel of the mate file's last statistics pass (it runs behind K2e, the E part's only reader in that pass;
            // a second, votes-only pass reads it again: then the memset below does it)
            const bool ride = clean_dense && !two_pass && e->fin_partials.p && e->plane_used[m];
            a.no_lean = test_env("BK_NO_LEAN_FINALIZE") != nullptr;
            a.lean_e_list = e->lean_e_list.p; a.lean_n_list = e->lean_n_list.p;
            a.zero_e = ride ? e->counters[m].p : nullptr;
            a.zero_e_n = ride ? (size_t)std::min<uint64_t>(e->v_off, e->plane_len) : 0;
            if (ride) e->plane_used[m] = false;
            if (pass == 0) { if (int rc = zero_plane_if_stale(e, m)) return rc; }
            if (e->pending.on && e->pending.mate == m) {
                // the mate file's reads were one launch: the regional finalize takes the V counts from the scan's items
                const bk::BinArgs& pb = e->pending.b;
                a.f_items = pb.items; a.f_tab = pb.tab; a.f_gext = pb.gext; a.f_ov = pb.ov; a.f_ov_n = pb.ov_n; a.f_ov_cap = pb.ov_cap; a.f_ov_par = pb.ov_par;
                a.f_n_wg = pb.n_wg; a.f_ig = pb.ig; a.f_touch = e->fuse_touch[m].p;
                if (clean_dense && !two_pass && bk::finalize_runs_by_region(a)) { e->pending.on = false; e->touch_used[m] = false; }
                else { a.f_items = nullptr; if (int rc = flush_pending_items(e)) return rc; }
            }
            bk_engine::Span sp(e, 1);
            if (gather && pass == 1 && m == 0) {   // (both mate files' counts at once: it stores)
                const unsigned long long* c1 = n_mates == 2 ? e->counters[1].p : nullptr;
                // many genomes that share their k-mers: the voters once per (k-mer, window position), not once per occurrence
                const bool by_table = bk::vote_table_fits(a) && e->row_bits.p && e->vote_tab.n >= bk::vote_table_words(a.ix);
                if (by_table) bk::launch_gather_votes_table(a, c1, e->vote_tab.p, e->row_bits.p, e->stream);
                else bk::launch_gather_votes(a, c1, e->stream);
                if (int rc = dbg_sync("gather_votes")) return rc;
            }
            if (gather && pass == 1 && e->n_merged_slots) { bk::launch_merged_votes(a, e->stream); if (int rc = dbg_sync("merged_votes")) return rc; }
            bk::launch_finalize(a, e->stream);
            if (gather && pass == 1) if (int rc = dbg_sync("alias-only general kernels")) return rc;
        }
        if (two_pass && pass == 0) {
            bk::CallArgs c{};
            c.n_files = e->n_files; c.n_mates = n_mates; c.stats = e->stats.p; c.present = e->present.p; c.genome_len = e->genome_len.p;
            c.out = e->sel_out.p;
            bk_engine::Span sp(e, 1);
            bk::launch_select_genome(c, e->stream);
            if (gather) bk::launch_copy_int(e->last_sel.p, &e->sel_out.p->file_id, e->stream);   // (the rows the next sample zeroes)
        }
    }
    if (clean_dense) {   // K2a zeroed the V counters; the E part (two counters per reference k-mer) goes here
        for (int m = 0; m < n_mates; m++) {
            if (e->plane_used[m]) {
                bk_engine::Span sp(e, 2);
                BK_HIP(hipMemsetAsync(e->counters[m].p, 0, (size_t)std::min<uint64_t>(e->v_off, e->plane_len) * sizeof(unsigned long long), e->stream));
            }
            e->plane_used[m] = false;
        }
    }
    if (e->sparse) {   // the maps are done: what they read is zeroed again, the planes are all zero for the next sample
        for (int m = 0; m < n_mates; m++) {
            bk_engine::Span sp(e, 1);
            bk::launch_clear_touched(e->counters[m].p, e->v_off, bk::v_real_len(e->n_full, e->v_span), (uint32_t)e->v_span + 1u, e->v_list[m].p,
                                     e->p_list[m].p, e->e_list[m].p, e->n_list[m].p, e->n_u, e->stream);
            e->plane_used[m] = false;
        }
    }
    if (e->ktab_keys.p) {
        bk_engine::Span sp(e, 1);
        bk::launch_ktab_stats(e->ktab_keys.p, e->ktab_cnt.p, e->ktab_log2, e->params.ci, e->params.cx, e->ktab_out.p, e->stream);
    }
    BK_HIP(hipGetLastError());
    e->in_sample = false;
    e->finalized_mates = n_mates;
    if (e->dbg.p) {   // BK_L2_STATS (testing build)
        unsigned long long h[32];
        unsigned int nd[2] = {0, 0};
        BK_HIP(hipMemcpyAsync(nd, e->n_deferred.p, sizeof nd, hipMemcpyDeviceToHost, e->stream));
        std::vector<unsigned long long> clk(4 * 1024);
        BK_HIP(hipMemcpyAsync(h, e->dbg.p, sizeof h, hipMemcpyDeviceToHost, e->stream));
        BK_HIP(hipMemcpyAsync(clk.data(), e->dbg.p + 32, clk.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
        BK_HIP(hipMemsetAsync(e->dbg.p, 0, (32 + 4 * 1024) * sizeof(unsigned long long), e->stream));
        BK_HIP(hipStreamSynchronize(e->stream));
        {   // the scan's workgroups on the clock (the sample's last launch): when each started, had its reference, ran out of tiles, ended
            unsigned long long t0 = ~0ull;
            int n_wg = 0;
            for (int b = 0; b < 512; b++) if (clk[4 * b]) { t0 = std::min(t0, clk[4 * b]); n_wg = b + 1; }   // (the second half holds the prologue clocks)
            if (n_wg) {
                double mx[4] = {0, 0, 0, 0}, mean[4] = {0, 0, 0, 0}, mn[4] = {1e30, 1e30, 1e30, 1e30};
                for (int b = 0; b < n_wg; b++)
                    for (int j = 0; j < 4; j++) {
                        const double us = (double)(clk[4 * b + j] - t0) * 0.01;
                        mx[j] = std::max(mx[j], us); mn[j] = std::min(mn[j], us); mean[j] += us / n_wg;
                    }
                fprintf(stderr, "[bk] scan workgroups (%d), us after the first start, min / mean / max: start %.1f / %.1f / %.1f, reference staged %.1f / %.1f / %.1f, "
                        "tiles done %.1f / %.1f / %.1f, end %.1f / %.1f / %.1f\n", n_wg, mn[0], mean[0], mx[0], mn[1], mean[1], mx[1], mn[2], mean[2], mx[2], mn[3], mean[3], mx[3]);
                {
                    double m2[4] = {0, 0, 0, 0};
                    int n2 = 0;
                    for (int b = 0; b < std::min(n_wg, 512); b++) if (clk[2048 + 4 * b]) { n2++; for (int j = 0; j < 4; j++) m2[j] += (double)(clk[2048 + 4 * b + j] - t0) * 0.01; }
                    if (n2) fprintf(stderr, "[bk]   ... mean: first tile's copy sent %.1f, window's loads stored %.1f, wave 0 has its first tile %.1f, buckets written out %.1f\n",
                                    m2[0] / n2, m2[1] / n2, m2[2] / n2, m2[3] / n2);
                }
                if (test_env("BK_L2_STATS")[0] == '2')
                    for (int b = 0; b < n_wg; b++) fprintf(stderr, "[bk]   wg %d: %.1f %.1f %.1f %.1f\n", b, (clk[4 * b] - t0) * 0.01, (clk[4 * b + 1] - t0) * 0.01, (clk[4 * b + 2] - t0) * 0.01, (clk[4 * b + 3] - t0) * 0.01);
            }
        }
        fprintf(stderr, "[bk] finalize: %u + %u k-mers deferred to the general kernel\n", nd[0], nd[1]);
        if (e->gather_mode) {
            unsigned int ah[2] = {0u, 0u};
            BK_HIP(hipMemcpy(ah, e->n_alias_hits.p, sizeof ah, hipMemcpyDeviceToHost));
            fprintf(stderr, "[bk] votes gathered cell by cell (bk_gather.hip); alias hits among the deferred k-mers: %u + %u\n", ah[0], ah[1]);
        }
        if (e->sparse) {
            unsigned int nl[8];
            BK_HIP(hipMemcpy(nl, e->n_list[0].p, sizeof nl, hipMemcpyDeviceToHost));
            fprintf(stderr, "[bk] sparse finalize (mate file 0): %u V rows of %llu, %u pseudo rows of %llu, %u reference k-mers of %u and %u pseudo k-mers of %u touched\n",
                    nl[0], (unsigned long long)bk::v_real_rows(e->n_full, e->v_span), nl[4], (unsigned long long)e->n_prows, nl[2], e->n_full, nl[3], e->n_u - e->n_full);
        }
        fprintf(stderr, "[bk] scan: %llu mismatches counted, %llu items processed, %llu E gaps before a mismatch, %llu behind the last\n", h[23], h[24], h[25], h[26]);
        fprintf(stderr, "[bk] scan N batches: %llu with %llu pieces (%.1f per batch), %llu of them forced by a tile's end\n", h[20], h[21], h[20] ? (double)h[21] / (double)h[20] : 0.0, h[22]);
        fprintf(stderr, "[bk] scan marked: no-diagonal %llu, dirty-head %llu, clean-head %llu, pairs %llu | level 2: k-mers %llu in %llu chunks, simple %llu, dead %llu, "
                "dirty answers %llu (one difference but id unknown: %llu), neither half present %llu, slow %llu (diffs 0/1/2/3+ with a diagonal: %llu/%llu/%llu/%llu) -> member %llu, neighbour %llu, nothing %llu\n",
                h[0], h[1], h[2], h[3], h[4], h[11], h[5], h[6], h[16], h[17], h[18], h[7], h[12], h[13], h[14], h[15], h[8], h[9], h[10]);
    }
    return BK_OK;
}

int bk_sample_finalize(bk_engine* e, int n_mates) {
    if (!e) return fail(BK_ERR_INVALID, "null engine");
    if (e->reduced_shards[0] > 1 || e->reduced_shards[1] > 1) return fail(BK_ERR_STATE, "this sample's planes went through bk_shard_transport: finalize it with bk_sample_finalize_shard");
    return finalize_part(e, n_mates, 0, e->plane_len);
}

int bk_sample_finalize_shard(bk_engine* e, int n_mates, int shard, int n_shards) {
    if (!e) return fail(BK_ERR_INVALID, "null engine");
    if (n_shards < 1 || (int)bk::kMaxShards % n_shards != 0 || shard < 0 || shard >= n_shards)
        return fail(BK_ERR_INVALID, "n_shards must divide %u and 0 <= shard < n_shards", bk::kMaxShards);
    if (e->ktab_keys.p && n_shards > 1 && !e->ktab_exchanged)
        return fail(BK_ERR_STATE, "full_kmer_stats with a sharded finalize: exchange the ranks' k-mer statistics tables first "
                                  "(bk_kmer_table_partition, all-to-all, bk_kmer_table_replace)");
    const uint64_t part = e->plane_len / (uint64_t)n_shards;
    for (int m = 0; m < n_mates; m++)
        if (e->reduced_shards[m] > 0 && (e->reduced_shards[m] != n_shards || e->reduced_shard[m] != shard))
            return fail(BK_ERR_STATE, "bk_sample_finalize_shard(%d of %d): mate file %d received part %d of %d (bk_shard_received)", shard, n_shards, m,
                        e->reduced_shard[m], e->reduced_shards[m]);
    int rc = finalize_part(e, n_mates, part * shard, part * (shard + 1));
    if (rc != BK_OK) return rc;
    if (e->ktab_keys.p) bk::launch_ktab_totals_to_kstats(e->ktab_out.p, e->kstats.p, n_mates, e->stream);   // (the ranks' totals add up)
    for (int m = 0; m < n_mates; m++) {   // the records this rank pushed join the device tally, so that the sum over ranks is the sample's
        if (e->pushed_records[m]) bk::launch_add_const_u64(e->kstats.p + m * 4 + 0, e->pushed_records[m], e->stream);
        e->pushed_records[m] = 0;
    }
    bk::launch_pack_sums(e->shard_sums.p, e->stats.p, e->present.p, e->kstats.p, e->n_files, e->xport_flag.p, e->stream);
    BK_HIP(hipGetLastError());
    return BK_OK;
}

int bk_kmer_table_partition(bk_engine* e, int n_parts, void** d_keys, void** d_counts, uint64_t* part_off) {
    if (!e || !d_keys || !d_counts || !part_off) return fail(BK_ERR_INVALID, "null argument");
    if (n_parts < 1 || n_parts > (int)bk::kMaxShards) return fail(BK_ERR_INVALID, "1 <= n_parts <= %u", bk::kMaxShards);
    if (!e->ktab_keys.p) return fail(BK_ERR_STATE, "the engine was created without full_kmer_stats");
    if (!e->in_sample) return fail(BK_ERR_STATE, "bk_kmer_table_partition comes between the pushes and the finalize of a sample");
    BK_HIP(hipSetDevice(e->device));
    if (!e->xchg_cursors.p) BK_HIP(e->xchg_cursors.alloc(bk::kMaxShards));
    BK_HIP(hipMemsetAsync(e->xchg_cursors.p, 0, bk::kMaxShards * sizeof(unsigned long long), e->stream));
    bk::launch_ktab_count_parts(e->ktab_keys.p, e->ktab_log2, (uint32_t)n_parts, e->xchg_cursors.p, e->stream);
    unsigned long long counts[bk::kMaxShards];
    BK_HIP(hipMemcpyAsync(counts, e->xchg_cursors.p, (size_t)n_parts * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    BK_HIP(hipStreamSynchronize(e->stream));
    unsigned long long first[bk::kMaxShards];
    part_off[0] = 0;
    for (int r = 0; r < n_parts; r++) { first[r] = part_off[r]; part_off[r + 1] = part_off[r] + counts[r]; }
    const uint64_t total = part_off[n_parts];
    if (e->xchg_keys.n < total || !e->xchg_keys.p) {
        BK_HIP(e->xchg_keys.alloc(std::max<uint64_t>(total + total / 4, 1024)));
        BK_HIP(e->xchg_cnt.alloc(e->xchg_keys.n));
    }
    BK_HIP(hipMemcpyAsync(e->xchg_cursors.p, first, (size_t)n_parts * sizeof(unsigned long long), hipMemcpyHostToDevice, e->stream));
    bk::launch_ktab_scatter_parts(e->ktab_keys.p, e->ktab_cnt.p, e->ktab_log2, (uint32_t)n_parts, e->xchg_cursors.p, e->xchg_keys.p, e->xchg_cnt.p, e->stream);
    BK_HIP(hipGetLastError());
    BK_HIP(hipStreamSynchronize(e->stream));   // (`first` is read by the copy above; the caller reads the arrays on its own stream)
    *d_keys = e->xchg_keys.p; *d_counts = e->xchg_cnt.p;
    return BK_OK;
}

int bk_kmer_table_replace(bk_engine* e, const void* d_keys, const void* d_counts, uint64_t n) {
    if (!e || (n && (!d_keys || !d_counts))) return fail(BK_ERR_INVALID, "null argument");
    if (!e->ktab_keys.p) return fail(BK_ERR_STATE, "the engine was created without full_kmer_stats");
    if (!e->in_sample) return fail(BK_ERR_STATE, "bk_kmer_table_replace comes between the pushes and the finalize of a sample");
    BK_HIP(hipSetDevice(e->device));
    // an empty table with room for the n entries (and, like after any push, for what finalize adds: the load stays below a half)
    BK_HIP(hipMemsetAsync(e->ktab_keys.p, 0xff, e->ktab_keys.n * sizeof(unsigned long long), e->stream));
    BK_HIP(hipMemsetAsync(e->ktab_cnt.p, 0, e->ktab_cnt.n * sizeof(unsigned int), e->stream));
    BK_HIP(hipMemsetAsync(e->ktab_out.p + 8, 0, bk::ktab_fill_words() * sizeof(unsigned long long), e->stream));
    if (e->fill_pending) { BK_HIP(hipEventSynchronize(e->fill_ev)); e->fill_pending = false; }
    e->fill_known = 0; e->fill_unknown_upper = 0;
    if (int rc = ensure_ktab_room(e, n)) return rc;
    bk::launch_ktab_import(static_cast<const unsigned long long*>(d_keys), static_cast<const unsigned int*>(d_counts), n, e->ktab_keys.p, e->ktab_cnt.p,
                           e->ktab_log2, e->ktab_out.p + 4, e->stream);
    BK_HIP(hipGetLastError());
    e->ktab_exchanged = true;
    return note_ktab_fill(e);
}

int bk_shard_sums_device_ptr(bk_engine* e, void** d_ptr, uint64_t* len) {
    if (!e || !d_ptr || !len) return fail(BK_ERR_INVALID, "null argument");
    *d_ptr = e->shard_sums.p;
    *len = (uint64_t)2 * e->n_files * 5 + 9;
    return BK_OK;
}

int bk_sample_merge_shards(bk_engine* e) {
    if (!e) return fail(BK_ERR_INVALID, "null engine");
    BK_HIP(hipSetDevice(e->device));
    bk::launch_unpack_sums(e->shard_sums.p, e->stats.p, e->present.p, e->kstats.p, e->n_files, e->xport_flag.p, e->stream);
    BK_HIP(hipGetLastError());
    return BK_OK;
}

static int shard_args_ok(bk_engine* e, int mate, int n_shards, int width) {
    if (!e || mate < 0 || mate > 1) return fail(BK_ERR_INVALID, "bad argument");
    if (n_shards < 1 || (int)bk::kMaxShards % n_shards != 0) return fail(BK_ERR_INVALID, "n_shards must divide %u", bk::kMaxShards);
    if (width != 16 && width != 32 && width != 64) return fail(BK_ERR_INVALID, "width must be 16, 32 or 64");
    if (e->sparse) return fail(BK_ERR_UNSUPPORTED, "an index this large keeps its counter planes sparse: shard whole samples over GPUs, not one sample's reads");
    if (!e->in_sample) return fail(BK_ERR_STATE, "the transport of a plane comes between the pushes and the finalize of a sample");
    return BK_OK;
}

int bk_shard_measure(bk_engine* e, int mate, void** d_max) {
    if (!d_max) return fail(BK_ERR_INVALID, "null argument");
    if (int rc = shard_args_ok(e, mate, 1, 64)) return rc;
    BK_HIP(hipSetDevice(e->device));
    if (int rc = flush_pending_items(e)) return rc;
    if (int rc = zero_plane_if_stale(e, mate)) return rc;
    BK_HIP(hipMemsetAsync(e->xport_flag.p + 2, 0, 2 * sizeof(unsigned long long), e->stream));
    bk::launch_xport_measure(e->counters[mate].p, e->plane_len, e->v_off, e->xport_flag.p + 2, e->stream);
    BK_HIP(hipGetLastError());
    *d_max = e->xport_flag.p + 2;
    return BK_OK;
}

int bk_shard_transport(bk_engine* e, int mate, int n_shards, int width, void** d_send, uint64_t* part_bytes, void** d_recv) {
    if (!d_send || !part_bytes || !d_recv) return fail(BK_ERR_INVALID, "null argument");
    if (int rc = shard_args_ok(e, mate, n_shards, width)) return rc;
    BK_HIP(hipSetDevice(e->device));
    if (int rc = flush_pending_items(e)) return rc;
    if (int rc = zero_plane_if_stale(e, mate)) return rc;   // (a mate file nothing was pushed for: its plane is zeroed lazily -- now)
    e->plane_used[mate] = true;
    e->xport_ever = true;
    if (e->reduced_shards[0] == 0 && e->reduced_shards[1] == 0)   // first transport of this sample
        BK_HIP(hipMemsetAsync(e->xport_flag.p, 0, sizeof(unsigned long long), e->stream));
    const uint64_t pb = bk::xport_part_bytes(e->plane_len, e->v_off, (uint32_t)n_shards, width);
    // Width 16 spends four lanes on an E count: with many shards (or a plane that is mostly E counts) its part is no smaller than
    // the 32-bit one -- it would send more, not less.  Refused, so that nobody packs a plane for nothing ("auto" falls back on 32).
    if (width == 16 && pb >= bk::xport_part_bytes(e->plane_len, e->v_off, (uint32_t)n_shards, 32))
        return fail(BK_ERR_INVALID, "width 16 does not shrink the plane at %d shards (every E count takes four 16-bit lanes): use width 32", n_shards);
    // The buffers are sized ONCE, for the worst case over every shard count and width (the whole plane for `reduced`; the
    // largest packed plane and part for the transport), and stay where they are for the engine's lifetime: a host may keep views.
    if (e->reduced[mate].n < e->plane_len) {
        BK_HIP(hipStreamSynchronize(e->stream));
        BK_HIP(e->reduced[mate].alloc(e->plane_len));
    }
    *part_bytes = pb;
    if (width == 64) {   // nothing to pack: the plane itself is the send buffer and the received part is the reduced part
        *d_send = e->counters[mate].p;
        *d_recv = e->reduced[mate].p;
        return BK_OK;
    }
    if (!e->xport_send.p) {
        uint64_t max_part = 0, max_all = 0;
        for (uint32_t n = 1; n <= bk::kMaxShards; n *= 2)
            for (int w : {16, 32}) {
                const uint64_t b = bk::xport_part_bytes(e->plane_len, e->v_off, n, w);
                max_part = std::max(max_part, b);
                max_all = std::max(max_all, b * n);
            }
        BK_HIP(hipStreamSynchronize(e->stream));
        BK_HIP(e->xport_send.alloc(max_all));
        BK_HIP(e->xport_recv.alloc(max_part));
    }
    bk_engine::Span sp(e, 2);
    bk::launch_xport_pack(e->counters[mate].p, e->plane_len, e->v_off, (uint32_t)n_shards, width, e->xport_send.p, e->xport_flag.p, e->stream);
    BK_HIP(hipGetLastError());
    *d_send = e->xport_send.p;
    *d_recv = e->xport_recv.p;
    return BK_OK;
}

int bk_shard_received(bk_engine* e, int mate, int shard, int n_shards, int width) {
    if (int rc = shard_args_ok(e, mate, n_shards, width)) return rc;
    if (shard < 0 || shard >= n_shards) return fail(BK_ERR_INVALID, "0 <= shard < n_shards");
    const uint64_t part = e->plane_len / (uint64_t)n_shards;
    if (e->reduced[mate].n < part || (width != 64 && !e->xport_recv.p)) return fail(BK_ERR_STATE, "bk_shard_received without bk_shard_transport");
    BK_HIP(hipSetDevice(e->device));
    if (width != 64) {
        bk_engine::Span sp(e, 2);
        bk::launch_xport_unpack(e->xport_recv.p, e->plane_len, e->v_off, (uint32_t)n_shards, (uint32_t)shard, width, e->reduced[mate].p, e->stream);
        BK_HIP(hipGetLastError());
    }
    e->reduced_shards[mate] = n_shards;
    e->reduced_shard[mate] = shard;
    return BK_OK;
}

int bk_transport_overflow(bk_engine* e, int* overflowed) {
    if (!e || !overflowed) return fail(BK_ERR_INVALID, "null argument");
    BK_HIP(hipSetDevice(e->device));
    unsigned long long f[2] = {0, 0};
    BK_HIP(hipMemcpyAsync(f, e->xport_flag.p, sizeof f, hipMemcpyDeviceToHost, e->stream));
    BK_HIP(hipMemsetAsync(e->xport_flag.p + 1, 0, sizeof(unsigned long long), e->stream));
    BK_HIP(hipStreamSynchronize(e->stream));
    *overflowed = f[1] != 0;
    return BK_OK;
}

int bk_sample_download(bk_engine* e, int n_mates, uint64_t* fwd_depth, uint64_t* rev_depth, uint64_t* fwd_nk, uint64_t* rev_nk,
                       uint64_t* stats, uint8_t* present, uint64_t* kmer_stats) {
    if (!e) return fail(BK_ERR_INVALID, "null engine");
    if (n_mates < 1 || n_mates > 2) return fail(BK_ERR_INVALID, "n_mates must be 1 or 2");
    BK_HIP(hipSetDevice(e->device));
    const size_t plane = (size_t)e->total_cells * 4;
    uint64_t* dst[4] = {fwd_depth, rev_depth, fwd_nk, rev_nk};
    {
        bk_engine::Span sp(e, 2);
        for (int i = 0; i < 4; i++)
            if (dst[i] && plane) BK_HIP(hipMemcpyAsync(dst[i], e->pileup.p + (size_t)i * plane, plane * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
        if (stats) BK_HIP(hipMemcpyAsync(stats, e->stats.p, (size_t)n_mates * e->n_files * 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
        if (present) BK_HIP(hipMemcpyAsync(present, e->present.p, (size_t)n_mates * e->n_files, hipMemcpyDeviceToHost, e->stream));
        if (kmer_stats) BK_HIP(hipMemcpyAsync(kmer_stats, e->kstats.p, (size_t)n_mates * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    }
    unsigned long long kt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (kmer_stats) BK_HIP(hipMemcpyAsync(kt, e->ktab_out.p, sizeof kt, hipMemcpyDeviceToHost, e->stream));
    unsigned long long xf = 0;
    if (e->xport_ever) {
        BK_HIP(hipMemcpyAsync(&xf, e->xport_flag.p + 1, sizeof xf, hipMemcpyDeviceToHost, e->stream));
        BK_HIP(hipMemsetAsync(e->xport_flag.p + 1, 0, sizeof xf, e->stream));
    }
    unsigned int ah[2] = {0u, 0u};
    if (e->gather_mode) BK_HIP(hipMemcpyAsync(ah, e->n_alias_hits.p, sizeof ah, hipMemcpyDeviceToHost, e->stream));
    BK_HIP(hipStreamSynchronize(e->stream));
    if (ah[0] > bk_engine::kAliasCap || ah[1] > bk_engine::kAliasCap)
        return fail(BK_ERR_RANGE, "more than %u k-mers of this sample reach a bucket through the 64-bit wrap of its id: the list of them overflowed, the pileup is incomplete", bk_engine::kAliasCap);
    if (xf) return fail(BK_ERR_RANGE, "a counter of this (or an earlier, unchecked) sample did not fit the width its plane was exchanged at: the results are "
                                      "invalid -- repeat the sample with a wider bk_shard_transport (bk_shard_measure tells which width is safe)");
    if (kmer_stats) {
        for (int m = 0; m < n_mates; m++) {
            kmer_stats[m * 4 + 0] += e->pushed_records[m];   // + the device-side tally of bk_push_reads_ascii batches
            if (e->ktab_keys.p) {
                // index-touching k-mers are in the counter plane (kept tally in [3], distinct tally in [2] by finalize);
                // the rest are in the hash table
                if (kt[4] || kmer_stats[m * 4 + 2] >= (1ull << 56)) { kmer_stats[m * 4 + 2] = kmer_stats[m * 4 + 3] = ~0ull; }   // (2^56: a rank's table overflowed, sharded finalize)
                else { kmer_stats[m * 4 + 2] += kt[m * 2 + 0]; kmer_stats[m * 4 + 3] += kt[m * 2 + 1]; }
            } else {
                kmer_stats[m * 4 + 2] = 0;
            }
        }
    }
    return BK_OK;
}

int bk_sample_finish(bk_engine* e, int n_mates, uint64_t* fwd_depth, uint64_t* rev_depth, uint64_t* fwd_nk, uint64_t* rev_nk,
                     uint64_t* stats, uint8_t* present, uint64_t* kmer_stats) {
    int rc = bk_sample_finalize(e, n_mates);
    if (rc != BK_OK) return rc;
    return bk_sample_download(e, n_mates, fwd_depth, rev_depth, fwd_nk, rev_nk, stats, present, kmer_stats);
}

// ---- after the pileup, on the device (bk_caller.hip) -----------------------------------------------------------
void bk_call_params_default(bk_call_params* p) {
    if (!p) return;
    p->k = 21;                          // consts.rs:3
    p->no_end_filter = 0; p->no_strand_filter = 0; p->no_strand_balance_filter = 0;
    p->min_af = 0.03;                   // consts.rs:8
    p->strand_balance_ratio = 0.1;      // consts.rs:10
    p->strand_odds_max = 6.0;           // cli.rs --strand_odds
    p->variant_multiplier = 1.5;        // consts.rs:15
    p->n_per_strand = 2; p->min_depth = 300; p->min_variant_depth = 3;
}

int bk_sample_call(bk_engine* e, int n_mates, const bk_call_params* p) {
    if (!e || !p) return fail(BK_ERR_INVALID, "null argument");
    if (n_mates < 1 || n_mates > 2) return fail(BK_ERR_INVALID, "n_mates must be 1 or 2");
    if (e->in_sample) return fail(BK_ERR_STATE, "bk_sample_call comes after bk_sample_finalize");
    if (e->finalized_mates == 0) return fail(BK_ERR_STATE, "bk_sample_call: no sample has been finalized on this engine");
    if (e->finalized_mates != n_mates) return fail(BK_ERR_STATE, "bk_sample_call(n_mates = %d): the sample was finalized with %d mate file(s)", n_mates, e->finalized_mates);
    BK_HIP(hipSetDevice(e->device));
    const uint64_t cap = std::max<uint64_t>(3 * e->max_file_cells, 1);   // at most three alternative bases per position
    if (!e->call_out.p) {
        BK_HIP(e->call_noise.alloc((size_t)e->total_cells));
        const size_t mc = std::max<uint64_t>(e->max_file_cells, 1);
        BK_HIP(e->noise_maf.alloc(mc * 3));
        BK_HIP(e->noise_tbl.alloc((mc + 64 * (size_t)std::max(e->max_seqs_per_file, 1) + 64) * 10));
        BK_HIP(e->noise_state.alloc(mc));
        BK_HIP(e->noise_sums.alloc(mc * 2));
        BK_HIP(e->noise_cnt.alloc(mc));
        BK_HIP(e->call_records.alloc((size_t)cap));
        BK_HIP(e->call_out.alloc(1));
    }
    bk::CallArgs a{};
    a.prm = *p;
    a.n_files = e->n_files; a.n_mates = n_mates;
    a.stats = e->stats.p; a.present = e->present.p;
    a.genome_len = e->genome_len.p; a.seq_first = e->seq_first.p; a.n_seqs = e->n_seqs_d.p; a.seq_cell = e->seq_cell.p; a.seq_len = e->seq_len_d.p;
    a.ref_words = e->ref_words.p + bk::scan_ref_pad_words();
    a.pileup = e->pileup.p; a.plane = (size_t)e->total_cells * 4;
    a.noise = e->call_noise.p; a.records = e->call_records.p; a.record_cap = cap; a.out = e->call_out.p;
    a.noise_maf = e->noise_maf.p; a.noise_tbl = e->noise_tbl.p; a.noise_sums = e->noise_sums.p; a.noise_cnt = e->noise_cnt.p; a.noise_state = e->noise_state.p;
    if (const char* ns = test_env("BK_NOISE_SERIAL")) a.noise_serial = atoi(ns);
    bk_engine::Span sp(e, 1);
    bk::launch_call(a, e->max_seqs_per_file, e->max_file_cells, e->stream);
    BK_HIP(hipGetLastError());
    return BK_OK;
}

int bk_sample_download_calls(bk_engine* e, bk_call_summary* summary, bk_call_record* records, uint64_t cap) {
    if (!e || !summary) return fail(BK_ERR_INVALID, "null argument");
    if (!e->call_out.p) return fail(BK_ERR_STATE, "bk_sample_download_calls comes after bk_sample_call");
    BK_HIP(hipSetDevice(e->device));
    BK_HIP(hipMemcpyAsync(summary, e->call_out.p, sizeof *summary, hipMemcpyDeviceToHost, e->stream));
    BK_HIP(hipStreamSynchronize(e->stream));
    const uint64_t n = std::min<uint64_t>(std::min<uint64_t>(summary->n_records, cap), e->call_records.n);
    if (n && records) {
        BK_HIP(hipMemcpy(records, e->call_records.p, (size_t)n * sizeof(bk_call_record), hipMemcpyDeviceToHost));
        std::sort(records, records + n, [](const bk_call_record& x, const bk_call_record& y) {
            if (x.seq_id != y.seq_id) return x.seq_id < y.seq_id;
            if (x.pos != y.pos) return x.pos < y.pos;
            return x.alt_base < y.alt_base;
        });
    }
    return BK_OK;
}

int bk_sample_download_noise(bk_engine* e, double* out, uint64_t cap, uint64_t* n) {
    if (!e || !n) return fail(BK_ERR_INVALID, "null argument");
    if (!e->call_out.p) return fail(BK_ERR_STATE, "bk_sample_download_noise comes after bk_sample_call");
    BK_HIP(hipSetDevice(e->device));
    bk_call_summary summ;
    BK_HIP(hipMemcpyAsync(&summ, e->call_out.p, sizeof summ, hipMemcpyDeviceToHost, e->stream));
    BK_HIP(hipStreamSynchronize(e->stream));
    *n = 0;
    if (summ.file_id < 0 || summ.file_id >= e->n_files) return BK_OK;
    const uint64_t lo = e->file_cell_lo[(size_t)summ.file_id], hi = summ.file_id + 1 < e->n_files ? e->file_cell_lo[(size_t)summ.file_id + 1] : e->total_cells;
    *n = hi - lo;
    if (out && cap) BK_HIP(hipMemcpy(out, e->call_noise.p + lo, (size_t)std::min<uint64_t>(cap, hi - lo) * sizeof(double), hipMemcpyDeviceToHost));
    return BK_OK;
}

// ---- K0 host packer ------------------------------------------------------------------------------------------
namespace {
struct Packer {
    int k; uint32_t stride; uint32_t* words; uint16_t* lens; uint64_t cap; uint64_t n = 0;
    void emit(const uint8_t* s, uint64_t len) {   // one record of <= 16*stride ACGT symbols
        if (n < cap) {
            uint32_t* w = words + n * stride;
            std::memset(w, 0, (size_t)stride * 4);
            for (uint64_t i = 0; i < len; i++) w[i >> 4] |= (uint32_t)bronko::acgt_code(s[i]) << (2 * (i & 15));
            lens[n] = (uint16_t)len;
        }
        n++;
    }
    void run(const uint8_t* s, uint64_t len) {    // one maximal ACGT run
        if (len < (uint64_t)k) return;
        const uint64_t maxb = std::min<uint64_t>((uint64_t)stride * 16, 65535);
        uint64_t pos = 0;
        for (;;) {
            const uint64_t take = std::min(maxb, len - pos);
            emit(s + pos, take);
            if (pos + take >= len) break;
            pos += take - (uint64_t)(k - 1);      // next chunk re-reads k-1 bases: no k-mer lost or doubled
        }
    }
    void read(const uint8_t* s, uint64_t len) {
        uint64_t start = 0;
        for (uint64_t i = 0; i <= len; i++) {
            if (i == len || bronko::acgt_code(s[i]) < 0) { run(s + start, i - start); start = i + 1; }
        }
    }
};
}  // namespace

uint64_t bk_pack_reads(const uint8_t* const* reads, const uint64_t* read_lens, uint64_t n_reads, int32_t k, uint32_t stride_words,
                       uint32_t* out_words, uint16_t* out_lens, uint64_t cap_records) {
    if (k < 1 || stride_words == 0 || (uint64_t)stride_words * 16 < (uint64_t)k) return 0;
    Packer p{k, stride_words, out_words, out_lens, (out_words && out_lens) ? cap_records : 0};
    for (uint64_t r = 0; r < n_reads; r++) p.read(reads[r], read_lens[r]);
    return p.n;
}

uint64_t bk_pack_reads_flat(const uint8_t* buf, const uint64_t* offsets, uint64_t n_reads, int32_t k, uint32_t stride_words,
                            uint32_t* out_words, uint16_t* out_lens, uint64_t cap_records) {
    if (k < 1 || stride_words == 0 || (uint64_t)stride_words * 16 < (uint64_t)k) return 0;
    Packer p{k, stride_words, out_words, out_lens, (out_words && out_lens) ? cap_records : 0};
    for (uint64_t r = 0; r < n_reads; r++) p.read(buf + offsets[r], offsets[r + 1] - offsets[r]);
    return p.n;
}

// ---- measurement ---------------------------------------------------------------------------------------------
int bk_timing_enable(bk_engine* e, int on) {
    if (!e) return fail(BK_ERR_INVALID, "null engine");
    e->timing = on != 0;
    e->timing_kinds = (on & 0xff) == 1 ? 0xfu : (unsigned)(on >> 1) & 0xfu;   // on = 1: all kinds; otherwise bit (1 + kind) selects a kind
    e->timing_every = std::max(1u, ((unsigned)on >> 8) & 0xffu);                // bits 8..15: every N-th launch of a kind only
    for (auto& c : e->timing_seen) c = 0;
    return BK_OK;
}

int bk_timing_read(bk_engine* e, double ms[4], uint64_t n[4], int reset) {
    if (!e || !ms || !n) return fail(BK_ERR_INVALID, "bad argument");
    BK_HIP(hipSetDevice(e->device));
    BK_HIP(hipStreamSynchronize(e->stream));
    for (int i = 0; i < 4; i++) { ms[i] = 0.0; n[i] = 0; }
    for (auto& s : e->spans) {
        float t = 0.f;
        BK_HIP(hipEventElapsedTime(&t, s.a, s.b));
        ms[s.kind] += t;
        n[s.kind] += 1;
    }
    if (reset) {
        for (auto& s : e->spans) { e->free_events.push_back(s.a); e->free_events.push_back(s.b); }
        e->spans.clear();
    }
    return BK_OK;
}

}  // extern "C"
