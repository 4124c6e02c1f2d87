// bk_build.hip -- build_indexes (/root/reference/src/build.rs:145-231) on the device.
//
// The reference walks every k-mer of every sequence, canonicalises it, computes its k bucket ids (assign_buckets, lcb.rs:1-45)
// and appends a BucketInfo to each bucket's list (build.rs:191-204); the lists of the files are concatenated in file order
// (build.rs:223-228).  So a bucket's list is ordered by (file, sequence, location) -- the order in which the pairs are generated
// when the k-mers are taken in (file, sequence, location) order.  Here: one thread per k-mer writes its k (bucket id, BucketInfo)
// pairs at their place in that generation order, a STABLE radix sort by bucket id (rocPRIM, a plain library sort) groups them
// without disturbing the order inside a bucket, and the host cuts the sorted run into buckets while it copies it out.  100 strains
// at k = 31 (93 M pairs): well under a second against 5 s on 32 host threads.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/bronko_hip.h"
#include "../host/lcb.hpp"
#include "bk_device.h"

namespace {

struct SeqRow { unsigned long long byte_off, len, kmer_off; unsigned int file, seq; };   // kmer_off: k-mers of the sequences before it

// pair value: file << 46 | seq << 38 | location << 6 | idx << 1 | canonical
__global__ __launch_bounds__(256) void gen_pairs_kernel(const unsigned char* __restrict__ bases, const SeqRow* __restrict__ rows, int n_rows,
                                                        unsigned long long n_kmers, int k, unsigned long long* __restrict__ keys,
                                                        unsigned long long* __restrict__ vals) {
    for (unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x; g < n_kmers; g += (unsigned long long)gridDim.x * 256) {
        int lo = 0, hi = n_rows - 1;                 // the last sequence whose k-mers start at or before g
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (rows[mid].kmer_off <= g) lo = mid; else hi = mid - 1;
        }
        const SeqRow r = rows[lo];
        const unsigned long long i = g - r.kmer_off;                              // location of the k-mer in its sequence
        const unsigned char* s = bases + r.byte_off + i;
        unsigned long long fwd = 0;
        for (int t = 0; t < k; ++t) fwd = (fwd << 2) | bronko::nt_to_bits(s[t]);  // kmer_to_u64 (lcb.rs:67-74; non-ACGT -> A)
        const bronko::Canon c = bronko::canonical_u64(fwd, k);                    // build.rs:193
        unsigned long long ids[32];
        bronko::assign_buckets(c.kmer, k, reinterpret_cast<uint64_t*>(ids));      // build.rs:194
        const unsigned long long base = ((unsigned long long)r.file << 46) | ((unsigned long long)r.seq << 38) | (i << 6) | (c.rc ? 1ull : 0ull);
        for (int j = 0; j < k; ++j) {                                             // build.rs:196-204
            keys[g * (unsigned long long)k + j] = ids[j];
            vals[g * (unsigned long long)k + j] = base | ((unsigned long long)j << 1);
        }
    }
}

thread_local char g_build_err[512];

}  // namespace

extern "C" {

const char* bk_build_last_error(void) { return g_build_err; }

void bk_built_index_free(bk_built_index* ix) {
    if (!ix) return;
    free(ix->bucket_ids); free(ix->bucket_off); free(ix->entries);
    ix->bucket_ids = nullptr; ix->bucket_off = nullptr; ix->entries = nullptr; ix->n_buckets = ix->n_entries = 0;
}

int bk_build_index(int32_t k, int32_t n_files, const int32_t* n_seqs, const uint64_t* seq_lens, const uint8_t* const* seqs, int32_t device,
                   bk_built_index* out) {
#define BB_FAIL(code, ...) do { snprintf(g_build_err, sizeof g_build_err, __VA_ARGS__); return (code); } while (0)
#define BB_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); BB_FAIL(BK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
    if (!out || !n_seqs || (n_files > 0 && (!seq_lens || !seqs))) BB_FAIL(BK_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof *out);
    if (k < 3 || k > 31 || (k & 1) == 0) BB_FAIL(BK_ERR_INVALID, "Invalid kmer size %d", k);
    if (n_files < 0 || n_files > 65536) BB_FAIL(BK_ERR_INVALID, "n_files out of range (file_id is u16)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) BB_FAIL(BK_ERR_NO_DEVICE, "no HIP device %d", device);
    unsigned char* d_bases = nullptr; SeqRow* d_rows = nullptr;
    unsigned long long *d_k0 = nullptr, *d_k1 = nullptr, *d_v0 = nullptr, *d_v1 = nullptr; void* d_tmp = nullptr;
    hipStream_t stream = nullptr;   // a stream of its own: a process with samples in flight on this device is not stalled by the build
    auto cleanup = [&] {
        (void)hipFree(d_bases); (void)hipFree(d_rows); (void)hipFree(d_k0); (void)hipFree(d_k1); (void)hipFree(d_v0); (void)hipFree(d_v1); (void)hipFree(d_tmp);
        d_bases = nullptr; d_rows = nullptr; d_k0 = d_k1 = d_v0 = d_v1 = nullptr; d_tmp = nullptr;
        if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
    };
    (void)hipSetDevice(device);

    std::vector<SeqRow> rows;
    unsigned long long bytes = 0, n_kmers = 0;
    size_t sq = 0;
    for (int f = 0; f < n_files; f++) {
        if (n_seqs[f] < 0 || n_seqs[f] > 256) BB_FAIL(BK_ERR_INVALID, "file %d: more than 256 sequences (seq_id is u8)", f);
        for (int s = 0; s < n_seqs[f]; s++, sq++) {
            const unsigned long long len = seq_lens[sq];
            if (len >= (1ull << 32)) BB_FAIL(BK_ERR_UNSUPPORTED, "sequence longer than 2^32 (location is u32)");
            if (len >= (unsigned long long)k) {   // (upstream would panic on a shorter one; nothing to index)
                rows.push_back(SeqRow{bytes, len, n_kmers, (unsigned)f, (unsigned)s});
                n_kmers += len - (unsigned long long)k + 1;
            }
            bytes += len;
        }
    }
    const unsigned long long n_pairs = n_kmers * (unsigned long long)k;
    if (n_pairs >= (1ull << 33)) BB_FAIL(BK_ERR_UNSUPPORTED, "index too large for the device build (%llu pairs)", n_pairs);
    if (n_pairs == 0) {
        out->bucket_off = (uint64_t*)malloc(sizeof(uint64_t));
        if (!out->bucket_off) BB_FAIL(BK_ERR_INVALID, "out of memory");
        out->bucket_off[0] = 0;
        return BK_OK;
    }
    {   // 32 bytes per pair for keys and values in two buffers each, the sort's scratch on top: refuse early instead of failing in the middle
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (unsigned long long)free_b < 40ull * n_pairs + bytes + (64ull << 20))
            BB_FAIL(BK_ERR_UNSUPPORTED, "not enough free device memory for the device build (%llu pairs need about %llu MB, %llu MB free)", n_pairs,
                    (40ull * n_pairs + bytes) >> 20, (unsigned long long)free_b >> 20);
    }
    BB_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

    std::vector<unsigned char> h_bases(bytes);
    {
        size_t q = 0; unsigned long long at = 0;
        for (int f = 0; f < n_files; f++) for (int s = 0; s < n_seqs[f]; s++, q++) { std::memcpy(h_bases.data() + at, seqs[q], seq_lens[q]); at += seq_lens[q]; }
    }
    BB_HIP(hipMalloc((void**)&d_bases, bytes ? bytes : 1));
    BB_HIP(hipMalloc((void**)&d_rows, rows.size() * sizeof(SeqRow)));
    BB_HIP(hipMalloc((void**)&d_k0, n_pairs * 8)); BB_HIP(hipMalloc((void**)&d_k1, n_pairs * 8));
    BB_HIP(hipMalloc((void**)&d_v0, n_pairs * 8)); BB_HIP(hipMalloc((void**)&d_v1, n_pairs * 8));
    BB_HIP(hipMemcpyAsync(d_bases, h_bases.data(), bytes, hipMemcpyHostToDevice, stream));
    BB_HIP(hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(SeqRow), hipMemcpyHostToDevice, stream));
    const unsigned grid = (unsigned)std::min<unsigned long long>((n_kmers + 255) / 256, 1u << 16);
    hipLaunchKernelGGL(gen_pairs_kernel, dim3(grid), dim3(256), 0, stream, d_bases, d_rows, (int)rows.size(), n_kmers, (int)k, d_k0, d_v0);
    BB_HIP(hipGetLastError());
    size_t tmp_bytes = 0;
    BB_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_k0, d_k1, d_v0, d_v1, (size_t)n_pairs, 0, 64, stream));
    BB_HIP(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
    BB_HIP(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_k0, d_k1, d_v0, d_v1, (size_t)n_pairs, 0, 64, stream));   // stable: keeps (file, seq, location) inside a bucket
    // (no value-initialisation of 1.5 GB of host memory, and the cutting into buckets on all host threads: the serial form of this
    // tail was 0.6 s of the 1.06 s a 100-strain index took)
    std::unique_ptr<unsigned long long[]> h_k(new unsigned long long[n_pairs]), h_v(new unsigned long long[n_pairs]);
    BB_HIP(hipMemcpyAsync(h_k.get(), d_k1, n_pairs * 8, hipMemcpyDeviceToHost, stream));
    BB_HIP(hipMemcpyAsync(h_v.get(), d_v1, n_pairs * 8, hipMemcpyDeviceToHost, stream));
    BB_HIP(hipStreamSynchronize(stream));
    cleanup();

    const unsigned nt = (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 128u), n_pairs / 65536 + 1));
    std::vector<uint64_t> first_b(nt + 1, 0);   // buckets that start in the chunks before chunk t
    auto chunk = [&](unsigned t) { return n_pairs * t / nt; };
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t] {
            uint64_t c = 0;
            for (unsigned long long i = chunk(t); i < chunk(t + 1); i++) c += (i == 0 || h_k[i] != h_k[i - 1]);
            first_b[t + 1] = c;
        });
        for (auto& x : th) x.join();
    }
    for (unsigned t = 0; t < nt; t++) first_b[t + 1] += first_b[t];
    const uint64_t nb = first_b[nt];
    out->bucket_ids = (uint64_t*)malloc(std::max<uint64_t>(nb, 1) * sizeof(uint64_t));
    out->bucket_off = (uint64_t*)malloc((nb + 1) * sizeof(uint64_t));
    out->entries = (bk_bucket_info*)malloc(n_pairs * sizeof(bk_bucket_info));
    if (!out->bucket_ids || !out->bucket_off || !out->entries) { bk_built_index_free(out); BB_FAIL(BK_ERR_INVALID, "out of memory"); }
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t] {
            uint64_t b = first_b[t];
            for (unsigned long long i = chunk(t); i < chunk(t + 1); i++) {
                if (i == 0 || h_k[i] != h_k[i - 1]) { out->bucket_ids[b] = h_k[i]; out->bucket_off[b] = i; b++; }
                const unsigned long long v = h_v[i];
                bk_bucket_info& e = out->entries[i];
                std::memset(&e, 0, sizeof e);   // (padding bytes zero)
                e.file_id = (uint16_t)(v >> 46); e.seq_id = (uint8_t)(v >> 38); e.location = (uint32_t)(v >> 6); e.idx = (uint8_t)((v >> 1) & 31u); e.canonical = (uint8_t)(v & 1u);
            }
        });
        for (auto& x : th) x.join();
    }
    out->bucket_off[nb] = n_pairs;
    out->n_buckets = nb; out->n_entries = n_pairs;
    return BK_OK;
#undef BB_HIP
#undef BB_FAIL
}

}  // extern "C"

// ---- a plain library sort for bk_engine_create's large host arrays ------------------------------------------------------------
// order[i] = index of the i-th smallest key (stable: equal keys keep their order); optionally the sorted keys as well.  n u64 keys
// go up (8 n bytes), n u32 indices come back: 15 M keys in ~50 ms where std::sort on 32 host threads over an indirect comparison
// took 1.5 s (profiles/r05_create_timing.txt).
namespace bk {
__global__ __launch_bounds__(256) void iota_u32_kernel(unsigned int* v, unsigned long long n) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) v[i] = (unsigned int)i;
}
hipError_t device_sort_order(const unsigned long long* h_keys, size_t n, int end_bit, unsigned int* h_order, unsigned long long* h_sorted_keys) {
    if (n == 0) return hipSuccess;
    unsigned long long *d_k0 = nullptr, *d_k1 = nullptr;
    unsigned int *d_v0 = nullptr, *d_v1 = nullptr;
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0;
    hipError_t e = hipSuccess;
    auto done = [&]() { (void)hipFree(d_k0); (void)hipFree(d_k1); (void)hipFree(d_v0); (void)hipFree(d_v1); (void)hipFree(d_tmp); return e; };
    if ((e = hipMalloc(reinterpret_cast<void**>(&d_k0), n * 8)) || (e = hipMalloc(reinterpret_cast<void**>(&d_k1), n * 8)) ||
        (e = hipMalloc(reinterpret_cast<void**>(&d_v0), n * 4)) || (e = hipMalloc(reinterpret_cast<void**>(&d_v1), n * 4))) return done();
    if ((e = hipMemcpy(d_k0, h_keys, n * 8, hipMemcpyHostToDevice))) return done();
    hipLaunchKernelGGL(iota_u32_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, nullptr, d_v0, (unsigned long long)n);
    if ((e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_k0, d_k1, d_v0, d_v1, n, 0, (unsigned)end_bit, (hipStream_t) nullptr))) return done();
    if ((e = hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16)))) return done();
    if ((e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_k0, d_k1, d_v0, d_v1, n, 0, (unsigned)end_bit, (hipStream_t) nullptr))) return done();
    if ((e = hipMemcpy(h_order, d_v1, n * 4, hipMemcpyDeviceToHost))) return done();
    if (h_sorted_keys && (e = hipMemcpy(h_sorted_keys, d_k1, n * 8, hipMemcpyDeviceToHost))) return done();
    return done();
}

// ---- bk_engine_create: the window bucket of every k-mer of U at every window position --------------------------------------------
// 15 M k-mers (14.5 M of them pseudo k-mers, k = 31) x 26 positions = 400 M probes into 1.7 GB of window tables: DRAM latency on
// the host (2.0 s on 256 threads), nothing on the device.  out[i * W + t] = slot or `empty`; valid[i] = bit t set where a slot was found.
__global__ __launch_bounds__(256) void lookup_slots_kernel(const TableSlot* __restrict__ table, uint32_t log2s, const unsigned long long* __restrict__ keys,
                                                           unsigned long long n, int W, int wstart, int k, uint32_t empty, uint32_t* __restrict__ out,
                                                           uint32_t* __restrict__ valid) {
    const size_t S = (size_t)1 << log2s;
    for (unsigned long long w = (unsigned long long)blockIdx.x * 256 + threadIdx.x; w < n * (unsigned long long)W; w += (unsigned long long)gridDim.x * 256) {
        const unsigned long long i = w / (unsigned)W;
        const int t = (int)(w - i * (unsigned)W);
        const unsigned long long key = keys[i] & ~(3ull << (2 * (k - 1 - (wstart + t))));
        const TableSlot* sub = table + (size_t)t * S;
        uint32_t h = hash_key(key, log2s);
        uint32_t slot = empty;
        for (;;) {
            const uint4 e = *reinterpret_cast<const uint4*>(sub + h);
            const unsigned long long kk = (unsigned long long)e.x | ((unsigned long long)e.y << 32);
            if (kk == key) { slot = e.z; break; }
            if (kk == kEmptyKey) break;
            h = (h + 1) & (uint32_t)(S - 1);
        }
        out[w] = slot;
        if (slot != empty) atomicOr(valid + i, 1u << t);
    }
}
// The window tables built where they stay: slot s (key keys[s], window position ts[s]) into sub-table ts[s] by open addressing;
// of equal keys the lowest slot stays (k = 31: an alias key that coincides with a real key; any other k: *dup is raised).
// `table` comes in filled with 0xff bytes (key = kEmptyKey, slot = ~0).
__global__ __launch_bounds__(256) void build_table_kernel(TableSlot* table, uint32_t log2s, const unsigned long long* __restrict__ keys,
                                                          const unsigned char* __restrict__ ts, unsigned long long n, unsigned int* dup) {
    const size_t S = (size_t)1 << log2s;
    for (unsigned long long s = (unsigned long long)blockIdx.x * 256 + threadIdx.x; s < n; s += (unsigned long long)gridDim.x * 256) {
        const unsigned long long key = keys[s];
        TableSlot* sub = table + (size_t)ts[s] * S;
        uint32_t h = hash_key(key, log2s);
        for (;;) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&sub[h].key), (unsigned long long)kEmptyKey, key);
            if (old == (unsigned long long)kEmptyKey || old == key) {
                atomicMin(&sub[h].slot, (unsigned int)s);
                if (old == key) *dup = 1u;
                break;
            }
            h = (h + 1) & (uint32_t)(S - 1);
        }
    }
}
hipError_t device_build_table(TableSlot* d_table, size_t n_table, uint32_t log2s, const unsigned long long* h_keys, const unsigned char* h_ts, size_t n, bool* dup) {
    unsigned long long* d_keys = nullptr;
    unsigned char* d_ts = nullptr;
    unsigned int* d_dup = nullptr;
    unsigned int h_dup = 0;
    hipError_t e = hipSuccess;
    auto done = [&]() { (void)hipFree(d_keys); (void)hipFree(d_ts); (void)hipFree(d_dup); return e; };
    if ((e = hipMemset(d_table, 0xff, n_table * sizeof(TableSlot)))) return done();
    if (n) {
        if ((e = hipMalloc(reinterpret_cast<void**>(&d_keys), n * 8)) || (e = hipMalloc(reinterpret_cast<void**>(&d_ts), n)) || (e = hipMalloc(reinterpret_cast<void**>(&d_dup), 4))) return done();
        if ((e = hipMemcpy(d_keys, h_keys, n * 8, hipMemcpyHostToDevice)) || (e = hipMemcpy(d_ts, h_ts, n, hipMemcpyHostToDevice)) || (e = hipMemset(d_dup, 0, 4))) return done();
        hipLaunchKernelGGL(build_table_kernel, dim3(256 * 16), dim3(256), 0, nullptr, d_table, log2s, d_keys, d_ts, (unsigned long long)n, d_dup);
        if ((e = hipGetLastError()) || (e = hipMemcpy(&h_dup, d_dup, 4, hipMemcpyDeviceToHost))) return done();
    }
    *dup = h_dup != 0;
    return done();
}
// dst[id_of[i] * W + t] = src[i * W + t]
__global__ __launch_bounds__(256) void permute_rows_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ id_of, unsigned long long n, int W, uint32_t* __restrict__ dst) {
    for (unsigned long long w = (unsigned long long)blockIdx.x * 256 + threadIdx.x; w < n * (unsigned long long)W; w += (unsigned long long)gridDim.x * 256) {
        const unsigned long long i = w / (unsigned)W;
        dst[(unsigned long long)id_of[i] * (unsigned)W + (w - i * (unsigned)W)] = src[w];
    }
}
hipError_t device_lookup_slots(const TableSlot* d_table, uint32_t log2s, const unsigned long long* h_keys, size_t n, int W, int wstart, int k, uint32_t empty,
                               uint32_t* d_out, uint32_t* h_valid) {
    if (n == 0 || W <= 0) return hipSuccess;
    unsigned long long* d_keys = nullptr;
    uint32_t* d_valid = nullptr;
    hipError_t e = hipSuccess;
    auto done = [&]() { (void)hipFree(d_keys); (void)hipFree(d_valid); return e; };
    if ((e = hipMalloc(reinterpret_cast<void**>(&d_keys), n * 8)) || (e = hipMalloc(reinterpret_cast<void**>(&d_valid), n * 4))) return done();
    if ((e = hipMemcpy(d_keys, h_keys, n * 8, hipMemcpyHostToDevice)) || (e = hipMemset(d_valid, 0, n * 4))) return done();
    hipLaunchKernelGGL(lookup_slots_kernel, dim3(256 * 16), dim3(256), 0, nullptr, d_table, log2s, d_keys, (unsigned long long)n, W, wstart, k, empty, d_out, d_valid);
    if ((e = hipGetLastError())) return done();
    e = hipMemcpy(h_valid, d_valid, n * 4, hipMemcpyDeviceToHost);
    return done();
}
hipError_t device_permute_rows(const uint32_t* d_src, const uint32_t* h_id_of, size_t n, int W, uint32_t* d_dst) {
    if (n == 0 || W <= 0) return hipSuccess;
    uint32_t* d_id = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_id), n * 4);
    if (!e) e = hipMemcpy(d_id, h_id_of, n * 4, hipMemcpyHostToDevice);
    if (!e) { hipLaunchKernelGGL(permute_rows_kernel, dim3(256 * 16), dim3(256), 0, nullptr, d_src, d_id, (unsigned long long)n, W, d_dst); e = hipGetLastError(); }
    if (!e) e = hipDeviceSynchronize();
    (void)hipFree(d_id);
    return e;
}
}  // namespace bk
