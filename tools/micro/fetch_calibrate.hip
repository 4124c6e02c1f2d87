// fetch_calibrate.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report per TRUE byte for the access patterns of this repo's kernels.
// MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE is exactly half of the bytes of a wide coalesced read; "other access widths and
// WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel below moves a known
// number of bytes of a 1 GiB buffer (four times the Infinity Cache, each byte touched once); run under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...   (tools/fetch_calibrate.sh)
// tools/fetch_calibrate.py divides: factor = true bytes / reported bytes per pattern -> profiles/r04_fetch_calibration.json, which
// tools/make_pmc_traffic.py applies kernel by kernel instead of a blanket x2.
//   rd_wide16      16 B per lane, consecutive lanes consecutive          (LDS-DMA record staging of scan_items_kernel, K0's line staging)
//   rd_stride40_4  4 B per lane at a 40-byte lane stride, ten passes     (scan_count_kernel's record words: every byte once, a word at a time)
//   rd_seq8        8 B per lane, consecutive                             (the V plane's rows in K2a / finalize_vbin_kernel / bin_count's read-modify-write)
//   rd_scatter8    8 B per lane at pseudo-random 8-byte slots             (BucketInfo / id_rec lookups of the many-genome finalize)
//   rd_scatter16   16 B per lane at pseudo-random 16-byte slots           (bin_count_kernel reading other workgroups' buckets, seed-table / slot_rec loads)
//   wr_wide16      16 B per lane stores, consecutive                      (scan_items_kernel's bucket area, K0's records)
//   wr_seq8        8 B per lane stores, consecutive                       (bin_count_kernel's V plane stores, K2a's zeroing)
//   wr_stride40_4  4 B stores at a 40-byte lane stride, ten passes        (round 3's K0)
//   at_scatter8    8-byte atomic adds at pseudo-random slots              (vote / V-row atomics)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr size_t kBytes = 1ull << 30;
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; return x ^ (x >> 33); }

__global__ void rd_wide16(const uint4* p, size_t n16, unsigned int* sink) {
    unsigned int s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; s += v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x12345678u) *sink = s;
}
__global__ void rd_stride40_4(const uint32_t* p, size_t n_rec, unsigned int* sink) {   // record r = words [10 r, 10 r + 10)
    unsigned int s = 0;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += (size_t)gridDim.x * blockDim.x)
        for (int j = 0; j < 10; ++j) s += p[r * 10 + j];
    if (s == 0x12345678u) *sink = s;
}
__global__ void rd_seq8(const unsigned long long* p, size_t n8, unsigned int* sink) {
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 0x12345678ull) *sink = (unsigned int)s;
}
__global__ void rd_scatter8(const unsigned long long* p, size_t n8, size_t n_loads, unsigned int* sink) {   // (a permutation-like walk: slot = hash(i) mod n8)
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += (size_t)gridDim.x * blockDim.x) s += p[mix(i) % n8];
    if (s == 0x12345678ull) *sink = (unsigned int)s;
}
__global__ void rd_scatter16(const uint4* p, size_t n16, size_t n_loads, unsigned int* sink) {
    unsigned int s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[mix(i) % n16]; s += v.x ^ v.w; }
    if (s == 0x12345678u) *sink = s;
}
__global__ void wr_wide16(uint4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void wr_seq8(unsigned long long* p, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) p[i] = i;
}
__global__ void wr_stride40_4(uint32_t* p, size_t n_rec) {
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += (size_t)gridDim.x * blockDim.x)
        for (int j = 0; j < 10; ++j) p[r * 10 + j] = (uint32_t)(r + j);
}
__global__ void at_scatter8(unsigned long long* p, size_t n8, size_t n_ops) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ops; i += (size_t)gridDim.x * blockDim.x) atomicAdd(p + mix(i) % n8, 1ull);
}

int main() {
    void* buf = nullptr;
    unsigned int* sink = nullptr;
    if (hipMalloc(&buf, kBytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, kBytes);
    const dim3 g(2048), b(256);
    const size_t n_scatter = 1ull << 24;   // 16 M scattered accesses
    for (int rep = 0; rep < 2; ++rep) {     // (the second repetition is the one to read: the first also pays page-table warm-up)
        hipLaunchKernelGGL(rd_wide16, g, b, 0, 0, (const uint4*)buf, kBytes / 16, sink);
        hipLaunchKernelGGL(rd_stride40_4, g, b, 0, 0, (const uint32_t*)buf, kBytes / 40, sink);
        hipLaunchKernelGGL(rd_seq8, g, b, 0, 0, (const unsigned long long*)buf, kBytes / 8, sink);
        hipLaunchKernelGGL(rd_scatter8, g, b, 0, 0, (const unsigned long long*)buf, kBytes / 8, n_scatter, sink);
        hipLaunchKernelGGL(rd_scatter16, g, b, 0, 0, (const uint4*)buf, kBytes / 16, n_scatter, sink);
        hipLaunchKernelGGL(wr_wide16, g, b, 0, 0, (uint4*)buf, kBytes / 16);
        hipLaunchKernelGGL(wr_seq8, g, b, 0, 0, (unsigned long long*)buf, kBytes / 8);
        hipLaunchKernelGGL(wr_stride40_4, g, b, 0, 0, (uint32_t*)buf, kBytes / 40);
        hipLaunchKernelGGL(at_scatter8, g, b, 0, 0, (unsigned long long*)buf, kBytes / 8, n_scatter);
        (void)hipDeviceSynchronize();
    }
    printf("true bytes: rd_wide16 %zu rd_stride40_4 %zu rd_seq8 %zu rd_scatter8 %zu rd_scatter16 %zu wr_wide16 %zu wr_seq8 %zu wr_stride40_4 %zu at_scatter8 %zu\n",
           kBytes, kBytes / 40 * 40, kBytes, n_scatter * 8, n_scatter * 16, kBytes, kBytes, kBytes / 40 * 40, n_scatter * 8);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
