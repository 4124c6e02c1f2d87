// atomics_probe.hip -- how fast are coalesced global atomics from 256 workgroups into one small array?
// (Would the scan's epilogue do better adding its per-cell counts straight into a shared array than writing 256 slabs
// that a second kernel folds?)  Build: hipcc --offload-arch=gfx950 -O3 -o atomics_probe atomics_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(1024) void add_u64(unsigned long long* dst, unsigned n, int rotate) {
    const unsigned start = rotate ? (blockIdx.x * 7919u * 64u) % n : 0u;
    for (unsigned i = threadIdx.x; i < n; i += 1024) {
        unsigned j = start + i; if (j >= n) j -= n;
        atomicAdd(dst + j, (unsigned long long)(i + 1));
    }
}
__global__ __launch_bounds__(1024) void add_u32(unsigned* dst, unsigned n, int rotate) {
    const unsigned start = rotate ? (blockIdx.x * 7919u * 64u) % n : 0u;
    for (unsigned i = threadIdx.x; i < n; i += 1024) {
        unsigned j = start + i; if (j >= n) j -= n;
        atomicAdd(dst + j, i + 1);
    }
}
__global__ __launch_bounds__(1024) void add_u64_wg(unsigned long long* dst, unsigned n, int rotate) {   // L2-local (workgroup scope)
    const unsigned start = rotate ? (blockIdx.x * 7919u * 64u) % n : 0u;
    for (unsigned i = threadIdx.x; i < n; i += 1024) {
        unsigned j = start + i; if (j >= n) j -= n;
        __hip_atomic_fetch_add(dst + j, (unsigned long long)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
__global__ __launch_bounds__(1024) void store_slab(unsigned* dst, unsigned n) {
    for (unsigned i = threadIdx.x; i < n; i += 1024) dst[(size_t)blockIdx.x * n + i] = i + blockIdx.x;
}
__global__ __launch_bounds__(256) void fold_slabs(const unsigned* slabs, unsigned n, unsigned n_slabs, unsigned long long* out) {
    const unsigned per = (n_slabs + gridDim.y - 1) / gridDim.y, b0 = blockIdx.y * per, b1 = min(n_slabs, b0 + per);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        unsigned long long s = 0;
        for (unsigned b = b0; b < b1; ++b) s += slabs[(size_t)b * n + i];
        if (s) atomicAdd(out + i, s);
    }
}

int main() {
    const unsigned n = 29903, grid = 256;
    unsigned long long* d64; unsigned* d32; unsigned* slabs;
    hipMalloc(&d64, n * 8); hipMalloc(&d32, n * 4); hipMalloc(&slabs, (size_t)grid * n * 4);
    hipMemset(d64, 0, n * 8); hipMemset(d32, 0, n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto timeit = [&](const char* name, auto&& launch) {
        for (int i = 0; i < 3; i++) launch();
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 20; i++) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s %8.2f us per launch\n", name, ms / 20 * 1e3);
    };
    timeit("u64 agent-scope atomics, same order", [&] { hipLaunchKernelGGL(add_u64, dim3(grid), dim3(1024), 0, 0, d64, n, 0); });
    timeit("u64 agent-scope atomics, rotated start", [&] { hipLaunchKernelGGL(add_u64, dim3(grid), dim3(1024), 0, 0, d64, n, 1); });
    timeit("u32 agent-scope atomics, rotated start", [&] { hipLaunchKernelGGL(add_u32, dim3(grid), dim3(1024), 0, 0, d32, n, 1); });
    timeit("u64 workgroup-scope atomics, rotated", [&] { hipLaunchKernelGGL(add_u64_wg, dim3(grid), dim3(1024), 0, 0, d64, n, 1); });
    timeit("slab store (256 x 120 KB)", [&] { hipLaunchKernelGGL(store_slab, dim3(grid), dim3(1024), 0, 0, slabs, n); });
    timeit("slab fold (read 30.6 MB)", [&] { hipLaunchKernelGGL(fold_slabs, dim3(117, 16), dim3(256), 0, 0, slabs, n, grid, d64); });
    return 0;
}
