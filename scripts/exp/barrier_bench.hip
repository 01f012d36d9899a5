// Experiment: cost of a grid-wide barrier on MI355X (persistent kernel, one atomic counter) and whether a matrix that was
// just streamed is served faster on the second pass (L2 / Infinity Cache residency). Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, long long deadline) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);   // system scope default; use agent below
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() > deadline) { ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counter, int iters, int* data, int* bad, long long budget) {
    const long long deadline = wall_clock64() + budget;
    const unsigned nb = gridDim.x;
    for (int i = 0; i < iters; ++i) {
        if (threadIdx.x == 0) data[blockIdx.x] = i + 1;
        if (!grid_barrier(counter, (unsigned)(i + 1) * nb, deadline)) { if (threadIdx.x == 0) atomicAdd(bad, 1000000); return; }
        if (threadIdx.x == 0) {
            const int v = __builtin_nontemporal_load(&data[(blockIdx.x + 77) % nb]);
            if (v != i + 1 && v != i + 2) atomicAdd(bad, 1);
        }
        // second barrier so that nobody overwrites data before everyone has read (two per iteration)
        if (!grid_barrier(counter + 32, (unsigned)(i + 1) * nb, deadline)) { if (threadIdx.x == 0) atomicAdd(bad, 1000000); return; }
    }
}

__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ w, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v = w[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    unsigned* counter; int *data, *bad;
    CK(hipMalloc(&counter, 4096)); CK(hipMalloc(&data, 4096 * 4)); CK(hipMalloc(&bad, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, barrier_kernel, 256, 0));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("CUs %d, occupancy %d blocks/CU, wall clock rate %d kHz\n", p.multiProcessorCount, occ, p.clockRate);
    for (int grid : {64, 256, 512, 1024}) {
        if (grid > occ * p.multiProcessorCount) continue;
        const int iters = 2000;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(counter, 0, 4096)); CK(hipMemset(bad, 0, 4));
            CK(hipEventRecord(e0));
            barrier_kernel<<<grid, 256>>>(counter, iters, data, bad, 200000000LL /* 2 s at 100 MHz */);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            if (rep) printf("grid %4d: %.3f us per barrier (bad=%d)\n", grid, ms * 1e3 / (2 * iters), hb);
        }
    }
    // residency: stream `mb` MB twice back to back
    unsigned* sink; CK(hipMalloc(&sink, 4));
    for (size_t mb : {8, 24, 64, 128, 256, 512}) {
        const size_t bytes = mb << 20; uint4* w; CK(hipMalloc(&w, bytes)); CK(hipMemset(w, 1, bytes));
        uint4* flush; CK(hipMalloc(&flush, (size_t)1 << 30)); CK(hipMemset(flush, 2, (size_t)1 << 30));
        float t[3];
        for (int rep = 0; rep < 3; ++rep) {
            if (rep == 0) { stream_kernel<<<2048, 256>>>(flush, ((size_t)1 << 30) / 16, sink); }
            CK(hipEventRecord(e0));
            stream_kernel<<<2048, 256>>>(w, bytes / 16, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t[rep], e0, e1));
        }
        printf("stream %4zu MB: cold %.1f us (%.2f TB/s), 2nd %.1f us (%.2f TB/s), 3rd %.1f us\n", mb, t[0] * 1e3, bytes / t[0] / 1e9,
               t[1] * 1e3, bytes / t[1] / 1e9, t[2] * 1e3);
        CK(hipFree(w)); CK(hipFree(flush));
    }
    return 0;
}
