// Experiment (round 4): is a per-layer PERSISTENT decode kernel (phases separated by an XCD-hierarchical grid barrier) faster than the
// chain of dependent launches the decode graph uses today -- alone on the chip, and while a second stream runs a UNet-like chain of
// short kernels (the two-stream schedule of the headline)? Nothing here computes anything real: a "phase" streams the bytes of one
// decode GEMV (Qwen2.5-7B layer: qkv 33 MB, attention 3.3 MB, combine 1 MB, o 25.7 MB, gate/up 271.6 MB, down 135.8 MB) with
// non-temporal 16-byte loads, reads the 256 partial results of the previous phase (the "activation vector" every block needs) and
// writes its own. Chain = one launch per phase in a hipGraph; persistent = one launch per layer (6 phases, 5 barriers).
// Build: hipcc --offload-arch=gfx950 -O3 -o persist_vs_chain persist_vs_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NWG = 256, NT = 256, NPH = 6, NLAYER = 28;
__constant__ unsigned long long c_phase_bytes[NPH];

struct Bar {                                   // every word on a 128-byte line of its own
    unsigned xcc_cnt[8][32];
    unsigned top[32];
    unsigned gen[8][32];
    unsigned pop[8][32];
    unsigned census[32];
    unsigned abort_[32];
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u; }   // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// XCD-hierarchical barrier: arrive on the XCC's counter; the XCC's last arriver arrives on the top counter, waits for all XCCs, then
// bumps its XCC's generation word, which the other workgroups of that XCC poll (L2-local). Bounded spins: a timeout sets abort_.
__device__ __forceinline__ bool grid_barrier(Bar* bar, unsigned x, unsigned epoch, unsigned n_xcc, long long deadline) {
    __syncthreads();
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        bool ok = true;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned pop = ld_sc1(&bar->pop[x][0]);
        const unsigned old = __hip_atomic_fetch_add(&bar->xcc_cnt[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == pop * epoch) {          // last of this XCC
            __hip_atomic_fetch_add(&bar->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_sc1(&bar->top[0]) < n_xcc * epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() > deadline || ld_sc1(&bar->abort_[0])) { ok = false; break; }
            }
            __hip_atomic_store(&bar->gen[x][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_sc1(&bar->gen[x][0]) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() > deadline || ld_sc1(&bar->abort_[0])) { ok = false; break; }
            }
        }
        if (!ok) __hip_atomic_store(&bar->abort_[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ok_s = ok ? 1 : 0;
    }
    __syncthreads();
    return ok_s != 0;
}

// one phase of one workgroup: stream `bytes / NWG` bytes starting at this workgroup's slice, 8 x 16-byte loads in flight per lane
__device__ __forceinline__ unsigned stream_slice(const uint4* __restrict__ w, size_t off_vec, size_t n_vec, unsigned acc) {
    const u32x4* p = reinterpret_cast<const u32x4*>(w) + off_vec;
    size_t i = threadIdx.x;
    for (; i + 7 * NT < n_vec; i += 8 * NT) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * NT);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n_vec; i += NT) { u32x4 v = __builtin_nontemporal_load(p + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    return acc;
}

// the same slice with its first 8 loads per lane already in registers (issued before the barrier: weights do not depend on activations)
__device__ __forceinline__ void prefetch8(const uint4* __restrict__ w, size_t off_vec, size_t n_vec, u32x4 (&v)[8]) {
    const u32x4* p = reinterpret_cast<const u32x4*>(w) + off_vec;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const size_t i = threadIdx.x + (size_t)u * NT;
        v[u] = __builtin_nontemporal_load(p + (i < n_vec ? i : 0));
    }
}
__device__ __forceinline__ unsigned stream_slice_pf(const uint4* __restrict__ w, size_t off_vec, size_t n_vec, unsigned acc, u32x4 (&v0)[8]) {
    const u32x4* p = reinterpret_cast<const u32x4*>(w) + off_vec;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v0[u].x ^ v0[u].y ^ v0[u].z ^ v0[u].w;
    size_t i = threadIdx.x + (size_t)8 * NT;
    for (; i + 7 * NT < n_vec; i += 8 * NT) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * NT);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n_vec; i += NT) { u32x4 v = __builtin_nontemporal_load(p + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    return acc;
}

__device__ __forceinline__ size_t phase_base(size_t w_vecs, int layer, int ph, int wg, size_t per) {
    return ((size_t)(layer * NPH + ph) * 7919u * 4096u + (size_t)wg * per) % (w_vecs - per - 1);
}

__device__ __forceinline__ void phase_body(const uint4* __restrict__ w, size_t w_vecs, int layer, int ph, unsigned* act, int wg) {
    // dependency: read every workgroup's result of the previous phase (1 KB, like the activation vector), then stream, then publish
    const unsigned* prev = act + (size_t)((layer * NPH + ph + 1) & 1) * NWG;
    unsigned a = prev[threadIdx.x % NWG];
    const size_t per = (size_t)(c_phase_bytes[ph] / 16 / NWG);
    a = stream_slice(w, phase_base(w_vecs, layer, ph, wg, per), per, a);
    a ^= __shfl_xor(a, 32, 64); a ^= __shfl_xor(a, 16, 64);
    if (threadIdx.x == 0) act[(size_t)((layer * NPH + ph) & 1) * NWG + wg] = a;
}

__global__ __launch_bounds__(NT) void phase_kernel(const uint4* __restrict__ w, size_t w_vecs, int layer, int ph, unsigned* act) {
    phase_body(w, w_vecs, layer, ph, act, blockIdx.x);
}

__global__ __launch_bounds__(NT) void census_kernel(Bar* bar) {
    if (threadIdx.x == 0) atomicAdd(&bar->pop[xcc_id()][0], 1u);
}

// persistent layer with the next phase's first loads issued BEFORE the barrier
__global__ __launch_bounds__(NT) void layer_pf_kernel(const uint4* __restrict__ w, size_t w_vecs, int layer, unsigned* act, Bar* bar,
                                                      unsigned epoch0, unsigned n_xcc, long long budget) {
    const long long deadline = wall_clock64() + budget;
    const unsigned x = xcc_id();
    const int wg = blockIdx.x;
    u32x4 v0[8];
    {
        const size_t per = (size_t)(c_phase_bytes[0] / 16 / NWG);
        prefetch8(w, phase_base(w_vecs, layer, 0, wg, per), per, v0);
    }
    for (int ph = 0; ph < NPH; ++ph) {
        const unsigned* prev = act + (size_t)((layer * NPH + ph + 1) & 1) * NWG;
        unsigned a = prev[threadIdx.x % NWG];
        const size_t per = (size_t)(c_phase_bytes[ph] / 16 / NWG);
        a = stream_slice_pf(w, phase_base(w_vecs, layer, ph, wg, per), per, a, v0);
        a ^= __shfl_xor(a, 32, 64); a ^= __shfl_xor(a, 16, 64);
        if (threadIdx.x == 0) act[(size_t)((layer * NPH + ph) & 1) * NWG + wg] = a;
        if (ph + 1 < NPH) {
            const size_t pern = (size_t)(c_phase_bytes[ph + 1] / 16 / NWG);
            prefetch8(w, phase_base(w_vecs, layer, ph + 1, wg, pern), pern, v0);
            if (!grid_barrier(bar, x, epoch0 + ph + 1, n_xcc, deadline)) return;
        }
    }
}

// one layer (NPH phases) per launch; epoch0 = barriers passed before this launch
__global__ __launch_bounds__(NT) void layer_kernel(const uint4* __restrict__ w, size_t w_vecs, int layer, unsigned* act, Bar* bar,
                                                   unsigned epoch0, unsigned n_xcc, long long budget) {
    const long long deadline = wall_clock64() + budget;
    const unsigned x = xcc_id();
    for (int ph = 0; ph < NPH; ++ph) {
        phase_body(w, w_vecs, layer, ph, act, blockIdx.x);
        if (ph + 1 < NPH) {
            if (!grid_barrier(bar, x, epoch0 + ph + 1, n_xcc, deadline)) return;
        }
    }
}

// UNet-like background load: a short kernel with LDS use and a dependent FMA loop (~10 us), launched as a dependent chain
__global__ __launch_bounds__(512) void bg_kernel(float* out, int iters, const uint4* __restrict__ mem, size_t mem_vecs, size_t vecs_per_block, int seq) {
    extern __shared__ float sm[];
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f;
    {   // memory part: every block reads vecs_per_block 16-byte vectors (activations + weights of a UNet kernel)
        const size_t base = ((size_t)seq * 104729u * 1024u + (size_t)blockIdx.x * vecs_per_block) % (mem_vecs - vecs_per_block - 1);
        unsigned acc = 0;
        for (size_t i = threadIdx.x; i < vecs_per_block; i += 512) { const uint4 v = mem[base + i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        a += (float)(acc & 1);
    }
    for (int i = threadIdx.x; i < 8192; i += 512) sm[i] = a + i;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        a = fmaf(a, 1.0001f, sm[(threadIdx.x * 7 + i) & 8191]);
        b = fmaf(b, 0.9999f, a);
    }
    if (a + b == 123.456f) out[blockIdx.x] = a;
}

int main(int argc, char** argv) {
    const unsigned long long phase_bytes[NPH] = {33030144ull, 3300000ull, 1000000ull, 25690112ull, 271581184ull, 135790592ull};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(c_phase_bytes), phase_bytes, sizeof(phase_bytes)));
    const size_t w_bytes = (size_t)1 << 30;
    uint4* w; CK(hipMalloc(&w, w_bytes)); CK(hipMemset(w, 1, w_bytes));
    unsigned* act; CK(hipMalloc(&act, 2 * NWG * 4)); CK(hipMemset(act, 0, 2 * NWG * 4));
    Bar* bar; CK(hipMalloc(&bar, sizeof(Bar))); CK(hipMemset(bar, 0, sizeof(Bar)));
    float* bgout; CK(hipMalloc(&bgout, 4096 * 4));
    hipStream_t sL, sU; CK(hipStreamCreate(&sL)); CK(hipStreamCreate(&sU));
    census_kernel<<<NWG, NT, 0, sL>>>(bar); CK(hipStreamSynchronize(sL));
    Bar hb; CK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
    unsigned n_xcc = 0; printf("census (workgroups per XCC of a %d-block grid):", NWG);
    for (int x = 0; x < 8; ++x) { printf(" %u", hb.pop[x][0]); n_xcc += hb.pop[x][0] > 0; }
    printf("  -> %u XCCs\n", n_xcc);
    const size_t w_vecs = w_bytes / 16;
    double total_mb = 0; for (int p = 0; p < NPH; ++p) total_mb += phase_bytes[p] / 1e6;

    // graphs: chain = NLAYER x NPH launches; persistent = NLAYER launches
    hipGraph_t g; hipGraphExec_t chain, pers, bg;
    CK(hipStreamBeginCapture(sL, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < NLAYER; ++l) for (int p = 0; p < NPH; ++p) phase_kernel<<<NWG, NT, 0, sL>>>(w, w_vecs, l, p, act);
    CK(hipStreamEndCapture(sL, &g)); CK(hipGraphInstantiate(&chain, g, nullptr, nullptr, 0));
    // persistent: the barrier epochs continue across launches and replays, so the epoch base is read from a device word? keep it simple:
    // epochs restart every replay -> counters are reset by a memset node at the head of the graph
    CK(hipStreamBeginCapture(sL, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(bar->xcc_cnt, 0, sizeof(hb.xcc_cnt) + sizeof(hb.top) + sizeof(hb.gen), sL));
    const long long budget = 100000000ll * 2;   // wall_clock64 ticks at 100 MHz: 2 s
    for (int l = 0; l < NLAYER; ++l)
        layer_kernel<<<NWG, NT, 0, sL>>>(w, w_vecs, l, act, bar, (unsigned)(l * (NPH - 1)), n_xcc, budget);
    CK(hipStreamEndCapture(sL, &g)); CK(hipGraphInstantiate(&pers, g, nullptr, nullptr, 0));
    hipGraphExec_t pers_pf;
    CK(hipStreamBeginCapture(sL, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(bar->xcc_cnt, 0, sizeof(hb.xcc_cnt) + sizeof(hb.top) + sizeof(hb.gen), sL));
    for (int l = 0; l < NLAYER; ++l)
        layer_pf_kernel<<<NWG, NT, 0, sL>>>(w, w_vecs, l, act, bar, (unsigned)(l * (NPH - 1)), n_xcc, budget);
    CK(hipStreamEndCapture(sL, &g)); CK(hipGraphInstantiate(&pers_pf, g, nullptr, nullptr, 0));
    const int bg_iters = argc > 1 ? atoi(argv[1]) : 2500;
    const double bg_mb = argc > 2 ? atof(argv[2]) : 0.0;       // MB read per background kernel
    const int bg_lds = argc > 3 ? atoi(argv[3]) : 32768;       // dynamic LDS bytes per background block
    CK(hipFuncSetAttribute((const void*)bg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    const size_t bg_vpb = (size_t)(bg_mb * 1e6 / 16 / 256);
    CK(hipStreamBeginCapture(sU, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 370; ++i) bg_kernel<<<256, 512, bg_lds, sU>>>(bgout, bg_iters, w, w_vecs, bg_vpb, i);
    CK(hipStreamEndCapture(sU, &g)); CK(hipGraphInstantiate(&bg, g, nullptr, nullptr, 0));

    hipEvent_t e0, e1, b0, b1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    auto time_graph = [&](hipGraphExec_t ge, hipStream_t st, int reps) {
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
    };
    const float bg_alone = time_graph(bg, sU, 5);
    printf("background chain alone: %.3f ms per 370 launches (%.1f us each)\n", bg_alone, bg_alone * 1e3 / 370);
    printf("background kernel: %d fma iterations, %.1f MB read, %d B LDS per block\n", bg_iters, bg_mb, bg_lds);
    for (int mode = 0; mode < 3; ++mode) {
        hipGraphExec_t ge = mode == 2 ? pers_pf : (mode ? pers : chain);
        const char* name = mode == 2 ? "persistent + loads hoisted over the barrier" : (mode ? "persistent (1 launch / layer, 5 barriers)" : "chain (6 launches / layer)");
        const float alone = time_graph(ge, sL, 10);
        // co-run: keep the background stream busy for the whole measurement
        const int reps = 10, bg_reps = (int)(alone * reps * 2.5f / bg_alone) + 4;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(b0, sU));
        for (int r = 0; r < bg_reps; ++r) CK(hipGraphLaunch(bg, sU));
        CK(hipEventRecord(b1, sU));
        CK(hipGraphLaunch(ge, sL));
        CK(hipEventRecord(e0, sL));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, sL));
        CK(hipEventRecord(e1, sL)); CK(hipEventSynchronize(e1));
        const bool bg_still = hipEventQuery(b1) == hipErrorNotReady;
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const float corun = ms / reps;
        CK(hipDeviceSynchronize());
        float bms; CK(hipEventElapsedTime(&bms, b0, b1));
        CK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
        printf("%-44s alone %.3f ms/token-body (%.1f us/layer, %.2f TB/s)   co-run %.3f ms (%.1f us/layer)%s   background %.3f ms per replay (alone %.3f)  abort=%u\n",
               name, alone, alone * 1e3 / NLAYER, total_mb * NLAYER / alone * 1e-3, corun, corun * 1e3 / NLAYER,
               bg_still ? "" : " [background ended early]", bms / bg_reps, bg_alone, hb.abort_[0]);
    }
    return 0;
}
