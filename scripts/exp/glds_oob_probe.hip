// Probe (MI355X): does `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer descriptor) write ZEROS into LDS for lanes whose
// offset is out of range? (needed to fold the conv halo / M,N,K tails into the LDS-DMA staging of the implicit-GEMM kernel)
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 scripts/exp/glds_oob_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__global__ void k(const uint16_t* src, uint32_t bytes, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) char lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) ((uint32_t*)lds)[i] = 0xDEADBEEFu;   // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(src), 0, bytes, 0x00020000);
    uint32_t off = lane * 16;
    if (lane & 1) off |= 0xFFFFFFFFu;          // odd lanes: out of range
    if (lane >= 48) off = bytes - 8 + (lane - 48) * 16;   // straddles / beyond the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = ((uint32_t*)lds)[lane * 4 + j];
}
int main() {
    const int n = 4096;
    std::vector<uint16_t> h(n);
    for (int i = 0; i < n; ++i) h[i] = (uint16_t)(i + 1);
    uint16_t* d; uint32_t* o;
    hipMalloc(&d, n * 2); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, 1024, o);   // descriptor covers 1024 bytes only
    std::vector<uint32_t> r(256);
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        uint32_t off = l * 16; bool oob = l & 1;
        if (l >= 48) { off = 1024 - 8 + (l - 48) * 16; oob = false; }
        for (int j = 0; j < 4; ++j) {
            uint32_t byte = off + j * 4;
            uint32_t exp = (oob || byte + 4 > 1024) ? 0u : ((uint32_t)h[byte / 2] | ((uint32_t)h[byte / 2 + 1] << 16));
            if (r[l * 4 + j] != exp) { if (bad < 12) printf("lane %d dword %d: got %08x expected %08x\n", l, j, r[l * 4 + j], exp); ++bad; }
        }
    }
    printf("glds OOB probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "OOB lanes write zeros, in-range lanes write data", bad);
    return 0;
}
