// Shared device helpers for the spider_hip kernels (gfx950 / CDNA4 only).
// Wavefront = 64 lanes; all bf16 tensors are passed as raw uint16_t bit patterns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spider {

typedef uint16_t bf16_t;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;  // 16-byte global/LDS access unit
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
__device__ __forceinline__ float bf16lo_to_f32(uint32_t packed) {
    return __uint_as_float(packed << 16);
}
__device__ __forceinline__ float bf16hi_to_f32(uint32_t packed) {
    return __uint_as_float(packed & 0xffff0000u);
}
// round-to-nearest-even f32 -> bf16 through the hardware converter (v_cvt_pk_bf16_f32 on gfx950; NaN stays NaN).
// A software rounding sequence costs ~7 VALU ops per element -- it dominated the flash-attention softmax.
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const hw_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}

// ---- 16-bit storage / operand type of the diffusion-side translation units (gemm, attention, UNet ops, fused cross-attention) ----
// Those files are compiled twice: as bf16 (LLM prefill, towers, CLIP at the BASELINE dtype) and, with -DSPIDER_F16, as IEEE
// half -- the reference's own diffusion dtype (torch_dtype=torch.float16, spider_decoder.py:109,114; base_model.py:211;
// Comic_Generation.py:313). gfx950 runs mfma_f32_*_f16 at the bf16 rate; three more significand bits bring one UNet
// evaluation from 1.5e-2 to 1.6e-3 of the fp32 oracle. The h16_* helpers below are the only place
// the two instantiations differ; entry points are named through SPIDER_FN: spider_gemm -> spider_gemm_bf16 / spider_gemm_f16.
typedef uint16_t h16_t;
typedef bf16x8 h16x8;
typedef bf16x4 h16x4;
#define SPIDER_CAT2_(a, b) a##_##b
#define SPIDER_CAT2(a, b) SPIDER_CAT2_(a, b)
#ifdef SPIDER_F16
#define SPIDER_DT f16
#define SPIDER_DT_NAME "f16"
typedef _Float16 hw_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 hw_f16x8 __attribute__((ext_vector_type(8)));
constexpr uint32_t H16_ONE = 0x3C00u;   // 1.0
__device__ __forceinline__ float h16_to_f32(h16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ float h16lo_to_f32(uint32_t packed) { return (float)__builtin_bit_cast(hw_f16x2, packed)[0]; }
__device__ __forceinline__ float h16hi_to_f32(uint32_t packed) { return (float)__builtin_bit_cast(hw_f16x2, packed)[1]; }
// round-to-nearest-even (v_cvt_f16_f32 / v_cvt_pk_f16_f32); values beyond 65504 become +-inf as in torch.float16
__device__ __forceinline__ h16_t f32_to_h16(float f) { return __builtin_bit_cast(h16_t, (_Float16)f); }
__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi) {
    const hw_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_f16x2));
}
__device__ __forceinline__ float dot2_h16(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(hw_f16x2, a), __builtin_bit_cast(hw_f16x2, b), acc, false);
}
__device__ __forceinline__ f32x4 mfma_16x16x32_h16(h16x8 a, h16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hw_f16x8, a), __builtin_bit_cast(hw_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x16_h16(h16x8 a, h16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hw_f16x8, a), __builtin_bit_cast(hw_f16x8, b), c, 0, 0, 0);
}
typedef _Float16 hw_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_16x16x16_h16(h16x4 a, h16x4 b, f32x4 c) {     // K = 16: 4 values per lane
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(hw_f16x4, a), __builtin_bit_cast(hw_f16x4, b), c, 0, 0, 0);
}
#else
#define SPIDER_DT bf16
#define SPIDER_DT_NAME "bf16"
constexpr uint32_t H16_ONE = 0x3F80u;   // 1.0
__device__ __forceinline__ float h16_to_f32(h16_t v) { return bf16_to_f32(v); }
__device__ __forceinline__ float h16lo_to_f32(uint32_t packed) { return bf16lo_to_f32(packed); }
__device__ __forceinline__ float h16hi_to_f32(uint32_t packed) { return bf16hi_to_f32(packed); }
__device__ __forceinline__ h16_t f32_to_h16(float f) { return f32_to_bf16(f); }
__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi) { return pack_bf16x2(lo, hi); }
__device__ __forceinline__ float dot2_h16(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hw_bf16x2, a), __builtin_bit_cast(hw_bf16x2, b), acc, false);
}
__device__ __forceinline__ f32x4 mfma_16x16x32_h16(h16x8 a, h16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x16_h16(h16x8 a, h16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_16x16x16_h16(h16x4 a, h16x4 b, f32x4 c) {     // K = 16: 4 values per lane
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
#endif
#define SPIDER_FN(name) SPIDER_CAT2(name, SPIDER_DT)

// acc += a.lo*b.lo + a.hi*b.hi on packed bf16 pairs (v_dot2c_f32_bf16): no unpacking, 1 VALU op per 2 MACs
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hw_bf16x2, a), __builtin_bit_cast(hw_bf16x2, b), acc, false);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blocks of NW waves; `red` is >= NW floats of LDS. All threads get the result.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}

// x * rcp(...) instead of x / (...): an IEEE fp32 division is a ~10-instruction sequence, v_rcp_f32 is 1 ulp
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
// exact-erf GELU (torch F.gelu default, diffusers GEGLU / CLIP-H / CLAP). erf through Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, two orders below bf16 resolution): branch-free, 2 transcendentals + 7 FMAs, where libdevice's
// erff is a ~50-instruction divergent polynomial -- it was ~25 % of the fused GEGLU GEMM's time.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.f - p * t * __expf(-ax * ax);
    return copysignf(e, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float quick_gelu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * x)); }

// XCD-aware bijective remap of a linear block id: blocks that share an XCD (id % 8) get a
// contiguous chunk of the logical grid, so neighbouring tiles hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace spider

// Dynamic LDS above 64 KiB must be enabled per function AND per device (one process may drive engines on several GPUs): raise it
// once for each device the function is launched on. `done_mask` is the caller's static bit mask (bit = device ordinal mod 32).
template <typename F>
inline void raise_dynamic_lds(F* fn, int bytes, unsigned& done_mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (!(done_mask & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        done_mask |= bit;
    }
}

// ---- error plumbing shared by the C-ABI translation units ----
extern "C" void spider_set_error(const char* msg);
extern "C" int g_spider_ws_inlaunch;      // capi.cpp: split-K combine form of the streaming conv (spider_set_ws_inlaunch)
#define SPIDER_CHECK(cond, msg)            \
    do {                                   \
        if (!(cond)) {                     \
            spider_set_error(msg);         \
            return -1;                     \
        }                                  \
    } while (0)
#define SPIDER_LAUNCH_OK()                                        \
    do {                                                          \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess) {                                  \
            spider_set_error(hipGetErrorString(e__));             \
            return -2;                                            \
        }                                                         \
    } while (0)
