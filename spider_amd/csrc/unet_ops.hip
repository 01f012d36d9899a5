// HBM-bound UNet / VAE / text-encoder operators on NHWC bf16 activations (gfx950).
//
// Reference call sites: the diffusion loop of spider/models/custom_sd.py:627-652 calls
// UNet2DConditionModel.forward (external diffusers==0.25.0); these kernels implement its
// GroupNorm(32)+SiLU (ResnetBlock2D.norm1/norm2, Transformer2DModel.norm, conv_norm_out), LayerNorm
// (BasicTransformerBlock.norm1/2/3), GEGLU (FeedForward), skip concat (UpBlock2D), conv_in / conv_out,
// plus the classifier-free-guidance combine (custom_sd.py:642-644) and the scheduler linear update
// (custom_sd.py:647). All statistics are fp32; activations are vectorised 16 B per lane.
#include "common.hpp"

using namespace spider;

namespace {

// ------------------------------------------------------------------------------------------
// GroupNorm, pass 1: per (batch, pixel-chunk) partial sums of every group.
// Block = (C/8)*KP threads: thread -> fixed 8-channel vector, pixel lane; so per-channel sums live
// in registers and are reduced over pixel lanes through LDS. partial: [B, nchunk, G, 2] fp32.
// ------------------------------------------------------------------------------------------
// Skip-concat form (x2 != nullptr): the input is the channel concatenation [x | x2] (UpBlock2D's cat([hidden, skip])) with
// C1 channels in x; the kernel reads the two sources in place and writes the concatenated tensor to `cat` on the way (the
// shortcut conv and pass 2 read it) -- the separate concat launch and its read pass are gone.
// X32: the input is the fp32 master of the residual stream (precise mode: no cat form)
template <bool X32>
__global__ void gn_stats_kernel(const h16_t* __restrict__ x, float* __restrict__ partial, int HW, int C, int G,
                                int nchunk, int KP, const h16_t* __restrict__ x2, int C1, h16_t* __restrict__ cat) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sm_s = reinterpret_cast<float*>(smem);   // [KP][C]
    float* sm_q = sm_s + KP * C;                    // [KP][C]
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int cv = C / 8;
    const int v = threadIdx.x % cv, pl = threadIdx.x / cv;
    const int per = (HW + nchunk - 1) / nchunk;
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    // this thread's 8-channel vector comes from x (row stride ld = C or C1) or, past C1, from x2 (row stride C - C1)
    const bool second = x2 != nullptr && v * 8 >= C1;
    const int ld = second ? C - C1 : (x2 ? C1 : C);
    const h16_t* xb = (second ? x2 + (size_t)b * HW * ld + (v * 8 - C1) : x + (size_t)b * HW * ld + v * 8);
    // 4 independent 16-byte loads in flight per thread (a runtime-trip loop with one load per iteration would
    // serialise the L2/HBM round trips)
    if (X32) {
        const float* xf = reinterpret_cast<const float*>(x) + (size_t)b * HW * C + v * 8;
        for (int px = p0 + pl; px < p1; px += 4 * KP) {
            f32x4 a[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pp = px + u * KP;
                const bool ok = pp < p1;
                a[u][0] = ok ? *reinterpret_cast<const f32x4*>(xf + (size_t)pp * C) : f32x4{0.f, 0.f, 0.f, 0.f};
                a[u][1] = ok ? *reinterpret_cast<const f32x4*>(xf + (size_t)pp * C + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float t = a[u][j >> 2][j & 3]; s[j] += t; q[j] += t * t; }
        }
    } else
    for (int px = p0 + pl; px < p1; px += 4 * KP) {
        u32x4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pp = px + u * KP;
            a[u] = pp < p1 ? *reinterpret_cast<const u32x4*>(xb + (size_t)pp * ld) : u32x4{0u, 0u, 0u, 0u};
        }
        if (cat) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pp = px + u * KP;
                if (pp < p1) *reinterpret_cast<u32x4*>(cat + ((size_t)b * HW + pp) * C + v * 8) = a[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t aw[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = h16lo_to_f32(aw[j]), hi = h16hi_to_f32(aw[j]);
                s[2 * j] += lo; q[2 * j] += lo * lo;
                s[2 * j + 1] += hi; q[2 * j + 1] += hi * hi;
            }
        }
    }
    *reinterpret_cast<f32x4*>(sm_s + pl * C + v * 8) = f32x4{s[0], s[1], s[2], s[3]};
    *reinterpret_cast<f32x4*>(sm_s + pl * C + v * 8 + 4) = f32x4{s[4], s[5], s[6], s[7]};
    *reinterpret_cast<f32x4*>(sm_q + pl * C + v * 8) = f32x4{q[0], q[1], q[2], q[3]};
    *reinterpret_cast<f32x4*>(sm_q + pl * C + v * 8 + 4) = f32x4{q[4], q[5], q[6], q[7]};
    __syncthreads();
    // channel totals over the KP pixel lanes, kept in row 0
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float ts = 0.f, tq = 0.f;
        for (int k = 0; k < KP; ++k) { ts += sm_s[k * C + c]; tq += sm_q[k * C + c]; }
        sm_s[c] = ts;
        sm_q[c] = tq;
    }
    __syncthreads();
    const int cpg = C / G;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        float gs = 0.f, gq = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { gs += sm_s[c]; gq += sm_q[c]; }
        float* dst = partial + (((size_t)b * nchunk + chunk) * G + g) * 2;
        dst[0] = gs;
        dst[1] = gq;
    }
}

// GroupNorm, pass 2: y = (x - mean_g) * rstd_g * gamma_c + beta_c, optional SiLU; output bf16.
// Rounds once after the affine (as torch's GroupNorm does) and once more after SiLU.
// X32 (precise mode): x is the fp32 master, the value is rounded ONCE on the way out (no 16-bit stop between the affine and SiLU),
// and y32 (optional) receives it unrounded for a consumer that splits it into hi / lo operand halves.
template <bool SILU, bool X32 = false>
__global__ void gn_apply_kernel(const h16_t* __restrict__ x, const float* __restrict__ partial,
                                const h16_t* __restrict__ gamma, const h16_t* __restrict__ beta,
                                h16_t* __restrict__ y, int HW, int C, int G, int nchunk,
                                float eps, int pix_per_block, int KP, float* __restrict__ y32 = nullptr, int ysplit = 0) {
    // Block = (C/8)*KP threads like pass 1: a thread owns one 8-channel vector for all its pixels, so the per-channel
    // scale a_c = rstd_g * gamma_c and shift b_c = beta_c - mean_g * a_c live in 16 registers and the inner loop is one fma
    // (+ SiLU) per element -- no per-element group lookup (an integer division by a runtime C/G and two LDS reads before).
    __shared__ float mean[64], rstd[64];
    __shared__ float ps[1024], pq[1024];
    const int b = blockIdx.y;
    const int cpg = C / G;
    const int nthr = blockDim.x;
    {
        const int parts = nthr / G;                 // >= 2 for every supported shape (nthr >= 128, G <= 64)
        const int g = threadIdx.x % G, part = threadIdx.x / G;
        float gs = 0.f, gq = 0.f;
        if (part < parts) {
            for (int c0 = part; c0 < nchunk; c0 += 8 * parts) {   // 8 independent loads per round
                float2 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + u * parts;
                    t[u] = c < nchunk ? *reinterpret_cast<const float2*>(partial + (((size_t)b * nchunk + c) * G + g) * 2)
                                      : float2{0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { gs += t[u].x; gq += t[u].y; }
            }
        }
        ps[threadIdx.x] = gs;
        pq[threadIdx.x] = gq;
        __syncthreads();
        if (threadIdx.x < G) {
            gs = 0.f; gq = 0.f;
            for (int k = 0; k < parts; ++k) { gs += ps[k * G + threadIdx.x]; gq += pq[k * G + threadIdx.x]; }
            const float n = (float)HW * (float)cpg;
            const float mu = gs / n;
            const float var = fmaxf(gq / n - mu * mu, 0.f);
            mean[threadIdx.x] = mu;
            rstd[threadIdx.x] = rsqrtf(var + eps);
        }
    }
    __syncthreads();
    const int cv = C / 8;
    const int v = threadIdx.x % cv, pl = threadIdx.x / cv;
    float ca[8], cb[8];
    {
        const u32x4 gq = *reinterpret_cast<const u32x4*>(gamma + v * 8);
        const u32x4 bq = *reinterpret_cast<const u32x4*>(beta + v * 8);
        const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w}, bw[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (v * 8 + j) / cpg;
            const float ga = (j & 1) ? h16hi_to_f32(gw[j >> 1]) : h16lo_to_f32(gw[j >> 1]);
            const float be = (j & 1) ? h16hi_to_f32(bw[j >> 1]) : h16lo_to_f32(bw[j >> 1]);
            ca[j] = rstd[g] * ga;
            cb[j] = be - mean[g] * ca[j];
        }
    }
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(HW, p0 + pix_per_block);
    const h16_t* xb = x + (size_t)b * HW * C + v * 8;
    h16_t* yb = y + (size_t)b * HW * C + v * 8;
    if (X32) {
        const float* xf = reinterpret_cast<const float*>(x) + (size_t)b * HW * C + v * 8;
        float* y32b = y32 ? y32 + (size_t)b * HW * C + v * 8 : nullptr;
        for (int px = p0 + pl; px < p1; px += 4 * KP) {
            f32x4 a[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pp = px + u * KP;
                const bool ok = pp < p1;
                a[u][0] = ok ? *reinterpret_cast<const f32x4*>(xf + (size_t)pp * C) : f32x4{0.f, 0.f, 0.f, 0.f};
                a[u][1] = ok ? *reinterpret_cast<const f32x4*>(xf + (size_t)pp * C + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pp = px + u * KP;
                if (pp >= p1) break;
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float t = fmaf(a[u][j >> 2][j & 3], ca[j], cb[j]);
                    if (SILU) t = silu_f(t);
                    o[j] = t;
                }
                if (y32b) {
                    *reinterpret_cast<f32x4*>(y32b + (size_t)pp * C) = f32x4{o[0], o[1], o[2], o[3]};
                    *reinterpret_cast<f32x4*>(y32b + (size_t)pp * C + 4) = f32x4{o[4], o[5], o[6], o[7]};
                }
                if (y && ysplit) {
                    // operand-split form: y is [B, HW, 2C], channels [0, C) the rounded value, [C, 2C) the rounded remainder
                    h16_t* ys = y + ((size_t)b * HW + pp) * 2 * C + v * 8;
                    uint32_t hh[4], ll[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        hh[j] = pack_h16x2(o[2 * j], o[2 * j + 1]);
                        ll[j] = pack_h16x2(o[2 * j] - h16lo_to_f32(hh[j]), o[2 * j + 1] - h16hi_to_f32(hh[j]));
                    }
                    *reinterpret_cast<u32x4*>(ys) = u32x4{hh[0], hh[1], hh[2], hh[3]};
                    *reinterpret_cast<u32x4*>(ys + C) = u32x4{ll[0], ll[1], ll[2], ll[3]};
                } else if (y) {
                    u32x4 ov;
                    ov.x = pack_h16x2(o[0], o[1]); ov.y = pack_h16x2(o[2], o[3]);
                    ov.z = pack_h16x2(o[4], o[5]); ov.w = pack_h16x2(o[6], o[7]);
                    *reinterpret_cast<u32x4*>(yb + (size_t)pp * C) = ov;
                }
            }
        }
        return;
    }
    for (int px = p0 + pl; px < p1; px += 4 * KP) {
        u32x4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pp = px + u * KP;
            a[u] = pp < p1 ? *reinterpret_cast<const u32x4*>(xb + (size_t)pp * C) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pp = px + u * KP;
            if (pp >= p1) break;
            const uint32_t aw[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = (j & 1) ? h16hi_to_f32(aw[j >> 1]) : h16lo_to_f32(aw[j >> 1]);
                float t = fmaf(xv, ca[j], cb[j]);
                if (SILU) t = silu_f(h16_to_f32(f32_to_h16(t)));
                o[j] = t;
            }
            u32x4 ov;
            ov.x = pack_h16x2(o[0], o[1]); ov.y = pack_h16x2(o[2], o[3]);
            ov.z = pack_h16x2(o[4], o[5]); ov.w = pack_h16x2(o[6], o[7]);
            *reinterpret_cast<u32x4*>(yb + (size_t)pp * C) = ov;
        }
    }
}

// GroupNorm for small feature maps (HW <= 256): ONE launch, one block per (group, batch); the group's data
// (HW x cpg values, 4-byte pairs, L2-resident) is read twice by the same block. Replaces the stats + apply pair
// whose two launches dominate at 16x16 / 8x8 latents.
template <int MAXP, bool SILU>
__global__ __launch_bounds__(256) void gn_small_kernel(const h16_t* __restrict__ x, const h16_t* __restrict__ gamma,
                                                       const h16_t* __restrict__ beta, h16_t* __restrict__ y, int HW, int C,
                                                       int G, float eps, const h16_t* __restrict__ x2, int C1,
                                                       h16_t* __restrict__ cat) {
    __shared__ float red[4];
    // MAXP channel pairs per thread: HW * cpg / 2 <= 256 * MAXP (host-checked)
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / G, hp = cpg / 2;          // channel pairs per group
    const int n = HW * hp;
    const h16_t* xb = x + (size_t)b * HW * C + g * cpg;
    h16_t* yb = y + (size_t)b * HW * C + g * cpg;
    // skip-concat form (see gn_stats_kernel): channel c < C1 from x (row stride C1), else from x2 (row stride C - C1)
    const h16_t* x1b = x + (size_t)b * HW * C1;
    const h16_t* x2b = x2 ? x2 + (size_t)b * HW * (C - C1) : nullptr;
    h16_t* catb = cat ? cat + (size_t)b * HW * C + g * cpg : nullptr;
    // the group's data is read ONCE into registers (all loads independent and in flight together)
    uint32_t v[MAXP];
    int off[MAXP];          // element offset of the pair inside the batch image; the pair index cp rides in bits 25..31 (cp < 128: C / G <= 256, and HW * C < 2^25: host-checked)
    // element i = threadIdx.x + 256 u lives at pixel i / hp, pair i % hp: one division for u = 0, then a carry-step per u (the 20
    // runtime divisions + 20 modulos per thread of the first version were most of the kernel's instructions, in front of its loads)
    int px = (int)threadIdx.x / hp, cp = (int)threadIdx.x - px * hp;
    const int dq = 256 / hp, dr = 256 - dq * hp;
#pragma unroll
    for (int u = 0; u < MAXP; ++u) {
        const int i = threadIdx.x + u * 256;
        const int o = px * C + 2 * cp;
        off[u] = o | (cp << 25);
        if (x2) {
            const int c = g * cpg + 2 * cp;
            const h16_t* src = c < C1 ? x1b + (size_t)px * C1 + c : x2b + (size_t)px * (C - C1) + (c - C1);
            v[u] = i < n ? *reinterpret_cast<const uint32_t*>(src) : 0u;
            if (catb && i < n) *reinterpret_cast<uint32_t*>(catb + o) = v[u];
        } else {
            v[u] = i < n ? *reinterpret_cast<const uint32_t*>(xb + o) : 0u;
        }
        px += dq; cp += dr;
        if (cp >= hp) { cp -= hp; ++px; }
    }
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int u = 0; u < MAXP; ++u) {
        const float lo = h16lo_to_f32(v[u]), hi = h16hi_to_f32(v[u]);   // padding entries are zeros
        s += lo + hi;
        q += lo * lo + hi * hi;
    }
    const float cnt = (float)HW * (float)cpg;
    const float mu = block_sum<4>(s, red) / cnt;
    const float var = fmaxf(block_sum<4>(q, red) / cnt - mu * mu, 0.f);
    const float rs = rsqrtf(var + eps);
#pragma unroll
    for (int u = 0; u < MAXP; ++u) {
        const int i = threadIdx.x + u * 256;
        if (i < n) {
            const int cp2 = (int)((unsigned)off[u] >> 25) * 2, o = off[u] & 0x1FFFFFF;     // channel offset inside the group, element offset
            const uint32_t gv = *reinterpret_cast<const uint32_t*>(gamma + g * cpg + cp2);
            const uint32_t bv = *reinterpret_cast<const uint32_t*>(beta + g * cpg + cp2);
            float lo = (h16lo_to_f32(v[u]) - mu) * rs * h16lo_to_f32(gv) + h16lo_to_f32(bv);
            float hi = (h16hi_to_f32(v[u]) - mu) * rs * h16hi_to_f32(gv) + h16hi_to_f32(bv);
            if (SILU) {
                lo = silu_f(h16_to_f32(f32_to_h16(lo)));
                hi = silu_f(h16_to_f32(f32_to_h16(hi)));
            }
            *reinterpret_cast<uint32_t*>(yb + o) = pack_h16x2(lo, hi);
        }
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm over the last dim (C <= 8192, multiple of 8), affine, fp32 statistics (two-pass in registers).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const h16_t* __restrict__ x, const h16_t* __restrict__ gamma,
                                                        const h16_t* __restrict__ beta, h16_t* __restrict__ y, int C,
                                                        float eps) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const int nv = C / 8;
    const u32x4* xv = reinterpret_cast<const u32x4*>(x + row * C);
    float h[4][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < nv) {
            const u32x4 a = xv[i];
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h[it][2 * j] = h16lo_to_f32(aw[j]);
                h[it][2 * j + 1] = h16hi_to_f32(aw[j]);
                s += h[it][2 * j] + h[it][2 * j + 1];
            }
        }
    }
    const float mu = block_sum<4>(s, red) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < nv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float dlt = h[it][j] - mu; q += dlt * dlt; }
        }
    }
    const float rs = rsqrtf(block_sum<4>(q, red) / (float)C + eps);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < nv) {
            const u32x4 gq = *reinterpret_cast<const u32x4*>(gamma + i * 8);
            const u32x4 bq = *reinterpret_cast<const u32x4*>(beta + i * 8);
            const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w}, bw[4] = {bq.x, bq.y, bq.z, bq.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = (h[it][2 * j] - mu) * rs * h16lo_to_f32(gw[j]) + h16lo_to_f32(bw[j]);
                const float hi = (h[it][2 * j + 1] - mu) * rs * h16hi_to_f32(gw[j]) + h16hi_to_f32(bw[j]);
                o[j] = pack_h16x2(lo, hi);
            }
            u32x4 ov;
            ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
            *reinterpret_cast<u32x4*>(y + row * C + i * 8) = ov;
        }
    }
}

// LayerNorm, one WAVE per row (C <= 2048): no LDS, no barriers; 4 rows per block. A 256-thread block per row leaves
// most lanes idle at C = 320 (40 vectors).
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_wave_kernel(const h16_t* __restrict__ x, const h16_t* __restrict__ gamma,
                                                             const h16_t* __restrict__ beta, h16_t* __restrict__ y,
                                                             int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = C / 8;
    const u32x4* xv = reinterpret_cast<const u32x4*>(x + (size_t)row * C);
    float h[VPL][8];
    float s = 0.f;
    // gamma / beta are requested together with the row (they do not depend on the statistics): one memory round trip, not two
    u32x4 gq[VPL], bq[VPL];
#pragma unroll
    for (int it = 0; it < VPL; ++it) {
        const int i = min(lane + it * 64, nv - 1);
        gq[it] = *reinterpret_cast<const u32x4*>(gamma + i * 8);
        bq[it] = *reinterpret_cast<const u32x4*>(beta + i * 8);
    }
#pragma unroll
    for (int it = 0; it < VPL; ++it) {
        const int i = lane + it * 64;
        if (i < nv) {
            const u32x4 a = xv[i];
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h[it][2 * j] = h16lo_to_f32(aw[j]);
                h[it][2 * j + 1] = h16hi_to_f32(aw[j]);
                s += h[it][2 * j] + h[it][2 * j + 1];
            }
        }
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < VPL; ++it) {
        if (lane + it * 64 < nv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float dlt = h[it][j] - mu; q += dlt * dlt; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int it = 0; it < VPL; ++it) {
        const int i = lane + it * 64;
        if (i < nv) {
            const uint32_t gw[4] = {gq[it].x, gq[it].y, gq[it].z, gq[it].w}, bw[4] = {bq[it].x, bq[it].y, bq[it].z, bq[it].w};
            u32x4 ov;
            ov.x = pack_h16x2((h[it][0] - mu) * rs * h16lo_to_f32(gw[0]) + h16lo_to_f32(bw[0]), (h[it][1] - mu) * rs * h16hi_to_f32(gw[0]) + h16hi_to_f32(bw[0]));
            ov.y = pack_h16x2((h[it][2] - mu) * rs * h16lo_to_f32(gw[1]) + h16lo_to_f32(bw[1]), (h[it][3] - mu) * rs * h16hi_to_f32(gw[1]) + h16hi_to_f32(bw[1]));
            ov.z = pack_h16x2((h[it][4] - mu) * rs * h16lo_to_f32(gw[2]) + h16lo_to_f32(bw[2]), (h[it][5] - mu) * rs * h16hi_to_f32(gw[2]) + h16hi_to_f32(bw[2]));
            ov.w = pack_h16x2((h[it][6] - mu) * rs * h16lo_to_f32(gw[3]) + h16lo_to_f32(bw[3]), (h[it][7] - mu) * rs * h16hi_to_f32(gw[3]) + h16hi_to_f32(bw[3]));
            *reinterpret_cast<u32x4*>(y + (size_t)row * C + i * 8) = ov;
        }
    }
}

// GEGLU: y[m, n] = bf16( x[m, n] * bf16(gelu(x[m, inner + n])) ), x [M, 2*inner]
__global__ __launch_bounds__(256) void geglu_kernel(const h16_t* __restrict__ x, h16_t* __restrict__ y, size_t total_vec,
                                                    int inner) {
    const int iv = inner / 8;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total_vec; idx += (size_t)gridDim.x * 256) {
        const size_t m = idx / iv;
        const int v = (int)(idx % iv);
        const u32x4 a = *reinterpret_cast<const u32x4*>(x + m * 2 * inner + v * 8);
        const u32x4 g = *reinterpret_cast<const u32x4*>(x + m * 2 * inner + inner + v * 8);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = h16lo_to_f32(aw[j]) * h16_to_f32(f32_to_h16(gelu_erf_f(h16lo_to_f32(gw[j]))));
            const float hi = h16hi_to_f32(aw[j]) * h16_to_f32(f32_to_h16(gelu_erf_f(h16hi_to_f32(gw[j]))));
            o[j] = pack_h16x2(lo, hi);
        }
        u32x4 ov;
        ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
        *reinterpret_cast<u32x4*>(y + m * inner + v * 8) = ov;
    }
}

// SwiGLU on a fused [gate | up] projection: y[m, n] = bf16( bf16(silu(x[m, n])) * x[m, inner + n] )
// (prefill form of modeling_llama3.py:197-199; the decode form is fused into gemv_swiglu)
__global__ __launch_bounds__(256) void swiglu_kernel(const h16_t* __restrict__ x, h16_t* __restrict__ y, size_t total_vec,
                                                     int inner) {
    const int iv = inner / 8;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total_vec; idx += (size_t)gridDim.x * 256) {
        const size_t m = idx / iv;
        const int v = (int)(idx % iv);
        const u32x4 g = *reinterpret_cast<const u32x4*>(x + m * 2 * inner + v * 8);
        const u32x4 u = *reinterpret_cast<const u32x4*>(x + m * 2 * inner + inner + v * 8);
        const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, uw[4] = {u.x, u.y, u.z, u.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = h16_to_f32(f32_to_h16(silu_f(h16lo_to_f32(gw[j])))) * h16lo_to_f32(uw[j]);
            const float hi = h16_to_f32(f32_to_h16(silu_f(h16hi_to_f32(gw[j])))) * h16hi_to_f32(uw[j]);
            o[j] = pack_h16x2(lo, hi);
        }
        u32x4 ov;
        ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
        *reinterpret_cast<u32x4*>(y + m * inner + v * 8) = ov;
    }
}

// channel concat of two NHWC tensors: y[r, :C1] = a[r], y[r, C1:] = b[r]
__global__ __launch_bounds__(256) void concat_kernel(const h16_t* __restrict__ a, const h16_t* __restrict__ b,
                                                     h16_t* __restrict__ y, size_t rows, int C1, int C2) {
    const int cv = (C1 + C2) / 8, c1v = C1 / 8;
    const size_t total = rows * cv;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t r = idx / cv;
        const int v = (int)(idx % cv);
        const u32x4 t = v < c1v ? *reinterpret_cast<const u32x4*>(a + r * C1 + v * 8)
                                : *reinterpret_cast<const u32x4*>(b + r * C2 + (v - c1v) * 8);
        *reinterpret_cast<u32x4*>(y + idx * 8) = t;
    }
}

// elementwise unary on bf16 (act: 1 silu, 2 gelu, 3 quick-gelu, 5 leaky-relu(param), 6 relu, 7 tanh), n multiple of 8
__global__ __launch_bounds__(256) void act_kernel(const h16_t* __restrict__ x, h16_t* __restrict__ y, size_t nvec, int act,
                                                  float param) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < nvec; idx += (size_t)gridDim.x * 256) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(x + idx * 8);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float lo = h16lo_to_f32(aw[j]), hi = h16hi_to_f32(aw[j]);
            if (act == 1) { lo = silu_f(lo); hi = silu_f(hi); }
            else if (act == 2) { lo = gelu_erf_f(lo); hi = gelu_erf_f(hi); }
            else if (act == 3) { lo = quick_gelu_f(lo); hi = quick_gelu_f(hi); }
            else if (act == 5) { lo = lo > 0.f ? lo : lo * param; hi = hi > 0.f ? hi : hi * param; }
            else if (act == 6) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
            else if (act == 7) { lo = tanhf(lo); hi = tanhf(hi); }
            o[j] = pack_h16x2(lo, hi);
        }
        u32x4 ov;
        ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
        *reinterpret_cast<u32x4*>(y + idx * 8) = ov;
    }
}

// y = bf16((a + b) * scale) elementwise
// "precise" operand split as its own pass: hi = round16(x), lo = round16(x - hi) of an fp32 tensor (8 values per thread). Lets the
// LDS-DMA / 256^2 kernels -- which never see A in registers -- take the hi / lo halves as two ordinary 16-bit operands: conv(hi) and
// conv(lo) accumulated in fp32 (res32 / c32d) is the same sum the A32 form makes with two MFMAs per K step.
__global__ __launch_bounds__(256) void split_hilo_kernel(const float* __restrict__ x, h16_t* __restrict__ hi, h16_t* __restrict__ lo, size_t nvec) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < nvec; idx += (size_t)gridDim.x * 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + idx * 8), b = *reinterpret_cast<const f32x4*>(x + idx * 8 + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        uint32_t h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = pack_h16x2(v[2 * j], v[2 * j + 1]);
            l[j] = pack_h16x2(v[2 * j] - h16lo_to_f32(h[j]), v[2 * j + 1] - h16hi_to_f32(h[j]));
        }
        *reinterpret_cast<u32x4*>(hi + idx * 8) = u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4*>(lo + idx * 8) = u32x4{l[0], l[1], l[2], l[3]};
    }
}

// Operand split by rows (ABI v4; DESIGN.md section 4, "split once, doubled K"): x [M, K] fp32 -> y [M, 2K] 16-bit with
// y[m, 0:K] = round16(v) and y[m, K:2K] = round16(v - round16(v)), v = x (LN = false) or LayerNorm(x) * gamma + beta with fp32
// two-pass statistics (LN = true). A GEMM / conv over y against the weight repeated along K ([W | W]) then accumulates
// hi . W + lo . W in its fp32 accumulator -- the hi / lo product of spider_gemm_a32 / spider_gemm_ln_a32 at the speed of the 16-bit
// tile kernels. One wave per row, 8 elements per lane and trip (NIT trips cover K <= 512 NIT), the row held in registers.
template <int NIT, bool LN>
__global__ __launch_bounds__(256) void row_split_kernel(const float* __restrict__ x, const h16_t* __restrict__ gamma,
                                                        const h16_t* __restrict__ beta, h16_t* __restrict__ y, long M, int K, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long m = (long)blockIdx.x * 4 + wave; m < M; m += (long)gridDim.x * 4) {
        const float* xr = x + (size_t)m * K;
        float v[NIT][8];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 8;
            if (c < K) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(xr + c), b = *reinterpret_cast<const f32x4*>(xr + c + 4);
                v[it][0] = a[0]; v[it][1] = a[1]; v[it][2] = a[2]; v[it][3] = a[3];
                v[it][4] = b[0]; v[it][5] = b[1]; v[it][6] = b[2]; v[it][7] = b[3];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[it][j] = 0.f;
            }
        }
        if (LN) {
            float s = 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[it][j];
            const float mu = wave_sum(s) / (float)K;
            float q = 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const bool ok = (it * 64 + lane) * 8 < K;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = v[it][j] - mu;
                    q += ok ? d * d : 0.f;
                }
            }
            const float rstd = rsqrtf(wave_sum(q) / (float)K + eps);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = (it * 64 + lane) * 8;
                if (c < K) {
                    const u32x4 gq = *reinterpret_cast<const u32x4*>(gamma + c);
                    const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w};
                    uint32_t bw[4] = {0u, 0u, 0u, 0u};
                    if (beta) {
                        const u32x4 bq = *reinterpret_cast<const u32x4*>(beta + c);
                        bw[0] = bq.x; bw[1] = bq.y; bw[2] = bq.z; bw[3] = bq.w;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float ga = (j & 1) ? h16hi_to_f32(gw[j >> 1]) : h16lo_to_f32(gw[j >> 1]);
                        const float be = (j & 1) ? h16hi_to_f32(bw[j >> 1]) : h16lo_to_f32(bw[j >> 1]);
                        v[it][j] = fmaf((v[it][j] - mu) * rstd, ga, be);
                    }
                }
            }
        }
        h16_t* yr = y + (size_t)m * 2 * K;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 8;
            if (c < K) {
                uint32_t h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h[j] = pack_h16x2(v[it][2 * j], v[it][2 * j + 1]);
                    l[j] = pack_h16x2(v[it][2 * j] - h16lo_to_f32(h[j]), v[it][2 * j + 1] - h16hi_to_f32(h[j]));
                }
                *reinterpret_cast<u32x4*>(yr + c) = u32x4{h[0], h[1], h[2], h[3]};
                *reinterpret_cast<u32x4*>(yr + K + c) = u32x4{l[0], l[1], l[2], l[3]};
            }
        }
    }
}

__global__ __launch_bounds__(256) void add_scaled_kernel(const h16_t* __restrict__ a, const h16_t* __restrict__ b,
                                                         h16_t* __restrict__ y, size_t nvec, float scale) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < nvec; idx += (size_t)gridDim.x * 256) {
        const u32x4 p = *reinterpret_cast<const u32x4*>(a + idx * 8);
        const u32x4 q = *reinterpret_cast<const u32x4*>(b + idx * 8);
        const uint32_t pw[4] = {p.x, p.y, p.z, p.w}, qw[4] = {q.x, q.y, q.z, q.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pack_h16x2((h16lo_to_f32(pw[j]) + h16lo_to_f32(qw[j])) * scale, (h16hi_to_f32(pw[j]) + h16hi_to_f32(qw[j])) * scale);
        u32x4 ov;
        ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
        *reinterpret_cast<u32x4*>(y + idx * 8) = ov;
    }
}

// y = bf16(alpha * a + beta * b) elementwise
__global__ __launch_bounds__(256) void axpby_kernel(const h16_t* __restrict__ a, const h16_t* __restrict__ b,
                                                    h16_t* __restrict__ y, size_t nvec, float alpha, float beta) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < nvec; idx += (size_t)gridDim.x * 256) {
        const u32x4 p = *reinterpret_cast<const u32x4*>(a + idx * 8);
        const u32x4 q = *reinterpret_cast<const u32x4*>(b + idx * 8);
        const uint32_t pw[4] = {p.x, p.y, p.z, p.w}, qw[4] = {q.x, q.y, q.z, q.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pack_h16x2(alpha * h16lo_to_f32(pw[j]) + beta * h16lo_to_f32(qw[j]),
                               alpha * h16hi_to_f32(pw[j]) + beta * h16hi_to_f32(qw[j]));
        u32x4 ov;
        ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
        *reinterpret_cast<u32x4*>(y + idx * 8) = ov;
    }
}

// y[b, c] = mean over t of x[b, t, c]   (TextFcLayerMoE router input, layers.py:254)
__global__ __launch_bounds__(256) void mean_tokens_kernel(const h16_t* __restrict__ x, h16_t* __restrict__ y, int B, int T, int C) {
    const size_t total = (size_t)B * C;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const size_t b = idx / C;
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += h16_to_f32(x[(b * T + t) * C + c]);
        y[idx] = f32_to_h16(s / (float)T);
    }
}

// TextFcLayerMoE mixing (layers.py:255-267): r = sigmoid(logits[b, :]) / sum, out[b, t, :] = sum_e r[e] * x_e[b, t, :]
struct MoeCombine {
    const h16_t* x[8];
    int E;
};
__global__ __launch_bounds__(256) void moe_combine_kernel(MoeCombine mc, const h16_t* __restrict__ logits, int ld, h16_t* __restrict__ y,
                                                          int B, size_t per_batch) {
    const size_t total = (size_t)B * per_batch;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t b = idx / per_batch;
        float r[8], rs = 0.f;
        for (int e = 0; e < mc.E; ++e) {
            r[e] = 1.f / (1.f + __expf(-h16_to_f32(logits[b * ld + e])));
            rs += r[e];
        }
        float acc = 0.f;
        for (int e = 0; e < mc.E; ++e) acc += h16_to_f32(f32_to_h16(h16_to_f32(mc.x[e][idx]) * h16_to_f32(f32_to_h16(r[e] / rs))));
        y[idx] = f32_to_h16(acc);
    }
}

// ConvTranspose1d overlap-add (see spider_col2im1d_f32_bf16): one thread per (b, t, 4 output channels)
__global__ __launch_bounds__(256) void col2im1d_kernel(const float* __restrict__ cols, const h16_t* __restrict__ bias,
                                                       h16_t* __restrict__ y, int B, int L_in, int L_out, int k, int stride,
                                                       int pad, int Cout) {
    const int c4 = Cout / 4;
    const size_t total = (size_t)B * L_out * c4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int cq = (int)(idx % c4);
        const int t = (int)((idx / c4) % L_out);
        const int b = (int)(idx / ((size_t)c4 * L_out));
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        // taps j == (t + pad) mod stride, input position i = (t + pad - j) / stride
        for (int j = (t + pad) % stride; j < k; j += stride) {
            const int i = (t + pad - j) / stride;
            if (t + pad - j < 0 || i >= L_in) continue;
            const f32x4 q = *reinterpret_cast<const f32x4*>(cols + (((size_t)b * L_in + i) * k + j) * Cout + cq * 4);
            v[0] += q[0]; v[1] += q[1]; v[2] += q[2]; v[3] += q[3];
        }
        if (bias) {
            const u32x2 bq = *reinterpret_cast<const u32x2*>(bias + cq * 4);
            v[0] += h16lo_to_f32(bq.x); v[1] += h16hi_to_f32(bq.x);
            v[2] += h16lo_to_f32(bq.y); v[3] += h16hi_to_f32(bq.y);
        }
        u32x2 o;
        o.x = pack_h16x2(v[0], v[1]);
        o.y = pack_h16x2(v[2], v[3]);
        *reinterpret_cast<u32x2*>(y + (((size_t)b * L_out + t) * Cout) + cq * 4) = o;
    }
}

// row-wise L2 normalisation, one block per row
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const h16_t* __restrict__ x, h16_t* __restrict__ y, int n, float eps) {
    __shared__ float red[4];
    const h16_t* xr = x + (size_t)blockIdx.x * n;
    h16_t* yr = y + (size_t)blockIdx.x * n;
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const float v = h16_to_f32(xr[i]); ss += v * v; }
    const float inv = 1.f / fmaxf(sqrtf(block_sum<4>(ss, red)), eps);
    for (int i = threadIdx.x; i < n; i += 256) yr[i] = f32_to_h16(h16_to_f32(xr[i]) * inv);
}

// y = bf16(a + b) elementwise
__global__ __launch_bounds__(256) void add_kernel(const h16_t* __restrict__ a, const h16_t* __restrict__ b,
                                                  h16_t* __restrict__ y, size_t nvec) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < nvec; idx += (size_t)gridDim.x * 256) {
        const u32x4 p = *reinterpret_cast<const u32x4*>(a + idx * 8);
        const u32x4 q = *reinterpret_cast<const u32x4*>(b + idx * 8);
        const uint32_t pw[4] = {p.x, p.y, p.z, p.w}, qw[4] = {q.x, q.y, q.z, q.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pack_h16x2(h16lo_to_f32(pw[j]) + h16lo_to_f32(qw[j]), h16hi_to_f32(pw[j]) + h16hi_to_f32(qw[j]));
        u32x4 ov;
        ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
        *reinterpret_cast<u32x4*>(y + idx * 8) = ov;
    }
}

// ------------------------------------------------------------------------------------------
// Small convolutions that do not fit the MFMA tile (Cin = 4 conv_in, Cout = 3/4 conv_out).
// ------------------------------------------------------------------------------------------
// Cin <= 8: one thread per (pixel, 8 output channels). w [Cout, ks, ks, Cin] staged in LDS as fp32.
// x32 (optional, precise mode): the input as fp32 NHWC instead of x; y32 (optional): fp32 copy of the result (the stream's master)
__global__ __launch_bounds__(256) void conv_small_cin_kernel(const h16_t* __restrict__ x, const h16_t* __restrict__ w,
                                                             const h16_t* __restrict__ bias, h16_t* __restrict__ y,
                                                             int B, int H, int W, int Cin, int Cout, int ks,
                                                             const float* __restrict__ x32 = nullptr, float* __restrict__ y32 = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ws = reinterpret_cast<float*>(smem);
    const int kk = ks * ks * Cin;
    for (int i = threadIdx.x; i < Cout * kk; i += 256) ws[(i % kk) * Cout + i / kk] = h16_to_f32(w[i]);
    __syncthreads();
    const int cov = Cout / 8;
    const size_t total = (size_t)B * H * W * cov;
    const int pad = ks / 2;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int v = (int)(idx % cov);
        const size_t pix = idx / cov;
        const int ox = (int)(pix % W), oy = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bias ? h16_to_f32(bias[v * 8 + j]) : 0.f;
        for (int ky = 0; ky < ks; ++ky) {
            const int iy = oy + ky - pad;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < ks; ++kx) {
                const int ix = ox + kx - pad;
                if (ix < 0 || ix >= W) continue;
                const size_t so = (((size_t)b * H + iy) * W + ix) * Cin;
                for (int c = 0; c < Cin; ++c) {
                    const float xv = x32 ? x32[so + c] : h16_to_f32(x[so + c]);
                    const float* wk = ws + ((ky * ks + kx) * Cin + c) * Cout + v * 8;
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wk), w1 = *reinterpret_cast<const f32x4*>(wk + 4);
                    acc[0] += xv * w0[0]; acc[1] += xv * w0[1]; acc[2] += xv * w0[2]; acc[3] += xv * w0[3];
                    acc[4] += xv * w1[0]; acc[5] += xv * w1[1]; acc[6] += xv * w1[2]; acc[7] += xv * w1[3];
                }
            }
        }
        u32x4 ov;
        ov.x = pack_h16x2(acc[0], acc[1]); ov.y = pack_h16x2(acc[2], acc[3]);
        ov.z = pack_h16x2(acc[4], acc[5]); ov.w = pack_h16x2(acc[6], acc[7]);
        *reinterpret_cast<u32x4*>(y + pix * Cout + v * 8) = ov;
        if (y32) {
            *reinterpret_cast<f32x4*>(y32 + pix * Cout + v * 8) = f32x4{acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<f32x4*>(y32 + pix * Cout + v * 8 + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
        }
    }
}

// Cout <= 4: one wave per output pixel, lanes split K = ks*ks*Cin (Cin % 8 == 0), shuffle reduce.
// Output fp32 [B, H, W, Cout] (conv_out feeds the scheduler / image, keep full precision) or bf16.
__global__ __launch_bounds__(256) void conv_small_cout_kernel(const h16_t* __restrict__ x, const h16_t* __restrict__ w,
                                                              const h16_t* __restrict__ bias, float* __restrict__ y32,
                                                              h16_t* __restrict__ y16, int B, int H, int W, int Cin,
                                                              int Cout, int ks, const float* __restrict__ x32 = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16_t* ws = reinterpret_cast<h16_t*>(smem);  // [Cout][ks*ks*Cin]
    const int kk = ks * ks * Cin;
    for (int i = threadIdx.x; i < Cout * kk / 8; i += 256)
        reinterpret_cast<u32x4*>(ws)[i] = reinterpret_cast<const u32x4*>(w)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t npix = (size_t)B * H * W;
    const int pad = ks / 2;
    const int cvn = Cin / 8;
    for (size_t pix = (size_t)blockIdx.x * 4 + wave; pix < npix; pix += (size_t)gridDim.x * 4) {
        const int ox = (int)(pix % W), oy = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = lane; i < ks * ks * cvn; i += 64) {
            const int tap = i / cvn, cvi = i % cvn;
            const int iy = oy + tap / ks - pad, ix = ox + tap % ks - pad;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const size_t xo = (((size_t)b * H + iy) * W + ix) * Cin + cvi * 8;
            float xa[8];
            if (x32) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(x32 + xo), a1 = *reinterpret_cast<const f32x4*>(x32 + xo + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { xa[j] = a0[j]; xa[4 + j] = a1[j]; }
            } else {
                const u32x4 a = *reinterpret_cast<const u32x4*>(x + xo);
                const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) { xa[2 * j] = h16lo_to_f32(aw[j]); xa[2 * j + 1] = h16hi_to_f32(aw[j]); }
            }
#pragma unroll
            for (int co = 0; co < 8; ++co) {
                if (co < Cout) {
                    const u32x4 wq = *reinterpret_cast<const u32x4*>(ws + (size_t)co * kk + tap * Cin + cvi * 8);
                    const uint32_t ww[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[co] += xa[2 * j] * h16lo_to_f32(ww[j]) + xa[2 * j + 1] * h16hi_to_f32(ww[j]);
                }
            }
        }
#pragma unroll
        for (int co = 0; co < 8; ++co) {
            if (co < Cout) {
                const float t = wave_sum(acc[co]) + (bias ? h16_to_f32(bias[co]) : 0.f);
                if (lane == 0) {
                    if (y32) y32[pix * Cout + co] = t;
                    else y16[pix * Cout + co] = f32_to_h16(t);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Latent plumbing around the UNet call (custom_sd.py:631-647). Latents stay fp32 NCHW on the scheduler
// side (the reference keeps them in the pipeline dtype; fp32 here so 40 scheduler updates do not
// accumulate bf16 rounding) and are converted to bf16 NHWC only as UNet input.
// ------------------------------------------------------------------------------------------
// out[rep, b, y, x, c] = bf16(lat[b, c, y, x] * scale) for rep < reps   (torch.cat([latents]*2) + scale_model_input)
__global__ __launch_bounds__(256) void latent_in_kernel(const float* __restrict__ lat, h16_t* __restrict__ out, int B, int C,
                                                        int HW, int reps, float scale) {
    const size_t total = (size_t)B * C * HW;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const size_t px = (idx / C) % HW;
        const size_t b = idx / ((size_t)C * HW);
        const h16_t v = f32_to_h16(lat[(b * C + c) * HW + px] * scale);
        for (int r = 0; r < reps; ++r) out[(size_t)r * total + idx] = v;
    }
}

// eps[b, c, px] = e_u + g * (e_c - e_u) from UNet output e [2, B, HW, C] (fp32 NHWC); out fp32 NCHW.
__global__ __launch_bounds__(256) void latent_in32_kernel(const float* __restrict__ lat, float* __restrict__ out, int B, int C, int HW,
                                                          int reps, float scale) {
    const size_t total = (size_t)B * C * HW;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const size_t px = (idx / C) % HW;
        const size_t b = idx / ((size_t)C * HW);
        const float v = lat[(b * C + c) * HW + px] * scale;
        for (int r = 0; r < reps; ++r) out[(size_t)r * total + idx] = v;
    }
}

__global__ __launch_bounds__(256) void cfg_combine_kernel(const float* __restrict__ e, float* __restrict__ out, int B, int C,
                                                          int HW, float g) {
    const size_t total = (size_t)B * C * HW;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const size_t px = (idx / C) % HW;
        const size_t b = idx / ((size_t)C * HW);
        const float eu = e[idx], ec = e[total + idx];
        out[(b * C + c) * HW + px] = eu + g * (ec - eu);
    }
}

struct LinComb {
    const float* in[6];
    float coef[6];
    int n;
};
// out = sum_j coef[j] * in[j]   (scheduler.step as a linear update of the sample and stored eps history)
__global__ __launch_bounds__(256) void lincomb_kernel(LinComb lc, float* __restrict__ out, size_t total) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        float a = 0.f;
        for (int j = 0; j < lc.n; ++j) a += lc.coef[j] * lc.in[j][idx];
        out[idx] = a;
    }
}

// fp32 NHWC [B,HW,C] -> fp32 NCHW [B,C,HW] with affine (VAE image post-process: x/2 + 0.5, clamp 0..1)
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C,
                                                           int HW, float mul, float add, int clamp01) {
    const size_t total = (size_t)B * C * HW;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const size_t px = (idx / C) % HW;
        const size_t b = idx / ((size_t)C * HW);
        float v = x[idx] * mul + add;
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        y[(b * C + c) * HW + px] = v;
    }
}

// Row softmax: y[r, :] = bf16(softmax(scale * x[r, :])), x fp32 [rows, n] (scores of the VAE's single-head
// d=512 attention, diffusers AttnProcessor on AutoencoderKL.mid_block.attentions.0). One block per row.
// Columns n_valid..n-1 (row padding up to the 8-element granularity of the next GEMM) are written as zeros.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, h16_t* __restrict__ y, int ld, int n,
                                                           float scale) {
    __shared__ float red[4];
    const float* xr = x + (size_t)blockIdx.x * ld;
    h16_t* yr = y + (size_t)blockIdx.x * ld;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, xr[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += __expf(xr[i] * scale - m);
    const float inv = 1.f / block_sum<4>(s, red);
    for (int i = threadIdx.x; i < n; i += 256) yr[i] = f32_to_h16(__expf(xr[i] * scale - m) * inv);
    for (int i = n + threadIdx.x; i < ld; i += 256) yr[i] = 0;
}

inline int grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

#ifndef SPIDER_F16   // dtype-free: built once (bf16 translation unit)
// StoryDiffusion keep vector -> bit words (cal_attn_mask_xl reduced to its information content, gradio_utils.py:241-287):
// bit j of word w = (u[64 w + j] < thr) && (64 w + j < n_valid). One ballot per wave: no host round trip per UNet step.
__global__ __launch_bounds__(256) void pack_keep_kernel(const float* __restrict__ u, unsigned long long* __restrict__ words, int n,
                                                        int n_valid, float thr) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool keep = idx < n && idx < n_valid && u[idx] < thr;
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && idx < ((n + 63) / 64) * 64) words[idx >> 6] = m;
}
#endif

static int gn_nchunk(int HW) {
    int n = HW / 32;
    return n < 1 ? 1 : (n > 128 ? 128 : n);
}

extern "C" {

#ifndef SPIDER_F16
int spider_groupnorm_nchunk(int HW) { return gn_nchunk(HW); }
#endif

// Pass 1 of GroupNorm alone: partial [B, nchunk, G, 2] fp32 = (sum, sum of squares) per chunk of ceil(HW / nchunk) pixels and group.
int SPIDER_FN(spider_groupnorm_stats_nhwc)(const void* x, void* partial, int B, int HW, int C, int G, int nchunk, void* stream) {
    SPIDER_CHECK(B > 0 && HW > 0 && C > 0 && G > 0 && nchunk > 0 && C % 8 == 0 && C % G == 0 && C <= 8192, "groupnorm_stats: bad shape");
    const int cv = C / 8;
    int KP = 256 / cv;
    if (KP < 1) KP = 1;
    const int threads = cv * KP;
    SPIDER_CHECK(threads <= 1024, "groupnorm_stats: C too large");
    gn_stats_kernel<false><<<dim3(nchunk, B), threads, (size_t)2 * KP * C * sizeof(float), (hipStream_t)stream>>>(
        (const h16_t*)x, (float*)partial, HW, C, G, nchunk, KP, nullptr, C, nullptr);
    SPIDER_LAUNCH_OK();
    return 0;
}

// Pass 2 of GroupNorm alone, on partial statistics someone else produced (the producing conv's epilogue, spider_conv_nhwc_gn, or
// spider_groupnorm_stats_nhwc): y = GroupNorm(x) (+ SiLU), partial [B, nchunk, G, 2] reduced in a fixed order by every block.
int SPIDER_FN(spider_groupnorm_apply_nhwc)(const void* x, const void* partial, int nchunk, const void* gamma, const void* beta, void* y,
                                     int B, int HW, int C, int G, float eps, int silu, void* stream) {
    SPIDER_CHECK(B > 0 && HW > 0 && C > 0 && G > 0 && G <= 64 && 256 % G == 0 && nchunk > 0, "groupnorm_apply: G must divide 256 and be <= 64");
    SPIDER_CHECK(C % 8 == 0 && C % G == 0 && C <= 8192, "groupnorm_apply: C must be a multiple of 8 and of G");
    const int cv = C / 8;
    int KP = 256 / cv;
    if (KP < 1) KP = 1;
    const int threads = cv * KP;
    SPIDER_CHECK(threads <= 1024 && (threads >= 128 || threads >= 2 * G), "groupnorm_apply: unsupported channel count for the block layout");
    int ppb = 4 * KP;
    while ((long)B * ((HW + ppb - 1) / ppb) > 1024 && ppb < 64 * KP) ppb += 4 * KP;
    dim3 g2((HW + ppb - 1) / ppb, B);
    if (silu) gn_apply_kernel<true><<<g2, threads, 0, (hipStream_t)stream>>>((const h16_t*)x, (const float*)partial, (const h16_t*)gamma,
                                                                       (const h16_t*)beta, (h16_t*)y, HW, C, G, nchunk, eps, ppb, KP);
    else gn_apply_kernel<false><<<g2, threads, 0, (hipStream_t)stream>>>((const h16_t*)x, (const float*)partial, (const h16_t*)gamma,
                                                                    (const h16_t*)beta, (h16_t*)y, HW, C, G, nchunk, eps, ppb, KP);
    SPIDER_LAUNCH_OK();
    return 0;
}

// "precise" GroupNorm (ABI v4): x32 [B, HW, C] is the FP32 master of the residual stream (not its 16-bit
// shadow); statistics from `partial` ([B, nchunk, G, 2], e.g. the producing conv's) or, partial == NULL, from a pass over x32 into ws
// (>= B * spider_groupnorm_nchunk(HW) * G * 2 floats); the normalised (+ SiLU) value is rounded once into y (16-bit, optional) and /
// or stored unrounded into y32 (fp32, optional: the A operand of spider_gemm_a32 / spider_conv_nhwc_a32).
int SPIDER_FN(spider_groupnorm_f32in_nhwc)(const float* x32, const float* partial, int nchunk, const void* gamma, const void* beta, void* y,
                                     float* y32, float* ws, int B, int HW, int C, int G, float eps, int silu, void* stream) {
    SPIDER_CHECK(B > 0 && HW > 0 && C > 0 && G > 0 && G <= 64 && 256 % G == 0, "groupnorm_f32in: G must divide 256 and be <= 64");
    SPIDER_CHECK(C % 8 == 0 && C % G == 0 && C <= 8192 && x32 && (y || y32), "groupnorm_f32in: C % 8, C % G; x32 and an output are required");
    const int cv = C / 8;
    int KP = 256 / cv;
    if (KP < 1) KP = 1;
    const int threads = cv * KP;
    SPIDER_CHECK(threads <= 1024 && (threads >= 128 || threads >= 2 * G), "groupnorm_f32in: unsupported channel count for the block layout");
    if (!partial) {
        SPIDER_CHECK(ws, "groupnorm_f32in: ws is required without partial statistics");
        nchunk = gn_nchunk(HW);
        gn_stats_kernel<true><<<dim3(nchunk, B), threads, (size_t)2 * KP * C * sizeof(float), (hipStream_t)stream>>>(
            (const h16_t*)x32, ws, HW, C, G, nchunk, KP, nullptr, C, nullptr);
        SPIDER_LAUNCH_OK();
        partial = ws;
    }
    SPIDER_CHECK(nchunk > 0, "groupnorm_f32in: nchunk");
    int ppb = 4 * KP;
    while ((long)B * ((HW + ppb - 1) / ppb) > 1024 && ppb < 64 * KP) ppb += 4 * KP;
    dim3 g2((HW + ppb - 1) / ppb, B);
    if (silu) gn_apply_kernel<true, true><<<g2, threads, 0, (hipStream_t)stream>>>((const h16_t*)x32, partial, (const h16_t*)gamma,
                                                                             (const h16_t*)beta, (h16_t*)y, HW, C, G, nchunk, eps, ppb, KP, y32);
    else gn_apply_kernel<false, true><<<g2, threads, 0, (hipStream_t)stream>>>((const h16_t*)x32, partial, (const h16_t*)gamma,
                                                                          (const h16_t*)beta, (h16_t*)y, HW, C, G, nchunk, eps, ppb, KP, y32);
    SPIDER_LAUNCH_OK();
    return 0;
}

// GroupNorm of the channel concatenation [x1 | x2] (UpBlock2D / UpBlock3D: cat([hidden, skip]) -> ResnetBlock.norm1) without a
// concat launch: x1 [B, HW, C1], x2 [B, HW, C2] are read in place, y [B, HW, C1 + C2] is the normalised result and `cat`
// (same shape) receives the concatenated input for the resnet's 1x1 shortcut. x2 == nullptr: plain GroupNorm of x1 (C2 = 0).
static int groupnorm_impl(const void* x, const void* x2, const void* gamma, const void* beta, void* y, void* cat, void* ws, int B,
                          int HW, int C1, int C2, int G, float eps, int silu, void* stream) {
    const int C = C1 + C2;
    SPIDER_CHECK(B > 0 && HW > 0 && C > 0 && G > 0 && G <= 64 && 256 % G == 0, "groupnorm: G must divide 256 and be <= 64");
    SPIDER_CHECK(C % 8 == 0 && C % G == 0 && C <= 8192, "groupnorm: C must be a multiple of 8 and of G");
    // small feature map: single launch, data held in registers (its packed offsets need C / G <= 256 and HW * C < 2^25)
    if ((long)HW * (C / G) <= 10240 && (C / G) % 2 == 0 && C / G <= 256 && (long)HW * C < (1L << 25)) {
        // (measured: a 40-pair variant for 20K-element groups is no faster than the stats + apply pair -- 64 blocks cannot
        // pull enough bandwidth)
        if (silu) gn_small_kernel<20, true><<<dim3(G, B), 256, 0, (hipStream_t)stream>>>((const h16_t*)x, (const h16_t*)gamma,
                                                                                        (const h16_t*)beta, (h16_t*)y, HW, C, G, eps,
                                                                                        (const h16_t*)x2, C1, (h16_t*)cat);
        else gn_small_kernel<20, false><<<dim3(G, B), 256, 0, (hipStream_t)stream>>>((const h16_t*)x, (const h16_t*)gamma,
                                                                                     (const h16_t*)beta, (h16_t*)y, HW, C, G, eps,
                                                                                     (const h16_t*)x2, C1, (h16_t*)cat);
        SPIDER_LAUNCH_OK();
        return 0;
    }
    const int nchunk = gn_nchunk(HW);
    const int cv = C / 8;
    int KP = 256 / cv;
    if (KP < 1) KP = 1;
    const int threads = cv * KP;
    SPIDER_CHECK(threads <= 1024, "groupnorm: C too large");
    dim3 g1(nchunk, B);
    gn_stats_kernel<false><<<g1, threads, (size_t)2 * KP * C * sizeof(float), (hipStream_t)stream>>>((const h16_t*)x, (float*)ws, HW, C,
                                                                                       G, nchunk, KP, (const h16_t*)x2, C1, (h16_t*)cat);
    SPIDER_LAUNCH_OK();
    if (x2) x = cat;              // pass 2 reads the concatenated copy pass 1 has just written
    int ppb = 4 * KP;             // one round of 4 independent 16-byte loads per thread
    while ((long)B * ((HW + ppb - 1) / ppb) > 1024 && ppb < 64 * KP) ppb += 4 * KP;   // keep the grid at <= ~4 blocks per CU
    SPIDER_CHECK(threads >= 128 || threads >= 2 * G, "groupnorm: too few channels for the block layout");
    dim3 g2((HW + ppb - 1) / ppb, B);
    if (silu) gn_apply_kernel<true><<<g2, threads, 0, (hipStream_t)stream>>>((const h16_t*)x, (const float*)ws, (const h16_t*)gamma,
                                                                       (const h16_t*)beta, (h16_t*)y, HW, C, G, nchunk, eps, ppb, KP);
    else gn_apply_kernel<false><<<g2, threads, 0, (hipStream_t)stream>>>((const h16_t*)x, (const float*)ws, (const h16_t*)gamma,
                                                                    (const h16_t*)beta, (h16_t*)y, HW, C, G, nchunk, eps, ppb, KP);
    SPIDER_LAUNCH_OK();
    return 0;
}

// x, y [B, HW, C] bf16 (NHWC); ws >= B * nchunk * G * 2 floats, nchunk = spider_groupnorm_nchunk(HW)
int SPIDER_FN(spider_groupnorm_nhwc)(const void* x, const void* gamma, const void* beta, void* y, void* ws, int B, int HW,
                               int C, int G, float eps, int silu, void* stream) {
    return groupnorm_impl(x, nullptr, gamma, beta, y, nullptr, ws, B, HW, C, 0, G, eps, silu, stream);
}

int SPIDER_FN(spider_groupnorm_cat_nhwc)(const void* x1, const void* x2, const void* gamma, const void* beta, void* y, void* cat,
                                   void* ws, int B, int HW, int C1, int C2, int G, float eps, int silu, void* stream) {
    SPIDER_CHECK(x2 && cat && C1 > 0 && C2 > 0 && C1 % 8 == 0 && C2 % 8 == 0, "groupnorm_cat: two sources with channels % 8 == 0");
    return groupnorm_impl(x1, x2, gamma, beta, y, cat, ws, B, HW, C1, C2, G, eps, silu, stream);
}

int SPIDER_FN(spider_layernorm)(const void* x, const void* gamma, const void* beta, void* y, int rows, int C, float eps,
                          void* stream) {
    SPIDER_CHECK(rows > 0 && C > 0 && C % 8 == 0 && C <= 8192, "layernorm: C must be a multiple of 8 and <= 8192");
    if (C <= 2048) {
        const int vpl = (C / 8 + 63) / 64, grid = (rows + 3) / 4;
#define LNW(V_) layernorm_wave_kernel<V_><<<grid, 256, 0, (hipStream_t)stream>>>((const h16_t*)x, (const h16_t*)gamma, \
                                                                                (const h16_t*)beta, (h16_t*)y, rows, C, eps)
        if (vpl == 1) LNW(1); else if (vpl == 2) LNW(2); else if (vpl == 3) LNW(3); else LNW(4);
#undef LNW
        SPIDER_LAUNCH_OK();
        return 0;
    }
    layernorm_kernel<<<rows, 256, 0, (hipStream_t)stream>>>((const h16_t*)x, (const h16_t*)gamma, (const h16_t*)beta,
                                                           (h16_t*)y, C, eps);
    SPIDER_LAUNCH_OK();
    return 0;
}

int SPIDER_FN(spider_geglu)(const void* x, void* y, int M, int inner, void* stream) {
    SPIDER_CHECK(M > 0 && inner > 0 && inner % 8 == 0, "geglu: inner must be a multiple of 8");
    const size_t nvec = (size_t)M * (inner / 8);
    geglu_kernel<<<grid_for(nvec), 256, 0, (hipStream_t)stream>>>((const h16_t*)x, (h16_t*)y, nvec, inner);
    SPIDER_LAUNCH_OK();
    return 0;
}

int SPIDER_FN(spider_swiglu)(const void* x, void* y, int M, int inner, void* stream) {
    SPIDER_CHECK(M > 0 && inner > 0 && inner % 8 == 0, "swiglu: inner must be a multiple of 8");
    const size_t nvec = (size_t)M * (inner / 8);
    swiglu_kernel<<<grid_for(nvec), 256, 0, (hipStream_t)stream>>>((const h16_t*)x, (h16_t*)y, nvec, inner);
    SPIDER_LAUNCH_OK();
    return 0;
}

int SPIDER_FN(spider_concat_channels)(const void* a, const void* b, void* y, long rows, int C1, int C2, void* stream) {
    SPIDER_CHECK(rows > 0 && C1 > 0 && C2 > 0 && C1 % 8 == 0 && C2 % 8 == 0, "concat: channels must be multiples of 8");
    const size_t nvec = (size_t)rows * ((C1 + C2) / 8);
    concat_kernel<<<grid_for(nvec), 256, 0, (hipStream_t)stream>>>((const h16_t*)a, (const h16_t*)b, (h16_t*)y,
                                                                  (size_t)rows, C1, C2);
    SPIDER_LAUNCH_OK();
    return 0;
}

int SPIDER_FN(spider_act_ex)(const void* x, void* y, long n, int act, float param, void* stream) {
    SPIDER_CHECK(n > 0 && n % 8 == 0, "act: n must be a multiple of 8");
    SPIDER_CHECK(act == 1 || act == 2 || act == 3 || act == 5 || act == 6 || act == 7, "act: unknown activation");
    act_kernel<<<grid_for((size_t)n / 8), 256, 0, (hipStream_t)stream>>>((const h16_t*)x, (h16_t*)y, (size_t)n / 8, act, param);
    SPIDER_LAUNCH_OK();
    return 0;
}

int SPIDER_FN(spider_act)(const void* x, void* y, long n, int act, void* stream) {
    SPIDER_CHECK(act >= 1 && act <= 3, "act: act in 1..3 (spider_act_ex_bf16 has the parameterised forms)");
    return SPIDER_FN(spider_act_ex)(x, y, n, act, 0.f, stream);
}

// hi[i] = round16(x[i]), lo[i] = round16(x[i] - hi[i]) (ABI v4, "precise" forms: the operand split as a pass of its own, for the
// convs large enough to run on the LDS-DMA / 256^2 kernels twice -- conv(hi) then conv(lo) with res32 = the first result)
int SPIDER_FN(spider_split_hilo_f32)(const float* x, void* hi, void* lo, long n, void* stream) {
    SPIDER_CHECK(n > 0 && n % 8 == 0 && x && hi && lo, "split_hilo: n must be a positive multiple of 8");
    split_hilo_kernel<<<grid_for((size_t)n / 8), 256, 0, (hipStream_t)stream>>>(x, (h16_t*)hi, (h16_t*)lo, (size_t)n / 8);
    SPIDER_LAUNCH_OK();
    return 0;
}

// x32 [M, K] fp32 -> y2 [M, 2K] 16-bit = [round16(v) | round16(v - round16(v))], v = x32 (gamma == NULL) or
// LayerNorm(x32) * gamma (+ beta) (gamma [K], beta [K] or NULL, 16-bit). K % 8 == 0, K <= 8192.
int SPIDER_FN(spider_row_split_f32)(const float* x32, const void* gamma, const void* beta, void* y2, long M, int K, float eps, void* stream) {
    SPIDER_CHECK(M > 0 && K > 0 && K % 8 == 0 && K <= 8192 && x32 && y2, "row_split: K must be a multiple of 8 and <= 8192");
    SPIDER_CHECK(gamma || !beta, "row_split: beta without gamma");
    const long want = (M + 3) / 4;
    const int grid = (int)(want < 4096 ? want : 4096);
    const h16_t *g = (const h16_t*)gamma, *b = (const h16_t*)beta;
    h16_t* y = (h16_t*)y2;
    hipStream_t st = (hipStream_t)stream;
#define SPIDER_ROW_SPLIT(NIT)                                                                                   \
    if (gamma) row_split_kernel<NIT, true><<<grid, 256, 0, st>>>(x32, g, b, y, M, K, eps);                      \
    else row_split_kernel<NIT, false><<<grid, 256, 0, st>>>(x32, g, b, y, M, K, eps)
    if (K <= 512) { SPIDER_ROW_SPLIT(1); }
    else if (K <= 1024) { SPIDER_ROW_SPLIT(2); }
    else if (K <= 2048) { SPIDER_ROW_SPLIT(4); }
    else if (K <= 4096) { SPIDER_ROW_SPLIT(8); }
    else { SPIDER_ROW_SPLIT(16); }
#undef SPIDER_ROW_SPLIT
    SPIDER_LAUNCH_OK();
    return 0;
}

// GroupNorm (+ SiLU) of the fp32 tensor x32 [B, HW, C] written in the operand-split form y2 [B, HW, 2C] (see spider_row_split_f32);
// statistics as spider_groupnorm_f32in_nhwc (partial, or a pass over x32 into ws).
int SPIDER_FN(spider_groupnorm_f32in_split_nhwc)(const float* x32, const float* partial, int nchunk, const void* gamma, const void* beta,
                                                 void* y2, float* ws, int B, int HW, int C, int G, float eps, int silu, void* stream) {
    SPIDER_CHECK(B > 0 && HW > 0 && C > 0 && G > 0 && G <= 64 && 256 % G == 0, "groupnorm_f32in_split: G must divide 256 and be <= 64");
    SPIDER_CHECK(C % 8 == 0 && C % G == 0 && C <= 8192 && x32 && y2, "groupnorm_f32in_split: C % 8, C % G; x32 and y2 are required");
    const int cv = C / 8;
    int KP = 256 / cv;
    if (KP < 1) KP = 1;
    const int threads = cv * KP;
    SPIDER_CHECK(threads <= 1024 && (threads >= 128 || threads >= 2 * G), "groupnorm_f32in_split: unsupported channel count for the block layout");
    if (!partial) {
        SPIDER_CHECK(ws, "groupnorm_f32in_split: ws is required without partial statistics");
        nchunk = gn_nchunk(HW);
        gn_stats_kernel<true><<<dim3(nchunk, B), threads, (size_t)2 * KP * C * sizeof(float), (hipStream_t)stream>>>(
            (const h16_t*)x32, ws, HW, C, G, nchunk, KP, nullptr, C, nullptr);
        SPIDER_LAUNCH_OK();
        partial = ws;
    }
    SPIDER_CHECK(nchunk > 0, "groupnorm_f32in_split: nchunk");
    int ppb = 4 * KP;
    while ((long)B * ((HW + ppb - 1) / ppb) > 1024 && ppb < 64 * KP) ppb += 4 * KP;
    dim3 g2((HW + ppb - 1) / ppb, B);
    if (silu) gn_apply_kernel<true, true><<<g2, threads, 0, (hipStream_t)stream>>>((const h16_t*)x32, partial, (const h16_t*)gamma,
                                                                             (const h16_t*)beta, (h16_t*)y2, HW, C, G, nchunk, eps, ppb, KP, nullptr, 1);
    else gn_apply_kernel<false, true><<<g2, threads, 0, (hipStream_t)stream>>>((const h16_t*)x32, partial, (const h16_t*)gamma,
                                                                          (const h16_t*)beta, (h16_t*)y2, HW, C, G, nchunk, eps, ppb, KP, nullptr, 1);
    SPIDER_LAUNCH_OK();
    return 0;
}

int SPIDER_FN(spider_add_scaled)(const void* a, const void* b, void* y, long n, float scale, void* stream) {
    SPIDER_CHECK(n > 0 && n % 8 == 0, "add_scaled: n must be a multiple of 8");
    add_scaled_kernel<<<grid_for((size_t)n / 8), 256, 0, (hipStream_t)stream>>>((const h16_t*)a, (const h16_t*)b, (h16_t*)y,
                                                                               (size_t)n / 8, scale);
    SPIDER_LAUNCH_OK();
    return 0;
}

int SPIDER_FN(spider_axpby)(const void* a, const void* b, void* y, long n, float alpha, float beta, void* stream) {
    SPIDER_CHECK(n > 0 && n % 8 == 0, "axpby: n must be a multiple of 8");
    axpby_kernel<<<grid_for((size_t)n / 8), 256, 0, (hipStream_t)stream>>>((const h16_t*)a, (const h16_t*)b, (h16_t*)y,
                                                                          (size_t)n / 8, alpha, beta);
    SPIDER_LAUNCH_OK();
    return 0;
}

int SPIDER_FN(spider_mean_tokens)(const void* x, void* y, int B, int T, int C, void* stream) {
    SPIDER_CHECK(B > 0 && T > 0 && C > 0, "mean_tokens: bad shape");
    mean_tokens_kernel<<<grid_for((size_t)B * C), 256, 0, (hipStream_t)stream>>>((const h16_t*)x, (h16_t*)y, B, T, C);
    SPIDER_LAUNCH_OK();
    return 0;
}

// host_xs: HOST array of E device pointers, each [B, per_batch] bf16; logits [B, ld >= E] bf16 (router outputs before the sigmoid)
int SPIDER_FN(spider_moe_combine)(const void* const* host_xs, int E, const void* logits, int ld, void* y, int B, long per_batch,
                            void* stream) {
    SPIDER_CHECK(E >= 1 && E <= 8 && ld >= E && B > 0 && per_batch > 0, "moe_combine: 1..8 experts, ld >= E");
    MoeCombine mc{};
    mc.E = E;
    for (int e = 0; e < E; ++e) mc.x[e] = (const h16_t*)host_xs[e];
    moe_combine_kernel<<<grid_for((size_t)B * per_batch), 256, 0, (hipStream_t)stream>>>(mc, (const h16_t*)logits, ld, (h16_t*)y, B,
                                                                                        (size_t)per_batch);
    SPIDER_LAUNCH_OK();
    return 0;
}

// ConvTranspose1d, second half: overlap-add of the per-tap GEMM output. cols [B, L_in, k, Cout] fp32 (cols[b,i,j,:] =
// x[b,i,:] . w[:, :, j]); y[b, t, :] = bias + sum over (i, j) with i*stride - pad + j == t. L_out = (L_in-1)*stride - 2*pad + k.
int SPIDER_FN(spider_col2im1d_f32)(const float* cols, const void* bias, void* y, int B, int L_in, int k, int stride, int pad,
                             int Cout, void* stream) {
    SPIDER_CHECK(B > 0 && L_in > 0 && k >= 1 && stride >= 1 && pad >= 0 && Cout > 0 && Cout % 4 == 0, "col2im1d: bad shape (Cout % 4 == 0)");
    const int L_out = (L_in - 1) * stride - 2 * pad + k;
    SPIDER_CHECK(L_out > 0, "col2im1d: empty output");
    const size_t total = (size_t)B * L_out * (Cout / 4);
    col2im1d_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(cols, (const h16_t*)bias, (h16_t*)y, B, L_in, L_out, k,
                                                                      stride, pad, Cout);
    SPIDER_LAUNCH_OK();
    return 0;
}

// y[r, :] = x[r, :] / max(||x[r, :]||_2, eps)   (F.normalize of the CLAP text embedding, custom_ad.py:217-219)
int SPIDER_FN(spider_l2_normalize_rows)(const void* x, void* y, int rows, int n, float eps, void* stream) {
    SPIDER_CHECK(rows > 0 && n > 0 && n <= 8192, "l2_normalize_rows: n must be 1..8192");
    l2norm_rows_kernel<<<rows, 256, 0, (hipStream_t)stream>>>((const h16_t*)x, (h16_t*)y, n, eps);
    SPIDER_LAUNCH_OK();
    return 0;
}

int SPIDER_FN(spider_add)(const void* a, const void* b, void* y, long n, void* stream) {
    SPIDER_CHECK(n > 0 && n % 8 == 0, "add: n must be a multiple of 8");
    add_kernel<<<grid_for((size_t)n / 8), 256, 0, (hipStream_t)stream>>>((const h16_t*)a, (const h16_t*)b, (h16_t*)y,
                                                                        (size_t)n / 8);
    SPIDER_LAUNCH_OK();
    return 0;
}

// conv with Cin <= 8 (conv_in). x [B,H,W,Cin], w [Cout,ks,ks,Cin], y [B,H,W,Cout]; stride 1, same padding.
int SPIDER_FN(spider_conv2d_small_cin)(const void* x, const void* w, const void* bias, void* y, int B, int H, int W, int Cin,
                                 int Cout, int ks, void* stream) {
    SPIDER_CHECK(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 8 && Cout % 8 == 0, "conv_small_cin: Cin <= 8, Cout % 8 == 0");
    SPIDER_CHECK(ks == 1 || ks == 3, "conv_small_cin: kernel size must be 1 or 3");
    const size_t smem = (size_t)Cout * ks * ks * Cin * sizeof(float);
    SPIDER_CHECK(smem <= 160 * 1024, "conv_small_cin: weights exceed LDS");
    const size_t total = (size_t)B * H * W * (Cout / 8);
    // every block re-stages the whole weight tensor into LDS: keep the grid at ~2 blocks per CU and grid-stride
    int grid = grid_for(total);
    if (grid > 512) grid = 512;
    conv_small_cin_kernel<<<grid, 256, smem, (hipStream_t)stream>>>((const h16_t*)x, (const h16_t*)w,
                                                                               (const h16_t*)bias, (h16_t*)y, B, H, W,
                                                                               Cin, Cout, ks);
    SPIDER_LAUNCH_OK();
    return 0;
}

// conv with Cout <= 4 (conv_out). Exactly one of y32 (fp32) / y16 (bf16) is written, layout [B,H,W,Cout].
int SPIDER_FN(spider_conv2d_small_cout)(const void* x, const void* w, const void* bias, void* y32, void* y16, int B, int H,
                                  int W, int Cin, int Cout, int ks, void* stream) {
    SPIDER_CHECK(B > 0 && H > 0 && W > 0 && Cin % 8 == 0 && Cout >= 1 && Cout <= 8, "conv_small_cout: Cout <= 8, Cin % 8 == 0");
    SPIDER_CHECK(ks == 1 || ks == 3, "conv_small_cout: kernel size must be 1 or 3");
    SPIDER_CHECK((y32 != nullptr) != (y16 != nullptr), "conv_small_cout: give exactly one output");
    SPIDER_CHECK((Cout * ks * ks * Cin) % 8 == 0, "conv_small_cout: weight count must be a multiple of 8");
    const size_t smem = (size_t)Cout * ks * ks * Cin * sizeof(h16_t);
    SPIDER_CHECK(smem <= 160 * 1024, "conv_small_cout: weights exceed LDS");
    const size_t npix = (size_t)B * H * W;
    size_t grid = (npix + 3) / 4;
    if (grid > 2048) grid = 2048;   // each block stages all weights into LDS once, then grid-strides over pixels
    conv_small_cout_kernel<<<(int)grid, 256, smem, (hipStream_t)stream>>>((const h16_t*)x, (const h16_t*)w,
                                                                         (const h16_t*)bias, (float*)y32, (h16_t*)y16, B,
                                                                         H, W, Cin, Cout, ks);
    SPIDER_LAUNCH_OK();
    return 0;
}

// "precise" forms of conv_in / conv_out (ABI v4): the input is fp32 NHWC (the un-rounded latents; the fp32 GroupNorm + SiLU output),
// conv_in also leaves the fp32 master of its result.
int SPIDER_FN(spider_conv2d_small_cin_f32in)(const float* x32, const void* w, const void* bias, void* y, float* y32, int B, int H, int W,
                                       int Cin, int Cout, int ks, void* stream) {
    SPIDER_CHECK(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 8 && Cout % 8 == 0 && x32 && y, "conv_small_cin_f32in: Cin <= 8, Cout % 8 == 0");
    SPIDER_CHECK(ks == 1 || ks == 3, "conv_small_cin_f32in: kernel size must be 1 or 3");
    const size_t smem = (size_t)Cout * ks * ks * Cin * sizeof(float);
    SPIDER_CHECK(smem <= 160 * 1024, "conv_small_cin_f32in: weights exceed LDS");
    int grid = grid_for((size_t)B * H * W * (Cout / 8));
    if (grid > 512) grid = 512;
    conv_small_cin_kernel<<<grid, 256, smem, (hipStream_t)stream>>>(nullptr, (const h16_t*)w, (const h16_t*)bias, (h16_t*)y, B, H, W, Cin,
                                                                    Cout, ks, x32, y32);
    SPIDER_LAUNCH_OK();
    return 0;
}

int SPIDER_FN(spider_conv2d_small_cout_f32in)(const float* x32, const void* w, const void* bias, float* y32, int B, int H, int W, int Cin,
                                        int Cout, int ks, void* stream) {
    SPIDER_CHECK(B > 0 && H > 0 && W > 0 && Cin % 8 == 0 && Cout >= 1 && Cout <= 8 && x32 && y32, "conv_small_cout_f32in: Cout <= 8, Cin % 8 == 0");
    SPIDER_CHECK((ks == 1 || ks == 3) && (Cout * ks * ks * Cin) % 8 == 0, "conv_small_cout_f32in: kernel size 1 or 3");
    const size_t smem = (size_t)Cout * ks * ks * Cin * sizeof(h16_t);
    SPIDER_CHECK(smem <= 160 * 1024, "conv_small_cout_f32in: weights exceed LDS");
    size_t grid = ((size_t)B * H * W + 3) / 4;
    if (grid > 2048) grid = 2048;
    conv_small_cout_kernel<<<(int)grid, 256, smem, (hipStream_t)stream>>>(nullptr, (const h16_t*)w, (const h16_t*)bias, y32, nullptr, B, H, W,
                                                                         Cin, Cout, ks, x32);
    SPIDER_LAUNCH_OK();
    return 0;
}

#ifndef SPIDER_F16
// fp32 NCHW latents -> fp32 NHWC [reps * B, HW, C] (the un-rounded UNet input of the "precise" conv_in, ABI v4)
int spider_latent_to_nhwc_f32(const float* lat, float* out, int B, int C, int HW, int reps, float scale, void* stream) {
    SPIDER_CHECK(B > 0 && C > 0 && HW > 0 && reps >= 1, "latent_to_nhwc_f32: bad shape");
    latent_in32_kernel<<<grid_for((size_t)B * C * HW), 256, 0, (hipStream_t)stream>>>(lat, out, B, C, HW, reps, scale);
    SPIDER_LAUNCH_OK();
    return 0;
}
#endif

int SPIDER_FN(spider_latent_to_nhwc)(const float* lat, void* out, int B, int C, int HW, int reps, float scale, void* stream) {
    SPIDER_CHECK(B > 0 && C > 0 && HW > 0 && reps >= 1, "latent_to_nhwc: bad shape");
    latent_in_kernel<<<grid_for((size_t)B * C * HW), 256, 0, (hipStream_t)stream>>>(lat, (h16_t*)out, B, C, HW, reps, scale);
    SPIDER_LAUNCH_OK();
    return 0;
}

#ifndef SPIDER_F16
int spider_cfg_combine_f32(const float* eps2, float* out, int B, int C, int HW, float guidance, void* stream) {
    SPIDER_CHECK(B > 0 && C > 0 && HW > 0, "cfg_combine: bad shape");
    cfg_combine_kernel<<<grid_for((size_t)B * C * HW), 256, 0, (hipStream_t)stream>>>(eps2, out, B, C, HW, guidance);
    SPIDER_LAUNCH_OK();
    return 0;
}

int spider_lincomb_f32(const float* const* ins, const float* coefs, int n, float* out, long total, void* stream) {
    SPIDER_CHECK(n >= 1 && n <= 6 && total > 0, "lincomb: 1..6 terms");
    LinComb lc{};
    lc.n = n;
    for (int j = 0; j < n; ++j) { lc.in[j] = ins[j]; lc.coef[j] = coefs[j]; }
    lincomb_kernel<<<grid_for((size_t)total), 256, 0, (hipStream_t)stream>>>(lc, out, (size_t)total);
    SPIDER_LAUNCH_OK();
    return 0;
}

#endif

int SPIDER_FN(spider_softmax_rows_f32)(const float* x, void* y, int rows, int n, int n_valid, float scale, void* stream) {
    SPIDER_CHECK(rows > 0 && n > 0 && n_valid > 0 && n_valid <= n && scale > 0.f, "softmax_rows: bad shape");
    softmax_rows_kernel<<<rows, 256, 0, (hipStream_t)stream>>>(x, (h16_t*)y, n, n_valid, scale);
    SPIDER_LAUNCH_OK();
    return 0;
}

#ifndef SPIDER_F16
int spider_nhwc_to_nchw_f32(const float* x, float* y, int B, int C, int HW, float mul, float add, int clamp01, void* stream) {
    SPIDER_CHECK(B > 0 && C > 0 && HW > 0, "nhwc_to_nchw: bad shape");
    nhwc_to_nchw_kernel<<<grid_for((size_t)B * C * HW), 256, 0, (hipStream_t)stream>>>(x, y, B, C, HW, mul, add, clamp01);
    SPIDER_LAUNCH_OK();
    return 0;
}

// u [n] fp32 uniforms in [0, 1); words [ceil(n / 64)] uint64: bit j % 64 of word j / 64 = (u[j] < thr) for j < n_valid, else 0
int spider_pack_keep_bits_f32(const float* u, void* words, int n, int n_valid, float thr, void* stream) {
    SPIDER_CHECK(n > 0 && n_valid >= 0, "pack_keep_bits: bad length");
    pack_keep_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(u, (unsigned long long*)words, n, n_valid, thr);
    SPIDER_LAUNCH_OK();
    return 0;
}

#endif

}  // extern "C"
