// LLM autoregressive-decode kernels (HBM-bound, one token per sequence per step).
//
// Replaces the PyTorch op sequences of the reference's Llama forward at S=1:
//   RMSNorm            spider/models/modeling_llama3.py:68-82  (old: modeling_llama.py:57-74)
//   q/k/v/o, MLP GEMV  spider/models/modeling_llama3.py:186-199,240-313
//   RoPE + KV append   spider/models/modeling_llama3.py:128-183  (old: modeling_llama.py:77-123,190-193)
//   decode attention   spider/models/modeling_llama3.py:202-237  (repeat_kv + eager softmax in fp32)
//   lm_head + argmax   spider/models/modeling_llama3.py:870-871 + HF greedy loop (spider.py:1492-1508)
//
// Layouts: activations [B, H] bf16 row-major; weights [N, K] bf16 row-major (nn.Linear layout, never
// transposed); KV cache [B, n_kv, T_max, d] bf16. All reductions accumulate in fp32.
#include "common.hpp"
#include <stdlib.h>

using namespace spider;

// ----------------------------------------------------------------------------------------------
// Embedding gather: out[r, :] = table[ids[r], :]
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const bf16_t* __restrict__ table, const int* __restrict__ ids,
                                                    bf16_t* __restrict__ out, int H, int V) {
    const int r = blockIdx.x;
    int id = ids[r];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const u32x4* src = reinterpret_cast<const u32x4*>(table + (size_t)id * H);
    u32x4* dst = reinterpret_cast<u32x4*>(out + (size_t)r * H);
    for (int i = threadIdx.x; i < H / 8; i += 256) dst[i] = src[i];
}

// ----------------------------------------------------------------------------------------------
// In-place half-rotation RoPE on packed rows with a per-row angle table (vision tower):
//   x[r, h, i]       = bf16(x1 * cos[r, i] - x2 * sin[r, i])        x1 = x[r, h, i], x2 = x[r, h, i + d/2]
//   x[r, h, i + d/2] = bf16(x2 * cos[r, i] + x1 * sin[r, i])
// = transformers apply_rotary_pos_emb_vision (fp32 arithmetic, cos/sin = cos,sin(freqs) with the d/2 angles repeated
// for both halves, result cast back to the tensor dtype). cs [rows, d] fp32 = [cos(d/2) | sin(d/2)].
// One thread per 8 rotary pairs (two 16-byte accesses).
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_rows_kernel(bf16_t* __restrict__ x, const float* __restrict__ cs, long row_stride,
                                                        int rows, int heads, int d) {
    const int half = d >> 1, cpr = half >> 3;                 // 8-pair chunks per head
    const long total = (long)rows * heads * cpr;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % cpr);
        const int h = (int)((idx / cpr) % heads);
        const long r = idx / ((long)cpr * heads);
        bf16_t* p = x + r * row_stride + (long)h * d + c * 8;
        const u32x4 lo = *reinterpret_cast<const u32x4*>(p);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(p + half);
        const float* cr = cs + r * d + c * 8;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cr), c1 = *reinterpret_cast<const f32x4*>(cr + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(cr + half), s1 = *reinterpret_cast<const f32x4*>(cr + half + 4);
        const float co[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        const float si[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        const uint32_t lw[4] = {lo.x, lo.y, lo.z, lo.w}, hw[4] = {hi.x, hi.y, hi.z, hi.w};
        uint32_t ol[4], oh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a0 = bf16lo_to_f32(lw[j]), a1 = bf16hi_to_f32(lw[j]);
            const float b0 = bf16lo_to_f32(hw[j]), b1 = bf16hi_to_f32(hw[j]);
            // separately rounded products, then one add -- the operation order of `t * cos + rotate_half(t) * sin` -- so the
            // result is bit-identical to the fp32 torch expression. The empty asm hides each product from the backend's
            // mul+add -> fma contraction (-ffp-contract=fast ignores the source-level contract pragma).
            float p0 = a0 * co[2 * j], p1 = b0 * si[2 * j], p2 = a1 * co[2 * j + 1], p3 = b1 * si[2 * j + 1];
            float r0 = b0 * co[2 * j], r1 = a0 * si[2 * j], r2 = b1 * co[2 * j + 1], r3 = a1 * si[2 * j + 1];
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
            ol[j] = pack_bf16x2(p0 - p1, p2 - p3);
            oh[j] = pack_bf16x2(r0 + r1, r2 + r3);
        }
        u32x4 vl, vh;
        vl.x = ol[0]; vl.y = ol[1]; vl.z = ol[2]; vl.w = ol[3];
        vh.x = oh[0]; vh.y = oh[1]; vh.z = oh[2]; vh.w = oh[3];
        *reinterpret_cast<u32x4*>(p) = vl;
        *reinterpret_cast<u32x4*>(p + half) = vh;
    }
}

// ----------------------------------------------------------------------------------------------
// RMSNorm (+ optional residual add):  h = x (+ res);  res_out = bf16(h);  y = bf16(w * bf16(h * rsqrt(mean(h^2)+eps)))
// One block (256 threads) per row. Double rounding mirrors `self.weight * hidden_states.to(input_dtype)`.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                      const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                                      bf16_t* __restrict__ res_out, int H, float eps) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const u32x4* xv = reinterpret_cast<const u32x4*>(x + row * H);
    const u32x4* rv = res ? reinterpret_cast<const u32x4*>(res + row * H) : nullptr;
    const u32x4* wv = reinterpret_cast<const u32x4*>(w);
    u32x4* yv = reinterpret_cast<u32x4*>(y + row * H);
    u32x4* rov = res_out ? reinterpret_cast<u32x4*>(res_out + row * H) : nullptr;
    const int nv = H / 8;
    // H <= 8192*? : keep up to 4 vectors per thread in registers (H <= 8192)
    float h[4][8];
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < nv) {
            u32x4 a = xv[i];
            uint32_t aw[4] = {a.x, a.y, a.z, a.w};
            if (rv) {
                u32x4 b = rv[i];
                uint32_t bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // bf16 + bf16 -> bf16 (as `residual + hidden_states` does), then the norm sees the rounded sum
                    float lo = bf16_to_f32(f32_to_bf16(bf16lo_to_f32(aw[j]) + bf16lo_to_f32(bw[j])));
                    float hi = bf16_to_f32(f32_to_bf16(bf16hi_to_f32(aw[j]) + bf16hi_to_f32(bw[j])));
                    h[it][2 * j] = lo;
                    h[it][2 * j + 1] = hi;
                }
                if (rov) {
                    u32x4 o;
                    o.x = pack_bf16x2(h[it][0], h[it][1]);
                    o.y = pack_bf16x2(h[it][2], h[it][3]);
                    o.z = pack_bf16x2(h[it][4], h[it][5]);
                    o.w = pack_bf16x2(h[it][6], h[it][7]);
                    rov[i] = o;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h[it][2 * j] = bf16lo_to_f32(aw[j]);
                    h[it][2 * j + 1] = bf16hi_to_f32(aw[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += h[it][j] * h[it][j];
        }
    }
    const float tot = block_sum<4>(ss, red);
    const float rs = rsqrtf(tot / (float)H + eps);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < nv) {
            u32x4 wq = wv[i];
            uint32_t ww[4] = {wq.x, wq.y, wq.z, wq.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float lo = bf16_to_f32(f32_to_bf16(h[it][2 * j] * rs)) * bf16lo_to_f32(ww[j]);
                float hi = bf16_to_f32(f32_to_bf16(h[it][2 * j + 1] * rs)) * bf16hi_to_f32(ww[j]);
                o[j] = pack_bf16x2(lo, hi);
            }
            u32x4 ov;
            ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
            yv[i] = ov;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Weight-streaming GEMV:  out[b, n] = epilogue( sum_k xin[b, k] * W[n, k] ),   b < NB <= 8
//   xin = x, or RMSNorm(x) * norm_w when norm_w != nullptr (fused prologue; every block recomputes the
//         row norm of its <= 8 activation rows, which is free next to the weight stream)
//   epilogue: (+bias) -> bf16 -> (+res -> bf16);  GATEUP: W holds [gate rows | up rows] (2*N rows) and
//         out[b,n] = bf16(bf16(silu(bf16(g))) * bf16(u))     (modeling_llama3.py:197-199)
// Each wave owns R output columns and streams their weight rows with 16-B loads (1 KiB per wave
// instruction), U chunks deep; activations are staged once per block in LDS as bf16 (XLDS) or, when
// NB*K*2 bytes exceed the LDS budget, re-read through L2.
// ----------------------------------------------------------------------------------------------
template <int NB, int R, bool GATEUP, bool XLDS, int NWB = 4>
__global__ __launch_bounds__(NWB * 64) void gemv_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ x,
                                                   bf16_t* __restrict__ out, const bf16_t* __restrict__ bias,
                                                   const bf16_t* __restrict__ res, const bf16_t* __restrict__ norm_w,
                                                   float eps, int N, int K, int hoist) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);  // [NB][K] when XLDS
    __shared__ float red[NWB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NR = GATEUP ? 2 * R : R;  // weight rows per wave
    constexpr int U = (NR * NB <= 4) ? 4 : 2;  // chunks in flight per row

    // Row pointers first; with `hoist` the first weight chunks are requested BEFORE the activation prologue (the
    // weight stream does not depend on x), so their HBM latency overlaps the RMSNorm / LDS staging below.
    const int col0 = (blockIdx.x * NWB + wave) * R;
    const bf16_t* wrow[NR];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int n = col0 + r;
        n = n < N ? n : N - 1;
        wrow[r] = W + (size_t)n * K;
        if (GATEUP) wrow[R + r] = W + (size_t)(N + n) * K;
    }
    u32x4 w0[U][NR];
    if (hoist) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = lane * 8 + u * 512;
            if (k < K) {
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    w0[u][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r] + k));
            }
        }
    }

    if (XLDS) {
        const int nv = K / 8;
        for (int b = 0; b < NB; ++b) {
            const u32x4* xv = reinterpret_cast<const u32x4*>(x + (size_t)b * K);
            u32x4* sv = reinterpret_cast<u32x4*>(xs + (size_t)b * K);
            if (norm_w) {
                float ss = 0.f;
                for (int i = threadIdx.x; i < nv; i += NWB * 64) {
                    u32x4 a = xv[i];
                    uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float lo = bf16lo_to_f32(aw[j]), hi = bf16hi_to_f32(aw[j]);
                        ss += lo * lo + hi * hi;
                    }
                }
                const float rs = rsqrtf(block_sum<NWB>(ss, red) / (float)K + eps);
                const u32x4* wv = reinterpret_cast<const u32x4*>(norm_w);
                for (int i = threadIdx.x; i < nv; i += NWB * 64) {
                    u32x4 a = xv[i], wq = wv[i];
                    uint32_t aw[4] = {a.x, a.y, a.z, a.w}, ww[4] = {wq.x, wq.y, wq.z, wq.w}, o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float lo = bf16_to_f32(f32_to_bf16(bf16lo_to_f32(aw[j]) * rs)) * bf16lo_to_f32(ww[j]);
                        float hi = bf16_to_f32(f32_to_bf16(bf16hi_to_f32(aw[j]) * rs)) * bf16hi_to_f32(ww[j]);
                        o[j] = pack_bf16x2(lo, hi);
                    }
                    u32x4 ov;
                    ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
                    sv[i] = ov;
                }
            } else {
                for (int i = threadIdx.x; i < nv; i += NWB * 64) sv[i] = xv[i];
            }
        }
        __syncthreads();
    }

    if (col0 >= N) return;
    float acc[NB][NR];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[b][r] = 0.f;

    auto fma_w = [&](const u32x4 (&wq)[U][NR], int k0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 512;
            if (k < K) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    u32x4 xq = XLDS ? *reinterpret_cast<const u32x4*>(xs + (size_t)b * K + k)
                                    : *reinterpret_cast<const u32x4*>(x + (size_t)b * K + k);
                    const uint32_t xw[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const uint32_t ww[4] = {wq[u][r].x, wq[u][r].y, wq[u][r].z, wq[u][r].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[b][r] = dot2_bf16(ww[j], xw[j], acc[b][r]);
                    }
                }
            }
        }
    };
    int kstart = lane * 8;
    if (hoist) {
        fma_w(w0, kstart);
        kstart += 512 * U;
    }
    for (int k0 = kstart; k0 < K; k0 += 512 * U) {
        u32x4 wq[U][NR];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 512;
            if (k < K) {
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    wq[u][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r] + k));
            }
        }
        fma_w(wq, k0);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[b][r] = wave_sum(acc[b][r]);

    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = col0 + r;
            if (n < N) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    float v;
                    if (GATEUP) {
                        const float g = bf16_to_f32(f32_to_bf16(acc[b][r]));
                        const float u = bf16_to_f32(f32_to_bf16(acc[b][R + r]));
                        const float a = bf16_to_f32(f32_to_bf16(silu_f(g)));
                        v = a * u;
                    } else {
                        v = acc[b][r];
                        if (bias) v += bf16_to_f32(bias[n]);
                        if (res) v = bf16_to_f32(f32_to_bf16(v)) + bf16_to_f32(res[(size_t)b * N + n]);
                    }
                    out[(size_t)b * N + n] = f32_to_bf16(v);
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// lm_head + greedy argmax. Stage 1: each wave scores R vocabulary rows (logit rounded to bf16 as the
// reference's bf16 lm_head output is), keeps its best (value, lowest index); each block writes one
// partial. Stage 2: one block reduces the partials per batch row. Ties -> lowest token id.
// ----------------------------------------------------------------------------------------------
template <int NB, int R>
__global__ __launch_bounds__(256) void lmhead_partial_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ x,
                                                             const bf16_t* __restrict__ norm_w, float eps,
                                                             float* __restrict__ pval, int* __restrict__ pidx,
                                                             bf16_t* __restrict__ logits, int V, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    __shared__ float red[4];
    __shared__ float bval[4][NB];
    __shared__ int bidx[4][NB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = K / 8;
    for (int b = 0; b < NB; ++b) {
        const u32x4* xv = reinterpret_cast<const u32x4*>(x + (size_t)b * K);
        u32x4* sv = reinterpret_cast<u32x4*>(xs + (size_t)b * K);
        if (norm_w) {
            float ss = 0.f;
            for (int i = threadIdx.x; i < nv; i += 256) {
                u32x4 a = xv[i];
                uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float lo = bf16lo_to_f32(aw[j]), hi = bf16hi_to_f32(aw[j]);
                    ss += lo * lo + hi * hi;
                }
            }
            const float rs = rsqrtf(block_sum<4>(ss, red) / (float)K + eps);
            const u32x4* wv = reinterpret_cast<const u32x4*>(norm_w);
            for (int i = threadIdx.x; i < nv; i += 256) {
                u32x4 a = xv[i], wq = wv[i];
                uint32_t aw[4] = {a.x, a.y, a.z, a.w}, ww[4] = {wq.x, wq.y, wq.z, wq.w}, o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float lo = bf16_to_f32(f32_to_bf16(bf16lo_to_f32(aw[j]) * rs)) * bf16lo_to_f32(ww[j]);
                    float hi = bf16_to_f32(f32_to_bf16(bf16hi_to_f32(aw[j]) * rs)) * bf16hi_to_f32(ww[j]);
                    o[j] = pack_bf16x2(lo, hi);
                }
                u32x4 ov;
                ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
                sv[i] = ov;
            }
        } else {
            for (int i = threadIdx.x; i < nv; i += 256) sv[i] = xv[i];
        }
    }
    __syncthreads();

    float best[NB];
    int besti[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { best[b] = -INFINITY; besti[b] = 0x7fffffff; }

    // rows handled by this block: [blockIdx.x * rows_per_block, +rows_per_block), waves interleave groups of R
    const int rows_per_block = (V + gridDim.x - 1) / gridDim.x;
    const int rbeg = blockIdx.x * rows_per_block;
    const int rend = min(V, rbeg + rows_per_block);
    for (int n0 = rbeg + wave * R; n0 < rend; n0 += 4 * R) {
        float acc[NB][R];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[b][r] = 0.f;
        const bf16_t* wrow[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wrow[r] = W + (size_t)min(n0 + r, V - 1) * K;
        for (int k = lane * 8; k < K; k += 512) {
            u32x4 wq[R];
#pragma unroll
            for (int r = 0; r < R; ++r) wq[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r] + k));
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                u32x4 xq = *reinterpret_cast<const u32x4*>(xs + (size_t)b * K + k);
                const uint32_t xw[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t ww[4] = {wq[r].x, wq[r].y, wq[r].z, wq[r].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[b][r] = dot2_bf16(ww[j], xw[j], acc[b][r]);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float s = wave_sum(acc[b][r]);
                const int n = n0 + r;
                if (n < rend) {
                    const bf16_t lb = f32_to_bf16(s);
                    const float lv = bf16_to_f32(lb);
                    if (logits && lane == 0) logits[(size_t)b * V + n] = lb;
                    if (lv > best[b] || (lv == best[b] && n < besti[b])) { best[b] = lv; besti[b] = n; }
                }
            }
    }
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) { bval[wave][b] = best[b]; bidx[wave][b] = besti[b]; }
    }
    __syncthreads();
    if (threadIdx.x < NB) {
        const int b = threadIdx.x;
        float bv = bval[0][b];
        int bi = bidx[0][b];
        for (int w = 1; w < 4; ++w) {
            if (bval[w][b] > bv || (bval[w][b] == bv && bidx[w][b] < bi)) { bv = bval[w][b]; bi = bidx[w][b]; }
        }
        pval[(size_t)b * gridDim.x + blockIdx.x] = bv;
        pidx[(size_t)b * gridDim.x + blockIdx.x] = bi;
    }
}

__global__ __launch_bounds__(256) void argmax_final_kernel(const float* __restrict__ pval, const int* __restrict__ pidx,
                                                           int* __restrict__ out, int nparts) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int b = blockIdx.x;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        const float v = pval[(size_t)b * nparts + i];
        const int ix = pidx[(size_t)b * nparts + i];
        if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float v = sv[threadIdx.x + s];
            const int ix = si[threadIdx.x + s];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && ix < si[threadIdx.x])) {
                sv[threadIdx.x] = v;
                si[threadIdx.x] = ix;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[b] = si[0];
}

// End of a decode step, one launch instead of five elementwise ones: the token just chosen becomes the next input, the
// position / cache cursors advance, and the token is appended to the sequence's row of the on-device token history
// (hist [B, cap], n_hist [B] tokens already there) that generate() returns. All index math, one thread per sequence.
__global__ void decode_advance_kernel(const int* __restrict__ next_ids, int* __restrict__ cur_ids, int* __restrict__ pos,
                                      int* __restrict__ slot, int* __restrict__ kv_end, int* __restrict__ hist,
                                      int* __restrict__ n_hist, int cap, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int id = next_ids[b];
    cur_ids[b] = id;
    pos[b] += 1;
    slot[b] += 1;
    kv_end[b] += 1;
    if (hist) {
        const int n = n_hist[b];
        if (n < cap) hist[(size_t)b * cap + n] = id;
        n_hist[b] = n + 1;
    }
}

// ----------------------------------------------------------------------------------------------
// RoPE (half-rotation layout) on q and k + append of k, v into the KV cache.
//   qkv   [rows, (n_q + 2 n_kv) * d]  (rows = B*S, output of the fused QKV projection)
//   pos   [rows] rotary position;  slot [rows] cache index;  batch index = row / S
//   cs    [max_pos, d] fp32: first d/2 = cos, last d/2 = sin (host computes them in fp32 exactly as the
//         reference does, including llama3 scaling and attention_scaling, optionally pre-rounded to bf16)
//   q_out [rows, n_q, d];  kc/vc [B, n_kv, T_max, d]
// out = bf16( bf16(x*cos) + bf16(rot(x)*sin) )  -- bf16 arithmetic order of apply_rotary_pos_emb.
// ----------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void rope_kv_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ pos,
                                                      const int* __restrict__ slot, const float* __restrict__ cs,
                                                      bf16_t* __restrict__ q_out, bf16_t* __restrict__ kc,
                                                      bf16_t* __restrict__ vc, int S, int n_q, int n_kv, int T_max,
                                                      int mrope_s0, int mrope_s1) {
    // mrope_s1 > 0: multimodal RoPE -- pos is [3, rows] (temporal, height, width) and rotary pair i takes its angle from
    // component 0 (i < s0), 1 (i < s1) or 2 (transformers apply_multimodal_rotary_pos_emb, mrope_section = [s0, s1-s0, rest])
    const int row = blockIdx.x, rows = gridDim.x;
    const int b = row / S;
    const int p = pos[row], sl = slot[row];
    const int p1 = mrope_s1 > 0 ? pos[rows + row] : p, p2 = mrope_s1 > 0 ? pos[2 * rows + row] : p;
    const int nh = n_q + 2 * n_kv;
    const bf16_t* src = qkv + (size_t)row * nh * D;
    const float* c = cs + (size_t)p * D;
    constexpr int HALF = D / 2;
    // one thread per (head, pair index i < D/2)
    for (int idx = threadIdx.x; idx < nh * HALF; idx += 256) {
        const int h = idx / HALF, i = idx % HALF;
        const float x1 = bf16_to_f32(src[h * D + i]);
        const float x2 = bf16_to_f32(src[h * D + HALF + i]);
        if (h < n_q + n_kv) {
            const float* ci = mrope_s1 > 0 ? cs + (size_t)(i < mrope_s0 ? p : (i < mrope_s1 ? p1 : p2)) * D : c;
            const float co = ci[i], si = ci[HALF + i];
            const float o1 = bf16_to_f32(f32_to_bf16(x1 * co)) + bf16_to_f32(f32_to_bf16(-x2 * si));
            const float o2 = bf16_to_f32(f32_to_bf16(x2 * co)) + bf16_to_f32(f32_to_bf16(x1 * si));
            if (h < n_q) {
                bf16_t* dst = q_out + ((size_t)row * n_q + h) * D;
                dst[i] = f32_to_bf16(o1);
                dst[HALF + i] = f32_to_bf16(o2);
            } else {
                bf16_t* dst = kc + (((size_t)b * n_kv + (h - n_q)) * T_max + sl) * D;
                dst[i] = f32_to_bf16(o1);
                dst[HALF + i] = f32_to_bf16(o2);
            }
        } else {
            bf16_t* dst = vc + (((size_t)b * n_kv + (h - n_q - n_kv)) * T_max + sl) * D;
            dst[i] = src[h * D + i];
            dst[HALF + i] = src[h * D + HALF + i];
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Decode attention (one query token per sequence), GQA-aware, split over the KV length.
// Block = (split, kv head, batch). The G = n_q / n_kv query heads of a kv head share every K/V row read:
// a wave instruction loads 4 cache rows (16 lanes x 16 B each = 1 KiB, coalesced); the 16 lanes of a row
// reduce the dot products by shuffles; softmax is online per wave in fp32; the 4 waves and the 4 row
// sub-groups are merged through LDS. Partials (m, l, O) are combined by attn_combine_kernel.
//   q [B, n_q, D] bf16;  kc/vc [B, n_kv, T_max, D];  kv_beg/kv_end [B]: valid cache slots [beg, end)
// ----------------------------------------------------------------------------------------------
template <int D, int G>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc,
                                                          const bf16_t* __restrict__ vc, const int* __restrict__ kv_beg,
                                                          const int* __restrict__ kv_end, float* __restrict__ part_o,
                                                          float* __restrict__ part_ml, bf16_t* __restrict__ out,
                                                          int n_kv, int T_max, float scale, int nsplit) {
    static_assert(D == 128, "decode attention is specialised for head_dim 128");
    const int split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
    const int n_q = n_kv * G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4;        // which of the 4 rows of a wave instruction
    const int dl = (lane & 15) * 8;   // this lane's 8 head-dim elements
    const int beg = kv_beg ? kv_beg[b] : 0, end = kv_end[b];
    const int len = max(end - beg, 0);
    const int per = (len + nsplit - 1) / nsplit;
    const int t0 = beg + split * per, t1 = min(end, t0 + per);

    float qf[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(q + ((size_t)b * n_q + hk * G + g) * D + dl);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qf[g][2 * j] = bf16lo_to_f32(aw[j]) * scale;
            qf[g][2 * j + 1] = bf16hi_to_f32(aw[j]) * scale;
        }
    }
    float m[G], l[G], o[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[g][j] = 0.f;
    }
    const bf16_t* kbase = kc + ((size_t)b * n_kv + hk) * (size_t)T_max * D;
    const bf16_t* vbase = vc + ((size_t)b * n_kv + hk) * (size_t)T_max * D;

    // The 16 lanes that share a cache row run the same trip count, and the in-loop shuffles (xor 1..8)
    // never leave that 16-lane group, so row sub-groups may diverge at the tail.
    for (int t = t0 + wave * 4 + sub; t < t1; t += 16) {
        const u32x4 kq = *reinterpret_cast<const u32x4*>(kbase + (size_t)t * D + dl);
        const u32x4 vq = *reinterpret_cast<const u32x4*>(vbase + (size_t)t * D + dl);
        const uint32_t kw[4] = {kq.x, kq.y, kq.z, kq.w};
        const uint32_t vw[4] = {vq.x, vq.y, vq.z, vq.w};
        float kf[8], vf[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            kf[2 * j] = bf16lo_to_f32(kw[j]);
            kf[2 * j + 1] = bf16hi_to_f32(kw[j]);
            vf[2 * j] = bf16lo_to_f32(vw[j]);
            vf[2 * j + 1] = bf16hi_to_f32(vw[j]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += qf[g][j] * kf[j];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 8, 64);
            const float mn = fmaxf(m[g], s);
            const float alpha = __expf(m[g] - mn);  // m = -inf on the first row -> 0
            const float p = __expf(s - mn);
            l[g] = l[g] * alpha + p;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[g][j] = o[g][j] * alpha + p * vf[j];
            m[g] = mn;
        }
    }

    // merge the 4 row sub-groups of the wave (lanes differing in bits 4,5), then the 4 waves via LDS
    __shared__ float sm_m[4][G], sm_l[4][G];
    __shared__ float sm_o[4][G][D];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float m2 = __shfl_xor(m[g], off, 64);
            const float l2 = __shfl_xor(l[g], off, 64);
            const float mn = fmaxf(m[g], m2);
            const float a1 = (m[g] == -INFINITY) ? 0.f : __expf(m[g] - mn);
            const float a2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
            l[g] = l[g] * a1 + l2 * a2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float o2 = __shfl_xor(o[g][j], off, 64);
                o[g][j] = o[g][j] * a1 + o2 * a2;
            }
            m[g] = mn;
        }
        if (lane < 16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sm_o[wave][g][dl + j] = o[g][j];
            if (lane == 0) { sm_m[wave][g] = m[g]; sm_l[wave][g] = l[g]; }
        }
    }
    __syncthreads();
    // final: thread (g, dd) for g < G, dd < D  -> G*D <= 7*128 = 896 values over 256 threads
    for (int idx = threadIdx.x; idx < G * D; idx += 256) {
        const int g = idx / D, dd = idx % D;
        float mm = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mm = fmaxf(mm, sm_m[w][g]);
        float ll = 0.f, oo = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a = (sm_m[w][g] == -INFINITY) ? 0.f : __expf(sm_m[w][g] - mm);
            ll += sm_l[w][g] * a;
            oo += sm_o[w][g][dd] * a;
        }
        const int hq = hk * G + g;
        if (nsplit == 1) {
            out[((size_t)b * n_q + hq) * D + dd] = f32_to_bf16(ll > 0.f ? oo / ll : 0.f);
        } else {
            const size_t pi = ((size_t)b * n_q + hq) * nsplit + split;
            part_o[pi * D + dd] = oo;
            if (dd == 0) { part_ml[pi * 2] = mm; part_ml[pi * 2 + 1] = ll; }
        }
    }
}

template <int D>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                           bf16_t* __restrict__ out, int nsplit) {
    static_assert(D == 128, "one wave covers the head dim with 2 elements per lane");
    __shared__ float sm_m[4], sm_l[4], sm_o[4][D];
    const size_t bh = blockIdx.x;  // b * n_q + hq
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -INFINITY, l = 0.f, o0 = 0.f, o1 = 0.f;
    for (int s0 = wave; s0 < nsplit; s0 += 64) {       // 16 independent (m,l) + O loads in flight per wave: 64 splits = ONE round trip
        float2 mlv[16], ovv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int s = s0 + 4 * u;
            const bool ok = s < nsplit;
            const size_t pi = bh * nsplit + (ok ? s : 0);
            mlv[u] = *reinterpret_cast<const float2*>(part_ml + pi * 2);
            ovv[u] = *reinterpret_cast<const float2*>(part_o + pi * D + lane * 2);
            if (!ok) mlv[u] = float2{-INFINITY, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float ms = mlv[u].x, ls = mlv[u].y;
            const float mn = fmaxf(m, ms);
            const float a = (m == -INFINITY) ? 0.f : __expf(m - mn);
            const float bsc = (ms == -INFINITY) ? 0.f : __expf(ms - mn);
            l = l * a + ls * bsc;
            o0 = o0 * a + ovv[u].x * bsc;
            o1 = o1 * a + ovv[u].y * bsc;
            m = mn;
        }
    }
    if (lane == 0) { sm_m[wave] = m; sm_l[wave] = l; }
    sm_o[wave][lane * 2] = o0;
    sm_o[wave][lane * 2 + 1] = o1;
    __syncthreads();
    if (threadIdx.x < D) {
        const int dd = threadIdx.x;
        float mm = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
        float ll = 0.f, oo = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a = (sm_m[w] == -INFINITY) ? 0.f : __expf(sm_m[w] - mm);
            ll += sm_l[w] * a;
            oo += sm_o[w][dd] * a;
        }
        out[bh * D + dd] = f32_to_bf16(ll > 0.f ? oo / ll : 0.f);
    }
}

// ----------------------------------------------------------------------------------------------
// Fused decode attention: RoPE(q, k_new) + KV-cache append + split-KV attention + cross-block combine in ONE
// launch (replaces rope_kv_kernel + attn_decode_kernel + attn_combine_kernel: 3 launches -> 1 per layer).
//   qkv [B, (n_q + 2 n_kv) * D] : fused projection of the current token;  pos [B] rotary position
//   caches [B, n_kv, T_max, D]; valid slots [kv_beg, kv_end) where slot kv_end-1 is THIS token: it is rotated in
//   registers, used from registers by the last split and written to the cache by that block only, so no block
//   reads a cache row another block writes in this launch.
// Cross-block combine (split-KV): every block stores its partial (m, l, O) with plain stores, drains them,
// lane 0 issues an agent-scope release and takes a ticket on cnt[b, kv head]; the block that draws the last
// ticket issues an agent-scope acquire, reduces all partials and writes the output (placement-independent
// release/acquire hand-off; the counter is reset by the reducer, so graph replays need no memset).
// ----------------------------------------------------------------------------------------------
template <int D, int G, int NWV, int UR>
__global__ __launch_bounds__(NWV * 64) void attn_decode_fused_kernel(
    const bf16_t* __restrict__ qkv, const int* __restrict__ pos, const float* __restrict__ cs, bf16_t* __restrict__ kc,
    bf16_t* __restrict__ vc, const int* __restrict__ kv_beg, const int* __restrict__ kv_end, float* __restrict__ part_o,
    float* __restrict__ part_ml, int* __restrict__ cnt, bf16_t* __restrict__ out, int n_kv, int T_max, float scale,
    int nsplit, int inline_combine) {
    static_assert(D == 128, "decode attention is specialised for head_dim 128");
    constexpr int HALF = D / 2;
    const int split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
    const int n_q = n_kv * G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4;
    const int dl = (lane & 15) * 8;
    const bool hi_half = dl >= HALF;
    const int dp = hi_half ? dl - HALF : dl + HALF;   // rotary partner chunk
    const int beg = kv_beg ? kv_beg[b] : 0, end = kv_end[b];
    const int t_new = end - 1;                        // slot of the current token
    const int len_old = max(t_new - beg, 0);
    const int per = (len_old + nsplit - 1) / nsplit;
    const int t0 = beg + split * per, t1 = min(t_new, t0 + per);

    // cache rows of this lane group, requested first: nothing below depends on them until the score loop
    const bf16_t* kbase = kc + ((size_t)b * n_kv + hk) * (size_t)T_max * D;
    const bf16_t* vbase = vc + ((size_t)b * n_kv + hk) * (size_t)T_max * D;
    constexpr int RSTEP = 4 * NWV;                        // cache rows covered by one instruction of every wave of the block
    auto load_rows = [&](int tb_, u32x4 (&kr)[UR], u32x4 (&vr)[UR]) {
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int t = min(tb_ + RSTEP * u, t1 - 1);   // clamped rows are loaded but not used
            kr[u] = *reinterpret_cast<const u32x4*>(kbase + (size_t)t * D + dl);
            vr[u] = *reinterpret_cast<const u32x4*>(vbase + (size_t)t * D + dl);
        }
    };
    const int tb0 = t0 + wave * 4 + sub;
    u32x4 kq[UR], vq[UR];
    if (tb0 < t1) load_rows(tb0, kq, vq);

    const bf16_t* row = qkv + (size_t)b * (n_q + 2 * n_kv) * D;
    const float* c = cs + (size_t)pos[b] * D;
    float cosv[8], sinv[8];
    {
        const int ci = hi_half ? dl - HALF : dl;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(c + ci), c1 = *reinterpret_cast<const f32x4*>(c + ci + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(c + HALF + ci), s1 = *reinterpret_cast<const f32x4*>(c + HALF + ci + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { cosv[j] = c0[j]; cosv[4 + j] = c1[j]; sinv[j] = s0[j]; sinv[4 + j] = s1[j]; }
    }
    // rotate 8 elements of one head: out = bf16( bf16(x*cos) + bf16(+-partner*sin) )  (rope_kv_kernel's arithmetic)
    auto rope8 = [&](const bf16_t* head, float (&o)[8]) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(head + dl), pq = *reinterpret_cast<const u32x4*>(head + dp);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, pw[4] = {pq.x, pq.y, pq.z, pq.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xv = (j & 1) ? bf16hi_to_f32(aw[j >> 1]) : bf16lo_to_f32(aw[j >> 1]);
            const float pv = (j & 1) ? bf16hi_to_f32(pw[j >> 1]) : bf16lo_to_f32(pw[j >> 1]);
            const float t1_ = bf16_to_f32(f32_to_bf16(xv * cosv[j]));
            const float t2_ = bf16_to_f32(f32_to_bf16((hi_half ? pv : -pv) * sinv[j]));
            o[j] = bf16_to_f32(f32_to_bf16(t1_ + t2_));
        }
    };

    float qf[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        rope8(row + (size_t)(hk * G + g) * D, qf[g]);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[g][j] *= scale;
    }
    float m[G], l[G], o[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[g][j] = 0.f;
    }
    auto update = [&](const float (&kf)[8], const float (&vf)[8]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float sc = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sc += qf[g][j] * kf[j];
            sc += __shfl_xor(sc, 1, 64);
            sc += __shfl_xor(sc, 2, 64);
            sc += __shfl_xor(sc, 4, 64);
            sc += __shfl_xor(sc, 8, 64);
            const float mn = fmaxf(m[g], sc);
            const float alpha = __expf(m[g] - mn);
            const float pr = __expf(sc - mn);
            l[g] = l[g] * alpha + pr;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[g][j] = o[g][j] * alpha + pr * vf[j];
            m[g] = mn;
        }
    };

    // UR cache rows per lane group in flight (K and V): NWV * 4 * UR rows per block and chunk (8 waves x 8: 256 rows, i.e. the
    // whole split of a 2k-token context in ONE round trip). The first chunk was requested at the top of the kernel, before the q
    // loads and RoPE; later chunks are requested one iteration ahead.
    constexpr int CHUNK = RSTEP * UR;
    for (int tb = tb0; tb < t1; tb += CHUNK) {
        const bool more = tb + CHUNK < t1;
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            if (tb + RSTEP * u < t1) {                    // uniform within the 16-lane row group
                const uint32_t kw[4] = {kq[u].x, kq[u].y, kq[u].z, kq[u].w}, vw[4] = {vq[u].x, vq[u].y, vq[u].z, vq[u].w};
                float kf[8], vf[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    kf[2 * j] = bf16lo_to_f32(kw[j]); kf[2 * j + 1] = bf16hi_to_f32(kw[j]);
                    vf[2 * j] = bf16lo_to_f32(vw[j]); vf[2 * j + 1] = bf16hi_to_f32(vw[j]);
                }
                update(kf, vf);
            }
            if (more) {                                   // refill this slot for the next chunk (ring: no second register set)
                const int t = min(tb + CHUNK + RSTEP * u, t1 - 1);
                kq[u] = *reinterpret_cast<const u32x4*>(kbase + (size_t)t * D + dl);
                vq[u] = *reinterpret_cast<const u32x4*>(vbase + (size_t)t * D + dl);
            }
        }
    }
    // the current token: last split, wave 0, row sub-group 0 (16 lanes) -- from registers, and appended to the cache
    if (split == nsplit - 1 && wave == 0 && sub == 0 && t_new >= beg) {
        float kf[8], vf[8];
        rope8(row + (size_t)(n_q + hk) * D, kf);
        const u32x4 vq = *reinterpret_cast<const u32x4*>(row + (size_t)(n_q + n_kv + hk) * D + dl);
        const uint32_t vw[4] = {vq.x, vq.y, vq.z, vq.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { vf[2 * j] = bf16lo_to_f32(vw[j]); vf[2 * j + 1] = bf16hi_to_f32(vw[j]); }
        update(kf, vf);
        u32x4 ko;
        ko.x = pack_bf16x2(kf[0], kf[1]); ko.y = pack_bf16x2(kf[2], kf[3]);
        ko.z = pack_bf16x2(kf[4], kf[5]); ko.w = pack_bf16x2(kf[6], kf[7]);
        *reinterpret_cast<u32x4*>(kc + (((size_t)b * n_kv + hk) * T_max + t_new) * D + dl) = ko;
        *reinterpret_cast<u32x4*>(vc + (((size_t)b * n_kv + hk) * T_max + t_new) * D + dl) = vq;
    }

    // merge the 4 row sub-groups of the wave, then the 4 waves via LDS
    __shared__ float sm_m[NWV][G], sm_l[NWV][G];
    __shared__ float sm_o[NWV][G][D];
    __shared__ int sm_last;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float m2 = __shfl_xor(m[g], off, 64);
            const float l2 = __shfl_xor(l[g], off, 64);
            const float mn = fmaxf(m[g], m2);
            const float a1 = (m[g] == -INFINITY) ? 0.f : __expf(m[g] - mn);
            const float a2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
            l[g] = l[g] * a1 + l2 * a2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float o2 = __shfl_xor(o[g][j], off, 64);
                o[g][j] = o[g][j] * a1 + o2 * a2;
            }
            m[g] = mn;
        }
        if (lane < 16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sm_o[wave][g][dl + j] = o[g][j];
            if (lane == 0) { sm_m[wave][g] = m[g]; sm_l[wave][g] = l[g]; }
        }
    }
    __syncthreads();
    if (nsplit > 1 && inline_combine) {
        // ---- one launch: publish this split's partial with write-through (sc1) 16-byte stores, take a ticket; the block that
        // draws the last ticket of its (sequence, kv head) merges all splits in split order (so the result does not depend on
        // which block that is) and writes the output. Hand-off (MI355X_MICROARCH.md, valid forms): every byte stored sc1, each
        // storing wave drains (vmcnt(0)), workgroup barrier, ONE lane's agent-scope atomic add; the last arriver takes an
        // agent-scope acquire before its plain loads. The counter is reset by the reducer: graph replays need no memset.
        for (int idx = threadIdx.x; idx < G * (D / 4); idx += NWV * 64) {
            const int g = idx / (D / 4), d4 = (idx % (D / 4)) * 4;
            float mm = -INFINITY;
#pragma unroll
            for (int w = 0; w < NWV; ++w) mm = fmaxf(mm, sm_m[w][g]);
            float ll = 0.f;
            f32x4 oo = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NWV; ++w) {
                const float a = (sm_m[w][g] == -INFINITY) ? 0.f : __expf(sm_m[w][g] - mm);
                ll += sm_l[w][g] * a;
#pragma unroll
                for (int j = 0; j < 4; ++j) oo[j] += sm_o[w][g][d4 + j] * a;
            }
            const size_t pi = ((size_t)b * n_q + hk * G + g) * nsplit + split;
            float* po = part_o + pi * D + d4;
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(po), "v"(oo) : "memory");
            if (d4 == 0) {
                const float2 mlv = float2{mm, ll};
                float* pm = part_ml + pi * 2;
                asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(pm), "v"(mlv) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its stores
        __syncthreads();
        if (threadIdx.x == 0) {
            int ticket;
            if (inline_combine == 2) {
                // form 2 (round 6): NO fence. Every byte of the partials left this block write-through (sc1) and was drained above, so
                // a RELAXED agent-scope add is enough to publish them, and the reducer reads them with sc1 loads (past its L1 and the
                // local L2's stale lines) -- MI355X_MICROARCH.md hand-off table row 1 (one lane per storing workgroup adds to ONE counter;
                // the workgroup whose add came last loads behind a barrier the adding wave joins; stores 8 / 16 B sc1, loads 8 / 16 B
                // sc1). An acq_rel ticket costs one L2 write-back + invalidate per block: + 9 us per layer under the two-stream schedule.
                ticket = __hip_atomic_fetch_add(cnt + (size_t)b * n_kv + hk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // form 1: acq_rel at agent scope -- the compiler emits the write-back of this block's partials in front of the ticket and
                // the invalidate behind it (placement-independent release / acquire pair, MI355X_MICROARCH.md "Valid forms")
                ticket = __hip_atomic_fetch_add(cnt + (size_t)b * n_kv + hk, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            sm_last = (ticket == nsplit - 1) ? 1 : 0;
        }
        __syncthreads();
        if (!sm_last) return;
        // reducer: thread (g, 4 head dims) walks the splits in split order with a running maximum (the same sum whichever block is
        // last); 16 partial rows + their (m, l) in flight per thread; EVERY load of the handed-off bytes is an sc1 buffer load
        const uint32_t po_bytes = (uint32_t)((size_t)gridDim.z * n_q * nsplit * D * sizeof(float));
        const __amdgpu_buffer_rsrc_t r_o = __builtin_amdgcn_make_buffer_rsrc(part_o, 0, po_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_ml = __builtin_amdgcn_make_buffer_rsrc(part_ml, 0, po_bytes / D * 2, 0x00020000);
        typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(r_o, 0, 0, 0)) v128_t;
        typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(r_ml, 0, 0, 0)) v64_t;
        for (int idx = threadIdx.x; idx < G * (D / 4); idx += NWV * 64) {
            const int g = idx / (D / 4), d4 = (idx % (D / 4)) * 4;
            const size_t bh = (size_t)b * n_q + hk * G + g;
            float mx = -INFINITY, lsum = 0.f;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int s0 = 0; s0 < nsplit; s0 += 16) {
                f32x4 ov[16];
                float2 mv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int sidx = min(s0 + u, nsplit - 1);
                    const uint32_t row = (uint32_t)(bh * nsplit + sidx);
                    ov[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_o, (row * D + d4) * 4u, 0, 16));
                    mv[u] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r_ml, row * 8u, 0, 16));
                }
                float bm = mx;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (s0 + u < nsplit) bm = fmaxf(bm, mv[u].x);
                const float resc = (mx == -INFINITY) ? 0.f : __expf(mx - bm);
                lsum *= resc;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] *= resc;
                mx = bm;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float w = (s0 + u < nsplit && mv[u].x != -INFINITY) ? __expf(mv[u].x - mx) : 0.f;
                    lsum += mv[u].y * w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += ov[u][j] * w;
                }
            }
            const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
            u32x2 o2;
            o2.x = pack_bf16x2(acc[0] * inv, acc[1] * inv);
            o2.y = pack_bf16x2(acc[2] * inv, acc[3] * inv);
            *reinterpret_cast<u32x2*>(out + bh * D + d4) = o2;
        }
        if (threadIdx.x == 0) __hip_atomic_store(cnt + (size_t)b * n_kv + hk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    for (int idx = threadIdx.x; idx < G * D; idx += NWV * 64) {
        const int g = idx / D, dd = idx % D;
        float mm = -INFINITY;
#pragma unroll
        for (int w = 0; w < NWV; ++w) mm = fmaxf(mm, sm_m[w][g]);
        float ll = 0.f, oo = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            const float a = (sm_m[w][g] == -INFINITY) ? 0.f : __expf(sm_m[w][g] - mm);
            ll += sm_l[w][g] * a;
            oo += sm_o[w][g][dd] * a;
        }
        const int hq = hk * G + g;
        if (nsplit == 1) {
            out[((size_t)b * n_q + hq) * D + dd] = f32_to_bf16(ll > 0.f ? oo / ll : 0.f);
        } else {
            const size_t pi = ((size_t)b * n_q + hq) * nsplit + split;
            part_o[pi * D + dd] = oo;
            if (dd == 0) { part_ml[pi * 2] = mm; part_ml[pi * 2 + 1] = ll; }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Skinny MFMA GEMM for batched decode (2..16 sequences): out[m, n] = epilogue(sum_k x[m,k] W[n,k]).
// The FMA GEMV above spends NB x the VALU work per weight byte and turns VALU-bound beyond ~4 sequences; here the
// weight rows are the MFMA "A" operand (16 output columns per block) and the <= 16 activation rows the "B" operand,
// so the weight stream is read once at the HBM rate for any batch <= 16. Each of the NW waves takes 1/NW of K
// (deep unrolled 32-byte loads straight to registers: weights are used once, no LDS staging), partial tiles are
// summed through LDS. k is permuted identically for both operands (lane group g owns k = 16g..16g+15 of every
// 64-wide block) so every lane issues two adjacent 16-byte loads = full 128-byte lines per weight row.
// ----------------------------------------------------------------------------------------------
template <bool GATEUP, int NW>
__global__ __launch_bounds__(NW * 64) void skinny_gemm_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ x,
                                                          bf16_t* __restrict__ out, const bf16_t* __restrict__ bias,
                                                          const bf16_t* __restrict__ res, int B, int N, int K) {
    constexpr int TN = GATEUP ? 2 : 1;
    __shared__ float part[NW][TN][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int nrow = min(n0 + r16, N - 1);
    const bf16_t* wrow[TN];
    wrow[0] = W + (size_t)nrow * K + 16 * g;
    if (GATEUP) wrow[1] = W + (size_t)(N + nrow) * K + 16 * g;
    const bool x_ok = r16 < B;
    const bf16_t* xrow = x + (size_t)(x_ok ? r16 : 0) * K + 16 * g;

    const int nkb = K / 64;
    const int kb0 = (wave * nkb) / NW, kb1 = ((wave + 1) * nkb) / NW;
    f32x4 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4 zero = {0u, 0u, 0u, 0u};
    constexpr int UB = 4;   // 64-wide k blocks in flight
    for (int kb = kb0; kb < kb1; kb += UB) {
        u32x4 wq[UB][TN][2], xq[UB][2];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (kb + u < kb1) {
                const size_t ko = (size_t)(kb + u) * 64;
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    wq[u][t][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[t] + ko));
                    wq[u][t][1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[t] + ko + 8));
                }
                xq[u][0] = x_ok ? *reinterpret_cast<const u32x4*>(xrow + ko) : zero;
                xq[u][1] = x_ok ? *reinterpret_cast<const u32x4*>(xrow + ko + 8) : zero;
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (kb + u < kb1) {
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    const bf16x8 xf = __builtin_bit_cast(bf16x8, xq[u][sx]);
#pragma unroll
                    for (int t = 0; t < TN; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wq[u][t][sx]), xf, acc[t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) part[wave][t][lane][j] = acc[t][j];
    __syncthreads();
    if (wave != 0) return;
    // lane holds D[n = n0 + 4g + j][m = r16]
    float v[TN][4];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += part[w][t][lane][j];
            v[t][j] = a;
        }
    if (!x_ok) return;
    const int m = r16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + 4 * g + j;
        if (n < N) {
            float o;
            if (GATEUP) {
                const float gg = bf16_to_f32(f32_to_bf16(v[0][j])), uu = bf16_to_f32(f32_to_bf16(v[1][j]));
                o = bf16_to_f32(f32_to_bf16(silu_f(gg))) * uu;
            } else {
                o = v[0][j];
                if (bias) o += bf16_to_f32(bias[n]);
                if (res) o = bf16_to_f32(f32_to_bf16(o)) + bf16_to_f32(res[(size_t)m * N + n]);
            }
            out[(size_t)m * N + n] = f32_to_bf16(o);
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Skinny MFMA GEMM on a FRAGMENT-MAJOR weight copy (batched decode, 5..16 sequences; also the batched lm_head).
// Layout of Wfm (built once per weight at load time, spider_amd.ops.repack_fm16): for every group of 16 weight rows and every
// 64-wide k block, two 1 KiB pieces (k sub-steps sx = 0, 1); piece = [lane 0..63][8 bf16] with lane = 16 g + r holding
// W[16 rg + r][64 kb + 32 sx + 8 g + 0..7] -- exactly the A fragment of mfma_f32_16x16x32_bf16. One wave instruction
// therefore reads 1 KiB of CONTIGUOUS memory; the row-major form above gathers 16 rows x 64 B per instruction and was
// measured at 2.0 - 4.0 TB/s at 8 sequences where this form streams at 3.2 (N = 3584, one block per CU) - 5.9 TB/s.
// The <= 16 activation rows are the B operand (natural k order, rows >= B read as zeros through the buffer range);
// each of the NW waves takes 1/NW of K and keeps 2 k blocks in flight; partial tiles are summed through LDS.
// MODE 0: out = x.W^T (+bias) (+res);  1: W = [gate rows | up rows], out = silu(g) * u;
// MODE 2: lm_head -- logits rounded to bf16, running arg-max per sequence over the block's row groups (ties -> lowest id),
//         one (value, index) partial per block and sequence, optional bf16 logits.
// ----------------------------------------------------------------------------------------------
// NORM: x is the UN-normalised residual stream and Wfm was built from W * diag(norm_w): the kernel accumulates sum x^2 of
// every sequence from the x fragments it streams anyway and applies rsqrt(mean + eps) to the accumulators (RMSNorm folded
// into the projection: no rmsnorm launch, no normalised copy of x).
template <int MODE, int NW, bool NORM>
__global__ __launch_bounds__(NW * 64) void skinny_fm_kernel(const bf16_t* __restrict__ Wfm, const bf16_t* __restrict__ x,
                                                            bf16_t* __restrict__ out, const bf16_t* __restrict__ bias,
                                                            const bf16_t* __restrict__ res, float* __restrict__ pval,
                                                            int* __restrict__ pidx, int B, int N, int K, int NG,
                                                            uint32_t w_bytes, uint32_t x_bytes, float eps) {
    constexpr bool GATEUP = MODE == 1;
    constexpr int TN = GATEUP ? 2 : 1;
    constexpr int D = 2;
    __shared__ float part[NW][TN][64][4];
    __shared__ float ssq[NW][16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Wfm), 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, x_bytes, 0x00020000);
    const uint32_t xbase = (uint32_t)r16 * (uint32_t)K * 2u + 16u * g;      // rows >= B lie beyond x_bytes: read as zeros
    const int nkb = K / 64;
    const int kb0 = (wave * nkb) / NW, kb1 = ((wave + 1) * nkb) / NW;
    const uint32_t grp_bytes = (uint32_t)nkb * 2048u;
    float best = -INFINITY;
    int besti = 0x7fffffff;

    for (int rg = blockIdx.x; rg < NG; rg += gridDim.x) {
        uint32_t wbase[TN];
        wbase[0] = (uint32_t)rg * grp_bytes + 16u * lane;
        if (GATEUP) wbase[1] = (uint32_t)(NG + rg) * grp_bytes + 16u * lane;
        f32x4 acc[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float sq = 0.f;     // NORM: this lane's share of sum_k x[r16, k]^2 over the wave's K range
        u32x4 wq[D][TN][2], xq[D][2];
        auto issue = [&](int kb, int s) {
            const uint32_t inv = (uint32_t)((kb1 - 1 - kb) >> 31);          // all ones past this wave's K range: zeros
            const uint32_t kw = (uint32_t)kb * 2048u, kx = (uint32_t)kb * 128u;
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                wq[s][t][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (wbase[t] + kw) | inv, 0, 2));
                wq[s][t][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (wbase[t] + kw + 1024u) | inv, 0, 2));
            }
            xq[s][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (xbase + kx) | inv, 0, 0));
            xq[s][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (xbase + kx + 64u) | inv, 0, 0));
        };
#pragma unroll
        for (int s = 0; s < D; ++s) issue(kb0 + s, s);
        for (int kb = kb0; kb < kb1; kb += D) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    const bf16x8 xf = __builtin_bit_cast(bf16x8, xq[s][sx]);
#pragma unroll
                    for (int t = 0; t < TN; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wq[s][t][sx]), xf, acc[t], 0, 0, 0);
                    if (NORM) {
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const float lo = bf16lo_to_f32(xq[s][sx][d]), hi = bf16hi_to_f32(xq[s][sx][d]);
                            sq = fmaf(lo, lo, fmaf(hi, hi, sq));
                        }
                    }
                }
                issue(kb + D + s, s);
                __builtin_amdgcn_sched_barrier(0);      // keep "consume slot s, refill slot s" in this order
            }
        }
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[wave][t][lane][j] = acc[t][j];
        if (NORM) {
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (lane < 16) ssq[wave][lane] = sq;
        }
        __syncthreads();
        if (wave == 0) {
            // lane holds D[n = 16 rg + 4 g + j][m = r16]
            float v[TN][4];
            float rs = 1.f;
            if (NORM) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) tot += ssq[w][r16];
                rs = rsqrtf(tot / (float)K + eps);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) a += part[w][t][lane][j];
                    v[t][j] = a * rs;
                }
            const int m = r16;
            if (MODE == 2) {
                float lb = -INFINITY;
                int li = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nn = 16 * rg + 4 * g + j;
                    const bf16_t q = f32_to_bf16(v[0][j]);
                    const float lv = bf16_to_f32(q);
                    if (nn < N && m < B) {
                        if (out) out[(size_t)m * N + nn] = q;
                        if (lv > lb) { lb = lv; li = nn; }        // j ascending: the first maximum has the lowest index
                    }
                }
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {        // the 4 lane groups of a sequence
                    const float ov = __shfl_xor(lb, off, 64);
                    const int oi = __shfl_xor(li, off, 64);
                    if (ov > lb || (ov == lb && oi < li)) { lb = ov; li = oi; }
                }
                if (lb > best || (lb == best && li < besti)) { best = lb; besti = li; }
            } else if (m < B) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nn = 16 * rg + 4 * g + j;
                    if (nn < N) {
                        float o;
                        if (GATEUP) {
                            const float gg = bf16_to_f32(f32_to_bf16(v[0][j])), uu = bf16_to_f32(f32_to_bf16(v[TN - 1][j]));
                            o = bf16_to_f32(f32_to_bf16(silu_f(gg))) * uu;
                        } else {
                            o = v[0][j];
                            if (bias) o += bf16_to_f32(bias[nn]);
                            if (res) o = bf16_to_f32(f32_to_bf16(o)) + bf16_to_f32(res[(size_t)m * N + nn]);
                        }
                        out[(size_t)m * N + nn] = f32_to_bf16(o);
                    }
                }
            }
        }
        if (MODE == 2) __syncthreads();     // `part` is rewritten by the next row group
    }
    if (MODE == 2 && wave == 0 && lane < 16 && lane < B) {
        pval[(size_t)lane * gridDim.x + blockIdx.x] = best;
        pidx[(size_t)lane * gridDim.x + blockIdx.x] = besti;
    }
}

#define GEMV_ARGS (const bf16_t*)W, (const bf16_t*)x, (bf16_t*)out, (const bf16_t*)bias, (const bf16_t*)res, (const bf16_t*)norm_w, eps, N, K, hoist
// wide8: 8 waves per block share one LDS copy of the activations (long-K projections: the 38 KB copy of a K = 18944 vector per
// 4-row block is 25 % of the block's weight bytes and caps the resident waves per CU)
#define GEMV_LAUNCH(NB_, R_, GU_, XL_)                                                                         \
    do {                                                                                                        \
        if (wide8) gemv_kernel<NB_, R_, GU_, XL_, 8><<<(N + 8 * R_ - 1) / (8 * R_), 512, XL_ ? (size_t)NB_ * K * 2 : 0, (hipStream_t)stream>>>(GEMV_ARGS); \
        else gemv_kernel<NB_, R_, GU_, XL_, 4><<<grid, 256, XL_ ? (size_t)NB_ * K * 2 : 0, (hipStream_t)stream>>>(GEMV_ARGS); \
    } while (0)

static int gemv_env_r() {
    static const int v = [] { const char* e = getenv("SPIDER_GEMV_R"); return e ? atoi(e) : 0; }();
    return v;
}

template <int NB, bool GU>
static int gemv_dispatch(const void* W, const void* x, void* out, const void* bias, const void* res,
                         const void* norm_w, float eps, int N, int K, void* stream) {
    // tuning aid: SPIDER_GEMV_XLDS_MAX = largest activation footprint (bytes) staged in LDS; beyond it the waves read x through L2
    // (a 38 KB LDS block -- the down projection -- cannot share a CU with a 110-147 KB workgroup of the diffusion stream)
    static const size_t xlds_max = [] { const char* e = getenv("SPIDER_GEMV_XLDS_MAX"); return e ? (size_t)atol(e) : (size_t)64 * 1024; }();
    const bool xlds = (size_t)NB * K * 2 <= xlds_max || norm_w != nullptr;
    if (!xlds) SPIDER_CHECK(norm_w == nullptr, "gemv: fused RMSNorm needs batch*K*2 <= 64 KiB");
    SPIDER_CHECK((size_t)NB * K * 2 <= 64 * 1024 || norm_w == nullptr, "gemv: fused RMSNorm needs batch*K*2 <= 64 KiB");
    // rows per wave: 1 for the fused gate/up form (2 weight rows per output) and for small matrices (more blocks in
    // flight hide the per-block activation prologue), 2 otherwise
    static const int hoist = [] { const char* e = getenv("SPIDER_GEMV_HOIST"); return e ? atoi(e) : 1; }();
    int R = GU ? 1 : 2;
    if (!GU && (size_t)N * K * 2 < (size_t)48 << 20) R = 1;
    if (gemv_env_r()) R = gemv_env_r();
    if (GU && R > 2) R = 2;
    if (NB > 4 && R > 2) R = 2;
    const int grid = (N + 4 * R - 1) / (4 * R);
    static const int wide_env = [] { const char* e = getenv("SPIDER_GEMV_WIDE8"); return e ? atoi(e) : -1; }();
    const bool wide8 = wide_env >= 0 ? wide_env != 0 : (!GU && K >= 8192 && NB == 1);
#define GEMV_PICK(R_)                                   \
    do {                                                \
        if (xlds) GEMV_LAUNCH(NB, R_, GU, true);        \
        else GEMV_LAUNCH(NB, R_, GU, false);            \
    } while (0)
    if (R == 1) GEMV_PICK(1);
    else if (R == 2) GEMV_PICK(2);
    else GEMV_PICK(4);
#undef GEMV_PICK
    SPIDER_LAUNCH_OK();
    return 0;
}

template <bool GU>
static int gemv_batch(const void* W, const void* x, void* out, const void* bias, const void* res, const void* norm_w,
                      float eps, int B, int N, int K, void* stream) {
    // 5..16 sequences: skinny MFMA GEMM (weights streamed once whatever the batch); needs K % 64 == 0 and no fused norm.
    // Measured crossover on MI355X (scripts/prof_decode_batch.py): the dot2 FMA kernel with its fused RMSNorm wins up to
    // 4 sequences (3.5 / 3.9 / 4.1 ms per step at B = 2 / 3 / 4 vs 4.5 / 4.5 / 4.7), MFMA wins at 8 (5.5 vs 9.5 ms).
    static const int mfma_min_b = [] { const char* e = getenv("SPIDER_GEMV_MFMA_MIN_B"); return e ? atoi(e) : 5; }();
    if (B >= mfma_min_b && B <= 16 && K % 64 == 0 && norm_w == nullptr) {
        // 8 waves per block (each 1/8 of K): N/16 blocks alone would leave most CUs with a single wave of loads in flight
        skinny_gemm_kernel<GU, 8><<<(N + 15) / 16, 512, 0, (hipStream_t)stream>>>((const bf16_t*)W, (const bf16_t*)x, (bf16_t*)out,
                                                                                  (const bf16_t*)bias, (const bf16_t*)res, B, N, K);
        SPIDER_LAUNCH_OK();
        return 0;
    }
    switch (B) {
        case 1: return gemv_dispatch<1, GU>(W, x, out, bias, res, norm_w, eps, N, K, stream);
        case 2: return gemv_dispatch<2, GU>(W, x, out, bias, res, norm_w, eps, N, K, stream);
        case 3: return gemv_dispatch<3, GU>(W, x, out, bias, res, norm_w, eps, N, K, stream);
        case 4: return gemv_dispatch<4, GU>(W, x, out, bias, res, norm_w, eps, N, K, stream);
        case 5: return gemv_dispatch<5, GU>(W, x, out, bias, res, norm_w, eps, N, K, stream);
        case 6: return gemv_dispatch<6, GU>(W, x, out, bias, res, norm_w, eps, N, K, stream);
        case 7: return gemv_dispatch<7, GU>(W, x, out, bias, res, norm_w, eps, N, K, stream);
        case 8: return gemv_dispatch<8, GU>(W, x, out, bias, res, norm_w, eps, N, K, stream);
        default: spider_set_error("gemv: batch must be 1..8 (1..16 when K % 64 == 0 and norm_w == NULL; use spider_gemm_bf16 beyond that)"); return -1;
    }
}


#define SPIDER_ATTN_INLINE_DEFAULT 0
// split-KV combine form of the fused decode attention: -1 = not chosen yet (SPIDER_ATTN_INLINE is read once, at the first call)
static int g_attn_inline = -1;

// ==============================================================================================
// C ABI
// ==============================================================================================
extern "C" {

// 1 / 2: the split-KV combine of spider_attn_decode_fused_bf16 is done by the last-arriving block of the attention launch itself (1: acq_rel
// ticket; 2: write-through partials, relaxed ticket, sc1 loads -- no fence); 0: by the separate combine launch; returns the previous
// setting (-1: environment not read yet)
int spider_set_attn_inline(int on) {
    const int prev = g_attn_inline;
    g_attn_inline = on < 0 ? 0 : (on > 2 ? 2 : on);
    return prev;
}

int spider_embed_bf16(const void* table, const int* ids, void* out, int rows, int H, int V, void* stream) {
    SPIDER_CHECK(rows > 0 && H > 0 && H % 8 == 0 && V > 0, "embed: bad shape (H must be a multiple of 8)");
    embed_kernel<<<rows, 256, 0, (hipStream_t)stream>>>((const bf16_t*)table, ids, (bf16_t*)out, H, V);
    SPIDER_LAUNCH_OK();
    return 0;
}

// x: bf16 [rows, row_stride] with heads*d contiguous elements at the start of each row (a q or k column slice of a fused
// projection output); cos_sin: fp32 [rows, d] = [cos(d/2) | sin(d/2)] per row. In place.
int spider_rope_rows_bf16(void* x, const float* cos_sin, long row_stride, int rows, int heads, int d, void* stream) {
    SPIDER_CHECK(rows > 0 && heads > 0 && d > 0 && d % 16 == 0, "rope_rows: head_dim must be a multiple of 16");
    SPIDER_CHECK(row_stride % 8 == 0 && row_stride >= (long)heads * d, "rope_rows: row stride must be >= heads*d and keep 16-byte alignment");
    const long total = (long)rows * heads * (d / 16);
    long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    rope_rows_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>((bf16_t*)x, cos_sin, row_stride, rows, heads, d);
    SPIDER_LAUNCH_OK();
    return 0;
}

int spider_rmsnorm_bf16(const void* x, const void* res, const void* w, void* y, void* res_out, int rows, int H,
                        float eps, void* stream) {
    SPIDER_CHECK(rows > 0 && H > 0 && H % 8 == 0 && H <= 8192, "rmsnorm: H must be a multiple of 8 and <= 8192");
    rmsnorm_kernel<<<rows, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, (const bf16_t*)res, (const bf16_t*)w,
                                                          (bf16_t*)y, (bf16_t*)res_out, H, eps);
    SPIDER_LAUNCH_OK();
    return 0;
}

int spider_gemv_bf16(const void* W, const void* x, void* out, const void* bias, const void* res, const void* norm_w,
                     float eps, int B, int N, int K, void* stream) {
    SPIDER_CHECK(N > 0 && K > 0 && K % 8 == 0, "gemv: K must be a positive multiple of 8");
    return gemv_batch<false>(W, x, out, bias, res, norm_w, eps, B, N, K, stream);
}

int spider_gemv_swiglu_bf16(const void* W_gate_up, const void* x, void* out, const void* norm_w, float eps, int B,
                            int I, int K, void* stream) {
    SPIDER_CHECK(I > 0 && K > 0 && K % 8 == 0, "gemv_swiglu: K must be a positive multiple of 8");
    return gemv_batch<true>(W_gate_up, x, out, nullptr, nullptr, norm_w, eps, B, I, K, stream);
}

#define LMHEAD_LAUNCH(NB_)                                                                                     \
    lmhead_partial_kernel<NB_, 2><<<nparts, 256, (size_t)NB_ * K * 2, (hipStream_t)stream>>>(                   \
        (const bf16_t*)W, (const bf16_t*)x, (const bf16_t*)norm_w, eps, (float*)ws_val, (int*)ws_idx,          \
        (bf16_t*)logits, V, K)

// cur_ids <- next_ids; pos, slot, kv_end += 1; hist[b, n_hist[b]++] = next_ids[b] (hist / n_hist may be NULL). All int32 [B].
int spider_decode_advance_i32(const int* next_ids, int* cur_ids, int* pos, int* slot, int* kv_end, int* hist, int* n_hist,
                              int cap, int B, void* stream) {
    SPIDER_CHECK(next_ids && cur_ids && pos && slot && kv_end && B > 0, "decode_advance: cursors required");
    SPIDER_CHECK(!hist || (n_hist && cap > 0), "decode_advance: history needs its counters and capacity");
    decode_advance_kernel<<<(B + 63) / 64, 64, 0, (hipStream_t)stream>>>(next_ids, cur_ids, pos, slot, kv_end, hist, n_hist, cap, B);
    SPIDER_LAUNCH_OK();
    return 0;
}

int spider_lm_head_nparts(int V) {
    int n = (V + 63) / 64;  // 64 rows per block
    return n < 2048 ? (n < 1 ? 1 : n) : 2048;
}

int spider_lm_head_argmax_bf16(const void* W, const void* x, const void* norm_w, float eps, int* out_ids, void* logits,
                               void* ws_val, void* ws_idx, int B, int V, int K, void* stream) {
    SPIDER_CHECK(B >= 1 && B <= 8, "lm_head_argmax: batch must be 1..8");
    SPIDER_CHECK(V > 0 && K > 0 && K % 8 == 0 && (size_t)B * K * 2 <= 64 * 1024, "lm_head_argmax: bad K");
    const int nparts = spider_lm_head_nparts(V);
    switch (B) {
        case 1: LMHEAD_LAUNCH(1); break;
        case 2: LMHEAD_LAUNCH(2); break;
        case 3: LMHEAD_LAUNCH(3); break;
        case 4: LMHEAD_LAUNCH(4); break;
        case 5: LMHEAD_LAUNCH(5); break;
        case 6: LMHEAD_LAUNCH(6); break;
        case 7: LMHEAD_LAUNCH(7); break;
        default: LMHEAD_LAUNCH(8); break;
    }
    SPIDER_LAUNCH_OK();
    argmax_final_kernel<<<B, 256, 0, (hipStream_t)stream>>>((const float*)ws_val, (const int*)ws_idx, out_ids, nparts);
    SPIDER_LAUNCH_OK();
    return 0;
}

// Fragment-major forms (see skinny_fm_kernel): Wfm is the repacked copy of W [N, K] (K % 64 == 0; rows padded to a multiple
// of 16 with zeros; the gate/up form packs [gate | up] with I % 16 == 0). 1 <= B <= 16. fold_rmsnorm = 0: x is used as is;
// 1: x is the un-normalised residual stream, Wfm was repacked from W * diag(norm_w), and the kernel applies
// rsqrt(mean(x^2) + eps) per sequence in its epilogue (RMSNorm folded into the projection).
int spider_gemv_fm_bf16(const void* Wfm, const void* x, void* out, const void* bias, const void* res, int B, int N, int K,
                        int fold_rmsnorm, float eps, void* stream) {
    SPIDER_CHECK(B >= 1 && B <= 16 && N > 0 && K > 0 && K % 64 == 0, "gemv_fm: 1 <= B <= 16, K a positive multiple of 64");
    const int NG = (N + 15) / 16;
    const size_t wb = (size_t)NG * 16 * K * 2, xb = (size_t)B * K * 2;
    SPIDER_CHECK(wb < ((size_t)1 << 32), "gemv_fm: weight copy must be < 4 GiB");
#define FM_ARGS(OUT_, BIAS_, RES_, PV_, PI_, N_) (const bf16_t*)Wfm, (const bf16_t*)x, (bf16_t*)(OUT_), (const bf16_t*)(BIAS_), (const bf16_t*)(RES_), \
        (float*)(PV_), (int*)(PI_), B, N_, K, NG, (uint32_t)wb, (uint32_t)xb, eps
    if (fold_rmsnorm) skinny_fm_kernel<0, 8, true><<<NG, 512, 0, (hipStream_t)stream>>>(FM_ARGS(out, bias, res, nullptr, nullptr, N));
    else skinny_fm_kernel<0, 8, false><<<NG, 512, 0, (hipStream_t)stream>>>(FM_ARGS(out, bias, res, nullptr, nullptr, N));
    SPIDER_LAUNCH_OK();
    return 0;
}

int spider_gemv_swiglu_fm_bf16(const void* Wfm, const void* x, void* out, int B, int I, int K, int fold_rmsnorm, float eps,
                               void* stream) {
    SPIDER_CHECK(B >= 1 && B <= 16 && I > 0 && I % 16 == 0 && K > 0 && K % 64 == 0,
                 "gemv_swiglu_fm: 1 <= B <= 16, I a multiple of 16, K a positive multiple of 64");
    const int NG = I / 16;
    const size_t wb = (size_t)2 * I * K * 2, xb = (size_t)B * K * 2;
    SPIDER_CHECK(wb < ((size_t)1 << 32), "gemv_swiglu_fm: weight copy must be < 4 GiB");
    if (fold_rmsnorm) skinny_fm_kernel<1, 8, true><<<NG, 512, 0, (hipStream_t)stream>>>(FM_ARGS(out, nullptr, nullptr, nullptr, nullptr, I));
    else skinny_fm_kernel<1, 8, false><<<NG, 512, 0, (hipStream_t)stream>>>(FM_ARGS(out, nullptr, nullptr, nullptr, nullptr, I));
    SPIDER_LAUNCH_OK();
    return 0;
}

// Batched lm_head + greedy argmax on the fragment-major copy: x [B, K] is the NORMALISED last hidden state (1 <= B <= 16),
// logits (optional) [B, V] bf16. ws_val / ws_idx: >= B * spider_lm_head_nparts(V) entries.
int spider_lm_head_argmax_fm_bf16(const void* Wfm, const void* x, int* out_ids, void* logits, void* ws_val, void* ws_idx, int B,
                                  int V, int K, int fold_rmsnorm, float eps, void* stream) {
    SPIDER_CHECK(B >= 1 && B <= 16 && V > 0 && K > 0 && K % 64 == 0, "lm_head_argmax_fm: 1 <= B <= 16, K a positive multiple of 64");
    const int NG = (V + 15) / 16;
    const size_t wb = (size_t)NG * 16 * K * 2, xb = (size_t)B * K * 2;
    SPIDER_CHECK(wb < ((size_t)1 << 32), "lm_head_argmax_fm: weight copy must be < 4 GiB");
    int nparts = spider_lm_head_nparts(V);
    if (nparts > NG) nparts = NG;
    if (fold_rmsnorm) skinny_fm_kernel<2, 8, true><<<nparts, 512, 0, (hipStream_t)stream>>>(FM_ARGS(logits, nullptr, nullptr, ws_val, ws_idx, V));
    else skinny_fm_kernel<2, 8, false><<<nparts, 512, 0, (hipStream_t)stream>>>(FM_ARGS(logits, nullptr, nullptr, ws_val, ws_idx, V));
#undef FM_ARGS
    SPIDER_LAUNCH_OK();
    argmax_final_kernel<<<B, 256, 0, (hipStream_t)stream>>>((const float*)ws_val, (const int*)ws_idx, out_ids, nparts);
    SPIDER_LAUNCH_OK();
    return 0;
}

// pos3 [3, B*S] (temporal, height, width position of every token); sec_t / sec_h: rotary pairs of the half head dim that
// follow the temporal / height component, the remaining d/2 - sec_t - sec_h follow the width component (Qwen2.5-Omni:
// 16 / 24 / 24). Text tokens carry three equal components, for which this is exactly spider_rope_kv_append_bf16.
int spider_rope_kv_append_mrope_bf16(const void* qkv, const int* pos3, const int* slot, const float* cos_sin, void* q_out,
                                     void* k_cache, void* v_cache, int B, int S, int n_q, int n_kv, int d, int T_max,
                                     int sec_t, int sec_h, void* stream) {
    SPIDER_CHECK(B > 0 && S > 0 && n_q > 0 && n_kv > 0 && T_max > 0, "rope_kv_append: bad shape");
    SPIDER_CHECK(d == 128 || d == 64, "rope_kv_append: head_dim must be 64 or 128");
    SPIDER_CHECK(sec_t >= 0 && sec_h >= 0 && sec_t + sec_h <= d / 2 && (sec_t + sec_h > 0 || (sec_t == 0 && sec_h == 0)),
                 "rope_kv_append: mrope sections must fit the half head dim");
    const int s0 = sec_t, s1 = sec_t + sec_h;    // s1 == 0: plain 1-D RoPE, pos3 is then [B*S]
    if (d == 128)
        rope_kv_kernel<128><<<B * S, 256, 0, (hipStream_t)stream>>>((const bf16_t*)qkv, pos3, slot, cos_sin, (bf16_t*)q_out,
                                                                    (bf16_t*)k_cache, (bf16_t*)v_cache, S, n_q, n_kv, T_max, s0, s1);
    else
        rope_kv_kernel<64><<<B * S, 256, 0, (hipStream_t)stream>>>((const bf16_t*)qkv, pos3, slot, cos_sin, (bf16_t*)q_out,
                                                                   (bf16_t*)k_cache, (bf16_t*)v_cache, S, n_q, n_kv, T_max, s0, s1);
    SPIDER_LAUNCH_OK();
    return 0;
}

int spider_rope_kv_append_bf16(const void* qkv, const int* pos, const int* slot, const float* cos_sin, void* q_out,
                               void* k_cache, void* v_cache, int B, int S, int n_q, int n_kv, int d, int T_max,
                               void* stream) {
    return spider_rope_kv_append_mrope_bf16(qkv, pos, slot, cos_sin, q_out, k_cache, v_cache, B, S, n_q, n_kv, d, T_max, 0, 0,
                                            stream);
}

#define ATTN_DEC_LAUNCH(G_)                                                                                     \
    attn_decode_kernel<128, G_><<<grid, 256, 0, (hipStream_t)stream>>>(                                         \
        (const bf16_t*)q, (const bf16_t*)k_cache, (const bf16_t*)v_cache, kv_beg, kv_end, (float*)ws_o,          \
        (float*)ws_ml, (bf16_t*)out, n_kv, T_max, scale, nsplit)

// workspace: ws_o >= B*n_q*nsplit*d floats, ws_ml >= B*n_q*nsplit*2 floats (unused when nsplit == 1)
int spider_attn_decode_bf16(const void* q, const void* k_cache, const void* v_cache, const int* kv_beg,
                            const int* kv_end, void* out, void* ws_o, void* ws_ml, int B, int n_q, int n_kv, int d,
                            int T_max, float scale, int nsplit, void* stream) {
    SPIDER_CHECK(d == 128, "attn_decode: head_dim must be 128");
    SPIDER_CHECK(B > 0 && n_kv > 0 && n_q % n_kv == 0 && nsplit >= 1 && T_max > 0, "attn_decode: bad shape");
    SPIDER_CHECK(nsplit == 1 || (ws_o && ws_ml), "attn_decode: workspace required for nsplit > 1");
    const int G = n_q / n_kv;
    dim3 grid(nsplit, n_kv, B);
    switch (G) {
        case 1: ATTN_DEC_LAUNCH(1); break;
        case 2: ATTN_DEC_LAUNCH(2); break;
        case 3: ATTN_DEC_LAUNCH(3); break;      // (Llama-3.2-3B: 24 / 8)
        case 4: ATTN_DEC_LAUNCH(4); break;      // Llama-3-8B, DeepSeek-R1-Distill-Llama-8B: 32 / 8
        case 5: ATTN_DEC_LAUNCH(5); break;      // (Qwen2.5-14B / 32B: 40 / 8)
        case 6: ATTN_DEC_LAUNCH(6); break;
        case 7: ATTN_DEC_LAUNCH(7); break;      // Qwen2.5-Omni-7B thinker: 28 / 4
        case 8: ATTN_DEC_LAUNCH(8); break;      // Qwen2.5-Omni-3B thinker: 16 / 2
        default: spider_set_error("attn_decode: GQA group size must be 1 ... 8"); return -1;
    }
    SPIDER_LAUNCH_OK();
    if (nsplit > 1) {
        attn_combine_kernel<128><<<B * n_q, 256, 0, (hipStream_t)stream>>>((const float*)ws_o, (const float*)ws_ml,
                                                                          (bf16_t*)out, nsplit);
        SPIDER_LAUNCH_OK();
    }
    return 0;
}

#define ATTN_FUSED_ARGS                                                                                         \
    (const bf16_t*)qkv, pos, cos_sin, (bf16_t*)k_cache, (bf16_t*)v_cache, kv_beg, kv_end, (float*)ws_o, (float*)ws_ml, counters, \
        (bf16_t*)out, n_kv, T_max, scale, nsplit, inline_combine
// wide: 8 waves x 8 rows per lane group = 256 cache rows in flight per block (splits of >= ~100 rows); narrow: 4 x 4 = 64
#define ATTN_FUSED_LAUNCH(G_)                                                                                   \
    do {                                                                                                        \
        if (wide) attn_decode_fused_kernel<128, G_, 8, 8><<<grid, 512, 0, (hipStream_t)stream>>>(ATTN_FUSED_ARGS); \
        else attn_decode_fused_kernel<128, G_, 4, 4><<<grid, 256, 0, (hipStream_t)stream>>>(ATTN_FUSED_ARGS);   \
    } while (0)

// Fused decode attention (RoPE + KV append of the current token + split-KV attention + combine), one launch.
// kv_end[b] INCLUDES the current token (its slot is kv_end[b]-1). counters: int[B*n_kv], zero before first use.
int spider_attn_decode_fused_bf16(const void* qkv, const int* pos, const float* cos_sin, void* k_cache, void* v_cache,
                                  const int* kv_beg, const int* kv_end, void* out, void* ws_o, void* ws_ml,
                                  int* counters, int B, int n_q, int n_kv, int d, int T_max, float scale, int nsplit,
                                  void* stream) {
    SPIDER_CHECK(d == 128, "attn_decode_fused: head_dim must be 128");
    SPIDER_CHECK(B > 0 && n_kv > 0 && n_q % n_kv == 0 && nsplit >= 1 && T_max > 0, "attn_decode_fused: bad shape");
    SPIDER_CHECK(nsplit == 1 || (ws_o && ws_ml && counters), "attn_decode_fused: workspace + counters required for nsplit > 1");
    const int G = n_q / n_kv;
    dim3 grid(nsplit, n_kv, B);
    // combine inline (ticket + last-arriver reduce) or by the separate combine kernel (SPIDER_ATTN_INLINE=0/1)
    // split-KV combine by the last-arriving block of the same launch (1) or by attn_combine_kernel (0); read per call so that
    // tests can exercise both forms in one process
    // (the environment is read once per process; spider_set_attn_inline() switches the form afterwards: tests, tuning)
    if (g_attn_inline < 0) { const char* e = getenv("SPIDER_ATTN_INLINE"); g_attn_inline = e ? (atoi(e) < 0 ? 0 : (atoi(e) > 2 ? 2 : atoi(e))) : SPIDER_ATTN_INLINE_DEFAULT; }
    const int inline_combine = g_attn_inline;
    static const int wide_env = [] { const char* e = getenv("SPIDER_ATTN_WIDE"); return e ? atoi(e) : -1; }();
    const bool wide = wide_env >= 0 ? wide_env != 0 : (T_max / nsplit >= 96);
    switch (G) {
        case 1: ATTN_FUSED_LAUNCH(1); break;
        case 2: ATTN_FUSED_LAUNCH(2); break;
        case 3: ATTN_FUSED_LAUNCH(3); break;
        case 4: ATTN_FUSED_LAUNCH(4); break;
        case 5: ATTN_FUSED_LAUNCH(5); break;
        case 6: ATTN_FUSED_LAUNCH(6); break;
        case 7: ATTN_FUSED_LAUNCH(7); break;
        case 8: ATTN_FUSED_LAUNCH(8); break;
        default: spider_set_error("attn_decode_fused: GQA group size must be 1 ... 8"); return -1;
    }
    SPIDER_LAUNCH_OK();
    if (nsplit > 1 && !inline_combine) {
        attn_combine_kernel<128><<<B * n_q, 256, 0, (hipStream_t)stream>>>((const float*)ws_o, (const float*)ws_ml,
                                                                          (bf16_t*)out, nsplit);
        SPIDER_LAUNCH_OK();
    }
    return 0;
}

}  // extern "C"
