// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )   (both operands K-contiguous)
//
// One kernel family serves
//   * LLM prefill projections            (modeling_llama3.py:186-199,260-313 at S = prompt length)
//   * UNet / text-encoder / VAE linears  (diffusers-0.25 Attention.to_q/k/v/out, FeedForward, proj_in/out)
//   * conv2d as implicit GEMM on NHWC    (ResnetBlock2D conv1/conv2, Down/Upsample2D, conv_shortcut;
//                                         call sites custom_sd.py:634-639 -> UNet2DConditionModel.forward)
// Tiles: 128x128x64 (large problems) or 64x64x64 (small M*N), 256 threads = 4 waves (2x2); each wave owns a
// (BM/2)x(BN/2) sub-tile of mfma_f32_16x16x32_bf16 tiles. The UNet at CFG batch 2 has M = 8192 .. 128 rows and
// K up to 23040 (3x3 conv over 2560 channels): grids of a few dozen tiles would leave most of the 256 CUs idle,
// so the host picks the tile and a split-K factor that bring the grid to >= ~2 blocks per CU; split-K partials go
// to an fp32 workspace (plain stores, one slab per split) and a second kernel reduces them and applies the epilogue.
// Operand roles are swapped (W tile is the MFMA "A" operand) so each lane ends up with 4 consecutive output
// columns of one row -> 8-byte epilogue loads/stores. LDS rows are 128 B with an XOR chunk swizzle (conflict-free
// ds_read_b128 / ds_write_b128), double-buffered, with a 2-deep global->register prefetch ring.
#include "common.hpp"
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>

using namespace spider;

namespace {

constexpr int BK = 64;
constexpr int LDS_STRIDE = BK;      // elements; 128-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7
// LDS image: chunk c (16 B) of row r lives at r*128 + ((c ^ ((r >> 1) & 7)) * 16). A ds_read_b128 serves 16-lane groups
// that MIX two k-chunks of the 16x16x32 fragment (lanes {0-3,12-15} of chunk g with lanes {20-27} of chunk g+1), so
// row padding alone leaves 2-way conflicts (measured: SQ_LDS_BANK_CONFLICT = 33 % of SQ_LDS_IDX_ACTIVE); with this
// swizzle every group hits 16 distinct 16-byte slots, and so do the 8-lane groups of the ds_write_b128 stores.

struct GemmArgs {
    const h16_t* A;
    const h16_t* W;
    h16_t* C;
    const h16_t* bias;     // [N] or null
    const h16_t* res;      // [M, ldc] or null (added after rounding the GEMM result to bf16)
    const h16_t* rowbias;  // [M / rows_per_group, N] or null (time-embedding add)
    float* C32;             // optional fp32 output instead of bf16
    // fp32 residual stream: res32 [M, ldc] fp32 is added to the UNROUNDED fp32 GEMM result (instead of `res`,
    // which is added after rounding the result to 16 bits as the reference's separate ops do), and c32d [M, ldc] fp32 receives
    // the fp32 value that C gets rounded: the residual chain x <- x + f(x) is then carried in fp32 (one rounding per READ of the
    // 16-bit shadow C instead of one accumulating rounding per add). Either may be null.
    const float* res32;
    float* c32d;
    float* ws;              // split-K workspace [splits, M, N] fp32
    int M, N, K, lda, ldc;
    uint32_t a_bytes, w_bytes;   // buffer-descriptor ranges of A and W (< 4 GiB each)
    uint32_t c_bytes, rb_bytes;  // ranges of C / res ([M, ldc] bf16) and rowbias, for the branch-free epilogue (0: not usable)
    int rows_per_group;
    int act;                // 0 none, 1 silu, 2 gelu(erf), 3 quick-gelu, 5 leaky-relu(act_param), 6 relu, 7 tanh
    float act_param;
    int geglu;              // 1: W = [value rows (N) | gate rows (N)], out[m,n] = bf16(v) * bf16(gelu(bf16(g)))  (diffusers GEGLU)
                            // 2: W = [gate rows (N) | up rows (N)],    out[m,n] = bf16(silu(bf16(g))) * bf16(u)  (LlamaMLP's SwiGLU, prefill)
    float out_scale;        // multiplies the final value (1/rescale_output_factor)
    int splits, kt_per_split;
    // implicit-GEMM conv (NHWC): A is the image [B, Hin, Win, Cin]; K = kh*kw*Cin (tap-major, channel-minor).
    // ups: the conv reads a virtual nearest-2x upsampled image cropped to lim_h x lim_w (= 2*Hin or 2*Hin-1, ...);
    // without ups lim_h/lim_w = Hin/Win. cin64: a 64-wide K tile never straddles two taps (uniform tap per tile).
    int conv, Hin, Win, Cin, Hout, Wout, kh, kw, stride, pad_h, pad_w, dil, ups, lim_h, lim_w, cin64;
    // kcm (with cin64): the K tiles are walked CHANNEL-BLOCK major, tap minor, instead of in the weight's own (tap, channel) order:
    // the kh*kw consecutive K tiles of one 64-channel block read the same input rows shifted by the tap, so all but the first of
    // them hit in L2 (each XCD keeps ~50 KB per resident tile instead of cycling through the whole Cin-wide rows nine times:
    // FETCH_SIZE of the 64^2 3x3 convs was 2.5x their algorithmic bytes, more on the UNet3D / SDXL maps that exceed the L2).
    // The weight tile of a step is the walk's (tap, channel block): same products, another fp32 summation order.
    int kcm;
    // hbits (with cin64, no upsample, kh*kw <= 32): each A piece carries a bitmask "tap t reads outside the image for this
    // pixel", built once per block, so the per-K-tile address of a DMA piece is base + (wave-uniform tap offset) | masks --
    // 5 vector instructions instead of ~25 in the issue slot between two barriers of the LDS-DMA kernels.
    int hbits;
    int dbg;   // tuning aid (SPIDER_GEMM_DBG): 1 = DMA only, 2 = compute only (results are garbage)
    int epi_lds;   // LDS-DMA kernel: the block's output tile goes through LDS and leaves in row order (epilogue_lds below)
    // tile order: 0 = m fastest (neighbouring blocks share a W tile: LLM prefill, W >> A), 1 = n fastest (they share the A tile:
    // the UNet's 8192-row activations against 320..2560 output columns -- with m fastest every column tile re-streamed all of A
    // through its XCD's 4 MiB L2: FETCH_SIZE 58 MB per GEGLU projection whose operands are 7 MB)
    int n_fast;
    // W layout: 0 = row-major [N, K] (nn.Linear / OHWI conv weights as stored); 1 = TILE-MAJOR copy [ceil(N/64)][ceil(K/64)][64 rows][64 k]
    // (8 KiB per tile, zero-padded): a K tile of 64 weight rows is ONE contiguous 8 KiB read instead of 64 pieces of 128 B at a
    // row stride of 2 K bytes. The weight-streaming problems of the UNet (16^2 / 8^2 maps: M = 512 / 128 rows against 3 - 59 MB of
    // weights, every byte used once per step and cold in HBM) ran at ~2 TB/s on the strided form -- the same DRAM-granularity
    // limit the decode path's fragment-major weights removed.
    int w_tiled;
    // LayerNorm folded into the GEMM (LN instantiations): C = rstd[m] * (A.W'^T - mean[m] * colsum[n]) + colbias[n], with
    // W' = W * diag(gamma) (folded by the caller), colsum[n] = sum_k W'[n,k], colbias[n] = sum_k beta[k] W[n,k] + bias[n];
    // the row statistics of A (K = the whole normalised row) are accumulated by the block itself while it stages A.
    const float* ln_colsum;
    const float* ln_bias;
    float ln_eps;
    // the 256^2 LDS-DMA kernel never sees A in registers: its LN form reads {mean, rstd} per row from this array, filled by
    // ln_row_stats_kernel just before the launch (padded to a multiple of 256 rows)
    const float2* ln_rows;
    // GroupNorm statistics of the OUTPUT, written by the producer (round 4): gn_part [M / gn_cr, gn_G, 2] fp32 holds, per chunk of
    // gn_cr consecutive output rows (a chunk never straddles two images: HW % gn_cr == 0) and per group of gn_cpg = N / gn_G
    // channels, (sum, sum of squares) of the 16-bit values C receives -- the `partial` array gn_apply_kernel reduces in fixed order,
    // so the consumer GroupNorm needs no statistics pass of its own (deterministic: no atomics, graph replay == eager).
    float* gn_part;
    int gn_G, gn_cpg, gn_cr;
    // GroupNorm applied to the A operand on its way into LDS (register-staged kernel, plain linear): A is the UN-normalised
    // [M, K] tensor, gna_part [M / HW, gna_nchunk, gna_G, 2] its partial statistics, and the block computes a_c = rstd_g gamma_c,
    // b_c = beta_c - mean_g a_c for its image once and stores round16(fma(x, a_c, b_c)) -- the values gn_apply_kernel would have
    // written (Transformer2DModel.norm + proj_in in one launch).
    const float* gna_part;
    const h16_t* gna_gamma;
    const h16_t* gna_beta;
    int gna_nchunk, gna_G, gna_hw;
    float gna_eps;
    // weight-stationary streaming conv (wstream_kernel, w_tiled == 2): ws_ncb = Cin / 32 channel blocks, ws_cpb of them per K split
    // (blockIdx.y), ws_rows = rows of the activation slab (input pixels B * Hin * Win); ws_cnt: per-strip arrival counters of the
    // in-launch split-K combine (zero when idle; null: partial slabs are left to splitk_reduce_kernel)
    int ws_ncb, ws_cpb, ws_rows;
    unsigned* ws_cnt;
    // "precise" operand: A is fp32 ([M, lda] floats, or the NHWC fp32 image of a conv) and is split into hi + lo 16-bit halves inside
    // the register-staged kernel (gemm_kernel<..., A32>); a_bytes counts fp32 bytes
    int a32;
};

__device__ __forceinline__ float apply_act(const GemmArgs& p, float v) {
    if (p.act == 1) return silu_f(v);
    if (p.act == 2) return gelu_erf_f(v);
    if (p.act == 3) return quick_gelu_f(v);
    if (p.act == 5) return v > 0.f ? v : v * p.act_param;
    if (p.act == 6) return fmaxf(v, 0.f);
    if (p.act == 7) return tanhf(v);
    return v;
}

// byte offset of row n's first K tile in W, and the byte step from one K tile to the next (see GemmArgs::w_tiled)
__device__ __forceinline__ uint32_t w_row_byte(const GemmArgs& p, int n) {
    const uint32_t nk = (uint32_t)((p.K + BK - 1) / BK);
    return p.w_tiled ? ((uint32_t)(n >> 6) * nk * 8192u + (uint32_t)(n & 63) * 128u) : (uint32_t)n * (uint32_t)p.K * 2u;
}
__device__ __forceinline__ uint32_t w_tile_step(const GemmArgs& p) { return p.w_tiled ? 8192u : (uint32_t)(BK * 2); }

// Walk over the K tiles of an implicit-GEMM conv whose 64-wide K tiles never straddle taps (Cin % 64 == 0): the (channel, tap
// row, tap column) of the NEXT tile are carried along instead of being re-derived by integer divisions per tile. Order: see
// GemmArgs::kcm. wtile() = index of the tile in the weight's K order (tap-major, channel-minor).
struct TapWalk {
    int c, ky, kx;
    __device__ __forceinline__ void init(const GemmArgs& p, int kt0) {
        if (p.kcm) {
            const int nt = p.kh * p.kw, cb = kt0 / nt, tap = kt0 - cb * nt;
            c = cb * BK; ky = tap / p.kw; kx = tap - ky * p.kw;
        } else {
            const int kk0 = kt0 * BK, tap0 = kk0 / p.Cin;
            c = kk0 - tap0 * p.Cin; ky = tap0 / p.kw; kx = tap0 - ky * p.kw;
        }
    }
    __device__ __forceinline__ void next(const GemmArgs& p) {
        if (p.kcm) {
            if (++kx == p.kw) { kx = 0; if (++ky == p.kh) { ky = 0; c += BK; } }
        } else {
            c += BK;
            if (c >= p.Cin) { c = 0; if (++kx == p.kw) { kx = 0; ++ky; } }
        }
    }
    __device__ __forceinline__ uint32_t wtile(const GemmArgs& p) const { return (uint32_t)((ky * p.kw + kx) * (p.Cin >> 6) + (c >> 6)); }
};

// bias / rowbias / activation / residual / scale on 4 consecutive columns of one row, then store
// EPI selects what is compiled in: 0 = bias / rowbias / residual / scale only, 1 = + activation, 2 = GEGLU (kernel-level).
// One kernel carrying every epilogue was > 50 KB of code (the transcendental bodies inlined per output fragment), i.e.
// most of the instruction cache two CUs share, for paths the UNet's convs and projections never take.
template <int EPI>
__device__ __forceinline__ void epilogue_store(const GemmArgs& p, int m, int n, float v[4]) {
    constexpr bool ACT = EPI == 1 || EPI == 3;
    const int grp = p.rowbias ? m / p.rows_per_group : 0;
    if (n + 3 < p.N) {
        if (p.bias) {
            const u32x2 bq = *reinterpret_cast<const u32x2*>(p.bias + n);
            v[0] += h16lo_to_f32(bq.x); v[1] += h16hi_to_f32(bq.x);
            v[2] += h16lo_to_f32(bq.y); v[3] += h16hi_to_f32(bq.y);
        }
        if (p.rowbias) {
            const u32x2 bq = *reinterpret_cast<const u32x2*>(p.rowbias + (size_t)grp * p.N + n);
            v[0] += h16lo_to_f32(bq.x); v[1] += h16hi_to_f32(bq.x);
            v[2] += h16lo_to_f32(bq.y); v[3] += h16hi_to_f32(bq.y);
        }
        if (ACT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(p, v[e]);
        }
        if (p.res32) {
            const f32x4 rq = *reinterpret_cast<const f32x4*>(p.res32 + (size_t)m * p.ldc + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rq[e];
        } else if (p.res) {
            const u32x2 rq = *reinterpret_cast<const u32x2*>(p.res + (size_t)m * p.ldc + n);
            v[0] = h16_to_f32(f32_to_h16(v[0])) + h16lo_to_f32(rq.x);
            v[1] = h16_to_f32(f32_to_h16(v[1])) + h16hi_to_f32(rq.x);
            v[2] = h16_to_f32(f32_to_h16(v[2])) + h16lo_to_f32(rq.y);
            v[3] = h16_to_f32(f32_to_h16(v[3])) + h16hi_to_f32(rq.y);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.out_scale;
        if (p.c32d) *reinterpret_cast<f32x4*>(p.c32d + (size_t)m * p.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
        if (p.C32) {
            *reinterpret_cast<f32x4*>(p.C32 + (size_t)m * p.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            u32x2 o;
            o.x = pack_h16x2(v[0], v[1]);
            o.y = pack_h16x2(v[2], v[3]);
            *reinterpret_cast<u32x2*>(p.C + (size_t)m * p.ldc + n) = o;
        }
    } else {
        for (int e = 0; e < 4 && n + e < p.N; ++e) {
            float t = v[e];
            if (p.bias) t += h16_to_f32(p.bias[n + e]);
            if (p.rowbias) t += h16_to_f32(p.rowbias[(size_t)grp * p.N + n + e]);
            if (ACT) t = apply_act(p, t);
            if (p.res32) t += p.res32[(size_t)m * p.ldc + n + e];
            else if (p.res) t = h16_to_f32(f32_to_h16(t)) + h16_to_f32(p.res[(size_t)m * p.ldc + n + e]);
            t *= p.out_scale;
            if (p.c32d) p.c32d[(size_t)m * p.ldc + n + e] = t;
            if (p.C32) p.C32[(size_t)m * p.ldc + n + e] = t;
            else p.C[(size_t)m * p.ldc + n + e] = f32_to_h16(t);
        }
    }
}

// Lean epilogue of the common case (bf16 output, N % 4 == 0, operands < 2 GiB): operands go through buffer descriptors
// with 32-bit offsets; rows / columns outside the problem get an all-ones offset (loads return 0, the store is dropped),
// so there is no divergent control flow and no 64-bit address arithmetic per fragment -- only the scalar
// "is this operand present" branches. (Reading absent operands through zero-range descriptors instead was measured
// slower: 48 useless memory instructions per thread.)
struct EpiRsrc {
    __amdgpu_buffer_rsrc_t bias, rowbias, res, c, res32, c32d;
};

__device__ __forceinline__ EpiRsrc make_epi_rsrc(const GemmArgs& p) {
    EpiRsrc r;
    r.bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.bias), 0, p.bias ? p.N * 2 : 0, 0x00020000);
    r.rowbias = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.rowbias), 0, p.rowbias ? (int)p.rb_bytes : 0, 0x00020000);
    r.res = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.res), 0, p.res ? (int)p.c_bytes : 0, 0x00020000);
    r.c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)p.c_bytes, 0x00020000);
    // fp32 stream operands: twice the byte range of C (host-checked to stay below 2 GiB when they are given)
    r.res32 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res32), 0, p.res32 ? (int)(2 * p.c_bytes) : 0, 0x00020000);
    r.c32d = __builtin_amdgcn_make_buffer_rsrc(p.c32d, 0, p.c32d ? (int)(2 * p.c_bytes) : 0, 0x00020000);
    return r;
}

template <bool ACT>
__device__ __forceinline__ void epilogue_fast(const GemmArgs& p, const EpiRsrc& r, int m, int n, uint32_t rb_row_byte, float v[4]) {
    const uint32_t inv = (uint32_t)(((p.M - 1 - m) | (p.N - 1 - n)) >> 31);       // all ones outside the problem
    const uint32_t off = (((uint32_t)m * (uint32_t)p.ldc + (uint32_t)n) * 2u) | inv;
    const uint32_t noff = ((uint32_t)n * 2u) | inv;
    // operand presence is uniform (scalar branches); a missing operand costs nothing, a present one is one 8-byte load
    if (p.bias) {
        const u32x2 bq = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r.bias, noff, 0, 0));
        v[0] += h16lo_to_f32(bq.x); v[1] += h16hi_to_f32(bq.x); v[2] += h16lo_to_f32(bq.y); v[3] += h16hi_to_f32(bq.y);
    }
    if (p.rowbias) {
        const u32x2 rb = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r.rowbias, (rb_row_byte + (uint32_t)n * 2u) | inv, 0, 0));
        v[0] += h16lo_to_f32(rb.x); v[1] += h16hi_to_f32(rb.x); v[2] += h16lo_to_f32(rb.y); v[3] += h16hi_to_f32(rb.y);
    }
    if (ACT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(p, v[e]);
    }
    const uint32_t off32 = (off << 1) | inv;      // byte offset of the same element in an fp32 [M, ldc] array
    if (p.res32) {   // fp32 residual stream: added to the unrounded result
        const f32x4 rq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r.res32, off32, 0, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rq[e];
    } else if (p.res) {   // the GEMM result is rounded to bf16 before the residual add, as the reference's separate ops do
        const u32x2 rq = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r.res, off, 0, 0));
        v[0] = h16_to_f32(f32_to_h16(v[0])) + h16lo_to_f32(rq.x); v[1] = h16_to_f32(f32_to_h16(v[1])) + h16hi_to_f32(rq.x);
        v[2] = h16_to_f32(f32_to_h16(v[2])) + h16lo_to_f32(rq.y); v[3] = h16_to_f32(f32_to_h16(v[3])) + h16hi_to_f32(rq.y);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= p.out_scale;
    if (p.c32d)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(r.c32d, 0, 0, 0)), f32x4{v[0], v[1], v[2], v[3]}), r.c32d, off32, 0, 0);
    u32x2 o;
    o.x = pack_h16x2(v[0], v[1]);
    o.y = pack_h16x2(v[2], v[3]);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b64(r.c, 0, 0, 0)), o), r.c, off, 0, 0);
}

// 8 consecutive columns per lane (16-byte bias / residual loads and stores): see the tile-pair exchange in write_out
template <bool ACT>
__device__ __forceinline__ void epilogue_fast8(const GemmArgs& p, const EpiRsrc& r, int m, int n, uint32_t rb_row_byte, float v[8]) {
    const uint32_t inv = (uint32_t)(((p.M - 1 - m) | (p.N - 8 - n)) >> 31);       // all ones outside the problem (N % 8 == 0)
    const uint32_t off = (((uint32_t)m * (uint32_t)p.ldc + (uint32_t)n) * 2u) | inv;
    const uint32_t noff = ((uint32_t)n * 2u) | inv;
    auto add8 = [&](const u32x4 q) {
        v[0] += h16lo_to_f32(q.x); v[1] += h16hi_to_f32(q.x); v[2] += h16lo_to_f32(q.y); v[3] += h16hi_to_f32(q.y);
        v[4] += h16lo_to_f32(q.z); v[5] += h16hi_to_f32(q.z); v[6] += h16lo_to_f32(q.w); v[7] += h16hi_to_f32(q.w);
    };
    if (p.bias) add8(__builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r.bias, noff, 0, 0)));
    if (p.rowbias) add8(__builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r.rowbias, (rb_row_byte + (uint32_t)n * 2u) | inv, 0, 0)));
    if (ACT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = apply_act(p, v[e]);
    }
    const uint32_t off32 = (off << 1) | inv;      // byte offset of the same 8 elements in an fp32 [M, ldc] array (two 16-byte halves)
    if (p.res32) {   // fp32 residual stream: added to the unrounded result
        const f32x4 r0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r.res32, off32, 0, 0));
        const f32x4 r1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r.res32, (off32 + 16u) | inv, 0, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
    } else if (p.res) {   // the GEMM result is rounded to bf16 before the residual add, as the reference's separate ops do
        const u32x4 rq = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r.res, off, 0, 0));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = h16_to_f32(f32_to_h16(v[e]));
        add8(rq);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
    if (p.c32d) {
        typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(r.c32d, 0, 0, 0)) vec_t;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vec_t, f32x4{v[0], v[1], v[2], v[3]}), r.c32d, off32, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vec_t, f32x4{v[4], v[5], v[6], v[7]}), r.c32d, (off32 + 16u) | inv, 0, 0);
    }
    u32x4 o;
    o.x = pack_h16x2(v[0], v[1]); o.y = pack_h16x2(v[2], v[3]); o.z = pack_h16x2(v[4], v[5]); o.w = pack_h16x2(v[6], v[7]);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(r.c, 0, 0, 0)), o), r.c, off, 0, 0);
}

// ---- epilogue shared by the GEMM kernels: lane holds C[mb + i*16 + (lane&15)][nb + j*16 + (lane>>4)*4 + 0..3] ----
// Per-column fp32 constants of 4 consecutive columns n .. n+3 of an [N] array (columns past N: the last valid one, never stored).
// A lane's columns do not depend on its row tile i, so the epilogues below fetch them ONCE per wave tile: with the loads inside the
// row loop (one set per 16 x 16 tile) the LayerNorm-folded 92160 x 960 x 320 projection ran 150 us against 94 us for the plain one.
__device__ __forceinline__ void cols4(const float* arr, int n, int N, float o[4]) {
    if (n + 3 < N) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(arr + n);
        o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = arr[n + e < N ? n + e : N - 1];
    }
}
__device__ __forceinline__ void cols4_h16(const h16_t* arr, int n, int N, float o[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = h16_to_f32(arr[n + e < N ? n + e : N - 1]);
}
// LayerNorm fold: row statistics ms = {mean, rstd} (GemmArgs::ln_colsum)
__device__ __forceinline__ void ln_fix(const float2 ms, const float cs[4], const float cb[4], float v[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ms.y * (v[e] - ms.x * cs[e]) + cb[e];
}

// GEGLU epilogue shared by the register-staged and the 256^2 kernel: acc[i][j] is value tile j (j < NTH), acc[i][j + NTH] its gate
// tile; out[m][n] = bf16(value) * bf16(gelu(bf16(gate))) at n = nb + j*16 + (lane>>4)*4 + 0..3 of the [M, N] output (diffusers
// GEGLU: hidden * gelu(gate)). LN: LayerNorm-folded form, row statistics ln_stat[m - m_blk]. Column constants once per wave tile;
// with two value tiles (and N, ldc multiples of 8) the tile-pair exchange of write_out gives 16-byte stores, 64 B per row.
template <int MT, int NTH, bool LN>
__device__ __forceinline__ void geglu_out(const GemmArgs& p, f32x4 (&acc)[MT][2 * NTH], int mb, int nb, int lane,
                                          const float2* ln_stat, int m_blk) {
    const int g = lane >> 4;
    float cv[NTH][4], bv[NTH][4], cg[NTH][4], bg[NTH][4];
#pragma unroll
    for (int j = 0; j < NTH; ++j) {
        const int n = nb + j * 16 + g * 4;
        if (LN) {
            cols4(p.ln_colsum, n, p.N, cv[j]); cols4(p.ln_bias, n, p.N, bv[j]);
            cols4(p.ln_colsum + p.N, n, p.N, cg[j]); cols4(p.ln_bias + p.N, n, p.N, bg[j]);
        } else if (p.bias) {
            cols4_h16(p.bias, n, p.N, bv[j]); cols4_h16(p.bias + p.N, n, p.N, bg[j]);
        }
    }
    const bool wide = NTH == 2 && ((p.N | p.ldc) & 7) == 0;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = mb + i * 16 + (lane & 15);
        float2 ms = float2{0.f, 1.f};
        if (LN) ms = ln_stat[m - m_blk];
        float r[NTH][4];
#pragma unroll
        for (int j = 0; j < NTH; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[i][j][e], gt = acc[i][j + NTH][e];
                if (LN) {
                    v = ms.y * (v - ms.x * cv[j][e]) + bv[j][e];
                    gt = ms.y * (gt - ms.x * cg[j][e]) + bg[j][e];
                } else if (p.bias) { v += bv[j][e]; gt += bg[j][e]; }
                if (p.geglu == 3) {      // GEGLU, "precise" form: the product is rounded once (no 16-bit stops at value / gate / gelu)
                    gt = gelu_erf_f(gt);
                } else if (p.geglu == 2) {      // SwiGLU (LlamaMLP, modeling_llama3.py:197-199): first half = gate rows, second = up rows
                    v = h16_to_f32(f32_to_h16(silu_f(h16_to_f32(f32_to_h16(v)))));
                    gt = h16_to_f32(f32_to_h16(gt));
                } else {
                    v = h16_to_f32(f32_to_h16(v));
                    gt = h16_to_f32(f32_to_h16(gelu_erf_f(h16_to_f32(f32_to_h16(gt)))));
                }
                r[j][e] = v * gt;
            }
        }
        if (wide) {
            float w[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[0][e]), __float_as_uint(r[NTH - 1][e]), false, false);
                w[e] = __uint_as_float(sw[0]);
                w[4 + e] = __uint_as_float(sw[1]);
            }
            const int n = nb + (g & 1) * 16 + 4 * (g & 2);
            if (m < p.M && n < p.N) {
                u32x4 o;
                o.x = pack_h16x2(w[0], w[1]); o.y = pack_h16x2(w[2], w[3]); o.z = pack_h16x2(w[4], w[5]); o.w = pack_h16x2(w[6], w[7]);
                *reinterpret_cast<u32x4*>(p.C + (size_t)m * p.ldc + n) = o;
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < NTH; ++j) {
            const int n = nb + j * 16 + g * 4;
            if (m < p.M && n < p.N) {
                if (n + 3 < p.N) {
                    u32x2 o;
                    o.x = pack_h16x2(r[j][0], r[j][1]);
                    o.y = pack_h16x2(r[j][2], r[j][3]);
                    *reinterpret_cast<u32x2*>(p.C + (size_t)m * p.ldc + n) = o;
                } else {
                    for (int e = 0; e < 4 && n + e < p.N; ++e) p.C[(size_t)m * p.ldc + n + e] = f32_to_h16(r[j][e]);
                }
            }
        }
    }
}

// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), result in every lane: four row rotations
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

// GN (producer statistics, GemmArgs::gn_part): gn_cols points at this wave's [NT * 16] float2 array in LDS; the wave leaves there,
// per output column of its tile, (sum, sum of squares) over its MT * 16 rows of the 16-bit values it stored.
template <int MT, int NT, int EPI, bool LN = false, bool GN = false>
__device__ __forceinline__ void write_out(const GemmArgs& p, f32x4 (&acc)[MT][NT], int mb, int nb, int split, int lane,
                                          const float2* ln_stat = nullptr, int m_blk = 0, float2* gn_cols = nullptr) {
    if (EPI <= 1 && p.splits == 1) {
        const EpiRsrc er = make_epi_rsrc(p);
        float gs[GN ? NT : 1][4], gq[GN ? NT : 1][4];     // GN: column sums of this lane's row (slot j = tile j's 4 columns of this lane)
        if (GN) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { gs[j][e] = 0.f; gq[j][e] = 0.f; }
        }
        auto gn_add = [&](int j, const float* v, float rowok) {     // v: the 4 final fp32 values of slot j, before the 16-bit rounding
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float r = h16_to_f32(f32_to_h16(v[e])) * rowok;
                gs[j][e] += r;
                gq[j][e] = fmaf(r, r, gq[j][e]);
            }
        };
        // Wide form (N, ldc multiples of 8): the 4 lane groups g = lane >> 4 of a row hold 4 columns each of tile j and of tile
        // j + 1. One v_permlane16_swap per accumulator register exchanges [tile j, odd g] <-> [tile j + 1, even g]: afterwards
        // a lane of an even group owns columns 4g .. 4g+7 of tile j and a lane of an odd group columns 4(g-1) .. 4(g-1)+7 of tile
        // j + 1 -- 8 consecutive columns, so bias / residual / output move as 16-byte accesses, 64 contiguous bytes per row and
        // wave instruction instead of 32 (the 8-byte form is store-issue and partial-line bound on the UNet's M x 320 outputs).
        const bool wide = NT >= 2 && ((p.N | p.ldc) & 7) == 0;
        // LayerNorm fold: this lane's column constants, slot j = the 4 columns its values of tile j end up in (after the pair
        // exchange of the wide form: tile j -> columns 0..3, tile j + 1 -> columns 4..7 of the lane's 8)
        float lcs[LN ? NT : 1][4], lcb[LN ? NT : 1][4];
        if (LN) {
            const int g = lane >> 4;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const bool paired = wide && ((j | 1) < NT);
                const int n = paired ? nb + ((j & ~1) + (g & 1)) * 16 + 4 * (g & 2) + 4 * (j & 1) : nb + j * 16 + g * 4;
                cols4(p.ln_colsum, n, p.N, lcs[j]);
                cols4(p.ln_bias, n, p.N, lcb[j]);
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = mb + i * 16 + (lane & 15);
            uint32_t rb_row = 0;
            if (p.rowbias) rb_row = (uint32_t)((m < p.M ? m : 0) / p.rows_per_group) * (uint32_t)p.N * 2u;
            float2 lms = float2{0.f, 1.f};
            if (LN) lms = ln_stat[m - m_blk];
            if (wide) {
                const int g = lane >> 4;
#pragma unroll
                for (int j = 0; j + 1 < NT; j += 2) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i][j][e]), __float_as_uint(acc[i][j + 1][e]), false, false);
                        v[e] = __uint_as_float(sw[0]);
                        v[4 + e] = __uint_as_float(sw[1]);
                    }
                    const int n = nb + (j + (g & 1)) * 16 + 4 * (g & 2);
                    if (LN) {   // ln_fix works on 4 columns: two halves
                        ln_fix(lms, lcs[j], lcb[j], v);
                        ln_fix(lms, lcs[j + 1], lcb[j + 1], v + 4);
                    }
                    epilogue_fast8<EPI == 1>(p, er, m, n, rb_row, v);
                    if (GN) { const float ok = m < p.M ? 1.f : 0.f; gn_add(j, v, ok); gn_add(j + 1, v + 4, ok); }
                }
                if (NT & 1) {
                    const int j = NT - 1;
                    const int n = nb + j * 16 + (lane >> 4) * 4;
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    if (LN) ln_fix(lms, lcs[j], lcb[j], v);
                    epilogue_fast<EPI == 1>(p, er, m, n, rb_row, v);
                    if (GN) gn_add(j, v, m < p.M ? 1.f : 0.f);
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = nb + j * 16 + (lane >> 4) * 4;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (LN) ln_fix(lms, lcs[j], lcb[j], v);
                epilogue_fast<EPI == 1>(p, er, m, n, rb_row, v);
                if (GN) gn_add(j, v, m < p.M ? 1.f : 0.f);
            }
        }
        if (GN) {
            // sum over the 16 rows (lane & 15) of the tile rows, then the lanes of row 0 publish their columns: slot j holds, in the
            // wide form, the pair exchange's columns (see above), else tile j's own 4 columns of lane group g
            const int g = lane >> 4;
            // (DPP row rotations: one VALU add each; __shfl_xor would be 160 ds_bpermute round trips per wave -- measured +10 us on the
            // 8192 x 320 x 2880 conv)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gs[j][e] = row16_sum(gs[j][e]);
                    gq[j][e] = row16_sum(gq[j][e]);
                }
            if ((lane & 15) == 0) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const bool paired = wide && ((j | 1) < NT);
                    const int c0 = paired ? ((j & ~1) + (g & 1)) * 16 + 4 * (g & 2) + 4 * (j & 1) : j * 16 + g * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gn_cols[c0 + e] = float2{gs[j][e], gq[j][e]};
                }
            }
        }
        return;
    }
    if (EPI <= 1) {   // split-K partial sums of the fast instantiations (N % 4 == 0): plain fp32 vector stores
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = mb + i * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = nb + j * 16 + (lane >> 4) * 4;
                if (m < p.M && n < p.N)
                    *reinterpret_cast<f32x4*>(p.ws + ((size_t)split * p.M + m) * p.N + n) = acc[i][j];
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = mb + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = nb + j * 16 + (lane >> 4) * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.splits > 1) {
                float* dst = p.ws + ((size_t)split * p.M + m) * p.N + n;
                if (n + 3 < p.N) *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                else for (int e = 0; e < 4 && n + e < p.N; ++e) dst[e] = v[e];
            } else {
                epilogue_store<EPI>(p, m, n, v);     // EPI 3: ragged N / fp32 output / operands >= 4 GiB
            }
        }
    }
}

// GNA (GemmArgs::gna_part): GroupNorm applied to the A operand between its global load and its LDS image (plain linears only).
// A32 ("precise" operand): A is an fp32 tensor (the master of the residual stream, or a GroupNorm output kept in
// fp32) and is split on its way into LDS into hi = round16(x) and lo = round16(x - hi); every K step runs two MFMAs, W.hi + W.lo --
// the operand carries ~22 significand bits instead of 11 at the price of issue slots this latency-bound step does not use.
template <int BM, int BN, bool CONV, int EPI, bool LN = false, bool GNA = false, bool A32 = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {
    constexpr bool GEGLU = EPI == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16_t* lds = reinterpret_cast<h16_t*>(smem);
    constexpr int TILE_ELEMS = ((A32 ? 2 : 1) * BM + BN) * LDS_STRIDE;  // [A tile BM rows | W tile BN rows | A32: A lo tile BM rows]
    constexpr uint32_t AE = A32 ? 4u : 2u;              // bytes per element of A
    constexpr int MT = BM / 32, NT = BN / 32;           // 16x16 MFMA tiles per wave (wave sub-tile BM/2 x BN/2)
    constexpr int AC = BM / 32, WC = BN / 32;           // 16-byte chunks per thread per K tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int ncols = GEGLU ? 2 * p.N : p.N;   // GEGLU: a BN-wide W tile yields BN/2 output columns
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (ncols + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = p.n_fast ? bid / tiles_n : bid % tiles_m, tn = p.n_fast ? bid % tiles_n : bid / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y;

    const int chunk = tid & 7;       // 16-byte chunk within the 64-wide K tile
    const int lrow = tid >> 3;       // 0..31, rows lrow + 32*i
    // Operands are read through buffer descriptors: an out-of-range offset returns 0 in hardware, so M/N/K tails and
    // the conv halo need no branches. (With predicated loads the compiler lost track of the outstanding-load count and
    // drained the whole prefetch ring with s_waitcnt vmcnt(0) before every LDS store.)
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.W), 0, p.w_bytes, 0x00020000);
    // Validity is carried as all-ones / zero masks OR-ed into the offset (pure ALU): a select would be turned into a
    // divergent branch by the compiler, which again hides the loads from its vmcnt bookkeeping.
    uint32_t a_base[AC], a_inv[AC];   // byte offset of the row (linear) / image (conv); a_inv = ~0 for rows >= M
    uint32_t a_lin[AC];
    uint32_t a_hb[AC];                                // CONV, p.hbits: bit t = tap t of this row's pixel lies outside the image
    int a_oy[AC], a_ox[AC];
    uint32_t w_base[WC], w_inv[WC];
#pragma unroll
    for (int i = 0; i < AC; ++i) {
        const int m = m0 + lrow + 32 * i;
        const bool ok = m < p.M;
        const int mc = ok ? m : 0;
        a_inv[i] = ok ? 0u : 0xFFFFFFFFu;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = mc / hw, rem = mc % hw;
            a_oy[i] = (rem / p.Wout) * p.stride - p.pad_h;      // input row / column of tap (0, 0)
            a_ox[i] = (rem % p.Wout) * p.stride - p.pad_w;
            a_base[i] = (uint32_t)b * (uint32_t)(p.Hin * p.Win * p.Cin) * AE;
            a_lin[i] = a_base[i] + (uint32_t)(a_oy[i] * p.Win + a_ox[i]) * (uint32_t)p.Cin * AE;   // + tap offset = the tap's pixel (no upsample)
            a_hb[i] = 0u;
            if (p.hbits) {
                a_lin[i] += chunk * 8u * AE;              // this thread's 8-element chunk of the 64-channel block
                {
                    uint32_t hb = 0u;
                    for (int ky = 0; ky < p.kh; ++ky)
                        for (int kx = 0; kx < p.kw; ++kx) {
                            const int iy = a_oy[i] + ky * p.dil, ix = a_ox[i] + kx * p.dil;
                            const uint32_t out = (uint32_t)((iy | ix | (p.lim_h - 1 - iy) | (p.lim_w - 1 - ix)) >> 31) & 1u;
                            hb |= out << (ky * p.kw + kx);
                        }
                    a_hb[i] = hb;
                }
            }
        } else {
            a_hb[i] = 0u;
            a_base[i] = (uint32_t)mc * (uint32_t)p.lda * AE + chunk * 8u * AE;
            a_oy[i] = a_ox[i] = 0;
            a_lin[i] = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < WC; ++i) {
        int n = n0 + lrow + 32 * i;
        bool ok;
        if (GEGLU) {
            // each wave column holds BN/4 value rows followed by the BN/4 gate rows of the same output columns, so a
            // lane finds value and gate of one output element in its own accumulators (tiles j and j + NT/2)
            const int rr = lrow + 32 * i, wn_ = rr / (BN / 2), within = rr % (BN / 2);
            const int oc = (n0 / 2) + wn_ * (BN / 4) + within % (BN / 4);
            ok = oc < p.N;
            n = (within >= BN / 4 ? p.N : 0) + oc;
        } else {
            ok = n < p.N;
        }
        w_inv[i] = ok ? 0u : 0xFFFFFFFFu;
        w_base[i] = ok ? w_row_byte(p, n) + chunk * 16u : 0u;
    }
    const uint32_t w_step = w_tile_step(p);

    const int nk_total = (p.K + BK - 1) / BK;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(nk_total, kt0 + p.kt_per_split);
    // running tap state of the next K tile to load (CONV with Cin % 64 == 0): see load_tile
    TapWalk walk{0, 0, 0};
    if (CONV && p.cin64) walk.init(p, kt0);
    // global -> registers for K tile kt (two register sets R0/R1 form a 2-deep prefetch ring); branch-free
    // (A32: ra holds the first 4 fp32 values of the lane's 8 elements, ra2 the other 4)
    auto load_tile = [&](int kt, u32x4 (&ra)[AC], u32x4 (&rw)[WC], u32x4 (&ra2)[A32 ? AC : 1]) {
        const uint32_t kbyte = (uint32_t)kt * (BK * AE);
        auto lda = [&](int i, uint32_t lin, uint32_t inv) {      // inv: all ones = outside the problem (reads zeros)
            ra[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, lin | inv, 0, 0));
            if (A32) ra2[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (lin + 16u) | inv, 0, 0));
        };
        // all ones when this lane's 8 k-elements lie beyond K (ragged last tile) or the tile is past this split's range:
        // such loads return zeros, so the K loop below needs no conditionals at all (the compiler then counts the
        // outstanding loads exactly and waits only for the register set it is about to store)
        const uint32_t k_inv = (uint32_t)((p.K - 1 - (kt * BK + chunk * 8)) >> 31) | (uint32_t)((kt1 - 1 - kt) >> 31);
        uint32_t ktw = (uint32_t)kt;           // the weight's K tile of this step
        if (CONV) {
            // tap of this lane's 8 k-elements: uniform per tile when Cin % 64 == 0, per lane otherwise (Cin % 8 == 0
            // keeps a 16-byte chunk inside one tap). Tiles are requested in increasing kt order, so with Cin % 64 == 0 the
            // (tap row, tap column, channel) of the tile is carried along instead of being re-derived by two integer divisions
            // (~35 VALU instructions per K tile and wave, next to 20-40 MFMAs).
            int ky, kx;
            uint32_t cbyte;
            if (p.cin64) {
                ky = walk.ky; kx = walk.kx;
                cbyte = (uint32_t)(walk.c + chunk * 8) * AE;
                ktw = walk.wtile(p);
                walk.next(p);
            } else {
                const int kk = kt * BK + chunk * 8;
                const int tap = kk / p.Cin;
                cbyte = (uint32_t)(kk - tap * p.Cin) * AE;
                ky = tap / p.kw; kx = tap - ky * p.kw;
            }
            const int hlim = p.lim_h, wlim = p.lim_w;
            const int dy = ky * p.dil, dx = kx * p.dil;                                   // wave-uniform tap displacement
            const uint32_t tap_off = (uint32_t)(dy * p.Win + dx) * (uint32_t)p.Cin * AE + cbyte;
            if (p.hbits) {     // uniform tap offset + per-row masks (GemmArgs::hbits); cbyte's per-thread chunk is already in a_lin
                const int tap = ky * p.kw + kx;
                const uint32_t toff = tap_off - chunk * 8u * AE;
#pragma unroll
                for (int i = 0; i < AC; ++i) {
                    const uint32_t halo = 0u - ((a_hb[i] >> tap) & 1u);
                    lda(i, a_lin[i] + toff, halo | a_inv[i] | k_inv);
                }
            } else
#pragma unroll
            for (int i = 0; i < AC; ++i) {
                const int iy = a_oy[i] + dy, ix = a_ox[i] + dx;
                // sign bit set if any of iy, ix, hlim-1-iy, wlim-1-ix is negative -> halo mask
                const uint32_t halo = (uint32_t)((iy | ix | (hlim - 1 - iy) | (wlim - 1 - ix)) >> 31);
                uint32_t lin = a_lin[i] + tap_off;
                if (p.ups) lin = a_base[i] + (uint32_t)((iy >> 1) * p.Win + (ix >> 1)) * (uint32_t)p.Cin * AE + cbyte;
                lda(i, lin, halo | a_inv[i] | k_inv);
            }
        } else {
#pragma unroll
            for (int i = 0; i < AC; ++i) {
                lda(i, a_base[i] + kbyte, a_inv[i] | k_inv);
            }
        }
#pragma unroll
        for (int i = 0; i < WC; ++i) {
            const uint32_t off = (w_base[i] + ktw * w_step) | w_inv[i] | k_inv;
            rw[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, off, 0, 0));
        }
    };
    const int st_chunk = chunk ^ ((lrow >> 1) & 7);   // rows lrow + 32*i share bits 1..3 with lrow
    float ln_s[AC], ln_q[AC];     // LN: this thread's share (8 of every 64 k) of sum x and sum x^2 of rows lrow + 32*i
#pragma unroll
    for (int i = 0; i < AC; ++i) ln_s[i] = ln_q[i] = 0.f;
    // GNA: per-channel (scale, shift) of this tile's image, behind the two tile buffers (K float2; the tile lies in ONE image:
    // host-checked gna_hw % BM == 0). Group statistics are reduced from the producer's partials exactly like gn_apply_kernel does.
    float2* gna_tab = reinterpret_cast<float2*>(smem + (size_t)2 * TILE_ELEMS * sizeof(h16_t));
    auto store_tile = [&](int buf, const u32x4 (&ra_)[AC], const u32x4 (&rw)[WC], const u32x4 (&ra2_)[A32 ? AC : 1], int kt_ = 0) {
        h16_t* base = lds + buf * TILE_ELEMS;
        u32x4 ra[AC];
        u32x4 rlo[A32 ? AC : 1];
#pragma unroll
        for (int i = 0; i < AC; ++i) ra[i] = ra_[i];
        if (A32) {
            // fp32 operand: (GroupNorm on A first, in fp32; or, LN form, the LayerNorm weight gamma_k applied to A so that W stays the
            // exact un-folded weight) then hi = round16(x), lo = round16(x - hi)
            float ca[8], cb[8];
            if (GNA || LN) {
                int k0 = kt_ * BK + chunk * 8;
                k0 = k0 + 8 <= p.K ? k0 : p.K - 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float2 t = gna_tab[k0 + j]; ca[j] = t.x; cb[j] = t.y; }
            }
#pragma unroll
            for (int i = 0; i < AC; ++i) {
                float x[8];
#pragma unroll
                for (int d = 0; d < 4; ++d) { x[d] = __uint_as_float(ra_[i][d]); x[4 + d] = __uint_as_float(ra2_[A32 ? i : 0][d]); }
                if (LN) {       // row statistics of the RAW fp32 values (every K tile passes here exactly once; masked tiles are zeros)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { ln_s[i] += x[j]; ln_q[i] = fmaf(x[j], x[j], ln_q[i]); }
                }
                if (GNA || LN) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = fmaf(x[j], ca[j], cb[j]);
                }
                uint32_t hi4[4], lo4[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    hi4[d] = pack_h16x2(x[2 * d], x[2 * d + 1]);
                    lo4[d] = pack_h16x2(x[2 * d] - h16lo_to_f32(hi4[d]), x[2 * d + 1] - h16hi_to_f32(hi4[d]));
                }
                ra[i] = u32x4{hi4[0], hi4[1], hi4[2], hi4[3]};
                rlo[A32 ? i : 0] = u32x4{lo4[0], lo4[1], lo4[2], lo4[3]};
            }
        } else if (GNA) {   // y = round16(fma(x, a_c, b_c)): the value gn_apply_kernel writes for the same element
            int k0 = kt_ * BK + chunk * 8;
            k0 = k0 + 8 <= p.K ? k0 : p.K - 8;         // tiles past K carry zero weights: any finite A value will do
            float ca[8], cb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float2 t = gna_tab[k0 + j]; ca[j] = t.x; cb[j] = t.y; }
#pragma unroll
            for (int i = 0; i < AC; ++i) {
                uint32_t w4[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float lo = fmaf(h16lo_to_f32(w4[d]), ca[2 * d], cb[2 * d]);
                    const float hi = fmaf(h16hi_to_f32(w4[d]), ca[2 * d + 1], cb[2 * d + 1]);
                    w4[d] = pack_h16x2(lo, hi);
                }
                ra[i] = u32x4{w4[0], w4[1], w4[2], w4[3]};
            }
        }
        if (LN && !A32) {   // every K tile passes through here exactly once (masked tiles are zeros)
#pragma unroll
            for (int i = 0; i < AC; ++i) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float lo = h16lo_to_f32(ra[i][d]), hi = h16hi_to_f32(ra[i][d]);
                    ln_s[i] += lo + hi;
                    ln_q[i] = fmaf(lo, lo, fmaf(hi, hi, ln_q[i]));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < AC; ++i)
            *reinterpret_cast<u32x4*>(base + (lrow + 32 * i) * LDS_STRIDE + st_chunk * 8) = ra[i];
#pragma unroll
        for (int i = 0; i < WC; ++i)
            *reinterpret_cast<u32x4*>(base + (BM + lrow + 32 * i) * LDS_STRIDE + st_chunk * 8) = rw[i];
        if (A32) {
#pragma unroll
            for (int i = 0; i < AC; ++i)
                *reinterpret_cast<u32x4*>(base + (BM + BN + lrow + 32 * i) * LDS_STRIDE + st_chunk * 8) = rlo[A32 ? i : 0];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fg = lane >> 4, fswz = (frow >> 1) & 7;   // fragment rows are frow + 16*i: same swizzle
    auto compute = [&](int buf) {
        const h16_t* abase = lds + buf * TILE_ELEMS + (wm * (BM / 2) + frow) * LDS_STRIDE;
        const h16_t* wbase = lds + buf * TILE_ELEMS + (BM + wn * (BN / 2) + frow) * LDS_STRIDE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + fg) ^ fswz) * 8;
            h16x8 af[MT], wf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const h16x8*>(abase + i * 16 * LDS_STRIDE + coff);
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const h16x8*>(wbase + j * 16 * LDS_STRIDE + coff);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = mfma_16x16x32_h16(wf[j], af[i], acc[i][j]);
            if (A32) {      // the lo halves of the same rows, behind the W tile
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const h16x8*>(abase + ((BM + BN) + i * 16) * LDS_STRIDE + coff);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = mfma_16x16x32_h16(wf[j], af[i], acc[i][j]);
            }
        }
    };

    u32x4 a0[AC], w0[WC], a1[AC], w1[WC];
    u32x4 a0b[A32 ? AC : 1], a1b[A32 ? AC : 1];
    load_tile(kt0, a0, w0, a0b);
    load_tile(kt0 + 1, a1, w1, a1b);
    if (LN && A32) {      // gamma table of the LayerNorm applied on the A side: (gamma_k, 0)
        for (int c = tid; c < p.K; c += 256) gna_tab[c] = float2{h16_to_f32(p.gna_gamma[c]), 0.f};
        __syncthreads();
    }
    // (GNA: the statistics prologue runs with the first two K tiles already in flight)
    if (GNA) {
        float* g_mean = reinterpret_cast<float*>(gna_tab + p.K);          // [G], [G] behind the table
        float* g_rstd = g_mean + p.gna_G;
        float* ps = g_rstd + p.gna_G;                                     // [256], [256]
        float* pq = ps + 256;
        const int G = p.gna_G, parts = 256 / G, cpg = p.K / G;
        const int b_img = m0 / p.gna_hw;
        const int g = tid % G, part = tid / G;
        float sum = 0.f, sq = 0.f;
        if (part < parts) {
            for (int c0 = part; c0 < p.gna_nchunk; c0 += 8 * parts) {
                float2 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + u * parts;
                    t[u] = c < p.gna_nchunk ? *reinterpret_cast<const float2*>(p.gna_part + (((size_t)b_img * p.gna_nchunk + c) * G + g) * 2)
                                            : float2{0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { sum += t[u].x; sq += t[u].y; }
            }
        }
        ps[tid] = sum;
        pq[tid] = sq;
        __syncthreads();
        if (tid < G) {
            sum = 0.f; sq = 0.f;
            for (int k = 0; k < parts; ++k) { sum += ps[k * G + tid]; sq += pq[k * G + tid]; }
            const float cnt = (float)p.gna_hw * (float)cpg;
            const float mu = sum / cnt;
            const float var = fmaxf(sq / cnt - mu * mu, 0.f);
            g_mean[tid] = mu;
            g_rstd[tid] = rsqrtf(var + p.gna_eps);
        }
        __syncthreads();
        for (int c = tid; c < p.K; c += 256) {
            const int gg = c / cpg;
            const float ca = g_rstd[gg] * h16_to_f32(p.gna_gamma[c]);
            gna_tab[c] = float2{ca, h16_to_f32(p.gna_beta[c]) - g_mean[gg] * ca};
        }
        __syncthreads();
    }
    store_tile(0, a0, w0, a0b, kt0);
    load_tile(kt0 + 2, a0, w0, a0b);
    __syncthreads();
    // invariant at loop top: LDS buf0 = tile kt; R1 = tile kt+1 (in flight); R0 = tile kt+2 (in flight).
    // Tiles >= kt1 are all-zero (masked loads): an odd tail costs one wasted half-iteration, never a wrong sum.
    // sched_barrier(0) pins the issue order: each register set's loads go out right after the barrier that frees it,
    // a full compute phase before they are needed (left alone, hipcc sinks both load groups to the end of the body).
    for (int kt = kt0; kt < kt1; kt += 2) {
        compute(0);
        store_tile(1, a1, w1, a1b, kt + 1);
        __syncthreads();
        load_tile(kt + 3, a1, w1, a1b);
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        store_tile(0, a0, w0, a0b, kt + 2);
        __syncthreads();
        load_tile(kt + 4, a0, w0, a0b);
        __builtin_amdgcn_sched_barrier(0);
    }

    const float2* ln_stat = reinterpret_cast<const float2*>(smem);
    if (LN) {   // finish the row statistics: the 8 lanes tid & 7 of a row hold its partial sums
        __syncthreads();                       // the K loop's last fragment reads are done: the tile memory is free
        float2* st = reinterpret_cast<float2*>(smem);
        const float inv_k = 1.f / (float)p.K;
#pragma unroll
        for (int i = 0; i < AC; ++i) {
            float s_ = ln_s[i], q_ = ln_q[i];
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) { s_ += __shfl_xor(s_, o, 64); q_ += __shfl_xor(q_, o, 64); }
            if (chunk == 0) {
                const float mean = s_ * inv_k;
                const float var = fmaxf(q_ * inv_k - mean * mean, 0.f);
                st[lrow + 32 * i] = float2{mean, rsqrtf(var + p.ln_eps)};
            }
        }
        __syncthreads();
    }

    // ---- GEGLU epilogue: value tile j and gate tile j + NT/2 of the same lane ----
    if (GEGLU) {
        geglu_out<MT, NT / 2, LN>(p, acc, m0 + wm * (BM / 2), (n0 / 2) + wn * (BN / 4), lane, ln_stat, m0);
        return;
    }

    write_out<MT, NT, EPI, LN>(p, acc, m0 + wm * (BM / 2), n0 + wn * (BN / 2), split, lane, ln_stat, m0);
}

// ------------------------------------------------------------------------------------------------------------------
// Row-order epilogue of the LDS-DMA kernel (splits == 1, EPI <= 1, no producer statistics). In the accumulator layout a lane owns 4
// (8 after the pair exchange) consecutive columns of 16 different rows: one store instruction of a wave touches 16 rows with 64 B
// (16-bit output) or 4 x 16 B at a 32-byte stride (the fp32 master c32d / the fp32 residual res32) in each -- every 128-byte line of
// the output is written in pieces by several instructions. On the store-heavy linears of the tall maps (92160 x 320 x 320 + res32 +
// c32d: 354 MB of HBM traffic for 19 GFLOP) that pattern, not the K loop, sets the time: 2.9 TB/s whatever the tile. Here the block
// stages its BM x BN fp32 tile in the (now free) operand ring, half the rows at a time, and the 512 threads walk it in row order --
// consecutive lanes hold consecutive 16-byte chunks of a row, so a wave's access is 1 KiB (fp32) / 512 B (16-bit) of contiguous bytes.
// Staging image: row stride BN + 4 floats (656 / 528 / 272 B = 144 / 16 / 16 mod 256: the 16 rows of a 16-lane group fall into 16
// distinct 16-byte bank slots for ds_write_b128; the row-order reads are linear).
// ------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int MT, int NT, bool ACT>
__device__ __forceinline__ void epilogue_lds(const GemmArgs& p, f32x4 (&acc)[MT][NT], char* smem, int m0, int n0, int wm, int wn) {
    constexpr int LDT = BN + 4, HR = BM / 2, CH = BN / 4;
    float* T = reinterpret_cast<float*>(smem);
    const EpiRsrc er = make_epi_rsrc(p);
    const int tid = threadIdx.x, lane = tid & 63;
    const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();                      // pass 0: every wave has finished its fragment reads of the ring; pass 1: the rows are out
        if ((wm >> 1) == pass) {
            float* tw = T + ((wm & 1) * (BM / 4) + r16) * LDT + wn * (BN / 2) + 4 * g;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(tw + i * 16 * LDT + j * 16) = acc[i][j];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < (HR * CH + 511) / 512; ++it) {
            const int idx = it * 512 + tid;
            if ((HR * CH) % 512 != 0 && idx >= HR * CH) break;
            const int row = idx / CH, c4 = idx - row * CH;
            const int m = m0 + pass * HR + row, n = n0 + c4 * 4;
            const f32x4 q = *reinterpret_cast<const f32x4*>(T + row * LDT + c4 * 4);
            float v[4] = {q[0], q[1], q[2], q[3]};
            uint32_t rb_row = 0;
            if (p.rowbias) rb_row = (uint32_t)((m < p.M ? m : 0) / p.rows_per_group) * (uint32_t)p.N * 2u;
            epilogue_fast<ACT>(p, er, m, n, rb_row, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant for the large problems: 128 x BN x 64 tiles (BN = 160 or 128), 512 threads = 8 waves (4 along M x 2
// along N, wave tile 32 x BN/2), ONE block per CU. Operands go global -> LDS directly (`buffer_load_dwordx4 ... lds`,
// no VGPR staging, no ds_write pass) into a ring of NS stages; the prefetch of tiles kt+1 .. kt+NS-1 stays in flight
// across the single raw s_barrier per K tile behind a counted `s_waitcnt vmcnt(N)`. Why a second structure: the
// register-staged 128^2 kernel above spends, per K tile and CU, 512 cycles of L1 bandwidth + ~420 cycles of ds_write
// transfer + 256 of ds_read for 512 cycles of MFMA -- all co-limited behind two barriers; here the ds_write pass is
// gone, the wave tile needs 14 (BN=160) ds_read_b128 per 20 MFMAs, and two waves per SIMD let one wave's DMA issue
// overlap the other's MFMAs.
//   * LDS-DMA writes lane l's 16 bytes at (wave-uniform base) + 16*l: the image must be linear in lane order, so the
//     XOR chunk swizzle of the image is applied on the SOURCE side (lane at slot c of row r fetches global chunk
//     c ^ ((r>>1)&7)); the fragment reads use the same involution as the kernel above.
//   * out-of-range buffer offsets make the DMA write ZEROS (probed on MI355X, scripts/exp/glds_oob_probe.hip), so the
//     conv halo, the M/N/K tails and the tiles past this split's K range need no branches and the vmcnt count is exact.
//   * W pieces (8 rows x 128 B per wave instruction): BN/8 = 20 does not divide over 8 waves -- waves 0-3 issue 3, waves
//     4-7 issue 2 (a wave-uniform branch) and each half waits with its own count.
// ------------------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NS = 2 (one K tile of look-ahead, 72 KiB of LDS at 128 x 160): TWO blocks per CU. For the short-K, store-heavy linears of the
// 16-frame video maps (92160 x 320 x 320 with an fp32 residual read and an fp32 master written beside the 16-bit output: 354 MB of HBM
// traffic for 19 GFLOP) a block is prologue + 5 K tiles + a long epilogue; a second resident block runs its loads / MFMAs under the
// first one's epilogue stores.
template <int BN, int NS, bool CONV, int EPI, int BM = 128, bool GN = false>
__global__ __launch_bounds__(512, NS == 2 ? 2 : 1) void gemm_dma_kernel(GemmArgs p) {
    // BM = 128: wave tile 32 x BN/2. BM = 64 (wave tile 16 x BN/2): twice the row tiles for problems whose 128-row tiling leaves
    // the chip half empty -- the UNet's 3x3 convs at 8192 rows then need no split-K, i.e. no fp32 slab round trip through HBM
    // (42 MB written + re-read per conv against a 5 MB output) and no reduce launch.
    static_assert(BM == 128 || BM == 64, "row tile");
    constexpr int ROWS = BM + BN;
    constexpr int STAGE = ROWS * 128;                 // bytes per stage: [A tile BM rows | W tile BN rows], 128-byte rows
    constexpr int MT = BM / 64, NT = BN / 32;         // 16x16 MFMA tiles per wave (wave tile BM/4 x BN/2)
    constexpr int AJ = BM / 64;                       // A pieces per wave (8 rows each, 8 waves)
    constexpr int WP = BN / 8;                        // W pieces in all
    constexpr int WJ = (WP + 7) / 8;                  // W pieces per wave, upper bound
    constexpr bool W_RAGGED = (WP % 8) != 0;          // last W piece only on waves < WP % 8
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = p.n_fast ? bid / tiles_n : bid % tiles_m, tn = p.n_fast ? bid % tiles_n : bid / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y;

    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.W), 0, p.w_bytes, 0x00020000);

    // this lane's slot in a piece: row prow of 8, 16-byte slot `slot` of 8; the global chunk it fetches is slot ^ swizzle(row)
    const int prow = lane >> 3, slot = lane & 7;
    uint32_t a_base[AJ], a_inv[AJ], a_gch[AJ], a_lin[AJ];
    uint32_t a_hb[AJ];                                // CONV, p.hbits: bit t = tap t of this piece's pixel lies outside the image
    int a_oy[AJ], a_ox[AJ];
    uint32_t w_base[WJ], w_inv[WJ], w_gch[WJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int R = (j * 8 + wave) * 8 + prow;
        const int m = m0 + R;
        const bool ok = m < p.M;
        const int mc = ok ? m : 0;
        a_gch[j] = (uint32_t)(slot ^ ((R >> 1) & 7));
        a_inv[j] = ok ? 0u : 0xFFFFFFFFu;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = mc / hw, rem = mc % hw;
            a_oy[j] = (rem / p.Wout) * p.stride - p.pad_h;
            a_ox[j] = (rem % p.Wout) * p.stride - p.pad_w;
            a_base[j] = (uint32_t)b * (uint32_t)(p.Hin * p.Win * p.Cin) * 2u;
            a_lin[j] = a_base[j] + (uint32_t)(a_oy[j] * p.Win + a_ox[j]) * (uint32_t)p.Cin * 2u;
            a_hb[j] = 0u;
            if (p.hbits) {
                a_lin[j] += a_gch[j] * 16u;               // this lane's 16-byte chunk of the 64-channel block
                {
                    uint32_t hb = 0u;
                    for (int ky = 0; ky < p.kh; ++ky)
                        for (int kx = 0; kx < p.kw; ++kx) {
                            const int iy = a_oy[j] + ky * p.dil, ix = a_ox[j] + kx * p.dil;
                            const uint32_t out = (uint32_t)((iy | ix | (p.lim_h - 1 - iy) | (p.lim_w - 1 - ix)) >> 31) & 1u;
                            hb |= out << (ky * p.kw + kx);
                        }
                    a_hb[j] = hb;
                }
            }
        } else {
            a_hb[j] = 0u;
            a_base[j] = (uint32_t)mc * (uint32_t)p.lda * 2u + a_gch[j] * 16u;
            a_oy[j] = a_ox[j] = 0;
            a_lin[j] = 0;
        }
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int R = (j * 8 + wave) * 8 + prow;      // row inside the W tile (rows >= BN: ragged last piece, never issued)
        const int n = n0 + R;
        const bool ok = n < p.N && R < BN;
        w_gch[j] = (uint32_t)(slot ^ ((R >> 1) & 7));   // BM = 128 rows precede the W tile: (128 + R) >> 1 & 7 == R >> 1 & 7
        w_inv[j] = ok ? 0u : 0xFFFFFFFFu;
        w_base[j] = ok ? w_row_byte(p, n) + w_gch[j] * 16u : 0u;
    }
    const uint32_t w_step = w_tile_step(p);

    const int nk_total = (p.K + BK - 1) / BK;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(nk_total, kt0 + p.kt_per_split);

    // DMA of K tile kt into ring stage `stage` (wave-uniform); tiles >= kt1 / chunks >= K are all-ones offsets -> zeros
    // running tap state of the next K tile to issue (CONV with Cin % 64 == 0; tiles are issued in increasing kt order): the
    // tap row / column / channel are carried along instead of re-derived by two integer divisions per tile
    TapWalk walk{0, 0, 0};
    if (CONV && p.cin64) walk.init(p, kt0);
    auto issue = [&](int kt, int stage) {
        char* sb = smem + stage * STAGE;
        const uint32_t kbyte = (uint32_t)kt * (BK * 2);
        const uint32_t t_inv = (uint32_t)((kt1 - 1 - kt) >> 31);
        const int t_c = walk.c, t_ky = walk.ky, t_kx = walk.kx;
        uint32_t ktw = (uint32_t)kt;           // the weight's K tile of this step
        if (CONV && p.cin64) {
            ktw = walk.wtile(p);
            walk.next(p);
        }
        const bool fast_a = CONV && p.hbits;   // wave-uniform tap offset + per-piece masks (GemmArgs::hbits)
        if (fast_a) {
            const int tap = t_ky * p.kw + t_kx;
            const uint32_t tap_off = (uint32_t)((t_ky * p.Win + t_kx) * p.dil) * (uint32_t)p.Cin * 2u + (uint32_t)t_c * 2u;
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const uint32_t halo = 0u - ((a_hb[j] >> tap) & 1u);
                const uint32_t off = (a_lin[j] + tap_off) | halo | a_inv[j] | t_inv;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sb + (j * 8 + wave) * 1024), 16, off, 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            if (fast_a) break;
            const uint32_t k_inv = (uint32_t)((p.K - 1 - (kt * BK + (int)a_gch[j] * 8)) >> 31) | t_inv;
            uint32_t off;
            if (CONV) {
                int ky, kx;
                uint32_t cbyte;
                if (p.cin64) {
                    ky = t_ky; kx = t_kx;
                    cbyte = (uint32_t)(t_c + (int)a_gch[j] * 8) * 2u;
                } else {
                    const int kk = kt * BK + (int)a_gch[j] * 8;
                    const int tap = kk / p.Cin;
                    cbyte = (uint32_t)(kk - tap * p.Cin) * 2u;
                    ky = tap / p.kw; kx = tap - ky * p.kw;
                }
                const int dy = ky * p.dil, dx = kx * p.dil;
                const int iy = a_oy[j] + dy, ix = a_ox[j] + dx;
                const uint32_t halo = (uint32_t)((iy | ix | (p.lim_h - 1 - iy) | (p.lim_w - 1 - ix)) >> 31);
                uint32_t lin = a_lin[j] + (uint32_t)(dy * p.Win + dx) * (uint32_t)p.Cin * 2u + cbyte;
                if (p.ups) lin = a_base[j] + (uint32_t)((iy >> 1) * p.Win + (ix >> 1)) * (uint32_t)p.Cin * 2u + cbyte;
                off = lin | halo | a_inv[j] | k_inv;
            } else {
                off = (a_base[j] + kbyte) | a_inv[j] | k_inv;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sb + (j * 8 + wave) * 1024), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            if (W_RAGGED && j == WJ - 1 && wave >= WP % 8) break;
            const uint32_t k_inv = (uint32_t)((p.K - 1 - (kt * BK + (int)w_gch[j] * 8)) >> 31) | t_inv;
            const uint32_t off = (w_base[j] + ktw * w_step) | w_inv[j] | k_inv;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + (BM / 8 + j * 8 + wave) * 1024), 16, off, 0, 0, 0);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fg = lane >> 4, fswz = (frow >> 1) & 7;
    const int a_rd = (wm * (BM / 4) + frow) * 128, w_rd = (BM + wn * (BN / 2) + frow) * 128;
    h16x8 af[MT], wf[NT];
    auto read_frags = [&](int stage, int ks) {
        const char* sb = smem + stage * STAGE;
        const int coff = ((ks * 4 + fg) ^ fswz) * 16;
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const h16x8*>(sb + a_rd + i * 16 * 128 + coff);
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const h16x8*>(sb + w_rd + j * 16 * 128 + coff);
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[i][j] = mfma_16x16x32_h16(wf[j], af[i], acc[i][j]);
    };

    // pieces per tile issued by this wave (vmcnt bookkeeping)
    constexpr int L_HI = AJ + WJ, L_LO = AJ + WJ - (W_RAGGED ? 1 : 0);
    const bool lo_half = W_RAGGED && wave >= WP % 8;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(kt0 + s, s);
    int stage = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        // tile kt is the oldest of the NS-1 tiles in flight: leave the NS-2 younger ones outstanding
        if (lo_half) wait_vmcnt<(NS - 2) * L_LO>(); else wait_vmcnt<(NS - 2) * L_HI>();
        __builtin_amdgcn_s_barrier();          // every wave's share of tile kt has landed; stage (kt-1) % NS is free again
        int nxt = stage + NS - 1;
        if (nxt >= NS) nxt -= NS;
        // The DMA pieces (~100 cycles of issue stall each) are issued while this wave's fragment reads are in flight, and the
        // two waves that share a SIMD (w and w + 4) do it at different points of the K tile, so one wave's issue stall falls
        // into the other's MFMA cluster; in lockstep both would stall, then both compute.
        read_frags(stage, 0);
        if (wave < 4 && p.dbg != 2) { __builtin_amdgcn_sched_barrier(0); issue(kt + NS - 1, nxt); __builtin_amdgcn_sched_barrier(0); }
        if (p.dbg != 1) mfmas();
        read_frags(stage, 1);
        if (wave >= 4 && p.dbg != 2) { __builtin_amdgcn_sched_barrier(0); issue(kt + NS - 1, nxt); __builtin_amdgcn_sched_barrier(0); }
        if (p.dbg != 1) mfmas();
        if (++stage == NS) stage = 0;
    }
    wait_vmcnt<0>();                            // the masked tail DMAs must not outlive the workgroup's LDS allocation

    if (!GN) {
        if constexpr (EPI <= 1) {
            if (p.epi_lds && p.splits == 1) {           // block-uniform
                epilogue_lds<BM, BN, MT, NT, EPI == 1>(p, acc, smem, m0, n0, wm, wn);
                return;
            }
        }
        write_out<MT, NT, EPI>(p, acc, m0 + wm * (BM / 4), n0 + wn * (BN / 2), split, lane);
        return;
    }
    // ---- producer-side GroupNorm statistics (GemmArgs::gn_part; host-checked: splits == 1, gn_cr == 64, (BN / 2) % gn_cpg == 0,
    // N % gn_cpg == 0): every wave leaves the column sums of its BM/4 x BN/2 tile in LDS; per 64-row chunk and group the two (BM = 128)
    // or four (BM = 64) row waves and the group's gn_cpg columns are then summed in a fixed order and written to gn_part.
    __syncthreads();                            // the ring is free: every wave has finished its last fragment reads
    float2* cols = reinterpret_cast<float2*>(smem);                 // [8 waves][BN / 2]
    write_out<MT, NT, EPI, false, true>(p, acc, m0 + wm * (BM / 4), n0 + wn * (BN / 2), split, lane, nullptr, 0, cols + wave * (BN / 2));
    __syncthreads();
    {
        constexpr int NCH = BM / 64, WPC = 4 / NCH;                  // 64-row chunks per tile, row waves per chunk
        const int cpg = p.gn_cpg, gpt = BN / cpg;                   // groups per tile (BN % cpg == 0)
        for (int t = tid; t < NCH * gpt; t += 512) {
            const int c = t / gpt, gi = t - c * gpt;
            const int col0 = gi * cpg, ncol = n0 + col0;
            const int mrow = m0 + c * 64;
            if (ncol >= p.N || mrow >= p.M) continue;
            const int wn_ = col0 / (BN / 2), cw = col0 - wn_ * (BN / 2);   // the group lies inside one column wave ((BN/2) % cpg == 0)
            float ss = 0.f, qq = 0.f;
            for (int w = 0; w < WPC; ++w) {
                const float2* src = cols + ((c * WPC + w) * 2 + wn_) * (BN / 2) + cw;
                for (int k = 0; k < cpg; ++k) { ss += src[k].x; qq += src[k].y; }
            }
            float* dst = p.gn_part + ((size_t)(mrow >> 6) * p.gn_G + (ncol / cpg)) * 2;
            dst[0] = ss;
            dst[1] = qq;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tiles for the LARGE problems (>= one tile per CU): 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 (128 accumulator
// registers), ONE block per CU, both operands by LDS-DMA into two K-tile buffers of four 16 KiB half-tiles [A rows 0-127 | A rows
// 128-255 | W rows 0-127 | W rows 128-255]. A K tile is four phases of 16 MFMAs, one C quadrant each, in the order
// (a0,b0) (a0,b1) (a1,b1) (a1,b0), so each phase reads at most one new A sub-block (8 ds_read_b128) and one new W sub-block (4);
// each phase also issues ONE half-tile of DMA (2 instructions per wave). The two wave groups wr = 0 / 1 (which share the SIMDs
// pairwise) run half a phase apart -- one executes its read / DMA-issue slot while the other runs its MFMA cluster -- through one
// extra barrier at the start of group 1 and at the end of group 0 (equal barrier counts for every wave).
//   DMA schedule (K tile kt, buffers cur = kt & 1, nxt):  P1: A-h1(kt+1) -> nxt   P2: W-h1(kt+1) -> nxt
//                                                       P3: W-h0(kt+2) -> cur   P4: A-h0(kt+2) -> cur
//   WAR: a region is re-staged at least one phase after its last fragment read, and every read slot retires its reads (lgkmcnt(0))
//   before its barrier. RAW: before the barrier that ends phase 4 every wave waits vmcnt(4): all but its last two half-tile shares
//   (P3 / P4 of this K tile, for K tile kt+2) have landed, i.e. K tile kt+1 is complete when phase 1 reads it.
// The wave tile needs 24 fragment reads per 64 MFMAs (the 128 x 160 kernel above: 14 per 20).
// ------------------------------------------------------------------------------------------------------------------
template <bool CONV, int EPI, bool LN = false>
__global__ __launch_bounds__(512, 1) void gemm_p8_kernel(GemmArgs p) {
    constexpr bool GEGLU = EPI == 2;
    constexpr int HT = 128 * 128;                     // bytes of a half-tile (128 rows x 128 B)
    constexpr int KT_BYTES = 4 * HT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ncols = GEGLU ? 2 * p.N : p.N;     // GEGLU: a 256-row W tile yields 128 output columns
    const int tiles_m = (p.M + 255) / 256, tiles_n = (ncols + 255) / 256;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = p.n_fast ? bid / tiles_n : bid % tiles_m, tn = p.n_fast ? bid % tiles_n : bid / tiles_m;
    const int m0 = tm * 256, n0 = tn * 256;
    const int split = blockIdx.y;

    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.W), 0, p.w_bytes, 0x00020000);

    // a DMA piece = 8 rows x 128 B (one wave instruction); a half-tile = 16 pieces, this wave issues pieces wave and 8 + wave
    const int prow = lane >> 3, slot = lane & 7;
    uint32_t a_base[2][2], a_inv[2][2], a_lin[2][2], w_base[2][2], w_inv[2][2], gch[2];
    uint32_t a_hb[2][2] = {{0u, 0u}, {0u, 0u}};      // CONV, p.hbits: bit t = tap t of this piece's pixel lies outside the image
    int a_oy[2][2], a_ox[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int R = (j * 8 + wave) * 8 + prow;       // row inside the half-tile
        gch[j] = (uint32_t)(slot ^ ((R >> 1) & 7));    // source-side XOR swizzle of the 16-byte chunk (image linear in lane order)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + h * 128 + R;
            const bool okm = m < p.M;
            const int mc = okm ? m : 0;
            a_inv[h][j] = okm ? 0u : 0xFFFFFFFFu;
            if (CONV) {
                const int hw = p.Hout * p.Wout;
                const int b = mc / hw, rem = mc % hw;
                a_oy[h][j] = (rem / p.Wout) * p.stride - p.pad_h;
                a_ox[h][j] = (rem % p.Wout) * p.stride - p.pad_w;
                a_base[h][j] = (uint32_t)b * (uint32_t)(p.Hin * p.Win * p.Cin) * 2u;
                a_lin[h][j] = a_base[h][j] + (uint32_t)(a_oy[h][j] * p.Win + a_ox[h][j]) * (uint32_t)p.Cin * 2u;
                if (p.hbits) {
                    a_lin[h][j] += gch[j] * 16u;         // this lane's 16-byte chunk of the 64-channel block
                    uint32_t hb = 0u;
                    for (int ky = 0; ky < p.kh; ++ky)
                        for (int kx = 0; kx < p.kw; ++kx) {
                            const int iy = a_oy[h][j] + ky * p.dil, ix = a_ox[h][j] + kx * p.dil;
                            const uint32_t out = (uint32_t)((iy | ix | (p.lim_h - 1 - iy) | (p.lim_w - 1 - ix)) >> 31) & 1u;
                            hb |= out << (ky * p.kw + kx);
                        }
                    a_hb[h][j] = hb;
                }
            } else {
                a_base[h][j] = (uint32_t)mc * (uint32_t)p.lda * 2u + gch[j] * 16u;
                a_oy[h][j] = a_ox[h][j] = 0;
                a_lin[h][j] = 0;
            }
            int n = n0 + h * 128 + R;
            bool okn = n < p.N;
            if (GEGLU) {
                // each wave column (64 W-tile rows) holds the 32 value rows and then the 32 gate rows of the same 32 output columns,
                // so a lane finds value and gate of one output element in its own accumulators (n-tiles j and j + 2)
                const int wcol = 2 * h + R / 64, within = R % 64;
                const int oc = n0 / 2 + wcol * 32 + within % 32;
                okn = oc < p.N;
                n = (within >= 32 ? p.N : 0) + oc;
            }
            w_inv[h][j] = okn ? 0u : 0xFFFFFFFFu;
            w_base[h][j] = okn ? w_row_byte(p, n) + gch[j] * 16u : 0u;
        }
    }
    const uint32_t w_step = w_tile_step(p);
    const int nk_total = (p.K + BK - 1) / BK;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(nk_total, kt0 + p.kt_per_split);

    // running tap state of the next K tile of each A half (CONV with Cin % 64 == 0): each half's shares are issued in increasing
    // kt order, so tap row / column / channel are carried along instead of re-derived by two integer divisions per share
    TapWalk wa[2] = {{0, 0, 0}, {0, 0, 0}}, ww[2] = {{0, 0, 0}, {0, 0, 0}};       // per A half / W half (each is issued once per K tile, in order)
    if (CONV && p.cin64) {
        wa[0].init(p, kt0);
        wa[1] = wa[0]; ww[0] = wa[0]; ww[1] = wa[0];
    }
    // one half-tile share of this wave: kind 0 = A, 1 = W; tiles >= kt1 and chunks >= K are all-ones offsets (DMA writes zeros)
    auto issue = [&](int kind, int h, int kt, int buf) {
        char* sb = smem + buf * KT_BYTES + (kind * 2 + h) * HT;
        const uint32_t kbyte = (uint32_t)kt * (BK * 2);
        const uint32_t t_inv = (uint32_t)((kt1 - 1 - kt) >> 31);
        int t_c = 0, t_ky = 0, t_kx = 0;
        uint32_t ktw = (uint32_t)kt;           // the weight's K tile of this step
        if (CONV && p.cin64) {
            if (kind == 0) {
                t_c = wa[h].c; t_ky = wa[h].ky; t_kx = wa[h].kx;
                wa[h].next(p);
            } else {
                ktw = ww[h].wtile(p);
                ww[h].next(p);
            }
        }
        if (CONV && kind == 0 && p.hbits) {      // wave-uniform tap offset + per-piece masks (GemmArgs::hbits)
            const int tap = t_ky * p.kw + t_kx;
            const uint32_t tap_off = (uint32_t)((t_ky * p.Win + t_kx) * p.dil) * (uint32_t)p.Cin * 2u + (uint32_t)t_c * 2u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t halo = 0u - ((a_hb[h][j] >> tap) & 1u);
                const uint32_t off = (a_lin[h][j] + tap_off) | halo | a_inv[h][j] | t_inv;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sb + (j * 8 + wave) * 1024), 16, off, 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t k_inv = (uint32_t)((p.K - 1 - (kt * BK + (int)gch[j] * 8)) >> 31) | t_inv;
            uint32_t off;
            if (kind == 0) {
                if (CONV) {
                    int ky, kx;
                    uint32_t cbyte;
                    if (p.cin64) {
                        ky = t_ky; kx = t_kx;
                        cbyte = (uint32_t)(t_c + (int)gch[j] * 8) * 2u;
                    } else {
                        const int kk = kt * BK + (int)gch[j] * 8;
                        const int tap = kk / p.Cin;
                        cbyte = (uint32_t)(kk - tap * p.Cin) * 2u;
                        ky = tap / p.kw; kx = tap - ky * p.kw;
                    }
                    const int dy = ky * p.dil, dx = kx * p.dil;
                    const int iy = a_oy[h][j] + dy, ix = a_ox[h][j] + dx;
                    const uint32_t halo = (uint32_t)((iy | ix | (p.lim_h - 1 - iy) | (p.lim_w - 1 - ix)) >> 31);
                    uint32_t lin = a_lin[h][j] + (uint32_t)(dy * p.Win + dx) * (uint32_t)p.Cin * 2u + cbyte;
                    if (p.ups) lin = a_base[h][j] + (uint32_t)((iy >> 1) * p.Win + (ix >> 1)) * (uint32_t)p.Cin * 2u + cbyte;
                    off = lin | halo | a_inv[h][j] | k_inv;
                } else {
                    off = (a_base[h][j] + kbyte) | a_inv[h][j] | k_inv;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sb + (j * 8 + wave) * 1024), 16, off, 0, 0, 0);
            } else {
                off = (w_base[h][j] + ktw * w_step) | w_inv[h][j] | k_inv;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + (j * 8 + wave) * 1024), 16, off, 0, 0, 0);
            }
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fg = lane >> 4, fswz = (frow >> 1) & 7;
    const int a_rd = (0 * 2 + wr) * HT + frow * 128;                                  // this wave's A half
    const int w_rd = (1 * 2 + (wc >> 1)) * HT + ((wc & 1) * 64 + frow) * 128;         // its 64 W rows inside their half
    h16x8 af[2][4], wf[2][2][2];          // A sub-block [ks][i]; W sub-blocks b0 / b1 [ks][j]
    auto read_a = [&](int buf, int qa) {
        const char* sb = smem + buf * KT_BYTES + a_rd + qa * 64 * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[ks][i] = *reinterpret_cast<const h16x8*>(sb + i * 16 * 128 + (((ks * 4 + fg) ^ fswz) * 16));
    };
    auto read_w = [&](int buf, int qb) {
        const char* sb = smem + buf * KT_BYTES + w_rd + qb * 32 * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                wf[qb][ks][j] = *reinterpret_cast<const h16x8*>(sb + j * 16 * 128 + (((ks * 4 + fg) ^ fswz) * 16));
    };
    auto mfmas = [&](int qa, int qb) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[qa * 4 + i][qb * 2 + j] = mfma_16x16x32_h16(wf[qb][ks][j], af[ks][i], acc[qa * 4 + i][qb * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto slot_end = [&]() {                 // fragment reads retired (their region may be re-staged from the next phase on), then the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // prologue: K tile kt0 whole, W-h0 / A-h0 of kt0 + 1 (the order the loop would have issued them in)
    issue(1, 0, kt0, 0); issue(0, 0, kt0, 0); issue(0, 1, kt0, 0); issue(1, 1, kt0, 0);
    issue(1, 0, kt0 + 1, 1); issue(0, 0, kt0 + 1, 1);
    wait_vmcnt<4>();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // group 1 runs half a phase behind group 0

    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const int nxt = cur ^ 1;
        // ---- phase 1: quadrant (a0, b0)
        read_w(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        issue(0, 1, kt + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
        slot_end();
        mfmas(0, 0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: quadrant (a0, b1)
        read_w(cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        issue(1, 1, kt + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
        slot_end();
        mfmas(0, 1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: quadrant (a1, b1)
        read_a(cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        issue(1, 0, kt + 2, cur);
        __builtin_amdgcn_sched_barrier(0);
        slot_end();
        mfmas(1, 1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 4: quadrant (a1, b0); K tile kt + 1 must have landed before the barrier that opens group 0's next read slot
        issue(0, 0, kt + 2, cur);
        __builtin_amdgcn_sched_barrier(0);
        if (wr == 1) wait_vmcnt<4>();
        slot_end();
        mfmas(1, 0);
        if (wr == 0) wait_vmcnt<4>();
        __builtin_amdgcn_s_barrier();
        cur = nxt;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();        // matches group 1's extra barrier at the start
    wait_vmcnt<0>();                                  // the masked tail DMAs must not outlive the workgroup's LDS allocation

    if (GEGLU) {      // value tile j and gate tile j + 2 of the same lane (diffusers GEGLU: hidden * gelu(gate))
        geglu_out<8, 2, LN>(p, acc, m0 + wr * 128, (n0 / 2) + wc * 32, lane, p.ln_rows, 0);
        return;
    }
    write_out<8, 4, EPI, LN>(p, acc, m0 + wr * 128, n0 + wc * 64, split, lane, p.ln_rows, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// 256 x 128 tiles for the MID-SIZE problems (too few 256^2 tiles to fill the chip, e.g. 4608 x 1280: 90 of them but 180 of these):
// 8 waves = 4 (M) x 2 (N), wave tile 64 x 64, a K tile = TWO phases of 16 MFMAs ((a, b0) then (a, b1)), a 3-stage ring of K tiles
// (A rows 0-127 | A rows 128-255 | W: 48 KiB each) filled two K tiles ahead by LDS-DMA -- every share of K tile kt+2 is issued
// during K tile kt into the stage K tile kt-1 has left, and one vmcnt(6) per K tile retires K tile kt+1 before it is read. The
// two wave groups (waves 0-3 / 4-7, which share the SIMDs) run half a phase apart as in gemm_p8_kernel.
// ------------------------------------------------------------------------------------------------------------------
template <bool CONV, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_p8h_kernel(GemmArgs p) {
    constexpr int HT = 128 * 128;                     // bytes of a half-tile (128 rows x 128 B)
    constexpr int KT_BYTES = 3 * HT;                  // [A-h0 | A-h1 | W]
    constexpr int NSTG = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;           // 64-row block of the tile, 64-column block
    const int grp = wave >> 2;
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 127) / 128;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = p.n_fast ? bid / tiles_n : bid % tiles_m, tn = p.n_fast ? bid % tiles_n : bid / tiles_m;
    const int m0 = tm * 256, n0 = tn * 128;
    const int split = blockIdx.y;

    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.W), 0, p.w_bytes, 0x00020000);

    const int prow = lane >> 3, slot = lane & 7;
    uint32_t a_base[2][2], a_inv[2][2], a_lin[2][2], w_base[2], w_inv[2], gch[2];
    uint32_t a_hb[2][2];                              // CONV, p.hbits: bit t = tap t of this piece's pixel lies outside the image
    int a_oy[2][2], a_ox[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int R = (j * 8 + wave) * 8 + prow;       // row inside the half-tile
        gch[j] = (uint32_t)(slot ^ ((R >> 1) & 7));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + h * 128 + R;
            const bool okm = m < p.M;
            const int mc = okm ? m : 0;
            a_inv[h][j] = okm ? 0u : 0xFFFFFFFFu;
            if (CONV) {
                const int hw = p.Hout * p.Wout;
                const int b = mc / hw, rem = mc % hw;
                a_oy[h][j] = (rem / p.Wout) * p.stride - p.pad_h;
                a_ox[h][j] = (rem % p.Wout) * p.stride - p.pad_w;
                a_base[h][j] = (uint32_t)b * (uint32_t)(p.Hin * p.Win * p.Cin) * 2u;
                a_lin[h][j] = a_base[h][j] + (uint32_t)(a_oy[h][j] * p.Win + a_ox[h][j]) * (uint32_t)p.Cin * 2u;
                a_hb[h][j] = 0u;
                if (p.hbits) {
                    a_lin[h][j] += gch[j] * 16u;         // this lane's 16-byte chunk of the 64-channel block
                    uint32_t hb = 0u;
                    for (int ky = 0; ky < p.kh; ++ky)
                        for (int kx = 0; kx < p.kw; ++kx) {
                            const int iy = a_oy[h][j] + ky * p.dil, ix = a_ox[h][j] + kx * p.dil;
                            const uint32_t out = (uint32_t)((iy | ix | (p.lim_h - 1 - iy) | (p.lim_w - 1 - ix)) >> 31) & 1u;
                            hb |= out << (ky * p.kw + kx);
                        }
                    a_hb[h][j] = hb;
                }
            } else {
                a_hb[h][j] = 0u;
                a_base[h][j] = (uint32_t)mc * (uint32_t)p.lda * 2u + gch[j] * 16u;
                a_oy[h][j] = a_ox[h][j] = 0;
                a_lin[h][j] = 0;
            }
        }
        const int n = n0 + R;
        const bool okn = n < p.N;
        w_inv[j] = okn ? 0u : 0xFFFFFFFFu;
        w_base[j] = okn ? w_row_byte(p, n) + gch[j] * 16u : 0u;
    }
    const uint32_t w_step = w_tile_step(p);
    const int nk_total = (p.K + BK - 1) / BK;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(nk_total, kt0 + p.kt_per_split);

    TapWalk walk{0, 0, 0};                            // tap state of the next K tile (CONV, Cin % 64 == 0): tiles are issued in order
    if (CONV && p.cin64) walk.init(p, kt0);
    int t_c = 0, t_ky = 0, t_kx = 0;                  // tap of the K tile currently being issued (set by begin_tile)
    uint32_t t_ktw = 0;                               // ... and its weight K tile (CONV, Cin % 64 == 0)
    auto begin_tile = [&]() {
        t_c = walk.c; t_ky = walk.ky; t_kx = walk.kx;
        if (CONV && p.cin64) {
            t_ktw = walk.wtile(p);
            walk.next(p);
        }
    };
    // this wave's share (2 DMA instructions) of one half-tile of K tile kt: part 0 / 1 = A rows 0-127 / 128-255, part 2 = W
    auto issue = [&](int part, int kt, int stg) {
        char* sb = smem + stg * KT_BYTES + part * HT;
        const uint32_t kbyte = (uint32_t)kt * (BK * 2);
        const uint32_t t_inv = (uint32_t)((kt1 - 1 - kt) >> 31);
        if (CONV && part < 2 && p.hbits) {       // wave-uniform tap offset + per-piece masks (GemmArgs::hbits)
            const int tap = t_ky * p.kw + t_kx;
            const uint32_t tap_off = (uint32_t)((t_ky * p.Win + t_kx) * p.dil) * (uint32_t)p.Cin * 2u + (uint32_t)t_c * 2u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t halo = 0u - ((a_hb[part][j] >> tap) & 1u);
                const uint32_t off = (a_lin[part][j] + tap_off) | halo | a_inv[part][j] | t_inv;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sb + (j * 8 + wave) * 1024), 16, off, 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t k_inv = (uint32_t)((p.K - 1 - (kt * BK + (int)gch[j] * 8)) >> 31) | t_inv;
            uint32_t off;
            if (part < 2) {
                const int h = part;
                if (CONV) {
                    int ky, kx;
                    uint32_t cbyte;
                    if (p.cin64) {
                        ky = t_ky; kx = t_kx;
                        cbyte = (uint32_t)(t_c + (int)gch[j] * 8) * 2u;
                    } else {
                        const int kk = kt * BK + (int)gch[j] * 8;
                        const int tap = kk / p.Cin;
                        cbyte = (uint32_t)(kk - tap * p.Cin) * 2u;
                        ky = tap / p.kw; kx = tap - ky * p.kw;
                    }
                    const int dy = ky * p.dil, dx = kx * p.dil;
                    const int iy = a_oy[h][j] + dy, ix = a_ox[h][j] + dx;
                    const uint32_t halo = (uint32_t)((iy | ix | (p.lim_h - 1 - iy) | (p.lim_w - 1 - ix)) >> 31);
                    uint32_t lin = a_lin[h][j] + (uint32_t)(dy * p.Win + dx) * (uint32_t)p.Cin * 2u + cbyte;
                    if (p.ups) lin = a_base[h][j] + (uint32_t)((iy >> 1) * p.Win + (ix >> 1)) * (uint32_t)p.Cin * 2u + cbyte;
                    off = lin | halo | a_inv[h][j] | k_inv;
                } else {
                    off = (a_base[h][j] + kbyte) | a_inv[h][j] | k_inv;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sb + (j * 8 + wave) * 1024), 16, off, 0, 0, 0);
            } else {
                off = (w_base[j] + ((CONV && p.cin64) ? t_ktw : (uint32_t)kt) * w_step) | w_inv[j] | k_inv;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + (j * 8 + wave) * 1024), 16, off, 0, 0, 0);
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fg = lane >> 4, fswz = (frow >> 1) & 7;
    const int a_rd = (wr >> 1) * HT + ((wr & 1) * 64 + frow) * 128;       // this wave's 64 A rows inside their half
    const int w_rd = 2 * HT + (wc * 64 + frow) * 128;                     // its 64 W rows
    h16x8 af[2][4], wf[2][2][2];
    auto read_a = [&](int stg) {
        const char* sb = smem + stg * KT_BYTES + a_rd;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[ks][i] = *reinterpret_cast<const h16x8*>(sb + i * 16 * 128 + (((ks * 4 + fg) ^ fswz) * 16));
    };
    auto read_w = [&](int stg, int qb) {
        const char* sb = smem + stg * KT_BYTES + w_rd + qb * 32 * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                wf[qb][ks][j] = *reinterpret_cast<const h16x8*>(sb + j * 16 * 128 + (((ks * 4 + fg) ^ fswz) * 16));
    };
    auto mfmas = [&](int qb) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][qb * 2 + j] = mfma_16x16x32_h16(wf[qb][ks][j], af[ks][i], acc[i][qb * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto slot_end = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // prologue: K tiles kt0 and kt0 + 1 in stages 0 and 1
    begin_tile(); issue(0, kt0, 0); issue(1, kt0, 0); issue(2, kt0, 0);
    begin_tile(); issue(0, kt0 + 1, 1); issue(1, kt0 + 1, 1); issue(2, kt0 + 1, 1);
    wait_vmcnt<6>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();       // group 1 runs half a phase behind group 0

    int stg = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        int fill = stg + 2;                           // stage of K tile kt + 2 = the stage K tile kt - 1 has left
        if (fill >= NSTG) fill -= NSTG;
        // ---- phase 1: (a, b0)
        read_w(stg, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(stg);
        __builtin_amdgcn_sched_barrier(0);
        begin_tile();
        issue(0, kt + 2, fill);
        __builtin_amdgcn_sched_barrier(0);
        slot_end();
        mfmas(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: (a, b1); K tile kt + 1 must have landed before the barrier that opens group 0's next read slot
        read_w(stg, 1);
        __builtin_amdgcn_sched_barrier(0);
        issue(1, kt + 2, fill);
        issue(2, kt + 2, fill);
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1) wait_vmcnt<6>();
        slot_end();
        mfmas(1);
        if (grp == 0) wait_vmcnt<6>();
        __builtin_amdgcn_s_barrier();
        if (++stg == NSTG) stg = 0;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();       // matches group 1's extra barrier at the start
    wait_vmcnt<0>();

    write_out<4, 4, EPI>(p, acc, m0 + wr * 64, n0 + wc * 64, split, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-stationary streaming 3x3 conv for the weight-bound levels of the UNet (M = B * Hout * Wout <= 512 output pixels against
// 15 - 59 MB of weights that are read once per evaluation: the 16^2 / 8^2 maps of SD-v1.5 at CFG batch 2; ResnetBlock2D conv1 / conv2
// and the 2x upsampler there, call site custom_sd.py:634-639). Built like the decode path, not like the tile kernels above:
//   * W is a FRAGMENT-MAJOR copy (ops.repack_fm_conv): for every group of 16 output channels and every (32-channel block cb, tap)
//     one 1 KiB piece = the A operand of mfma_f32_16x16x32, pieces of a channel block contiguous. Every wave streams its own pieces
//     global -> registers, 16 B per lane, a whole channel block (9 taps x 2 column tiles = 18 KiB) ahead of its use.
//   * the activations are the B operand: a channel block of ALL input pixels (the slab, 64 B per pixel, XOR-swizzled 16-byte
//     chunks) sits in LDS and serves the 9 taps -- the taps are gathered by LDS addressing (a shifted row, or a zero row outside
//     the image), so A travels L2 -> CU once per channel block instead of once per tap.
//   * 8 waves = WM groups of 128 output rows x WK K-groups. Each K-group owns every WK-th channel block of the split's range and
//     has its own double-buffered slab (wave-private at WM = 1: no barrier in the loop); a block = one 32-column strip x one K
//     split, so N / 32 x splits ~ 200-256 blocks keep every CU pulling. The K-groups are summed through LDS in a fixed order.
//   * split-K partials go to the fp32 workspace like the tile kernels' (write_out); combine: splitk_reduce_kernel, or in-launch by
//     the block that arrives last at the strip's counter (GemmArgs::ws_cnt), slabs summed in split order either way.
// ------------------------------------------------------------------------------------------------------------------
constexpr int WS_NTW = 2;          // 16-column tiles per wave (strip = 32 output channels)
constexpr int WS_TAPS = 9;

__device__ __forceinline__ uint32_t ws_swz(uint32_t row) {      // chunk permutation of slab row `row` (conflict-free ds_read_b128)
    const uint32_t q = (row >> 2) & 3u;                         // {0, 1, 2, 3} -> {0, 3, 2, 1}
    return (q ^ (q << 1)) & 3u;
}
constexpr int WS_ZERO = 7 * 1024 + 64;      // LDS bytes in front of the slabs: 64 zero bytes at every 1024 i (see the tap loop)

template <int WM, int WK, bool UPS>
__global__ __launch_bounds__(512, 1) void wstream_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTW = WS_NTW, T = WS_TAPS, TPF = 8 / WK;
    constexpr int SLAB = 8192 * WM;                  // bytes: 128 * WM pixel rows x 64 B
    constexpr int GS = 64 * WM;                      // threads of a K-group
    constexpr int AUXW = WM == 1 ? 2 : 0;            // nt on weights only one wave reads
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, kg = wave / WM;
    const int r = lane & 15, g = lane >> 4;
    const int strip = blockIdx.x, split = blockIdx.y;
    const int ncb = p.ws_ncb;
    const int cb0 = split * p.ws_cpb, cb1 = min(ncb, cb0 + p.ws_cpb);
    const int nit = (p.dbg & 16) ? 0 : (cb1 - cb0 + WK - 1) / WK;       // iterations of every K-group (a group past the range streams zeros)
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.W), 0, p.w_bytes, 0x00020000);

    // zero fragments: lanes whose tap lies outside the image read 16 B at 16 g + 1024 i -- the same immediate offsets as the slab reads
    if (tid < 128) reinterpret_cast<uint32_t*>(smem + 1024 * (tid >> 4))[tid & 15] = 0u;
    const uint32_t slab_off = (uint32_t)WS_ZERO + (uint32_t)kg * 2u * SLAB;

    // ---- per 16-row tile i (rows m = 128 wm + 16 i + r): one byte of flags -- 1 top row, 2 bottom row, 4 left column, 8 right column
    // of the (upsampled) image, 16 row past M; a tap is outside the image iff its static mask meets the flags. Without the upsample
    // the conv keeps the map size (3 x 3, pad 1, stride 1: host-checked) and tap (dy, dx) of row m is slab row m + (dy-1) Win + dx-1.
    uint32_t fl[2] = {0u, 0u};
    int uyx[UPS ? 8 : 1];        // UPS: (oy - 1) | (ox - 1) << 8 (signed bytes: maps of <= 512 pixels) | image << 16
    {
        const float inv_hw = 1.f / (float)(p.Hout * p.Wout), inv_w = 1.f / (float)p.Wout;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = wm * 128 + 16 * i + r;
            const int mc = m < p.M ? m : 0;
            const int b = (int)(((float)mc + 0.5f) * inv_hw), rem = mc - b * (p.Hout * p.Wout);     // exact for these sizes (< 2^20)
            const int oy = (int)(((float)rem + 0.5f) * inv_w), ox = rem - oy * p.Wout;
            const uint32_t f = (oy == 0 ? 1u : 0u) | (oy == p.Hout - 1 ? 2u : 0u) | (ox == 0 ? 4u : 0u) | (ox == p.Wout - 1 ? 8u : 0u) |
                               (m < p.M ? 0u : 16u);
            fl[i >> 2] |= f << (8 * (i & 3));
            if (UPS) uyx[i] = ((oy - 1) & 0xff) | (((ox - 1) & 0xff) << 8) | (b << 16);
        }
    }

    // ---- slab of channel block cb: 8 chunks of 16 B per thread of the K-group, global -> registers -> LDS, in two halves (the
    // staging registers of a half are live for half an iteration)
    const int tg = wm * 64 + lane;
    u32x4 sreg[4];
    auto slab_load = [&](int it, int half) {
        const int cb = cb0 + kg + it * WK;
        const uint32_t cinv = (uint32_t)((cb1 - 1 - cb) >> 31) | ((p.dbg & 4) ? 0xFFFFFFFFu : 0u);     // dbg 4: no slab traffic
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = tg + (half * 4 + j) * GS, row = c >> 2, ch = c & 3;
            const uint32_t rinv = (uint32_t)((p.ws_rows - 1 - row) >> 31);
            const uint32_t off = ((uint32_t)row * (uint32_t)p.lda + (uint32_t)cb * 32u + (uint32_t)ch * 8u) * 2u;
            sreg[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, off | cinv | rinv, 0, 0));
        }
    };
    auto slab_store = [&](int buf, int half) {
        char* dst = smem + slab_off + buf * SLAB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t c = (uint32_t)(tg + (half * 4 + j) * GS), row = c >> 2, ch = c & 3u;
            *reinterpret_cast<u32x4*>(dst + row * 64u + ((ch ^ ws_swz(row)) << 4)) = sreg[j];
        }
    };
    // ---- W pieces of (iteration it, tap t): one 16-byte load per lane and column tile
    u32x4 wq[T][NTW];
    const uint32_t piece0 = (uint32_t)(strip * NTW) * (uint32_t)(ncb * T);
    auto w_load = [&](int it, int t) {
        const int cb = cb0 + kg + it * WK;
        const uint32_t cinv = (uint32_t)((cb1 - 1 - cb) >> 31) | ((p.dbg & 2) ? 0xFFFFFFFFu : 0u);     // dbg 2: no W stream (tuning aid)
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const uint32_t off = (piece0 + (uint32_t)n * (uint32_t)(ncb * T) + (uint32_t)(cb * T + t)) * 1024u + (uint32_t)lane * 16u;
            wq[t][n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, off | cinv, 0, AUXW));
        }
    };

    f32x4 acc[8][NTW];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    {   // first slab: both halves in flight ahead of the W pieces (loads return in order: the slab must not queue behind 18 KiB of W)
        u32x4 s0[4];
        slab_load(0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) s0[j] = sreg[j];
        slab_load(0, 1);
#pragma unroll
        for (int t = 0; t < T; ++t) w_load(0, t);
        __syncthreads();                    // the zero fragments are written
        slab_store(0, 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) sreg[j] = s0[j];
        slab_store(0, 0);
    }
    if (WM > 1) __syncthreads();

    const int Win = p.Win, hw_in = p.Hin * p.Win;
    int row00 = wm * 128 + r - Win - 1;                 // slab row of tap (0, 0) of tile 0
    const uint32_t zoff = (uint32_t)g * 16u;
    // one channel block; PF: the next one is prefetched behind it (the last iteration issues no loads at all: every 16-byte wave load
    // costs 16 cycles of the CU's address path whether or not its offset is in range)
    auto iter = [&](int it, auto pf_tag) {
        constexpr bool PF = decltype(pf_tag)::value;
        const uint32_t cur_off = slab_off + (uint32_t)((it & 1) * SLAB);
        // (the per-tap addresses and masks do not depend on `it`: keep them from being hoisted out of this loop into ~100 registers)
        asm volatile("" : "+v"(row00), "+v"(fl[0]), "+v"(fl[1]));
        if (UPS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(uyx[i]));
        }
        if (PF) slab_load(it + 1, 0);   // (a K-group past the range streams zeros: the load counters stay exact)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int dy = t / 3, dx = t % 3;
            constexpr uint32_t M4 = 0x01010101u;
            if (p.dbg & 1) {                // dbg 1: loads only (the W pieces are kept alive, nothing is read from LDS or multiplied)
#pragma unroll
                for (int n = 0; n < NTW; ++n) asm volatile("" ::"v"(wq[t][n]));
                if (PF) {
                    w_load(it + 1, t);
                    if (t == 4) { slab_store((it + 1) & 1, 0); slab_load(it + 1, 1); }
                }
                continue;
            }
            const uint32_t tmask = (16u | (dy == 0 ? 1u : 0u) | (dy == 2 ? 2u : 0u) | (dx == 0 ? 4u : 0u) | (dx == 2 ? 8u : 0u)) * M4;
            h16x8 xf[8];
            if (UPS) {      // (4 tiles at a time: the per-tile source coordinates cost registers)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const int i = hh * 4 + ii;
                        const uint32_t bad = fl[i >> 2] & (tmask & (0xffu << (8 * (i & 3))));
                        const int uy = ((int)(signed char)(uyx[i] & 0xff) + dy) >> 1, ux = ((int)(signed char)((uyx[i] >> 8) & 0xff) + dx) >> 1;
                        const uint32_t row = (uint32_t)((uyx[i] >> 16) * hw_in + uy * Win + ux);
                        const uint32_t a_ok = cur_off + row * 64u + (((uint32_t)g ^ ws_swz(row)) << 4);
                        xf[ii] = *reinterpret_cast<const h16x8*>(smem + (bad ? zoff : a_ok));
                    }
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                        for (int n = 0; n < NTW; ++n)
                            acc[hh * 4 + ii][n] = mfma_16x16x32_h16(__builtin_bit_cast(h16x8, wq[t][n]), xf[ii], acc[hh * 4 + ii][n]);
                    __builtin_amdgcn_sched_barrier(0);      // (left alone, the scheduler hoists every tile's address arithmetic: spills)
                }
            } else {
                // tile i reads slab row (row_t + 16 i): the swizzle term depends on bits 2-3 of the row, which + 16 i leaves alone
                const uint32_t row_t = (uint32_t)(row00 + dy * Win + dx);
                const uint32_t base_t = cur_off + row_t * 64u + (((uint32_t)g ^ ws_swz(row_t)) << 4);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t bad = fl[i >> 2] & (tmask & (0xffu << (8 * (i & 3))));
                    const uint32_t sel = bad ? zoff : base_t;
                    xf[i] = *reinterpret_cast<const h16x8*>(smem + sel + 1024 * i);
                }
            }
            if (!UPS) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int n = 0; n < NTW; ++n)
                        acc[i][n] = mfma_16x16x32_h16(__builtin_bit_cast(h16x8, wq[t][n]), xf[i], acc[i][n]);
            }
            if (PF) {
                w_load(it + 1, t);          // the same slot, one channel block ahead
                if (t == 4) { slab_store((it + 1) & 1, 0); slab_load(it + 1, 1); }
            }
        }
        if (PF) {
            slab_store((it + 1) & 1, 1);
            if (WM > 1) __syncthreads();
        }
    };
    for (int it = 0; it + 1 < nit; ++it) iter(it, std::true_type{});
    if (nit > 0) iter(nit - 1, std::false_type{});

    // ---- K-groups summed through LDS in group order; the finalizer of tile i is K-group i / TPF
    if (p.dbg & 8) {                    // tuning aid: no combine, no output
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(acc[i][0]), "v"(acc[i][1]));
        return;
    }
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem + WS_ZERO);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int n = 0; n < NTW; ++n)
            red[((((wm * 8 + i) * WK + kg) * NTW + n) << 6) + lane] = acc[i][n];
    __syncthreads();
    f32x4 fin[TPF][NTW];
#pragma unroll
    for (int j = 0; j < TPF; ++j)
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < WK; ++kk) v += red[((((wm * 8 + kg * TPF + j) * WK + kk) * NTW + n) << 6) + lane];
            fin[j][n] = v;
        }
    if (p.splits == 1 || p.ws_cnt == nullptr) {       // no split, or partial slabs for splitk_reduce_kernel
        write_out<TPF, NTW, 0>(p, fin, wm * 128 + kg * TPF * 16, strip * (16 * NTW), split, lane);
        return;
    }

    // ---- in-launch split-K combine. Publish: the partial tile goes out with WRITE-THROUGH (sc1) 16-byte stores, every wave drains
    // them (vmcnt(0)), one barrier, then ONE lane adds to the strip's arrival counter -- no release fence (a fence per block writes the
    // whole L2 back: measured + 6 us at 128 rows, + 21 us at 512). The block whose add came last reads the slabs of ALL splits with
    // sc1 loads (served past L1 / the local L2's stale lines) in split order -- the same sum whichever block is last -- and applies the
    // epilogue. The counter is zero whenever no launch is in flight: the last arriver resets it. (cdna guide, Guideline 16 R1 /
    // MI355X_MICROARCH hand-off table row 1: one lane per storing workgroup adds after every wave's drain + a barrier; the other
    // waves of the reducer load behind a barrier that the adding wave joins.)
    const uint32_t ws_bytes_used = (uint32_t)((size_t)p.splits * p.M * p.N * sizeof(float));
    const __amdgpu_buffer_rsrc_t rsrc_ws = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, ws_bytes_used, 0x00020000);
    typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(rsrc_ws, 0, 0, 0)) vec_t;
    {
        const int mb = wm * 128 + kg * TPF * 16, nb = strip * (16 * NTW);
#pragma unroll
        for (int j = 0; j < TPF; ++j)
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int m = mb + j * 16 + (lane & 15), nn = nb + n * 16 + (lane >> 4) * 4;
                const uint32_t inv = (uint32_t)(((p.M - 1 - m) | (p.N - 4 - nn)) >> 31);
                const uint32_t off = (uint32_t)(((size_t)split * p.M + m) * p.N + nn) * 4u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vec_t, fin[j][n]), rsrc_ws, off | inv, 0, 16);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t* lastflag = reinterpret_cast<uint32_t*>(smem + 64);          // LDS word outside the zero fragments
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.ws_cnt + strip, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == (unsigned)p.splits - 1u;
        if (last) __hip_atomic_store(p.ws_cnt + strip, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lastflag = last ? 1u : 0u;
    }
    __syncthreads();
    if (*lastflag == 0u) return;
    {
        const EpiRsrc er = make_epi_rsrc(p);
        const int q = tid & 7, n = strip * (16 * NTW) + 4 * q;            // 8 column quads of the 32-column strip
        const uint32_t slab_bytes = (uint32_t)((size_t)p.M * p.N * sizeof(float));
        // every slab load of a batch of rows is in flight at once (a dependent loop of sc1 loads pays a memory round trip per step:
        // measured + 4 us at 128 rows, + 18 us at 512); at most 8 splits (host-checked), slabs past the split count read as zeros
        constexpr int RB = WM == 1 ? 2 : 4;
#pragma unroll
        for (int jb = 0; jb < 2 * WM; jb += RB) {
            f32x4 t8[RB][8];
#pragma unroll
            for (int jj = 0; jj < RB; ++jj) {
                const int m = (tid >> 3) + 64 * (jb + jj);
                const uint32_t inv = (uint32_t)(((p.M - 1 - m) | (p.N - 4 - n)) >> 31);
                const uint32_t off0 = (uint32_t)((size_t)m * p.N + n) * 4u;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t sinv = (uint32_t)((p.splits - 1 - u) >> 31);
                    t8[jj][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_ws, (off0 + (uint32_t)u * slab_bytes) | inv | sinv, 0, 16));
                }
            }
#pragma unroll
            for (int jj = 0; jj < RB; ++jj) {
                const int m = (tid >> 3) + 64 * (jb + jj);
                float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) { v[0] += t8[jj][u][0]; v[1] += t8[jj][u][1]; v[2] += t8[jj][u][2]; v[3] += t8[jj][u][3]; }
                uint32_t rb_row = 0;
                if (p.rowbias) rb_row = (uint32_t)((m < p.M ? m : 0) / p.rows_per_group) * (uint32_t)p.N * 2u;
                epilogue_fast<false>(p, er, m, n, rb_row, v);
            }
        }
    }
}

// {mean, rstd} of every row of A [M, K] (K % 8 == 0): one wave per row, fp32 sums -- the statistics the LN instantiations of
// gemm_kernel accumulate while staging A (same formulas: var = max(E[x^2] - mean^2, 0)), for the kernel that stages by DMA
__global__ __launch_bounds__(256) void ln_row_stats_kernel(const h16_t* __restrict__ A, float2* __restrict__ out, int M, int K, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const u32x4* a = reinterpret_cast<const u32x4*>(A + (size_t)row * K);
    float s_ = 0.f, q_ = 0.f;
    for (int c = lane; c < K / 8; c += 64) {
        const u32x4 v = a[c];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float lo = h16lo_to_f32(w[d]), hi = h16hi_to_f32(w[d]);
            s_ += lo + hi;
            q_ = fmaf(lo, lo, fmaf(hi, hi, q_));
        }
    }
    s_ = wave_sum(s_);
    q_ = wave_sum(q_);
    if (lane == 0) {
        const float mean = s_ / (float)K;
        const float var = fmaxf(q_ / (float)K - mean * mean, 0.f);
        out[row] = float2{mean, rsqrtf(var + eps)};
    }
}

// split-K: sum the fp32 slabs and apply the epilogue; one thread per 4 consecutive columns
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs p) {
    const int n4 = (p.N + 3) / 4;
    const size_t total = (size_t)p.M * n4;
    const size_t slab = (size_t)p.M * p.N;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int m = (int)(idx / n4), n = (int)(idx % n4) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const float* src = p.ws + (size_t)m * p.N + n;
        if (n + 3 < p.N) {
            // slabs four at a time with INDEPENDENT loads (a one-load-per-iteration loop serialises the round trips); the tail
            // re-reads the last slab with weight 0 instead of branching
            for (int s0 = 0; s0 < p.splits; s0 += 4) {
                f32x4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const f32x4*>(src + (size_t)min(s0 + u, p.splits - 1) * slab);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float w = s0 + u < p.splits ? 1.f : 0.f;
                    v[0] += w * t[u][0]; v[1] += w * t[u][1]; v[2] += w * t[u][2]; v[3] += w * t[u][3];
                }
            }
        } else {
            for (int s = 0; s < p.splits; ++s)
                for (int e = 0; e < 4 && n + e < p.N; ++e) v[e] += src[(size_t)s * slab + e];
        }
        epilogue_store<EPI>(p, m, n, v);
    }
}

// split-K reduce + epilogue + producer-side GroupNorm statistics (GemmArgs::gn_part): one block per (16 output rows, slab of `cs`
// columns = whole groups); thread -> (row r of 16, column quad q). All slab loads of a thread are in flight at once (<= 16 splits),
// then the 16 rows of a column and the columns of a group are summed through LDS in a fixed order. A chunk of gn_cr = 16 rows is one
// block; with gn_cr = 64 four row blocks would share a chunk, so this kernel is launched with gn_cr == 16 only (host-checked).
template <int EPI>
__global__ void splitk_reduce_gn_kernel(GemmArgs p, int cs) {
    extern __shared__ __attribute__((aligned(16))) char smem_r[];
    float2* sm = reinterpret_cast<float2*>(smem_r);          // [16][cs], then [cs] column totals
    float2* colsum = sm + 16 * cs;
    const int nq = cs / 4;
    const int q = threadIdx.x % nq, r = threadIdx.x / nq;    // r < 16
    const int n_slabs = (p.N + cs - 1) / cs;
    const int chunk = blockIdx.x / n_slabs, slab = blockIdx.x - chunk * n_slabs;
    const int n = slab * cs + q * 4, m = chunk * 16 + r;
    const size_t slab_elems = (size_t)p.M * p.N;
    float2 o[4] = {float2{0.f, 0.f}, float2{0.f, 0.f}, float2{0.f, 0.f}, float2{0.f, 0.f}};
    if (n < p.N && m < p.M) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const float* src = p.ws + (size_t)m * p.N + n;
        f32x4 t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = *reinterpret_cast<const f32x4*>(src + (size_t)min(u, p.splits - 1) * slab_elems);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float w = u < p.splits ? 1.f : 0.f;
            v[0] += w * t[u][0]; v[1] += w * t[u][1]; v[2] += w * t[u][2]; v[3] += w * t[u][3];
        }
        epilogue_store<EPI>(p, m, n, v);          // leaves the final fp32 values in v
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float rv = h16_to_f32(f32_to_h16(v[e]));
            o[e] = float2{rv, rv * rv};
        }
    }
    if (n < p.N) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sm[r * cs + q * 4 + e] = o[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cs; c += blockDim.x) {
        float ss = 0.f, qq = 0.f;
        if (slab * cs + c < p.N) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { const float2 t2 = sm[k * cs + c]; ss += t2.x; qq += t2.y; }
        }
        colsum[c] = float2{ss, qq};
    }
    __syncthreads();
    const int cpg = p.gn_cpg, gps = cs / cpg;
    for (int gi = threadIdx.x; gi < gps; gi += blockDim.x) {
        const int ncol = slab * cs + gi * cpg;
        if (ncol >= p.N) continue;
        float ss = 0.f, qq = 0.f;
        for (int c = 0; c < cpg; ++c) { const float2 t2 = colsum[gi * cpg + c]; ss += t2.x; qq += t2.y; }
        float* dst = p.gn_part + ((size_t)chunk * p.gn_G + ncol / cpg) * 2;
        dst[0] = ss;
        dst[1] = qq;
    }
}

#ifdef SPIDER_WS_DEV
// development aid (scripts/exp/ws_isa.sh): only the streaming kernel is instantiated, for a quick look at its ISA
template __global__ void wstream_kernel<1, 8, false>(GemmArgs);
template __global__ void wstream_kernel<4, 2, false>(GemmArgs);
template __global__ void wstream_kernel<4, 2, true>(GemmArgs);
}  // namespace
#else
template <int BM, int BN>
void launch_tile(const GemmArgs& a, int tiles, hipStream_t st) {
    const size_t smem = (size_t)2 * ((a.a32 ? 2 : 1) * BM + BN) * LDS_STRIDE * sizeof(h16_t);
    dim3 grid(tiles, a.splits);
    // epilogue instantiation: 2 GEGLU, 0/1 branch-free bf16 (without / with activation), 3 general (see epilogue_store)
    const bool fast_ok = a.C && !a.C32 && a.N % 4 == 0 && a.c_bytes != 0;
    const int epi = a.geglu ? 2 : (!fast_ok ? 3 : (a.act ? 1 : 0));
    if (a.a32) {           // fp32 A operand, hi / lo split (host-checked: epilogue 0 / GEGLU of the LN form)
        static unsigned done_a = 0, done_b = 0, done_c = 0, done_d = 0, done_e = 0;
        if (a.ln_colsum) {      // LayerNorm on the fp32 A operand, gamma applied to A (table behind the tiles), exact W
            const size_t extra = (size_t)a.K * sizeof(float2);
            if (a.geglu) {
                raise_dynamic_lds(&gemm_kernel<BM, BN, false, 2, true, false, true>, (int)(smem + extra), done_d);
                gemm_kernel<BM, BN, false, 2, true, false, true><<<grid, 256, smem + extra, st>>>(a);
            } else {
                raise_dynamic_lds(&gemm_kernel<BM, BN, false, 0, true, false, true>, (int)(smem + extra), done_e);
                gemm_kernel<BM, BN, false, 0, true, false, true><<<grid, 256, smem + extra, st>>>(a);
            }
        } else if (a.gna_part) {
            if constexpr (BM == 64 && BN == 64) {
                const size_t extra = (size_t)a.K * sizeof(float2) + (size_t)2 * a.gna_G * sizeof(float) + 2 * 256 * sizeof(float);
                raise_dynamic_lds(&gemm_kernel<64, 64, false, 0, false, true, true>, (int)(smem + extra), done_c);
                gemm_kernel<64, 64, false, 0, false, true, true><<<grid, 256, smem + extra, st>>>(a);
            }
        } else if (a.conv) {
            raise_dynamic_lds(&gemm_kernel<BM, BN, true, 0, false, false, true>, (int)smem, done_a);
            gemm_kernel<BM, BN, true, 0, false, false, true><<<grid, 256, smem, st>>>(a);
        } else {
            raise_dynamic_lds(&gemm_kernel<BM, BN, false, 0, false, false, true>, (int)smem, done_b);
            gemm_kernel<BM, BN, false, 0, false, false, true><<<grid, 256, smem, st>>>(a);
        }
        return;
    }
    if (a.gna_part) {      // GroupNorm applied to A on the way into LDS (host-checked: plain linear, 64^2 tiles, epilogue 0, no split-K)
        if constexpr (BM == 64 && BN == 64) {
            const size_t extra = (size_t)a.K * sizeof(float2) + (size_t)2 * a.gna_G * sizeof(float) + 2 * 256 * sizeof(float);
            static unsigned done_g = 0;
            raise_dynamic_lds(&gemm_kernel<64, 64, false, 0, false, true>, (int)(smem + extra), done_g);     // K > ~3.8k passes 64 KiB
            gemm_kernel<64, 64, false, 0, false, true><<<grid, 256, smem + extra, st>>>(a);
        }
        return;
    }
    if (a.ln_colsum) {     // LayerNorm-folded linears (checked by the caller: bf16 out, N % 4 == 0, no activation, no split-K)
        if (epi == 2) gemm_kernel<BM, BN, false, 2, true><<<grid, 256, smem, st>>>(a);
        else gemm_kernel<BM, BN, false, 0, true><<<grid, 256, smem, st>>>(a);
        return;
    }
    if (a.conv) {
        if (epi == 3) gemm_kernel<BM, BN, true, 3><<<grid, 256, smem, st>>>(a);
        else if (epi == 1) gemm_kernel<BM, BN, true, 1><<<grid, 256, smem, st>>>(a);
        else gemm_kernel<BM, BN, true, 0><<<grid, 256, smem, st>>>(a);
    } else {
        if (epi == 3) gemm_kernel<BM, BN, false, 3><<<grid, 256, smem, st>>>(a);
        else if (epi == 2) gemm_kernel<BM, BN, false, 2><<<grid, 256, smem, st>>>(a);
        else if (epi == 1) gemm_kernel<BM, BN, false, 1><<<grid, 256, smem, st>>>(a);
        else gemm_kernel<BM, BN, false, 0><<<grid, 256, smem, st>>>(a);
    }
}

// LDS-DMA kernel launch: BN = 160 or 128, NS-stage ring in dynamic LDS (> 64 KiB: raised once per instantiation)
template <int BN, int NS, bool CONV, int EPI, int BM = 128>
void launch_dma_inst(const GemmArgs& a, dim3 grid, hipStream_t st) {
    constexpr int smem = NS * (BM + BN) * 128;
    static unsigned done = 0;
    raise_dynamic_lds(&gemm_dma_kernel<BN, NS, CONV, EPI, BM>, smem, done);
    gemm_dma_kernel<BN, NS, CONV, EPI, BM><<<grid, 512, smem, st>>>(a);
}

template <bool CONV, int EPI, bool LN = false>
void launch_p8_inst(const GemmArgs& a, dim3 grid, hipStream_t st) {
    constexpr int smem = 2 * 4 * 128 * 128;
    static unsigned done = 0;
    raise_dynamic_lds(&gemm_p8_kernel<CONV, EPI, LN>, smem, done);
    gemm_p8_kernel<CONV, EPI, LN><<<grid, 512, smem, st>>>(a);
}

void launch_p8(const GemmArgs& a, hipStream_t st) {
    const int ncols = a.geglu ? 2 * a.N : a.N;
    dim3 grid(((a.M + 255) / 256) * ((ncols + 255) / 256), a.splits);
    const bool fast_ok = a.C && !a.C32 && a.N % 4 == 0 && a.c_bytes != 0;
    const int epi = !fast_ok ? 3 : (a.act ? 1 : 0);
    if (a.ln_colsum) {      // LayerNorm-folded linears: row statistics from ln_row_stats_kernel (launched by the caller)
        if (a.geglu) launch_p8_inst<false, 2, true>(a, grid, st);
        else if (epi == 3) launch_p8_inst<false, 3, true>(a, grid, st);
        else launch_p8_inst<false, 0, true>(a, grid, st);
        return;
    }
    if (a.geglu) { launch_p8_inst<false, 2>(a, grid, st); return; }
    if (a.conv) {
        if (epi == 3) launch_p8_inst<true, 3>(a, grid, st);
        else if (epi == 1) launch_p8_inst<true, 1>(a, grid, st);
        else launch_p8_inst<true, 0>(a, grid, st);
    } else {
        if (epi == 3) launch_p8_inst<false, 3>(a, grid, st);
        else if (epi == 1) launch_p8_inst<false, 1>(a, grid, st);
        else launch_p8_inst<false, 0>(a, grid, st);
    }
}

template <bool CONV, int EPI>
void launch_p8h_inst(const GemmArgs& a, dim3 grid, hipStream_t st) {
    constexpr int smem = 3 * 3 * 128 * 128;      // 144 KiB
    static unsigned done = 0;
    raise_dynamic_lds(&gemm_p8h_kernel<CONV, EPI>, smem, done);
    gemm_p8h_kernel<CONV, EPI><<<grid, 512, smem, st>>>(a);
}

void launch_p8h(const GemmArgs& a, hipStream_t st) {
    dim3 grid(((a.M + 255) / 256) * ((a.N + 127) / 128), a.splits);
    const bool fast_ok = a.C && !a.C32 && a.N % 4 == 0 && a.c_bytes != 0;
    const int epi = !fast_ok ? 3 : (a.act ? 1 : 0);
    if (a.conv) {
        if (epi == 3) launch_p8h_inst<true, 3>(a, grid, st);
        else if (epi == 1) launch_p8h_inst<true, 1>(a, grid, st);
        else launch_p8h_inst<true, 0>(a, grid, st);
    } else {
        if (epi == 3) launch_p8h_inst<false, 3>(a, grid, st);
        else if (epi == 1) launch_p8h_inst<false, 1>(a, grid, st);
        else launch_p8h_inst<false, 0>(a, grid, st);
    }
}

template <int BN, int NS, int BM>
void launch_dma_gn(const GemmArgs& a, dim3 grid, hipStream_t st) {
    constexpr int smem = NS * (BM + BN) * 128;
    static unsigned done = 0;
    raise_dynamic_lds(&gemm_dma_kernel<BN, NS, true, 0, BM, true>, smem, done);
    gemm_dma_kernel<BN, NS, true, 0, BM, true><<<grid, 512, smem, st>>>(a);
}

// can this launch of the 128/64 x 160 LDS-DMA kernel write the GroupNorm partials of its output in its epilogue?
bool dma_gn_ok(const GemmArgs& a) {
    const bool fast_ok = a.C && !a.C32 && a.N % 4 == 0 && a.c_bytes != 0;
    return a.gn_part && a.conv && a.splits == 1 && fast_ok && !a.act && a.gn_cr == 64 && a.gn_cpg > 0 && 80 % a.gn_cpg == 0 &&
           a.N % a.gn_cpg == 0 && a.M % 64 == 0 && a.rows_per_group % 64 == 0;
}

template <int BN, int NS, int BM = 128>
void launch_dma(const GemmArgs& a, int tiles, hipStream_t st) {
    dim3 grid(tiles, a.splits);
    const bool fast_ok = a.C && !a.C32 && a.N % 4 == 0 && a.c_bytes != 0;
    const int epi = !fast_ok ? 3 : (a.act ? 1 : 0);
    if constexpr (BN == 160 && (NS == 2 || NS == 3 || (NS == 4 && BM == 64))) {
        if (dma_gn_ok(a)) { launch_dma_gn<BN, NS, BM>(a, grid, st); return; }
    }
    if (a.conv) {
        if (epi == 3) launch_dma_inst<BN, NS, true, 3, BM>(a, grid, st);
        else if (epi == 1) launch_dma_inst<BN, NS, true, 1, BM>(a, grid, st);
        else launch_dma_inst<BN, NS, true, 0, BM>(a, grid, st);
    } else {
        if (epi == 3) launch_dma_inst<BN, NS, false, 3, BM>(a, grid, st);
        else if (epi == 1) launch_dma_inst<BN, NS, false, 1, BM>(a, grid, st);
        else launch_dma_inst<BN, NS, false, 0, BM>(a, grid, st);
    }
}

// Tile / split-K choice: fill >= ~2 blocks per CU (512) when the problem allows it, keep >= 4 K tiles per split.
int env_int(const char* name) {
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
}

// byte ranges of C / res and rowbias for the buffer-descriptor epilogue; 0 disables it (operands >= 2 GiB)
void set_epilogue_ranges(GemmArgs& a) {
    const size_t cb = (size_t)(a.M - 1) * a.ldc * 2 + (size_t)a.N * 2;
    const size_t rb = a.rowbias ? (size_t)((a.M + a.rows_per_group - 1) / a.rows_per_group) * a.N * 2 : 0;
    const bool ok = cb < ((size_t)1 << 31) && rb < ((size_t)1 << 31);
    a.c_bytes = ok ? (uint32_t)cb : 0;
    a.rb_bytes = ok ? (uint32_t)rb : 0;
}

uint32_t tiled_bytes(int N, int K) {
    return (uint32_t)((size_t)((N + 63) / 64) * ((K + BK - 1) / BK) * 8192);
}

// column slab of the statistics-producing split-K reduce: whole groups, a multiple of 4 columns, about 160 wide (0: not possible)
int gn_reduce_slab(int cpg, int target) {
    int k = target / cpg;
    if (k < 1) k = 1;
    while (k > 1 && (cpg * k) % 4 != 0) --k;
    const int cs = cpg * k;
    return (cs % 4 == 0 && cs / 4 <= 64) ? cs : 0;       // 16 rows x (cs / 4) column quads <= 1024 threads
}

// gn_done (optional): set to true when the launch wrote GemmArgs::gn_part (the caller otherwise runs a statistics pass)
template <int WM, int WK, bool UPS>
void launch_ws_inst(const GemmArgs& a, dim3 grid, hipStream_t st) {
    constexpr int smem = WS_ZERO + 16 * 8192;
    static unsigned done = 0;
    raise_dynamic_lds(&wstream_kernel<WM, WK, UPS>, smem, done);
    wstream_kernel<WM, WK, UPS><<<grid, 512, smem, st>>>(a);
}

// is this problem one the weight-stationary streaming kernel serves? (mirrored by ops._ws_eligible: the caller packs W for it)
bool ws_eligible(const GemmArgs& a) {
    const bool fast_ok = a.C && !a.C32 && a.N % 4 == 0 && a.c_bytes != 0;
    if (!a.conv || a.kh != 3 || a.kw != 3 || a.stride != 1 || a.dil != 1 || a.pad_h != 1 || a.pad_w != 1 || a.Cin % 32 != 0 || a.act ||
        a.geglu || a.ln_colsum || !fast_ok)
        return false;
    const int rows = a.M / (a.Hout * a.Wout) * a.Hin * a.Win;
    if (a.M <= 128) return rows <= 128 && !a.ups;
    return a.M <= 512 && rows <= 512;
}

int launch(GemmArgs a, long ws_bytes, void* stream, int* gn_done = nullptr) {
    hipStream_t st = (hipStream_t)stream;
    if (gn_done) *gn_done = 0;
    static const int dbg = env_int("SPIDER_GEMM_DBG");
    a.dbg = dbg;
    static const int epi_lds_env = getenv("SPIDER_EPI_LDS") ? atoi(getenv("SPIDER_EPI_LDS")) : 1;      // tuning aid: 0 = fragment-order stores, 2 = on every LDS-DMA launch
    if (a.w_tiled == 2) {
        // weight-stationary streaming conv: strips of 32 columns x K splits of whole 32-channel blocks, ~200-256 blocks of 8 waves
        const int WK = a.M <= 128 ? 8 : 2;
        a.ws_ncb = a.Cin / 32;
        a.ws_rows = a.M / (a.Hout * a.Wout) * a.Hin * a.Win;
        const int strips = (a.N + 31) / 32;
        static const int ws_blocks = env_int("SPIDER_WS_BLOCKS") ? env_int("SPIDER_WS_BLOCKS") : 256;
        int S = (ws_blocks + strips / 2) / strips;
        if (S < 1) S = 1;
        int cpb = (a.ws_ncb + S - 1) / S;
        cpb = (cpb + WK - 1) / WK * WK;                 // every K-group runs the same number of channel blocks
        S = (a.ws_ncb + cpb - 1) / cpb;
        if (!a.ws) { S = 1; cpb = a.ws_ncb; }
        while (S > 1 && (size_t)S * a.M * a.N * sizeof(float) > (size_t)ws_bytes) { cpb += WK; S = (a.ws_ncb + cpb - 1) / cpb; }
        a.ws_cpb = cpb;
        a.splits = S;
        a.kt_per_split = 0;
        // in-launch combine by the last-arriving block of a strip: its arrival counters are the 4096 bytes BEHIND the declared
        // workspace (zero when idle; contract of w_tiled = 2, see spider_hip.h)
        if (g_spider_ws_inlaunch < 0) g_spider_ws_inlaunch = getenv("SPIDER_WS_INLAUNCH") ? (atoi(getenv("SPIDER_WS_INLAUNCH")) != 0) : 1;
        a.ws_cnt = (g_spider_ws_inlaunch && S > 1 && S <= 8 && strips <= 1024) ? reinterpret_cast<unsigned*>(reinterpret_cast<char*>(a.ws) + ws_bytes) : nullptr;
        dim3 grid(strips, S);
        if (a.M <= 128) launch_ws_inst<1, 8, false>(a, grid, st);
        else if (a.ups) launch_ws_inst<4, 2, true>(a, grid, st);
        else launch_ws_inst<4, 2, false>(a, grid, st);
        SPIDER_LAUNCH_OK();
        if (a.ws_cnt) return 0;                                   // combined in the launch: no reduce kernel, no statistics
        if (a.gn_part && a.splits == 1) a.gn_part = nullptr;      // no statistics epilogue here: the caller runs its own pass
    } else {
    static const int force_tile = env_int("SPIDER_GEMM_TILE"), force_splits = env_int("SPIDER_GEMM_SPLITS");  // tuning aid
    const int nk = (a.K + BK - 1) / BK;
    const int ncols = a.geglu ? 2 * a.N : a.N;
    const int t128 = ((a.M + 127) / 128) * ((ncols + 127) / 128);
    const int t64 = ((a.M + 63) / 64) * ((ncols + 63) / 64);
    // Measured on MI355X (scripts/bench_gemm.py): K-heavy problems (3x3 convs, down_proj) want 128^2 tiles plus
    // split-K up to ~768 blocks; everything else with < 384 big tiles (or a short K) is faster on 64^2 tiles (more
    // blocks, all resident), split only when even those leave CUs idle.
    bool small;
    int splits = 1;
    if (nk >= 44 && t128 >= 24) {
        small = false;
        splits = 768 / t128;                       // fill ~3 blocks per CU ...
        if (splits < nk / 32) splits = nk / 32;    // ... and never leave one block with hundreds of K tiles (down_proj)
        if (splits > 8) splits = 8;
        if (splits > nk / 8) splits = nk / 8;
    } else if (t128 >= 320) {
        // (re-measured with the per-epilogue instantiations, scripts/bench_gemm_k.py: once the 128^2 kernel's code fits
        // the instruction cache it wins for every K >= 320 at these sizes -- 8192x2560x320: 40.8 vs 45.1 us)
        small = false;
    } else {
        small = true;
        if (t64 < 256 && nk >= 64) {               // short K: the extra reduce launch costs more than it hides
            splits = (512 + t64 - 1) / t64;
            if (splits > 16) splits = 16;
            if (splits > nk / 4) splits = nk / 4;
        }
    }
    if (force_tile) small = force_tile == 64;
    // SPIDER_GEMM_TILE = 160 / 161 (4-stage ring) / 129 (128 x 128 DMA tile): force the LDS-DMA kernel (tuning aid)
    static const int nfast_env = getenv("SPIDER_GEMM_NFAST") ? atoi(getenv("SPIDER_GEMM_NFAST")) : -1;
    a.n_fast = nfast_env >= 0 ? nfast_env : (a.M > ncols ? 1 : 0);
    int dma_bn = (force_tile == 160 || force_tile == 161 || force_tile == 162) ? 160 : (force_tile == 129 ? 128 : (force_tile == 65 ? 64 : 0));
    bool dma_ns2 = force_tile == 162;          // 2-stage ring, two blocks per CU (linears only)
    // Measured on MI355X (scripts/bench_gemm.py with GEMM_CFGS): with >= 2048 rows and >= 16 K tiles the LDS-DMA kernel wins
    // (UNet convs at 64^2 / 32^2: 33 vs 46 us, 48 vs 70, 32 vs 43, 47 vs 64; at 16^2 with 8 K splits 30 vs 41, 48 vs 63); below that the register-staged
    // tiles (more, smaller blocks) stay ahead, and at M = 1536 (LLM prefill) the two tie. Split K to ~256 blocks = 1 per CU.
    // Only where N fills the 160-wide tiles (<= 8 % padding: the VAE's 128 / 256 / 512 channels would waste 25 %), and for
    // plain linears only with a long K (at K = 1280 the tower / FF projections measured slower on it).
    const int n160 = (a.N + 159) / 160;
    int dma_bm = 128;
    static const int bm64_env = getenv("SPIDER_GEMM_BM64") ? atoi(getenv("SPIDER_GEMM_BM64")) : 1;
    // (a long-K linear with 512..2047 rows also belongs here: the UNet's 16^2 ff2, 512 x 1280 x 5120, 27.3 -> 20.6 us with 8 K splits)
    const bool dma_rows = a.conv ? a.M >= 512 : (a.M >= 2048 || (a.M >= 512 && nk >= 64));
    if (!force_tile && !a.geglu && dma_rows && nk >= (a.conv && a.M >= 2048 ? 16 : 40) && n160 * 160 * 25 <= a.N * 27) {
        // 64-row tiles when the 128-row tiling gives at most ~half a wave of blocks (<= 160) AND K is short (< 64 tiles): the
        // 8192 x 320 x 2880 convs -> 256 tiles, no split-K (measured 29.7 vs 33.8 us); with a longer K the 16 x 80 wave tile's
        // lower MFMA density costs more than the slab round trip of 2 splits (640 -> 320 at 64^2: 55 vs 49 us)
        if (bm64_env && ((a.M + 127) / 128) * n160 <= 160 && a.M >= 1024 && nk < 64) dma_bm = 64;
        const int tdma = ((a.M + dma_bm - 1) / dma_bm) * n160;
        int ds = 1;
        if (nk >= 40) {
            ds = (256 + tdma / 2) / tdma;
            if (ds > 8) ds = 8;
            if (ds > nk / 8) ds = nk / 8;
            if (ds < 1) ds = 1;
        }
        // one block per CU: a grid just above a multiple of 256 leaves most of the chip idle in its last round (SDXL 24^2 convs:
        // 288 tiles = 2 rounds at 56 % -> 260 us vs 199 us on the 2-blocks-per-CU kernel; 96^2: 1152 tiles = 90 % -> 190 vs 212)
        const int blocks = tdma * ds, rounds = (blocks + 255) / 256;
        if (blocks <= 256 || blocks * 100 >= rounds * 256 * 85) {
            dma_bn = 160;
            splits = ds;
        }
    }
    // 256^2 tiles (gemm_p8_kernel) for the large problems. Measured on MI355X (scripts/bench_gemm.py, tile 256): LLM prefill
    // gate/up 474 -> 345 us (1.21 PFLOP/s), down 262 -> 179, o 76 -> 58, qkv 79 -> 72; SDXL story linears 78 -> 68 / 83 -> 65 /
    // 98 -> 74 us and its 48^2 / 24^2 convs 165 -> 153, 319 -> 303, 342 -> 307, 184 -> 167; it loses where the 256-wide tiles
    // pad N (N = 320: 0.625 of the tile area used, 189 -> 235 us) or fewer than ~160 blocks exist (4608 x 1280 x 1280: 29 -> 32).
    static const int p8_env = getenv("SPIDER_GEMM_P8") ? atoi(getenv("SPIDER_GEMM_P8")) : 1;
    bool use_p8 = false;
    int p8_splits = 1, p8_rounds = 1;
    const bool p8_fused = a.geglu || a.ln_colsum;     // GEGLU / LayerNorm-folded forms: no split-K on this kernel
    const size_t ln_rows_bytes = (size_t)((a.M + 255) / 256) * 256 * sizeof(float2);
    if (p8_env && !force_tile && !force_splits && nk >= 8 && !(a.geglu && a.conv) &&
        (!a.ln_colsum || (a.ws && ln_rows_bytes <= (size_t)ws_bytes && a.act == 0))) {
        const long t256 = (long)((a.M + 255) / 256) * ((ncols + 255) / 256);
        if ((double)a.M * ncols >= 0.75 * 65536.0 * (double)t256) {
            int s = 1;
            if (t256 < 160 && a.ws && !p8_fused) {
                s = (int)((192 + t256 - 1) / t256);
                if (s > 3) s = 3;     // more splits: the slab round trip wins back nothing (2048 x 640 x 11520 conv, 8 splits: 58 vs 48 us)
                if (s > nk / 16) s = nk / 16;
                if (s < 1) s = 1;
                while (s > 1 && (size_t)s * a.M * a.N * sizeof(float) > (size_t)ws_bytes) --s;
            }
            if (t256 * s >= 160) { use_p8 = true; p8_splits = s; p8_rounds = (int)((t256 * s + 255) / 256); }
        }
    }
    // 256 x 128 tiles (gemm_p8h_kernel) for the linears that leave the 256^2 grid under-filled: measured (us, best other kernel ->
    // this one) SDXL 24^2 to_out 4608 x 1280 x 1280 28.5 -> 21.9, ff2 74.3 -> 64.2, qkv 66.7 -> 61.5, LLM prefill qkv 71.3 -> 58.8,
    // o 58.0 -> 48.8; it loses where the 256^2 grid fills its rounds (18432-row SDXL linears, gate/up), on long K that the 256^2
    // kernel splits (down projection) and on every conv. Cost model: rounds of 256 blocks, a 256 x 128 round = 0.625 of a 256^2 one.
    bool use_p8h = false;
    static const int p8h_env = getenv("SPIDER_GEMM_P8H") ? atoi(getenv("SPIDER_GEMM_P8H")) : 1;
    if (p8_env && p8h_env && !force_tile && !force_splits && !p8_fused && !a.conv && nk >= 5) {
        const long th = (long)((a.M + 255) / 256) * ((a.N + 127) / 128);
        const int rounds_h = (int)((th + 255) / 256);
        const bool fits = (double)a.M * a.N >= 0.8 * 32768.0 * (double)th && th >= 160 && th <= 768;   // <= 3 rounds: the measured range
        if (fits) {
            if (!use_p8) use_p8h = true;
            else if (p8_splits == 1) use_p8h = 0.625 * rounds_h < (double)p8_rounds;
            else use_p8h = nk < 128;                 // the 256^2 kernel would split K: worth it only for a long K
        }
    }
    // Very tall problems (UNet3D 16-frame maps: 92160 / 23040 rows and multiples; SDXL 96^2 / 48^2 maps): every kernel runs whole
    // rounds of one block per CU (two for the 128^2 register-staged tiles), so the choice is made by a cost model instead of the
    // size classes above:  time = ceil(tiles / resident blocks) x (t0 + nk x tk),  t0 = prologue + epilogue of a block, tk = one
    // 64-deep K tile. Constants fitted on MI355X (scripts/bench_gemm.py, SHAPES=v3d / sdxl, 16 shapes x 4 kernels, within ~10 %):
    // 256^2 8.4 + 1.40 nk (convs 1.47; 1.67 on the general address path), 256x128 7.0 + 0.78 nk, 128x160 LDS-DMA 5.5 + 0.68 nk, 128^2 5.7 + 1.0 nk at two blocks
    // per CU; the GEGLU epilogue adds ~2.2 us per block, the LayerNorm-folded form on the 256^2 kernel its row-statistics pass.
    // What the classes got wrong there: 270 tiles of 256^2 = two rounds at 53 % (23040 x 640 convs: 318 vs 201 us), and the K = 320
    // linears kept off the 256^2 kernel by its nk >= 8 rule (92160 x 960 x 320: 123 vs 93 us; GEGLU ff1 390 vs ~270).
    static const int model_env = getenv("SPIDER_GEMM_MODEL") ? atoi(getenv("SPIDER_GEMM_MODEL")) : 2;
    const bool fused = a.geglu || a.ln_colsum;
    if (model_env && (model_env >= 2 || !fused) && !force_tile && !force_splits && a.M >= 16384 && nk >= 4 && ncols >= 128) {
        const bool p8_ok = p8_env && !(a.geglu && a.conv) &&
                           (!a.ln_colsum || (a.ws && ln_rows_bytes <= (size_t)ws_bytes && a.act == 0));
        const float epi = a.geglu ? 2.2f : 0.f;
        const float ln_pass = a.ln_colsum ? 3.f + (float)((double)a.M * a.K * 2.0 / 5.0e6) : 0.f;   // ln_row_stats_kernel: one read of A
        // fp32 residual stream (res32 read / c32d written beside the 16-bit output): 4 + 4 extra bytes per output element leave in the
        // block's epilogue -- ~6 us per stream and 128 x 160 tile in fragment order (fitted: 92160 x 320 x 320 + both streams 126 us
        // on the LDS-DMA kernel against 49 us plain). Candidate 4 = the LDS-DMA kernel with a 2-stage ring, TWO blocks per CU and the
        // row-order epilogue through LDS (epilogue_lds): fitted on scripts/exp/lin_tiles.py (13 shapes), 9 us + 9 us per stream per
        // block, 1.06 us per K tile with two blocks sharing the CU (convs 1.2: scripts/exp/conv_tiles.py, tconv_tiles.py -- 92160 x 320
        // 3 x 3: 253 -> 207 us, (3,1,1): 120 -> 91, 65536 x 320 3 x 3: 148 -> 126); N must fill 160-wide tiles.
        static const int ns2_env = getenv("SPIDER_GEMM_NS2") ? atoi(getenv("SPIDER_GEMM_NS2")) : 1;
        const int nstream = (a.res32 ? 1 : 0) + (a.c32d ? 1 : 0);
        const bool n160_fit = n160 * 160 * 25 <= a.N * 27;
        // wider convs (N = 640): measured - 11 % (16384 rows: 512 tiles, one full round), - 5 % (23040: 1.4 rounds), + 24 % (18432: 1.125
        // rounds -- the 64 blocks past the full round cost a block time of their own): only with a full or well-filled last round
        const long ns2_tiles = (long)((a.M + 127) / 128) * n160;
        const bool ns2_tail_ok = ns2_tiles % 512 == 0 || ns2_tiles % 512 >= 205;
        struct Cand { int bm, bn, slots; float t0, tk, pre; bool ok; };
        const Cand cand[5] = {{256, 256, 256, 8.4f + epi, a.conv ? (a.hbits ? 1.47f : 1.67f) : 1.40f, ln_pass, p8_ok},
                              {256, 128, 256, 7.0f, a.conv ? 0.98f : 0.78f, 0.f, p8_env && p8h_env && !fused && (!a.conv || a.hbits)},
                              {128, 160, 256, 5.5f, 0.68f, 0.f, !fused},
                              {128, 128, 512, 5.7f + epi, 1.0f, 0.f, true},
                              {128, 160, 512, 9.0f + 9.0f * (float)nstream, a.conv ? 1.2f : 1.06f, 0.f,
                               ns2_env && !fused && n160_fit && a.N % 4 == 0 && (!a.conv || n160 <= 2 || ns2_tail_ok)}};
        // (the four older candidates are ranked among themselves as they were fitted, without the stream term; candidate 4 then
        // competes against the winner with the winner's stream cost added)
        int best = -1;
        float best_t = 0.f;
        auto cost = [&](int i, bool with_stream) {
            const long tiles_i = (long)((a.M + cand[i].bm - 1) / cand[i].bm) * ((ncols + cand[i].bn - 1) / cand[i].bn);
            const float stream = (with_stream && i != 4) ? 6.0f * (float)nstream * (float)(cand[i].bm * cand[i].bn) / 20480.f : 0.f;
            float rounds = (float)((tiles_i + cand[i].slots - 1) / cand[i].slots);
            // two blocks per CU: the blocks of a partial last round have their CU to themselves and finish early -- whole rounds
            // over-charge it (73728 x 320 3 x 3: 2.25 rounds, 179 us against 206 on the one-block kernel; charged 3 it loses on paper)
            if (i == 4) rounds = 0.5f * (rounds + (float)tiles_i / (float)cand[i].slots);
            return cand[i].pre + rounds * (cand[i].t0 + stream + (float)nk * cand[i].tk);
        };
        for (int i = 0; i < 4; ++i) {
            if (!cand[i].ok) continue;
            const float t = cost(i, false);
            if (best < 0 || t < best_t) { best = i; best_t = t; }
        }
        if (cand[4].ok && (best < 0 || cost(4, true) < cost(best, true))) best = 4;
        use_p8 = best == 0; use_p8h = best == 1;
        dma_bn = (best == 2 || best == 4) ? 160 : 0; dma_bm = 128;
        dma_ns2 = best == 4;
        small = false; splits = 1; p8_splits = 1;
    }
    if (a.gna_part) { use_p8 = use_p8h = false; dma_bn = 0; small = true; splits = 1; }   // GroupNorm-on-A exists on the 64^2 register-staged kernel
    if (a.a32) { use_p8 = use_p8h = false; dma_bn = 0; }                                   // the fp32 operand exists on the register-staged kernels
    if (use_p8h) { use_p8 = false; splits = 1; dma_bn = 0; }
    if (use_p8) { splits = p8_splits; dma_bn = 0; }
    if (a.ln_colsum) { dma_bn = 0; splits = 1; }      // the block must see whole rows of A (row statistics)
    if (!a.ws || splits < 1 || a.geglu) splits = 1;
    while (splits > 1 && (size_t)splits * a.M * a.N * sizeof(float) > (size_t)ws_bytes) --splits;
    const int tiles = small ? t64 : t128;
    if (force_splits && a.ws && !a.geglu && !a.ln_colsum) {
        splits = force_splits < nk ? force_splits : nk;
        while (splits > 1 && (size_t)splits * a.M * a.N * sizeof(float) > (size_t)ws_bytes) --splits;
    }
    a.kt_per_split = (nk + splits - 1) / splits;
    a.splits = (nk + a.kt_per_split - 1) / a.kt_per_split;
    static const int trace = env_int("SPIDER_GEMM_TRACE");       // tuning aid: one line per dispatch decision on stderr
    if (trace)
        fprintf(stderr, "spider_gemm_dispatch M=%d N=%d K=%d conv=%d(%dx%d s%d ups%d) geglu=%d ln=%d a32=%d -> %s splits=%d\n", a.M, a.N, a.K, a.conv,
                a.kh, a.kw, a.stride, a.ups, a.geglu, a.ln_colsum != nullptr, a.a32,
                use_p8h ? "p8h(256x128)" : use_p8 ? "p8(256x256)" : (dma_bn && !a.geglu) ? (dma_ns2 ? "dma2(128x160, 2 per CU)" : dma_bm == 64 ? "dma(64x160)" : "dma(128x160)")
                                                                   : small ? "reg(64x64)" : "reg(128x128)", a.splits);
    if (use_p8h || (force_tile == 257 && !a.geglu && !a.ln_colsum)) {
        launch_p8h(a, st);
    } else if (use_p8 || (force_tile == 256 && !a.geglu && !a.ln_colsum)) {
        if (a.ln_colsum) {
            a.ln_rows = reinterpret_cast<const float2*>(a.ws);
            ln_row_stats_kernel<<<(a.M + 3) / 4, 256, 0, st>>>(a.A, reinterpret_cast<float2*>(a.ws), a.M, a.K, a.ln_eps);
            SPIDER_LAUNCH_OK();
        }
        launch_p8(a, st);
    } else if (dma_bn && !a.geglu) {
        const int tdma = ((a.M + dma_bm - 1) / dma_bm) * ((a.N + dma_bn - 1) / dma_bn);
        // producer-side GroupNorm partials in the epilogue: chunks of 64 rows (GemmArgs::gn_cr is an OUTPUT of launch: the caller
        // learns the chunking from it); otherwise, with split-K, the reduce below writes 16-row chunks
        if (a.gn_part && a.splits == 1) { a.gn_cr = 64; if (!(dma_bn == 160 && force_tile != 161 && dma_gn_ok(a))) a.gn_part = nullptr; }
        if (gn_done && a.gn_part && a.splits == 1) *gn_done = 64;     // launch_dma takes the GN instantiation
        // row-order epilogue: pays with two blocks per CU (one's stores under the other's K loop) and fp32 streams to move; with one
        // block per CU the extra LDS round trip and its four barriers cost more than the coalescing returns (136 vs 126 us)
        const bool ns2 = dma_ns2;
        a.epi_lds = epi_lds_env == 2 || (epi_lds_env == 1 && ns2 && (a.res32 || a.c32d));
        if (dma_bn == 64) launch_dma<64, 6>(a, tdma, st);
        else if (ns2) launch_dma<160, 2>(a, tdma, st);
        else if (force_tile == 161) launch_dma<160, 4>(a, tdma, st);
        else if (dma_bn == 160 && dma_bm == 64) launch_dma<160, 4, 64>(a, tdma, st);
        else if (dma_bn == 160) launch_dma<160, 3>(a, tdma, st);
        else launch_dma<128, 4>(a, tdma, st);
    } else if (small) launch_tile<64, 64>(a, tiles, st);
    else launch_tile<128, 128>(a, tiles, st);
    SPIDER_LAUNCH_OK();
    }
    if (a.splits > 1) {
        const int cs = (a.gn_part && a.splits <= 16 && a.gn_cpg > 0 && a.N % a.gn_cpg == 0 && a.N % 4 == 0 && a.M % 16 == 0 && !a.C32 &&
                        (!a.conv || a.rows_per_group % 16 == 0))
                           ? gn_reduce_slab(a.gn_cpg, a.M >= 1024 ? 160 : 80) : 0;
        a.gn_cr = 16;
        if (cs) {      // reduce + epilogue + GroupNorm partials of the output in one launch
            const int nq = cs / 4;
            const int blocks = (a.M / 16) * ((a.N + cs - 1) / cs);
            const size_t sm = (size_t)17 * cs * sizeof(float2);
            if (a.act) splitk_reduce_gn_kernel<1><<<blocks, nq * 16, sm, st>>>(a, cs);
            else splitk_reduce_gn_kernel<0><<<blocks, nq * 16, sm, st>>>(a, cs);
            SPIDER_LAUNCH_OK();
            if (gn_done) *gn_done = 16;
            return 0;
        }
        const size_t total = (size_t)a.M * ((a.N + 3) / 4);
        size_t g = (total + 255) / 256;
        if (g > 2048) g = 2048;
        if (a.act) splitk_reduce_kernel<1><<<(int)g, 256, 0, st>>>(a);
        else splitk_reduce_kernel<0><<<(int)g, 256, 0, st>>>(a);
        SPIDER_LAUNCH_OK();
    }
    return 0;
}

}  // namespace

extern "C" {

// C = act(A[M,K] . W[N,K]^T + bias + rowbias[row / rows_per_group]) (+ res) , * out_scale
// ldc applies to C, C32 and res. ws/ws_bytes: optional fp32 split-K workspace (NULL disables split-K).
// w_tiled != 0: W is the tile-major copy [ceil(N/64)][ceil(K/64)][64][64] of the [N, K] weight (GemmArgs::w_tiled; built by the
// caller once per weight: pure data movement, like the OIHW -> OHWI conv repack).
// res32 / c32d (either may be NULL): the fp32 residual stream -- res32 [M, ldc] fp32 replaces `res` and is added to the unrounded
// result; c32d [M, ldc] fp32 receives the fp32 value that C rounds (GemmArgs::res32). Need the 16-bit output C.
int SPIDER_FN(spider_gemm)(const void* A, const void* W, void* C, void* C32, const void* bias, const void* res,
                     const void* rowbias, int rows_per_group, int M, int N, int K, int lda, int ldc, int act,
                     float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes, void* stream) {
    SPIDER_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem");
    SPIDER_CHECK(K % 8 == 0 && lda % 8 == 0, "gemm: K and lda must be multiples of 8 (16-byte rows)");
    const bool glu = act == 4 || act == 8 || act == 9;      // 4: GEGLU, 8: SwiGLU, 9: GEGLU rounded once (fused epilogues: the output has N / 2 columns)
    SPIDER_CHECK(ldc % 4 == 0 && ldc >= (glu ? N / 2 : N), "gemm: ldc must be >= the output width and a multiple of 4");
    SPIDER_CHECK((C != nullptr) != (C32 != nullptr), "gemm: exactly one of C (bf16) / C32 (fp32) must be given");
    SPIDER_CHECK(!rowbias || rows_per_group > 0, "gemm: rowbias needs rows_per_group > 0");
    SPIDER_CHECK(act >= 0 && act <= 9, "gemm: unknown activation");
    SPIDER_CHECK(!glu || (C && !res && !rowbias && N % 2 == 0), "gemm: GEGLU / SwiGLU epilogue needs bf16 output, even N, no res/rowbias");
    SPIDER_CHECK((!res32 && !c32d) || (C && !glu && !(res && res32) && (size_t)M * ldc * 4 < ((size_t)1 << 31)),
                 "gemm: the fp32 residual stream needs the 16-bit output, no GEGLU, at most one residual, and < 2 GiB of fp32 rows");
    GemmArgs a{};
    a.res32 = res32; a.c32d = c32d;
    a.A = (const h16_t*)A; a.W = (const h16_t*)W; a.C = (h16_t*)C; a.C32 = (float*)C32;
    a.bias = (const h16_t*)bias; a.res = (const h16_t*)res; a.rowbias = (const h16_t*)rowbias;
    a.rows_per_group = rows_per_group; a.M = M; a.K = K; a.lda = lda; a.ldc = ldc;
    a.geglu = act == 4 ? 1 : (act == 8 ? 2 : (act == 9 ? 3 : 0));
    a.N = a.geglu ? N / 2 : N;      // N counts W rows; the GEGLU / SwiGLU output has N/2 columns
    a.act = a.geglu ? 0 : act;
    a.act_param = 0.1f;             // leaky-relu slope of the GEMM form (HiFi-GAN); the conv form takes it as an argument
    a.out_scale = out_scale; a.conv = 0; a.ws = (float*)ws;
    SPIDER_CHECK((size_t)M * lda * 2 < ((size_t)1 << 32) && (size_t)N * K * 2 < ((size_t)1 << 32), "gemm: operands must be < 4 GiB");
    a.a_bytes = (uint32_t)((size_t)(M - 1) * lda * 2 + (size_t)K * 2);
    a.w_tiled = w_tiled ? 1 : 0;
    a.w_bytes = w_tiled ? tiled_bytes(N, K) : (uint32_t)((size_t)N * K * 2);
    set_epilogue_ranges(a);
    return launch(a, ws ? ws_bytes : 0, stream);
}

// C = LayerNorm(A; gamma, beta, eps) . W^T + bias (+ res), or its GEGLU form (act = 4), with the normalisation folded into the
// GEMM: Wf = W * diag(gamma) (bf16, [N, K]), colsum[n] = sum_k Wf[n,k] (fp32), colbias[n] = sum_k beta[k] W[n,k] + bias[n]
// (fp32) are prepared once per layer by the caller; the kernel computes the row statistics of A on the fly and applies
// C = rstd * (A.Wf^T - mean * colsum) + colbias in its epilogue. Replaces BasicTransformerBlock.norm1/2/3 + the projection
// that consumes it (diffusers-0.25 attention.py; call site custom_sd.py:634-639). lda must equal K (whole rows are normalised).
int SPIDER_FN(spider_gemm_ln)(const void* A, const void* Wf, void* C, const float* colsum, const float* colbias, const void* res,
                        int M, int N, int K, int ldc, int act, float eps, int w_tiled, void* ws, long ws_bytes, void* stream) {
    SPIDER_CHECK(M > 0 && N > 0 && K > 0, "gemm_ln: empty problem");
    SPIDER_CHECK(K % 8 == 0, "gemm_ln: K must be a multiple of 8 (16-byte rows)");
    SPIDER_CHECK(act == 0 || act == 4 || act == 9, "gemm_ln: only the plain and the GEGLU epilogues (4, 9 = rounded once) are built");
    SPIDER_CHECK(colsum && colbias && C, "gemm_ln: colsum, colbias and C are required");
    SPIDER_CHECK(act == 0 || (!res && N % 2 == 0), "gemm_ln: GEGLU epilogue needs even N and no residual");
    GemmArgs a{};
    a.A = (const h16_t*)A; a.W = (const h16_t*)Wf; a.C = (h16_t*)C; a.C32 = nullptr;
    a.bias = nullptr; a.res = (const h16_t*)res; a.rowbias = nullptr; a.rows_per_group = 0;
    a.M = M; a.K = K; a.lda = K; a.ldc = ldc;
    a.geglu = act == 4 ? 1 : (act == 9 ? 3 : 0);
    a.N = a.geglu ? N / 2 : N;
    SPIDER_CHECK(a.N % 4 == 0 && ldc % 4 == 0 && ldc >= a.N, "gemm_ln: output width and ldc must be multiples of 4, ldc >= width");
    a.act = 0; a.act_param = 0.f; a.out_scale = 1.f; a.conv = 0; a.ws = (float*)ws;     // ws: row statistics of the 256^2 form
    a.ln_colsum = colsum; a.ln_bias = colbias; a.ln_eps = eps;
    SPIDER_CHECK((size_t)M * K * 2 < ((size_t)1 << 31) && (size_t)N * K * 2 < ((size_t)1 << 32) && (size_t)M * ldc * 2 < ((size_t)1 << 31),
                 "gemm_ln: operands must be < 2 GiB");
    a.a_bytes = (uint32_t)((size_t)M * K * 2);
    a.w_tiled = w_tiled ? 1 : 0;
    a.w_bytes = w_tiled ? tiled_bytes(N, K) : (uint32_t)((size_t)N * K * 2);
    set_epilogue_ranges(a);
    return launch(a, ws ? ws_bytes : 0, stream);
}

// NHWC conv as implicit GEMM, general form. x [B, Hin, Win, Cin] bf16; w [Cout, kh, kw, Cin] bf16 (OHWI);
// y [B, Hout, Wout, Cout] bf16. up_h/up_w > 0: x is read through a fused nearest-2x upsample cropped to
// up_h x up_w (Upsample2D with an explicit output size, then the conv). 1-D convs are Hin = kh = 1; the
// (3,1,1) temporal conv of UNet3D is Hin = frames, Win = H*W, kh = 3, kw = 1.
// rowbias [B, Cout] is the per-image time-embedding add of ResnetBlock2D; res is [B,Hout,Wout,Cout].
static int conv_impl(const void* x, const void* w, void* y, const void* bias, const void* res,
                     const void* rowbias, int B, int Hin, int Win, int Cin, int Cout, int kh, int kw, int stride,
                     int pad_h, int pad_w, int dil, int up_h, int up_w, int act, float act_param,
                     float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes, void* stream,
                     float* gn_part, int gn_groups, int* produced, int a32 = 0) {
    if (produced) *produced = 0;
    SPIDER_CHECK(B > 0 && Hin > 0 && Win > 0 && Cin > 0 && Cout > 0, "conv: empty problem");
    SPIDER_CHECK(kh >= 1 && kw >= 1 && kh * kw <= 64 && dil >= 1, "conv: kernel taps must be 1..64, dilation >= 1");
    SPIDER_CHECK(stride == 1 || stride == 2, "conv: stride must be 1 or 2");
    SPIDER_CHECK(Cin % 8 == 0, "conv: Cin must be a multiple of 8 for the MFMA path (use conv2d_small)");
    SPIDER_CHECK(Cout % 4 == 0, "conv: Cout must be a multiple of 4");
    SPIDER_CHECK(pad_h >= 0 && pad_w >= 0, "conv: negative padding");
    SPIDER_CHECK(act == 0 || act == 1 || act == 2 || act == 3 || act == 5 || act == 6 || act == 7, "conv: unknown activation");
    const int ups = (up_h > 0 || up_w > 0) ? 1 : 0;
    if (ups) {
        SPIDER_CHECK(stride == 1, "conv: fused upsample requires stride 1");
        // the kernel reads source row iy >> 1: that is F.interpolate(size=..., mode="nearest") (floor(dst * in / out)) only
        // for out = 2*in and out = 2*in - 1 (the sizes Upsample2D's `upsample_size` rule produces for odd skip maps)
        SPIDER_CHECK((up_h == 2 * Hin || up_h == 2 * Hin - 1) && (up_w == 2 * Win || up_w == 2 * Win - 1),
                     "conv: fused nearest upsample supports output sizes 2*in and 2*in-1 per axis");
    }
    const int Hs = ups ? up_h : Hin, Ws = ups ? up_w : Win;
    const int Hout = (Hs + 2 * pad_h - dil * (kh - 1) - 1) / stride + 1, Wout = (Ws + 2 * pad_w - dil * (kw - 1) - 1) / stride + 1;
    SPIDER_CHECK(Hout > 0 && Wout > 0, "conv: kernel larger than the padded input");
    SPIDER_CHECK((!res32 && !c32d) || (!(res && res32) && (size_t)B * Hout * Wout * Cout * 4 < ((size_t)1 << 31)),
                 "conv: the fp32 residual stream takes at most one residual and < 2 GiB of fp32 output");
    GemmArgs a{};
    a.res32 = res32; a.c32d = c32d;
    a.A = (const h16_t*)x; a.W = (const h16_t*)w; a.C = (h16_t*)y; a.C32 = nullptr;
    a.bias = (const h16_t*)bias; a.res = (const h16_t*)res; a.rowbias = (const h16_t*)rowbias;
    a.rows_per_group = Hout * Wout; a.M = B * Hout * Wout; a.N = Cout; a.K = kh * kw * Cin; a.lda = Cin; a.ldc = Cout;
    a.act = act; a.act_param = act_param; a.out_scale = out_scale; a.ws = (float*)ws;
    a.conv = 1; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.kh = kh; a.kw = kw; a.stride = stride;
    a.pad_h = pad_h; a.pad_w = pad_w; a.dil = dil; a.ups = ups; a.lim_h = Hs; a.lim_w = Ws; a.cin64 = Cin % 64 == 0;
    static const int kcm_env = getenv("SPIDER_CONV_KCM") ? atoi(getenv("SPIDER_CONV_KCM")) : 1;
    a.kcm = (kcm_env && a.cin64 && kh * kw > 1) ? 1 : 0;
    static const int hb_env = getenv("SPIDER_CONV_HBITS") ? atoi(getenv("SPIDER_CONV_HBITS")) : 1;
    a.hbits = (hb_env && a.cin64 && !ups && kh * kw <= 32) ? 1 : 0;
    a.a32 = a32 ? 1 : 0;
    SPIDER_CHECK((size_t)B * Hin * Win * Cin * (a32 ? 4 : 2) < ((size_t)1 << 32) && (size_t)Cout * a.K * 2 < ((size_t)1 << 32), "conv: operands must be < 4 GiB");
    a.a_bytes = (uint32_t)((size_t)B * Hin * Win * Cin * (a32 ? 4 : 2));
    a.w_tiled = w_tiled == 2 ? 2 : (w_tiled ? 1 : 0);
    a.w_bytes = w_tiled == 2 ? (uint32_t)((size_t)((Cout + 31) / 32 * 32) * a.K * 2) : (w_tiled ? tiled_bytes(Cout, a.K) : (uint32_t)((size_t)Cout * a.K * 2));
    set_epilogue_ranges(a);
    if (w_tiled == 2) SPIDER_CHECK(ws_eligible(a), "conv: w_tiled = 2 (fragment-major weights) needs a 3x3 stride-1 conv with <= 512 output pixels, Cin % 32 == 0");
    if (gn_part) {
        SPIDER_CHECK(gn_groups > 0 && Cout % gn_groups == 0, "conv_gn: Cout must be a multiple of the groups");
        a.gn_part = gn_part; a.gn_G = gn_groups; a.gn_cpg = Cout / gn_groups; a.gn_cr = 0;
    }
    int done = 0;
    const int rc = launch(a, ws ? ws_bytes : 0, stream, &done);
    if (produced) *produced = done;
    return rc;
}

int SPIDER_FN(spider_conv_nhwc_ex)(const void* x, const void* w, void* y, const void* bias, const void* res,
                             const void* rowbias, int B, int Hin, int Win, int Cin, int Cout, int kh, int kw, int stride,
                             int pad_h, int pad_w, int dil, int up_h, int up_w, int act, float act_param,
                             float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes, void* stream) {
    return conv_impl(x, w, y, bias, res, rowbias, B, Hin, Win, Cin, Cout, kh, kw, stride, pad_h, pad_w, dil, up_h, up_w, act, act_param,
                     out_scale, w_tiled, res32, c32d, ws, ws_bytes, stream, nullptr, 0, nullptr);
}

// The same conv that ALSO leaves the GroupNorm partial statistics of its output (ResnetBlock2D: conv1 -> norm2, conv2 -> the next
// block's GroupNorm): gn_part (room for B * Hout * Wout / 16 * gn_groups * 2 floats) receives [B * Hout * Wout / cr, gn_groups, 2]
// fp32 = per chunk of cr consecutive output pixels and per group (sum, sum of squares) of the 16-bit values written to y, in the
// layout spider_groupnorm_apply_nhwc and spider_gemm_gn_in consume with nchunk = Hout * Wout / cr. *produced = cr: 64 when the
// LDS-DMA kernel's epilogue wrote them, 16 when the split-K reduce did, 0 when this shape's kernel cannot (the caller then runs
// spider_groupnorm_stats_nhwc).
int SPIDER_FN(spider_conv_nhwc_gn)(const void* x, const void* w, void* y, const void* bias, const void* res,
                             const void* rowbias, int B, int Hin, int Win, int Cin, int Cout, int kh, int kw, int stride,
                             int pad_h, int pad_w, int dil, int up_h, int up_w, int act, float act_param,
                             float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes, void* stream,
                             float* gn_part, int gn_groups, int* produced) {
    SPIDER_CHECK(gn_part && produced, "conv_gn: gn_part and produced are required");
    return conv_impl(x, w, y, bias, res, rowbias, B, Hin, Win, Cin, Cout, kh, kw, stride, pad_h, pad_w, dil, up_h, up_w, act, act_param,
                     out_scale, w_tiled, res32, c32d, ws, ws_bytes, stream, gn_part, gn_groups, produced);
}

// C = GroupNorm(A) . W^T + bias for a plain linear whose input is a GroupNorm output (Transformer2DModel: norm -> proj_in) without
// materialising the normalised tensor: A [M, K] is the UN-normalised NHWC tensor ([B, HW, K]), gn_part [B, nchunk, G, 2] its partial
// statistics (spider_conv_nhwc_gn / spider_groupnorm_stats_nhwc), gamma / beta [K]; the kernel stores round16(fma(x, a_c, b_c)) into
// its LDS image of A -- the values spider_groupnorm_nhwc(silu = 0) writes. c32d (optional): fp32 copy of the result (fp32 stream).
int SPIDER_FN(spider_gemm_gn_in)(const void* A, const void* W, void* C, const void* bias, int M, int N, int K, int ldc, int w_tiled,
                           const float* gn_part, int nchunk, const void* gamma, const void* beta, int G, float eps, int HW,
                           float* c32d, void* stream) {
    SPIDER_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 4 == 0 && ldc % 4 == 0 && ldc >= N, "gemm_gn_in: K % 64, N % 4, ldc % 4");
    SPIDER_CHECK(gn_part && gamma && beta && C && nchunk > 0, "gemm_gn_in: statistics, gamma, beta and C are required");
    SPIDER_CHECK(G > 0 && G <= 64 && 256 % G == 0 && K % G == 0 && HW % 64 == 0 && M % HW == 0, "gemm_gn_in: G must divide 256 and K; HW % 64 == 0");
    SPIDER_CHECK((size_t)M * K * 2 < ((size_t)1 << 31) && (size_t)N * K * 2 < ((size_t)1 << 32) && (size_t)M * ldc * 4 < ((size_t)1 << 31),
                 "gemm_gn_in: operands must be < 2 GiB");
    GemmArgs a{};
    a.A = (const h16_t*)A; a.W = (const h16_t*)W; a.C = (h16_t*)C; a.bias = (const h16_t*)bias; a.c32d = c32d;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = ldc; a.out_scale = 1.f;
    a.a_bytes = (uint32_t)((size_t)M * K * 2);
    a.w_tiled = w_tiled ? 1 : 0;
    a.w_bytes = w_tiled ? tiled_bytes(N, K) : (uint32_t)((size_t)N * K * 2);
    a.gna_part = gn_part; a.gna_nchunk = nchunk; a.gna_G = G; a.gna_hw = HW; a.gna_eps = eps;
    a.gna_gamma = (const h16_t*)gamma; a.gna_beta = (const h16_t*)beta;
    set_epilogue_ranges(a);
    SPIDER_CHECK(a.c_bytes != 0, "gemm_gn_in: output must be < 2 GiB");
    return launch(a, 0, stream);
}

// ---- "precise" operand forms (ABI v4): the A operand is FP32 -- the fp32 master of the residual stream
// (UNetEngine(stream32=True)) or a GroupNorm output kept in fp32 -- and is split into hi = round16(x), lo = round16(x - hi) on its
// way into LDS; every K step multiplies W by both halves (two MFMAs), so the operand carries ~22 significand bits instead of 11.
// Used at the sites the per-site attribution names (scripts/exp/precision_sites.py): the stream read by conv_shortcut, the down- /
// upsamplers, proj_out, and Transformer2DModel.norm -> proj_in. Same epilogue contract as spider_gemm / spider_conv_nhwc_ex
// (16-bit output C + optional fp32 residual stream operands); no activation, no GEGLU, W row-major or tile-major.
int SPIDER_FN(spider_gemm_a32)(const float* A32, const void* W, void* C, const void* bias, const void* res, int M, int N, int K, int lda,
                         int ldc, float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes, void* stream) {
    SPIDER_CHECK(M > 0 && N > 0 && K > 0 && K % 8 == 0 && lda % 8 == 0 && N % 4 == 0 && ldc % 4 == 0 && ldc >= N && C && A32,
                 "gemm_a32: K, lda multiples of 8; N, ldc multiples of 4; A and C required");
    SPIDER_CHECK(!(res && res32) && (size_t)M * ldc * 4 < ((size_t)1 << 31) && (size_t)M * lda * 4 < ((size_t)1 << 32) &&
                 (size_t)N * K * 2 < ((size_t)1 << 32), "gemm_a32: at most one residual; operands < 2 GiB");
    GemmArgs a{};
    a.a32 = 1; a.res32 = res32; a.c32d = c32d;
    a.A = (const h16_t*)A32; a.W = (const h16_t*)W; a.C = (h16_t*)C; a.bias = (const h16_t*)bias; a.res = (const h16_t*)res;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.out_scale = out_scale; a.ws = (float*)ws;
    a.a_bytes = (uint32_t)((size_t)(M - 1) * lda * 4 + (size_t)K * 4);
    a.w_tiled = w_tiled ? 1 : 0;
    a.w_bytes = w_tiled ? tiled_bytes(N, K) : (uint32_t)((size_t)N * K * 2);
    set_epilogue_ranges(a);
    SPIDER_CHECK(a.c_bytes != 0, "gemm_a32: output must be < 2 GiB");
    return launch(a, ws ? ws_bytes : 0, stream);
}

// LayerNorm(A32; gamma, beta, eps) . W^T + bias (+ its GEGLU form, act 4 / 9) on the FP32 rows A32 [M, K] (the master of the token
// stream): the kernel takes the row statistics from the fp32 values, applies gamma to A (fp32) before the hi / lo split and keeps
// W EXACT -- spider_gemm_ln's fold re-rounds W * gamma to 16 bits, the largest single error site of the SDXL evaluation
// (scripts/exp/precision_sites.py: 24 % of the variance). colsum[n] = sum_k gamma[k] W[n,k] (fp32, of the unrounded products),
// colbias[n] = sum_k beta[k] W[n,k] + bias[n], prepared once per layer by the caller (ops.fold_layernorm_exact).
int SPIDER_FN(spider_gemm_ln_a32)(const float* A32, const void* W, void* C, const void* gamma, const float* colsum, const float* colbias,
                            int M, int N, int K, int ldc, int act, float eps, int w_tiled, void* stream) {
    SPIDER_CHECK(M > 0 && N > 0 && K > 0 && K % 8 == 0, "gemm_ln_a32: K must be a multiple of 8");
    SPIDER_CHECK(act == 0 || act == 4 || act == 9, "gemm_ln_a32: plain or GEGLU (4, 9 = rounded once) epilogue");
    SPIDER_CHECK(A32 && W && C && gamma && colsum && colbias, "gemm_ln_a32: A32, W, C, gamma, colsum and colbias are required");
    SPIDER_CHECK(act == 0 || N % 2 == 0, "gemm_ln_a32: GEGLU needs even N");
    GemmArgs a{};
    a.a32 = 1;
    a.A = (const h16_t*)A32; a.W = (const h16_t*)W; a.C = (h16_t*)C;
    a.M = M; a.K = K; a.lda = K; a.ldc = ldc;
    a.geglu = act == 4 ? 1 : (act == 9 ? 3 : 0);
    a.N = a.geglu ? N / 2 : N;
    SPIDER_CHECK(a.N % 4 == 0 && ldc % 4 == 0 && ldc >= a.N, "gemm_ln_a32: output width and ldc must be multiples of 4, ldc >= width");
    a.out_scale = 1.f;
    a.ln_colsum = colsum; a.ln_bias = colbias; a.ln_eps = eps; a.gna_gamma = (const h16_t*)gamma;
    SPIDER_CHECK((size_t)M * K * 4 < ((size_t)1 << 32) && (size_t)N * K * 2 < ((size_t)1 << 32) && (size_t)M * ldc * 2 < ((size_t)1 << 31),
                 "gemm_ln_a32: operands must be < 2 GiB");
    SPIDER_CHECK(K <= 8192, "gemm_ln_a32: K <= 8192 (gamma table in LDS)");
    a.a_bytes = (uint32_t)((size_t)M * K * 4);
    a.w_tiled = w_tiled ? 1 : 0;
    a.w_bytes = w_tiled ? tiled_bytes(N, K) : (uint32_t)((size_t)N * K * 2);
    set_epilogue_ranges(a);
    return launch(a, 0, stream);
}

// GroupNorm(A) . W^T + bias as spider_gemm_gn_in, with A the FP32 tensor: normalised in fp32, then split hi / lo.
int SPIDER_FN(spider_gemm_gn_in_a32)(const float* A32, const void* W, void* C, const void* bias, int M, int N, int K, int ldc, int w_tiled,
                               const float* gn_part, int nchunk, const void* gamma, const void* beta, int G, float eps, int HW,
                               float* c32d, void* stream) {
    SPIDER_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 4 == 0 && ldc % 4 == 0 && ldc >= N, "gemm_gn_in_a32: K % 64, N % 4, ldc % 4");
    SPIDER_CHECK(gn_part && gamma && beta && C && A32 && nchunk > 0, "gemm_gn_in_a32: statistics, gamma, beta, A and C are required");
    SPIDER_CHECK(G > 0 && G <= 64 && 256 % G == 0 && K % G == 0 && HW % 64 == 0 && M % HW == 0, "gemm_gn_in_a32: G must divide 256 and K; HW % 64 == 0");
    SPIDER_CHECK((size_t)M * K * 4 < ((size_t)1 << 32) && (size_t)N * K * 2 < ((size_t)1 << 32) && (size_t)M * ldc * 4 < ((size_t)1 << 31),
                 "gemm_gn_in_a32: operands must be < 2 GiB");
    GemmArgs a{};
    a.a32 = 1;
    a.A = (const h16_t*)A32; a.W = (const h16_t*)W; a.C = (h16_t*)C; a.bias = (const h16_t*)bias; a.c32d = c32d;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = ldc; a.out_scale = 1.f;
    a.a_bytes = (uint32_t)((size_t)M * K * 4);
    a.w_tiled = w_tiled ? 1 : 0;
    a.w_bytes = w_tiled ? tiled_bytes(N, K) : (uint32_t)((size_t)N * K * 2);
    a.gna_part = gn_part; a.gna_nchunk = nchunk; a.gna_G = G; a.gna_hw = HW; a.gna_eps = eps;
    a.gna_gamma = (const h16_t*)gamma; a.gna_beta = (const h16_t*)beta;
    set_epilogue_ranges(a);
    SPIDER_CHECK(a.c_bytes != 0, "gemm_gn_in_a32: output must be < 2 GiB");
    return launch(a, 0, stream);
}

// NHWC conv with the FP32 image x32 [B, Hin, Win, Cin] as the A operand (conv_shortcut, Downsample2D / Upsample2D convs reading the
// stream's master); arguments as spider_conv_nhwc_ex without the activation.
int SPIDER_FN(spider_conv_nhwc_a32)(const float* x32, const void* w, void* y, const void* bias, const void* res, const void* rowbias,
                              int B, int Hin, int Win, int Cin, int Cout, int kh, int kw, int stride, int pad_h, int pad_w, int dil,
                              int up_h, int up_w, float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes,
                              void* stream) {
    SPIDER_CHECK(x32 && w_tiled != 2, "conv_a32: x32 is required; fragment-major weights are not an operand of this form");
    return conv_impl(x32, w, y, bias, res, rowbias, B, Hin, Win, Cin, Cout, kh, kw, stride, pad_h, pad_w, dil, up_h, up_w, 0, 0.f,
                     out_scale, w_tiled, res32, c32d, ws, ws_bytes, stream, nullptr, 0, nullptr, 1);
}

// Square-kernel form used by the SD / SDXL UNet and the VAE (ResnetBlock2D convs, Down/Upsample2D, shortcuts).
// ups=1 fuses the exact nearest-2x upsample.
int SPIDER_FN(spider_conv2d_nhwc)(const void* x, const void* w, void* y, const void* bias, const void* res,
                            const void* rowbias, int B, int Hin, int Win, int Cin, int Cout, int ks, int stride,
                            int pad, int ups, float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes,
                            void* stream) {
    SPIDER_CHECK(ks == 1 || ks == 3, "conv2d: kernel size must be 1 or 3");
    SPIDER_CHECK(Cin % 64 == 0, "conv2d: Cin must be a multiple of 64 for the MFMA path (use conv2d_small)");
    return SPIDER_FN(spider_conv_nhwc_ex)(x, w, y, bias, res, rowbias, B, Hin, Win, Cin, Cout, ks, ks, stride, pad, pad, 1,
                                    ups ? 2 * Hin : 0, ups ? 2 * Win : 0, 0, 0.f, out_scale, w_tiled, res32, c32d, ws, ws_bytes, stream);
}

}  // extern "C"
#endif  // SPIDER_WS_DEV
