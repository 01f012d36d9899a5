// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )   (both operands K-contiguous)
//
// One kernel serves
//   * LLM prefill projections            (modeling_llama3.py:186-199,260-313 at S = prompt length)
//   * UNet / text-encoder / VAE linears  (diffusers-0.25 Attention.to_q/k/v/out, FeedForward, proj_in/out)
//   * conv2d as implicit GEMM on NHWC    (ResnetBlock2D conv1/conv2, Down/Upsample2D, conv_shortcut;
//                                         call sites custom_sd.py:634-639 -> UNet2DConditionModel.forward)
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 tiles of mfma_f32_16x16x32_bf16.
// Operand roles are swapped (W tile is the MFMA "A" operand) so each lane ends up with 4 consecutive
// output columns of one row -> 8-byte epilogue loads/stores. LDS rows are padded to 144 B (odd multiple
// of 16 B: conflict-free ds_read_b128), double-buffered, global->register prefetch one K tile ahead.
#include "common.hpp"

using namespace spider;

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_STRIDE = BK + 8;  // elements; 144 bytes

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* W;
    bf16_t* C;
    const bf16_t* bias;     // [N] or null
    const bf16_t* res;      // [M, ldc] or null (added after rounding the GEMM result to bf16)
    const bf16_t* rowbias;  // [M / rows_per_group, N] or null (time-embedding add)
    float* C32;             // optional fp32 output instead of bf16
    int M, N, K, lda, ldc;
    int rows_per_group;
    int act;                // 0 none, 1 silu, 2 gelu(erf), 3 quick-gelu
    float out_scale;        // multiplies the final value (1/rescale_output_factor)
    // implicit-GEMM conv (NHWC): A is the image [B, Hin, Win, Cin]; K = ks*ks*Cin
    int conv, Hin, Win, Cin, Hout, Wout, ks, stride, pad, ups;
};

template <bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* lds = reinterpret_cast<bf16_t*>(smem);
    // layout: [2 buffers][A tile 128 rows | W tile 128 rows][LDS_STRIDE]
    constexpr int TILE_ELEMS = (BM + BN) * LDS_STRIDE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = bid % tiles_m, tn = bid / tiles_m;  // m fastest: neighbours share the W tile
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread staging assignment: 4 A chunks + 4 W chunks of 16 B per K tile ----
    const int chunk = tid & 7;       // 16-byte chunk within the 64-wide K tile
    const int lrow = tid >> 3;       // 0..31, rows lrow + 32*i
    const bf16_t* a_ptr[4];
    bool a_ok[4];
    int a_oy[4], a_ox[4];
    const bf16_t* w_ptr[4];
    bool w_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + lrow + 32 * i;
        a_ok[i] = m < p.M;
        const int mc = a_ok[i] ? m : 0;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = mc / hw, rem = mc % hw;
            a_oy[i] = rem / p.Wout;
            a_ox[i] = rem % p.Wout;
            a_ptr[i] = p.A + (size_t)b * p.Hin * p.Win * p.Cin;
        } else {
            a_ptr[i] = p.A + (size_t)mc * p.lda;
            a_oy[i] = a_ox[i] = 0;
        }
        const int n = n0 + lrow + 32 * i;
        w_ok[i] = n < p.N;
        w_ptr[i] = p.W + (size_t)(w_ok[i] ? n : 0) * p.K;
    }

    const int nk = (p.K + BK - 1) / BK;
    u32x4 ra[4], rw[4];
    const u32x4 zero = {0u, 0u, 0u, 0u};

    auto load_tile = [&](int kt) {
        const int k = kt * BK + chunk * 8;
        const bool kok = k < p.K;
        if (CONV) {
            const int tap = (kt * BK) / p.Cin;
            const int c = (kt * BK) % p.Cin + chunk * 8;
            const int ky = tap / p.ks, kx = tap % p.ks;
            const int hlim = p.ups ? p.Hin * 2 : p.Hin, wlim = p.ups ? p.Win * 2 : p.Win;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int iy = a_oy[i] * p.stride + ky - p.pad;
                int ix = a_ox[i] * p.stride + kx - p.pad;
                const bool ok = a_ok[i] && kok && iy >= 0 && iy < hlim && ix >= 0 && ix < wlim;
                if (p.ups) { iy >>= 1; ix >>= 1; }
                ra[i] = ok ? *reinterpret_cast<const u32x4*>(a_ptr[i] + ((size_t)iy * p.Win + ix) * p.Cin + c) : zero;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                ra[i] = (a_ok[i] && kok) ? *reinterpret_cast<const u32x4*>(a_ptr[i] + k) : zero;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rw[i] = (w_ok[i] && kok) ? *reinterpret_cast<const u32x4*>(w_ptr[i] + k) : zero;
    };
    auto store_tile = [&](int buf) {
        bf16_t* base = lds + buf * TILE_ELEMS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(base + (lrow + 32 * i) * LDS_STRIDE + chunk * 8) = ra[i];
            *reinterpret_cast<u32x4*>(base + (BM + lrow + 32 * i) * LDS_STRIDE + chunk * 8) = rw[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frow = lane & 15, fk = (lane >> 4) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const bf16_t* abase = lds + cur * TILE_ELEMS + (wm * 64 + frow) * LDS_STRIDE + fk;
        const bf16_t* wbase = lds + cur * TILE_ELEMS + (BM + wn * 64 + frow) * LDS_STRIDE + fk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(abase + i * 16 * LDS_STRIDE + ks * 32);
                wf[i] = *reinterpret_cast<const bf16x8*>(wbase + i * 16 * LDS_STRIDE + ks * 32);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + (lane>>4)*4 + 0..3] ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        const int grp = p.rowbias ? m / p.rows_per_group : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            const bool full = (n + 3 < p.N);
            if (full) {
                if (p.bias) {
                    const u32x2 bq = *reinterpret_cast<const u32x2*>(p.bias + n);
                    v[0] += bf16lo_to_f32(bq.x); v[1] += bf16hi_to_f32(bq.x);
                    v[2] += bf16lo_to_f32(bq.y); v[3] += bf16hi_to_f32(bq.y);
                }
                if (p.rowbias) {
                    const u32x2 bq = *reinterpret_cast<const u32x2*>(p.rowbias + (size_t)grp * p.N + n);
                    v[0] += bf16lo_to_f32(bq.x); v[1] += bf16hi_to_f32(bq.x);
                    v[2] += bf16lo_to_f32(bq.y); v[3] += bf16hi_to_f32(bq.y);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (p.act == 1) v[e] = silu_f(v[e]);
                    else if (p.act == 2) v[e] = gelu_erf_f(v[e]);
                    else if (p.act == 3) v[e] = quick_gelu_f(v[e]);
                }
                if (p.res) {
                    const u32x2 rq = *reinterpret_cast<const u32x2*>(p.res + (size_t)m * p.ldc + n);
                    v[0] = bf16_to_f32(f32_to_bf16(v[0])) + bf16lo_to_f32(rq.x);
                    v[1] = bf16_to_f32(f32_to_bf16(v[1])) + bf16hi_to_f32(rq.x);
                    v[2] = bf16_to_f32(f32_to_bf16(v[2])) + bf16lo_to_f32(rq.y);
                    v[3] = bf16_to_f32(f32_to_bf16(v[3])) + bf16hi_to_f32(rq.y);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= p.out_scale;
                if (p.C32) {
                    *reinterpret_cast<f32x4*>(p.C32 + (size_t)m * p.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    u32x2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(p.C + (size_t)m * p.ldc + n) = o;
                }
            } else {
                for (int e = 0; e < 4 && n + e < p.N; ++e) {
                    float t = v[e];
                    if (p.bias) t += bf16_to_f32(p.bias[n + e]);
                    if (p.rowbias) t += bf16_to_f32(p.rowbias[(size_t)grp * p.N + n + e]);
                    if (p.act == 1) t = silu_f(t);
                    else if (p.act == 2) t = gelu_erf_f(t);
                    else if (p.act == 3) t = quick_gelu_f(t);
                    if (p.res) t = bf16_to_f32(f32_to_bf16(t)) + bf16_to_f32(p.res[(size_t)m * p.ldc + n + e]);
                    t *= p.out_scale;
                    if (p.C32) p.C32[(size_t)m * p.ldc + n + e] = t;
                    else p.C[(size_t)m * p.ldc + n + e] = f32_to_bf16(t);
                }
            }
        }
    }
}

int launch(const GemmArgs& a, void* stream) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const size_t smem = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(bf16_t);  // 73,728 B
    if (a.conv) gemm_kernel<true><<<tiles, 256, smem, (hipStream_t)stream>>>(a);
    else gemm_kernel<false><<<tiles, 256, smem, (hipStream_t)stream>>>(a);
    SPIDER_LAUNCH_OK();
    return 0;
}

}  // namespace

extern "C" {

// C[M,N] = act(A[M,K] . W[N,K]^T + bias + rowbias[row / rows_per_group]) (+ res) , * out_scale
// ldc applies to C, C32 and res. N % 4 == 0 and ldc % 4 == 0 keep the 8-byte epilogue path aligned.
int spider_gemm_bf16(const void* A, const void* W, void* C, void* C32, const void* bias, const void* res,
                     const void* rowbias, int rows_per_group, int M, int N, int K, int lda, int ldc, int act,
                     float out_scale, void* stream) {
    SPIDER_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem");
    SPIDER_CHECK(K % 8 == 0 && lda % 8 == 0, "gemm: K and lda must be multiples of 8 (16-byte rows)");
    SPIDER_CHECK(ldc % 4 == 0 && ldc >= N, "gemm: ldc must be >= N and a multiple of 4");
    SPIDER_CHECK((C != nullptr) != (C32 != nullptr), "gemm: exactly one of C (bf16) / C32 (fp32) must be given");
    SPIDER_CHECK(!rowbias || rows_per_group > 0, "gemm: rowbias needs rows_per_group > 0");
    SPIDER_CHECK(act >= 0 && act <= 3, "gemm: unknown activation");
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.C = (bf16_t*)C; a.C32 = (float*)C32;
    a.bias = (const bf16_t*)bias; a.res = (const bf16_t*)res; a.rowbias = (const bf16_t*)rowbias;
    a.rows_per_group = rows_per_group; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.act = act;
    a.out_scale = out_scale; a.conv = 0;
    return launch(a, stream);
}

// NHWC conv2d as implicit GEMM. x [B, Hin, Win, Cin] bf16; w [Cout, ks, ks, Cin] bf16 (OHWI);
// y [B, Hout, Wout, Cout]. ups=1 reads x through a fused nearest-2x upsample (Upsample2D + conv).
// rowbias [B, Cout] is the per-image time-embedding add of ResnetBlock2D; res is [B,Hout,Wout,Cout].
int spider_conv2d_nhwc_bf16(const void* x, const void* w, void* y, const void* bias, const void* res,
                            const void* rowbias, int B, int Hin, int Win, int Cin, int Cout, int ks, int stride,
                            int pad, int ups, float out_scale, void* stream) {
    SPIDER_CHECK(B > 0 && Hin > 0 && Win > 0 && Cin > 0 && Cout > 0, "conv2d: empty problem");
    SPIDER_CHECK(ks == 1 || ks == 3, "conv2d: kernel size must be 1 or 3");
    SPIDER_CHECK(stride == 1 || stride == 2, "conv2d: stride must be 1 or 2");
    SPIDER_CHECK(Cin % 64 == 0, "conv2d: Cin must be a multiple of 64 for the MFMA path (use conv2d_small)");
    SPIDER_CHECK(Cout % 4 == 0, "conv2d: Cout must be a multiple of 4");
    SPIDER_CHECK(!(ups && stride != 1), "conv2d: fused upsample requires stride 1");
    const int Hs = ups ? Hin * 2 : Hin, Ws = ups ? Win * 2 : Win;
    const int Hout = (Hs + 2 * pad - ks) / stride + 1, Wout = (Ws + 2 * pad - ks) / stride + 1;
    GemmArgs a{};
    a.A = (const bf16_t*)x; a.W = (const bf16_t*)w; a.C = (bf16_t*)y; a.C32 = nullptr;
    a.bias = (const bf16_t*)bias; a.res = (const bf16_t*)res; a.rowbias = (const bf16_t*)rowbias;
    a.rows_per_group = Hout * Wout; a.M = B * Hout * Wout; a.N = Cout; a.K = ks * ks * Cin; a.lda = Cin; a.ldc = Cout;
    a.act = 0; a.out_scale = out_scale;
    a.conv = 1; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.ks = ks; a.stride = stride;
    a.pad = pad; a.ups = ups;
    return launch(a, stream);
}

}  // extern "C"
