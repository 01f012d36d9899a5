// Fused cross-attention sub-block of the UNet's BasicTransformerBlock (gfx950):
//     h_out = h + to_out( softmax( to_q(LayerNorm2(h)) . K^T / sqrt(d) ) . V )           one launch, Q / P / O never in HBM
// replaces norm2 -> attn2.to_q -> SDPA against the 77 text tokens -> attn2.to_out[0] (+ residual) of diffusers-0.25
// `BasicTransformerBlock.forward` (reached from custom_sd.py:634-639: encoder_hidden_states = prompt embeddings).
//
// The text K / V of a prompt are constant over all denoising steps, so they are folded into the projections once per prompt
// (spider_amd/unet.py:prepare):
//     Mq[b] [H*LP, C] = scale * K_h[b] . Wq_h . diag(gamma2)      (per head h, LP = 80 >= 77 keys, rows of padded keys are zero)
//     Mo[b] [C, H*LP] = Wo_h . V_h[b]^T
// and the block is two GEMMs with a softmax between them:
//     S[m, (h,l)] = rstd_m * (h[m,:] . Mq[b][(h,l),:] - mean_m * colsum[(h,l)]) + colbias[(h,l)]     (LayerNorm folded, see gemm.hip)
//     P = softmax over l (77 valid keys) per head;     out[m, :] = P[m, :] . Mo[b]^T + bias_o + h[m, :]
// Per block: BM token rows of one sample, 8 waves = 8 heads. Phase 1: wave h accumulates S^T for its head (keys on the
// accumulator rows, tokens on the lanes: the softmax statistics of a token need its own registers + 2 lane exchanges); the
// token tile streams through LDS in 64-wide K chunks (register-staged, XOR-swizzled image as in gemm.hip) and the row
// statistics of LayerNorm are taken from it on the way. P goes to LDS as bf16 (row stride 1408 B = 128 mod 256 with the same
// chunk swizzle: conflict-free ds_read_b128). Phase 2: wave w owns output column tiles w, w + 8, ...; P^T fragments from LDS.
// Both weight-like operands (Mq, Mo) are stored FRAGMENT-MAJOR (ops.repack_fm16: 1 KiB contiguous per wave instruction) and are
// read straight into registers: they are not shared between the waves of a block, and stay L2-resident across blocks.
#include "common.hpp"

using namespace spider;

namespace {

constexpr int XH = 8;            // heads (= waves per block)
constexpr int XLP = 80;          // padded keys per head
constexpr int XHL = XH * XLP;    // 640: K of the second GEMM
constexpr int XLT = XLP / 16;    // key tiles per head
constexpr int XPS = 1408;        // P image row stride in bytes

struct XArgs {
    const h16_t* x;        // [rows, C] residual stream (un-normalised)
    const h16_t* mq;       // [B2][XHL/16][C/64][2][64][8]   fragment-major Mq (gamma and scale folded)
    const h16_t* mo;       // [B2][C/16][XHL/64][2][64][8]   fragment-major Mo
    const float* colsum;    // [B2, XHL]
    const float* colbias;   // [B2, XHL]
    const h16_t* bias_o;   // [C]
    h16_t* out;            // [rows, C]
    const float* x32;      // optional fp32 master of the residual stream [rows, C]: added to the unrounded result instead of x
    float* out32;          // optional fp32 master of the output [rows, C] (UNetEngine stream32, see gemm.hip GemmArgs::res32)
    int rows, C, n_tok, n_keys;
    float eps;
};

template <int BM, int CTW>
__global__ __launch_bounds__(512) void xattn_fused_kernel(XArgs p) {
    constexpr int MT = BM / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;                                           // 2 x [BM][128 B] token chunk images
    float2* stat = reinterpret_cast<float2*>(smem + 2 * BM * 128);          // [BM] {mean, rstd}
    char* ps = smem + 2 * BM * 128 + BM * 8;                   // [BM][XPS] probabilities (bf16)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * BM;
    const int b = row0 / p.n_tok;
    const int C = p.C, nkc = C / 64;

    // ---- token chunk staging: item c -> row c >> 3, 16-byte chunk c & 7 of the 64-wide K chunk
    constexpr int ITEMS = BM * 8;
    const bool x_owner = tid < ITEMS;                          // BM = 64: every thread; 32 / 16: the first 256 / 128
    const int xr = tid >> 3, xc = tid & 7;
    const h16_t* xsrc = p.x + (size_t)(row0 + (x_owner ? xr : 0)) * C + xc * 8;
    const int xdst = xr * 128 + ((xc ^ ((xr >> 1) & 7)) * 16);
    float sum = 0.f, sq = 0.f;

    // ---- phase 1: S^T[(h,l), m] for head h = wave
    f32x4 acc[XLT][MT];
#pragma unroll
    for (int i = 0; i < XLT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4* mqf = reinterpret_cast<const u32x4*>(p.mq) + ((size_t)(b * (XHL / 16) + wave * XLT) * nkc) * 128 + lane;
    // fragment (lt, kc, ks) of this head: mqf[(lt * nkc + kc) * 128 + ks * 64]
    // Operand rings: PD1 K chunks (token chunk + Mq fragments) are requested ahead of their use. In the UNet these operands come
    // from HBM / the Infinity Cache, not from L2: with one chunk of look-ahead every K step paid a full memory latency
    // (35 us per 32^2 site in situ against 22 us with hot caches); the ring pays one latency per PD1 chunks.
    constexpr int PD1 = BM >= 64 ? 2 : 3;
    u32x4 aq[PD1][2][XLT];
    u32x4 xreg[PD1];
    auto issue1 = [&](int kc, int s) {
        const int kcc = min(kc, nkc - 1);                     // past the end: re-read the last chunk (never consumed)
        xreg[s] = u32x4{0u, 0u, 0u, 0u};
        if (x_owner) xreg[s] = *reinterpret_cast<const u32x4*>(xsrc + kcc * 64);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int lt = 0; lt < XLT; ++lt) aq[s][ks][lt] = mqf[(size_t)(lt * nkc + kcc) * 128 + ks * 64];
    };
    auto stage_x = [&](int buf, const u32x4& xr_) {
        if (x_owner) {
            *reinterpret_cast<u32x4*>(xs + buf * (BM * 128) + xdst) = xr_;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float lo = h16lo_to_f32(xr_[d]), hi = h16hi_to_f32(xr_[d]);
                sum += lo + hi;
                sq = fmaf(lo, lo, fmaf(hi, hi, sq));
            }
        }
    };
#pragma unroll
    for (int s = 0; s < PD1; ++s) issue1(s, s);
    const int fswz = (r16 >> 1) & 7;
    for (int kc0 = 0; kc0 < nkc; kc0 += PD1) {
#pragma unroll
        for (int s = 0; s < PD1; ++s) {
            const int kc = kc0 + s;
            if (kc < nkc) {                                   // block-uniform
                const int buf = kc & 1;
                stage_x(buf, xreg[s]);                        // buffer `buf` was last read two chunks ago, two barriers back
                __syncthreads();
                const char* xb = xs + buf * (BM * 128);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    h16x8 xf[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        xf[mt] = *reinterpret_cast<const h16x8*>(xb + (mt * 16 + r16) * 128 + (((ks * 4 + g) ^ fswz) * 16));
#pragma unroll
                    for (int lt = 0; lt < XLT; ++lt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[lt][mt] = mfma_16x16x32_h16(__builtin_bit_cast(h16x8, aq[s][ks][lt]), xf[mt], acc[lt][mt]);
                }
                issue1(kc + PD1, s);
            }
        }
    }
    __syncthreads();

    // ---- LayerNorm statistics of the BM rows (8 staging lanes per row)
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { sum += __shfl_xor(sum, o, 64); sq += __shfl_xor(sq, o, 64); }
    if (x_owner && xc == 0) {
        const float mean = sum / (float)C;
        const float var = fmaxf(sq / (float)C - mean * mean, 0.f);
        stat[xr] = float2{mean, rsqrtf(var + p.eps)};
    }
    __syncthreads();

    // ---- scores -> probabilities (per token: 20 keys in this lane, the other 60 in the lanes g' != g of the same r16)
    {
        const float* cs = p.colsum + (size_t)b * XHL + wave * XLP;
        const float* cb = p.colbias + (size_t)b * XHL + wave * XLP;
        f32x4 csv[XLT], cbv[XLT];
#pragma unroll
        for (int lt = 0; lt < XLT; ++lt) {
            csv[lt] = *reinterpret_cast<const f32x4*>(cs + lt * 16 + 4 * g);
            cbv[lt] = *reinterpret_cast<const f32x4*>(cb + lt * 16 + 4 * g);
        }
        constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mt * 16 + r16;
            const float2 ms = stat[m];
            float mx = -1e30f;
#pragma unroll
            for (int lt = 0; lt < XLT; ++lt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int l = lt * 16 + 4 * g + e;
                    float sv = (ms.y * (acc[lt][mt][e] - ms.x * csv[lt][e]) + cbv[lt][e]) * LOG2E;
                    sv = l < p.n_keys ? sv : -1e30f;
                    acc[lt][mt][e] = sv;
                    mx = fmaxf(mx, sv);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float den = 0.f;
#pragma unroll
            for (int lt = 0; lt < XLT; ++lt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(acc[lt][mt][e] - mx);
                    acc[lt][mt][e] = pv;
                    den += pv;
                }
            den += __shfl_xor(den, 16, 64);
            den += __shfl_xor(den, 32, 64);
            const float inv = __builtin_amdgcn_rcpf(den);
            char* prow = ps + m * XPS;
            const int pswz = (m >> 1) & 7;
#pragma unroll
            for (int lt = 0; lt < XLT; ++lt) {
                const int k = wave * XLP + lt * 16 + 4 * g;            // 4 consecutive keys: 8 bytes inside one 16-byte chunk
                u32x2 o2;
                o2.x = pack_h16x2(acc[lt][mt][0] * inv, acc[lt][mt][1] * inv);
                o2.y = pack_h16x2(acc[lt][mt][2] * inv, acc[lt][mt][3] * inv);
                *reinterpret_cast<u32x2*>(prow + (k >> 6) * 128 + ((((k & 63) >> 3) ^ pswz) * 16) + (k & 4) * 2) = o2;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: out^T[c, m] = Mo[b][c, :] . P[m, :] for the column tiles ct = wave + 8 i
    const int nct = C / 16;
    f32x4 acc2[CTW][MT];
#pragma unroll
    for (int i = 0; i < CTW; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4* mof = reinterpret_cast<const u32x4*>(p.mo) + (size_t)b * nct * (XHL / 64) * 128 + lane;
    // fragment (ct, k2) of Mo: mof[(ct * (XHL/64) + (k2 >> 1)) * 128 + (k2 & 1) * 64];  tiles past nct are clamped (results unused)
    int ctw[CTW];
#pragma unroll
    for (int i = 0; i < CTW; ++i) ctw[i] = min(wave + 8 * i, nct - 1);
    constexpr int PD2 = 4;
    constexpr int NK2 = XHL / 32;
    u32x4 mo_r[PD2][CTW];
    auto issue2 = [&](int k2, int s) {
        const int kk = min(k2, NK2 - 1);
#pragma unroll
        for (int i = 0; i < CTW; ++i) mo_r[s][i] = mof[(size_t)(ctw[i] * (XHL / 64) + (kk >> 1)) * 128 + (kk & 1) * 64];
    };
#pragma unroll
    for (int s = 0; s < PD2; ++s) issue2(s, s);
    for (int k20 = 0; k20 < NK2; k20 += PD2) {
#pragma unroll
        for (int s = 0; s < PD2; ++s) {
            const int k2 = k20 + s;                                  // NK2 = 20 is a multiple of PD2
            h16x8 pf[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 16 + r16;
                pf[mt] = *reinterpret_cast<const h16x8*>(ps + m * XPS + (k2 >> 1) * 128 + (((((k2 & 1) * 4) + g) ^ ((m >> 1) & 7)) * 16));
            }
#pragma unroll
            for (int i = 0; i < CTW; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc2[i][mt] = mfma_16x16x32_h16(__builtin_bit_cast(h16x8, mo_r[s][i]), pf[mt], acc2[i][mt]);
            issue2(k2 + PD2, s);
        }
    }

    // ---- epilogue: + bias, round, + residual (the block input), store; lane holds out[m][ct*16 + 4g + 0..3]
#pragma unroll
    for (int i = 0; i < CTW; ++i) {
        const int ct = wave + 8 * i;
        if (ct >= nct) continue;                                     // wave-uniform
        const int c = ct * 16 + 4 * g;
        const u32x2 bq = *reinterpret_cast<const u32x2*>(p.bias_o + c);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const size_t off = (size_t)(row0 + mt * 16 + r16) * C + c;
            float v0 = acc2[i][mt][0] + h16lo_to_f32(bq.x), v1 = acc2[i][mt][1] + h16hi_to_f32(bq.x);
            float v2 = acc2[i][mt][2] + h16lo_to_f32(bq.y), v3 = acc2[i][mt][3] + h16hi_to_f32(bq.y);
            if (p.x32) {      // fp32 residual stream: no rounding before the add
                const f32x4 rq = *reinterpret_cast<const f32x4*>(p.x32 + off);
                v0 += rq[0]; v1 += rq[1]; v2 += rq[2]; v3 += rq[3];
            } else {
                const u32x2 rq = *reinterpret_cast<const u32x2*>(p.x + off);
                v0 = h16_to_f32(f32_to_h16(v0)) + h16lo_to_f32(rq.x); v1 = h16_to_f32(f32_to_h16(v1)) + h16hi_to_f32(rq.x);
                v2 = h16_to_f32(f32_to_h16(v2)) + h16lo_to_f32(rq.y); v3 = h16_to_f32(f32_to_h16(v3)) + h16hi_to_f32(rq.y);
            }
            if (p.out32) *reinterpret_cast<f32x4*>(p.out32 + off) = f32x4{v0, v1, v2, v3};
            u32x2 o2;
            o2.x = pack_h16x2(v0, v1);
            o2.y = pack_h16x2(v2, v3);
            *reinterpret_cast<u32x2*>(p.out + off) = o2;
        }
    }
}

template <int BM, int CTW>
int launch_x(const XArgs& a, hipStream_t st) {
    constexpr int smem = 2 * BM * 128 + BM * 8 + BM * XPS;
    static unsigned done = 0;
    raise_dynamic_lds(&xattn_fused_kernel<BM, CTW>, smem, done);
    xattn_fused_kernel<BM, CTW><<<a.rows / BM, 512, smem, st>>>(a);
    SPIDER_LAUNCH_OK();
    return 0;
}

}  // namespace

extern "C" {

// x, out [rows, C] bf16 (rows = B2 * n_tok, sample-major); mq_fm / mo_fm: fragment-major folded projections (see the header);
// colsum / colbias [B2, 8 * 80] fp32; bias_o [C]. 8 heads, n_keys <= 80 text tokens, C a multiple of 64, C <= 1280 (<= 10
// output column tiles per wave), n_tok a multiple of the row tile (16; 32 / 64 are chosen when the grid still fills the chip).
int SPIDER_FN(spider_xattn_fused)(const void* x, const void* mq_fm, const void* mo_fm, const float* colsum, const float* colbias,
                            const void* bias_o, void* out, int B2, int n_tok, int C, int heads, int n_keys, float eps,
                            const float* x32, float* out32, void* stream) {
    SPIDER_CHECK(heads == XH, "xattn_fused: built for 8 heads");
    SPIDER_CHECK(B2 > 0 && n_tok > 0 && n_tok % 16 == 0, "xattn_fused: tokens per sample must be a multiple of 16");
    SPIDER_CHECK(C % 64 == 0 && C >= 64 && C <= 1280, "xattn_fused: C must be a multiple of 64, <= 1280");
    SPIDER_CHECK(n_keys >= 1 && n_keys <= XLP, "xattn_fused: at most 80 keys");
    SPIDER_CHECK((size_t)B2 * n_tok * C * 2 < ((size_t)1 << 31), "xattn_fused: activations must be < 2 GiB");
    XArgs a{};
    a.x = (const h16_t*)x; a.mq = (const h16_t*)mq_fm; a.mo = (const h16_t*)mo_fm; a.colsum = colsum; a.colbias = colbias;
    a.x32 = x32; a.out32 = out32;
    a.bias_o = (const h16_t*)bias_o; a.out = (h16_t*)out; a.rows = B2 * n_tok; a.C = C; a.n_tok = n_tok; a.n_keys = n_keys; a.eps = eps;
    hipStream_t st = (hipStream_t)stream;
    const int ctw = (C / 16 + 7) / 8;           // output column tiles per wave
    // row tile: the largest of 64 / 32 / 16 that divides n_tok and still gives >= ~192 blocks (or the smallest otherwise)
    int bm = 16;
    if (n_tok % 64 == 0 && a.rows / 64 >= 192 && ctw <= 3) bm = 64;
    else if (n_tok % 32 == 0 && a.rows / 32 >= 192 && ctw <= 5) bm = 32;
    static const int force_bm = [] { const char* e = getenv("SPIDER_XATTN_BM"); return e ? atoi(e) : 0; }();
    if (force_bm && n_tok % force_bm == 0 && ((force_bm == 64 && ctw <= 3) || (force_bm == 32 && ctw <= 5) || force_bm == 16)) bm = force_bm;
    if (bm == 64) return launch_x<64, 3>(a, st);
    if (bm == 32) return ctw <= 3 ? launch_x<32, 3>(a, st) : launch_x<32, 5>(a, st);
    if (ctw <= 3) return launch_x<16, 3>(a, st);
    if (ctw <= 5) return launch_x<16, 5>(a, st);
    return launch_x<16, 10>(a, st);
}

}  // extern "C"
