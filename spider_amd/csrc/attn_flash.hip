// Flash-style fused attention on MFMA (gfx950), one kernel for every dense attention on the path:
//   * LLM prefill, causal + GQA             (modeling_llama3.py:202-237; old modeling_llama.py:190-231)
//   * UNet self- / cross-attention          (diffusers-0.25 Attention + AttnProcessor2_0, gradio_utils.py:400-472)
//   * StoryDiffusion consistent self-attn   (Comic_Generation.py:129-196 with cal_attn_mask_xl masks,
//                                            gradio_utils.py:241-287)
//   * CLIP text encoder (causal, d = 64)    (custom_sd.py:306-310)
// The [Lq, Lk] score matrix is never materialised. Per block: 128 query rows (4 waves x 32), KV tiles of
// 64 keys staged global -> registers -> LDS (prefetch of tile t+1 overlaps the MFMAs of tile t).
//   S^T = K . Q^T  (mfma_f32_32x32x16_bf16; key on the accumulator row, query on the lane) so a query row's
//   softmax statistics are lane-local; the S^T accumulators, packed to bf16, are directly the B operand of
//   O^T += V^T . P^T, whose A operand comes from the row-major V tile through ds_read_b64_tr_b16.
// LDS: K rows padded to an odd multiple of 16 B (conflict-free ds_read_b128); V row stride = 64 or 192 mod
// 256 B (conflict-free transposed reads).
//
// The consistent-self-attention mask is column-structured (every row shares one random keep vector except
// for its own image block, gradio_utils.py:257-285), so it is passed as a bit vector over the keys plus the
// image block length instead of a dense [4N,4N] bool matrix.
#include "common.hpp"
#include <stdlib.h>

using namespace spider;

namespace {

struct AttnArgs {
    const h16_t* q; const h16_t* k; const h16_t* v; h16_t* o;
    long q_bs, q_hs, q_rs;   // element strides: batch, head, row
    long k_bs, k_hs, k_rs;
    long v_bs, v_hs, v_rs;
    long o_bs, o_hs, o_rs;
    int B, Hq, Hkv, Lq, Lk, d;
    float scale_log2e;
    int causal, kv_off;                   // causal: key j visible to query i iff j <= i + kv_off
    const int* kv_beg;                    // [B] or null: keys < kv_beg[b] are masked (left padding)
    const unsigned long long* keep_bits;  // [ceil(Lk/64)] or null: bit j%64 of word j/64 = key j kept
    int blk, q_off;                       // image block length / query row offset for the "own block" rule
    // packed variable-length mode (Qwen2.5-Omni vision windows / audio chunks: cu_seqlens segments of one packed
    // sequence): one record {q_start, q_len <= 128, k_start, k_len} per block instead of the regular 128-row grid
    const int* tiles;
    int n_tiles;
    // key-list mode (with `tiles`): a record's key range [k_start, k_start + k_len) counts POSITIONS of key_idx[], whose entries
    // are the K / V rows to visit -- the consistent-self-attention mask as a per-image list of visible keys (kept keys + the own
    // image block): masked keys are never loaded or scored instead of being scored and zeroed
    const int* key_idx;
    int idx_len;
    int xcd_order;   // 1: remap the launch order so that one XCD walks the q tiles of a (batch, head) (SPIDER_ATTN_XCD, default on)
};

// value of the partner lane (lane ^ 32) by v_permlane32_swap: a VALU op, where __shfl_xor(x, 32) is a ds_bpermute round trip
// through the LDS crossbar on the critical path of every tile's softmax (twice)
__device__ __forceinline__ float partner32(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // lanes < 32: {v[l], v[l+32]}; lanes >= 32: {v[l-32], v[l]}
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

template <int DP>
struct Cfg {
    static constexpr int KS = DP + 8;                                  // K LDS row stride (elements)
    static constexpr int VS = (DP == 64 || DP == 96) ? 96 : 160;       // V LDS row stride (elements)
    static constexpr int NCH = (64 * DP / 8 + 255) / 256;              // 16-B chunks per thread per operand
    static constexpr int CPR = DP / 8;                                 // chunks per (padded) row
};

// ONES (head_dim < DP, e.g. the UNet's d = 40 / 80 in 64 / 96-wide tiles): the first padded V column is staged as 1.0, so
// the PV MFMA accumulates the softmax denominator (of the bf16-rounded probabilities it actually multiplies) in an O^T
// row that would otherwise hold zeros -- 32 adds per lane and K tile leave a loop whose SIMD issue port is the bottleneck.
// PLAIN (dense attention: no causal mask, no keep vector, no left padding, no packed tiles -- every UNet / VAE / tower-free call):
// the mask arithmetic, its scalar state and the per-tile "is this tile fully visible" test are compiled out; every tile but a
// ragged last one takes the predicate-free path. The general instantiation spilled scalars into VGPR lanes (v_readlane in the
// tile loop) and spent ~60 SALU + ~80 non-essential VALU instructions per 64-key tile in a loop whose issue port is the limit.
template <int DP, bool ONES, bool PLAIN, bool KIDX = false>
__global__ __launch_bounds__(256, (DP <= 96 ? 2 : 1)) void attn_flash_kernel(AttnArgs p) {
    using C = Cfg<DP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two LDS images [K tile | V tile]: tile t+1 is written while tile t is still being read -> one barrier per tile
    constexpr int IMG = 64 * (C::KS + C::VS);
    h16_t* const lds0 = reinterpret_cast<h16_t*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h32 = lane >> 5, l32 = lane & 31;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs in launch order (x fastest), so with the plain
    // (q tile, head, batch) indices every XCD saw every head and pulled all of K / V through its own 4 MiB L2 (PMC: 89 MB fetched
    // for 16 MB of operands at the UNet's 64^2 self-attention). Remapped, an XCD owns a contiguous run of the logical grid: the
    // q tiles of one (batch, head) share an L2 and K / V come from HBM once.
    int qt, hq, b;
    {
        const int gx = gridDim.x, gy = gridDim.y;
        const int flat = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int L = p.xcd_order ? xcd_remap(flat, gx * gy * (int)gridDim.z) : flat;
        qt = L % gx;
        hq = (L / gx) % gy;
        b = L / (gx * gy);
    }
    const int hk = hq / (p.Hq / p.Hkv);
    int q0 = qt * 128, lq_end = p.Lq, seg_kbeg = 0, lk_end = p.Lk;
    if (!PLAIN && p.tiles) {
        const int* rec = p.tiles + 4 * qt;
        q0 = rec[0]; lq_end = rec[0] + rec[1]; seg_kbeg = rec[2]; lk_end = rec[2] + rec[3];
    }
    const int qi = q0 + wave * 32 + l32;  // this lane's query row
    const bool q_ok = qi < lq_end;

    const h16_t* qb = p.q + b * p.q_bs + hq * p.q_hs;
    const h16_t* kb = p.k + b * p.k_bs + hk * p.k_hs;
    const h16_t* vb = p.v + b * p.v_bs + hk * p.v_hs;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[qi][16*ks + 8*h32 + 0..7] ----
    h16x8 qf[DP / 16];
#pragma unroll
    for (int ks = 0; ks < DP / 16; ++ks) {
        const int dd = ks * 16 + h32 * 8;
        u32x4 t = {0u, 0u, 0u, 0u};
        if (q_ok && dd < p.d) t = *reinterpret_cast<const u32x4*>(qb + (long)qi * p.q_rs + dd);
        qf[ks] = __builtin_bit_cast(h16x8, t);
    }

    // ---- key range for this query tile ----
    const int kbeg = PLAIN ? 0 : (p.kv_beg ? p.kv_beg[b] : seg_kbeg);
    int kend = lk_end;
    if (!PLAIN && p.causal) kend = min(kend, q0 + 128 + p.kv_off);
    const int t_begin = kbeg / 64;
    const int t_end = (kend + 63) / 64;

    const int own_lo = (!PLAIN && p.blk > 0) ? ((qi + p.q_off) / p.blk) * p.blk : 0;
    const int own_hi = own_lo + p.blk;
    // the image block of this wave's first row, and whether its last row lies in the same one (wave-uniform scalars)
    const int w_first = __builtin_amdgcn_readfirstlane(q0 + wave * 32);
    const int w_own_lo = (!PLAIN && p.blk > 0) ? ((w_first + p.q_off) / p.blk) * p.blk : 0;
    const bool own_uniform = !PLAIN && p.blk > 0 && ((w_first + 31 + p.q_off) / p.blk) * p.blk == w_own_lo;
    const int caus_max = qi + p.kv_off;  // last visible key when causal

    f32x16 acc_o[DP / 32];
#pragma unroll
    for (int i = 0; i < DP / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    u32x4 rk[C::NCH], rv[C::NCH];
    // K / V rows through buffer descriptors: a chunk outside the tile's keys / the head dim gets an all-ones offset and reads
    // zeros in hardware -- no per-load branch (with `ok ? load : 0` the loop carried 4 divergent branches per tile).
    // Ranges are clamped to 4 GiB - 1; attention operands of this path are far below that.
    const int k_rows = KIDX ? p.Lk : lk_end;       // key-list mode: list entries address any row of the K / V view
    auto span = [&](long rs) { const long b_ = ((long)(k_rows - 1) * rs + p.d) * 2; return (uint32_t)(b_ < 0xFFFFFFFFl ? b_ : 0xFFFFFFFFl); };
    const __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(kb), 0, span(p.k_rs), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(vb), 0, span(p.v_rs), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(KIDX ? p.key_idx : nullptr), 0,
                                                                            KIDX ? (uint32_t)p.idx_len * 4u : 0u, 0x00020000);
    uint32_t ld_row[C::NCH], ld_cb[C::NCH], ld_inv[C::NCH];
#pragma unroll
    for (int i = 0; i < C::NCH; ++i) {
        const int c = tid + i * 256;
        ld_row[i] = (uint32_t)(c / C::CPR);
        ld_cb[i] = (uint32_t)(c % C::CPR) * 16u;
        ld_inv[i] = ((c < 64 * C::CPR) && (c % C::CPR) * 8 < p.d) ? 0u : 0xFFFFFFFFu;
    }
    // key-list mode: the K / V rows of the NEXT tile to load, fetched one tile ahead of the loads that use them
    uint32_t kidx[C::NCH];
    auto load_idx = [&](int t) {
#pragma unroll
        for (int i = 0; i < C::NCH; ++i) {
            const int pos = t * 64 + (int)ld_row[i];
            const uint32_t inv = (uint32_t)((lk_end - 1 - pos) >> 31);       // positions past the list read 0 (never used: masked)
            kidx[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc_i, ((uint32_t)pos * 4u) | inv, 0, 0);
        }
    };
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < C::NCH; ++i) {
            const int pos = t * 64 + (int)ld_row[i];
            const uint32_t key = KIDX ? kidx[i] : (uint32_t)pos;
            const uint32_t inv = ld_inv[i] | (uint32_t)((lk_end - 1 - pos) >> 31);
            rk[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_k, (key * (uint32_t)p.k_rs * 2u + ld_cb[i]) | inv, 0, 0));
            rv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, (key * (uint32_t)p.v_rs * 2u + ld_cb[i]) | inv, 0, 0));
        }
    };
    // `t` = the tile these registers hold. The ones column (V[key][d] = 1.0) is patched HERE, at the LDS write, not where the
    // loads are issued: patched at the load, the wave waited for the V loads at the top of every tile (vmcnt right behind the
    // buffer_load), i.e. their latency sat in front of the QK^T MFMAs instead of behind the whole tile's compute.
    auto store_tile = [&](int buf, int t) {
        h16_t* Kd = lds0 + buf * IMG;
        h16_t* Vd = Kd + 64 * C::KS;
#pragma unroll
        for (int i = 0; i < C::NCH; ++i) {
            const int c = tid + i * 256;
            if (c < 64 * C::CPR) {
                const int row = c / C::CPR, ch = c % C::CPR;
                u32x4 vv = rv[i];
                if (ONES) vv.x = (ld_cb[i] == (uint32_t)p.d * 2u && t * 64 + row < lk_end) ? H16_ONE : vv.x;
                *reinterpret_cast<u32x4*>(Kd + row * C::KS + ch * 8) = rk[i];
                *reinterpret_cast<u32x4*>(Vd + row * C::VS + ch * 8) = vv;
            }
        }
    };

    if (t_begin < t_end) {
        if (KIDX) load_idx(t_begin);
        load_tile(t_begin);
        if (KIDX) load_idx(t_begin + 1);
        store_tile(0, t_begin);
    }
    // Retire the Q loads HERE: their first use is the QK^T MFMA inside the tile loop, where the compiler cannot count how many
    // K / V prefetch loads were issued after them and waited with vmcnt(0) at the top of EVERY tile -- draining the prefetch of
    // tile t+1 in front of the MFMAs of tile t instead of behind them.
#pragma unroll
    for (int ks = 0; ks < DP / 16; ++ks) asm volatile("" ::"v"(qf[ks]));
    __syncthreads();

    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        const h16_t* Ks = lds0 + cur * IMG;
        const h16_t* Vs = Ks + 64 * C::KS;
        if (t + 1 < t_end) {
            load_tile(t + 1);
            if (KIDX) load_idx(t + 2);
        }

        // ---- S^T = K . Q^T for the two 32-key halves of the tile ----
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
            const h16_t* krow = Ks + (kt * 32 + l32) * C::KS + h32 * 8;
#pragma unroll
            for (int ks = 0; ks < DP / 16; ++ks) {
                const h16x8 kf = *reinterpret_cast<const h16x8*>(krow + ks * 16);
                s[kt] = mfma_32x32x16_h16(kf, qf[ks], s[kt]);
            }
        }

        // ---- mask + online softmax (lane = query column; the partner lane^32 holds the other 32 keys) ----
        // Fast path (wave-uniform): a tile entirely inside [kbeg, Lk), below the causal diagonal of every row of
        // this wave and without a keep vector needs no per-element predicates: max, one fma + exp2, add.
        const int wave_q0 = q0 + wave * 32;
        // Keep-vector tiles (consistent self-attention), classified per wave and tile: when the wave's 32 rows share one image
        // block (always, for block lengths that are multiples of 64), a tile inside that block is fully visible, and a tile
        // outside it is masked by the keep bits alone -- a per-KEY bias, the same for every query, applied inside the scale fma
        // (2 extra VALU per score instead of ~9 for the general row-and-key predicate).
        bool keep_only = false, own_tile = false;
        if (!PLAIN && p.keep_bits && own_uniform && !p.causal && t * 64 >= kbeg && t * 64 + 64 <= lk_end) {
            own_tile = t * 64 >= w_own_lo && t * 64 + 64 <= w_own_lo + p.blk;
            keep_only = t * 64 + 64 <= w_own_lo || t * 64 >= w_own_lo + p.blk;
        }
        const bool full_tile = PLAIN ? (t * 64 + 64 <= lk_end)
                                     : (own_tile || ((t * 64 >= kbeg) && (t * 64 + 64 <= lk_end) && !p.keep_bits &&
                                                     (!p.causal || t * 64 + 63 <= wave_q0 + p.kv_off)));
        float psum = 0.f, alpha;
        if (full_tile) {
            // four independent chains for the max and for the sum: a single 32-long dependent chain costs its full latency
            float tm[4] = {s[0][0], s[0][1], s[0][2], s[0][3]};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) tm[r & 3] = fmaxf(tm[r & 3], s[kt][r]);
            float tmax = fmaxf(fmaxf(tm[0], tm[1]), fmaxf(tm[2], tm[3]));
            tmax = fmaxf(tmax, partner32(tmax)) * p.scale_log2e;
            const float m_new = fmaxf(m_run, tmax);
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // <= 0: the bare v_exp_f32 (exp2f() adds a denormal-range fix-up)
            float ps4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(s[kt][r], p.scale_log2e, -m_new));
                    s[kt][r] = pv;
                    if (!ONES) ps4[r & 3] += pv;
                }
            if (!ONES) psum = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
            m_run = m_new;
        } else {
            // Branch-free masked tile: an invisible key's score becomes -1e30, so exp2 returns exactly 0 for it and no second
            // predicate is needed (per-element `ok ? exp2f(..) : 0` compiled to 32 divergent branches per tile -- the whole
            // consistent-self-attention path and every causal diagonal tile went through them).
            const int key0 = t * 64 + 4 * h32;                       // key = key0 + kt*32 + (r&3) + 8*(r>>2)
            const int hi_lim = (!PLAIN && p.causal) ? min(lk_end, caus_max + 1) : lk_end;
            float tmax = -1e30f;
            if (!PLAIN && keep_only) {
                const unsigned long long kl = p.keep_bits[t] >> (4 * h32);
                const uint32_t klo = (uint32_t)kl, khi = (uint32_t)(kl >> 32);
                float tm[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int off = kt * 32 + (r & 3) + 8 * (r >> 2);
                        // all ones where the key is kept -> bias 0, else -1e30 (0xF149F2CA)
                        const uint32_t kept = (uint32_t)__builtin_amdgcn_sbfe((int)(off < 32 ? klo : khi), off & 31, 1);
                        const float sv = fmaf(s[kt][r], p.scale_log2e, __uint_as_float(~kept & 0xF149F2CAu));
                        s[kt][r] = sv;
                        tm[r & 3] = fmaxf(tm[r & 3], sv);
                    }
                tmax = fmaxf(fmaxf(tm[0], tm[1]), fmaxf(tm[2], tm[3]));
            } else if (!PLAIN && p.keep_bits) {
                const unsigned long long kl = p.keep_bits[t] >> (4 * h32);
                const uint32_t klo = (uint32_t)kl, khi = (uint32_t)(kl >> 32);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int off = kt * 32 + (r & 3) + 8 * (r >> 2);
                        const int key = key0 + off;
                        const uint32_t keep = ((off < 32 ? klo : khi) >> (off & 31)) & 1u;
                        const bool ok = (key >= kbeg) & (key < hi_lim) & ((keep != 0u) | ((key >= own_lo) & (key < own_hi)));
                        const float sv = ok ? s[kt][r] * p.scale_log2e : -1e30f;
                        s[kt][r] = sv;
                        tmax = fmaxf(tmax, sv);
                    }
            } else {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2);
                        const bool ok = (key >= kbeg) & (key < hi_lim);
                        const float sv = ok ? s[kt][r] * p.scale_log2e : -1e30f;
                        s[kt][r] = sv;
                        tmax = fmaxf(tmax, sv);
                    }
            }
            tmax = fmaxf(tmax, partner32(tmax));
            const float m_new = fmaxf(m_run, tmax);
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            const float m_use = m_new < -1e29f ? 0.f : m_new;     // row with no visible key so far: keep every p at 0
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(s[kt][r] - m_use);
                    s[kt][r] = pv;
                    if (!ONES) psum += pv;
                }
            m_run = m_new;
        }
        if (!ONES) {
            psum += partner32(psum);
            l_run = l_run * alpha + psum;
        }
        // rescale O only when some row of the wave actually moved its max (alpha == 1 exactly otherwise)
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int i = 0; i < DP / 32; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[i][r] *= alpha;
        }

        // ---- P^T as the B operand: registers 8s..8s+7 of each half, packed to bf16 ----
        h16x8 pf[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                u32x4 t4;
                t4.x = pack_h16x2(s[kt][8 * sx + 0], s[kt][8 * sx + 1]);
                t4.y = pack_h16x2(s[kt][8 * sx + 2], s[kt][8 * sx + 3]);
                t4.z = pack_h16x2(s[kt][8 * sx + 4], s[kt][8 * sx + 5]);
                t4.w = pack_h16x2(s[kt][8 * sx + 6], s[kt][8 * sx + 7]);
                pf[kt][sx] = __builtin_bit_cast(h16x8, t4);
            }

        // ---- O^T += V^T . P^T ; V^T fragments by transposed LDS reads of the row-major V tile ----
        // lane i of a 16-lane group addresses row r0 + (i>>2), columns c0 + 4*(i&3); it receives column i.
        const int g16 = (lane >> 4) & 1, i16 = lane & 15;
#pragma unroll
        for (int db = 0; db < DP / 32; ++db) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    const int r0 = kt * 32 + 16 * sx + 4 * h32 + (i16 >> 2);
                    const int c0 = db * 32 + 16 * g16 + 4 * (i16 & 3);
                    const h16_t* a0 = Vs + r0 * C::VS + c0;
                    const h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) h16x4*)(a0));
                    const h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) h16x4*)(a0 + 8 * C::VS));
                    const h16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    acc_o[db] = mfma_32x32x16_h16(vf, pf[kt][sx], acc_o[db]);
                }
        }

        if (t + 1 < t_end) store_tile(cur ^ 1, t + 1);   // buffer cur^1 was last read in iteration t-1, before its barrier
        __syncthreads();
    }

    // ---- epilogue: lane holds O[qi][db*32 + 8*(r>>2) + 4*h32 + (r&3)] ----
    if (ONES) {   // the denominator sits in O^T row d: register 4*g of tile db on the lanes with h32 == 0
        float l = 0.f;
#pragma unroll
        for (int db = 0; db < DP / 32; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (db * 32 + 8 * g == p.d) l = acc_o[db][4 * g];
        l_run = __shfl(l, l32, 64);
    }
    if (q_ok) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        h16_t* ob = p.o + b * p.o_bs + hq * p.o_hs + (long)qi * p.o_rs;
#pragma unroll
        for (int db = 0; db < DP / 32; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dd = db * 32 + 8 * g + 4 * h32;
                if (dd < p.d) {
                    u32x2 o2;
                    o2.x = pack_h16x2(acc_o[db][4 * g + 0] * inv, acc_o[db][4 * g + 1] * inv);
                    o2.y = pack_h16x2(acc_o[db][4 * g + 2] * inv, acc_o[db][4 * g + 3] * inv);
                    *reinterpret_cast<u32x2*>(ob + dd) = o2;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Software-pipelined form for dense attention over whole 64-key tiles (UNet / VAE-free self-attention: Lk % 64 == 0, >= 2 tiles).
// With 2 x 8 heads x 4096 queries the grid has exactly two waves per SIMD, so a wave's serial chain QK^T -> softmax -> PV sets
// the time (the instruction count does not: removing 100 of 350 instructions per tile changed nothing). Here the QK^T MFMAs of
// tile t+1 are issued under the softmax VALU work of tile t -- two independent dependency chains in one wave:
//     iteration t:  S_next = K(t+1) . Q^T   ||   P = softmax(S_cur)      then      O^T += V(t)^T . P^T
// K is consumed one iteration ahead of V, so the two LDS images hold {K(t+1), V(t)} while {K(t+2), V(t+1)} are in registers;
// one barrier per tile as before.
// ------------------------------------------------------------------------------------------------------------------
// KQ = 16-wide k-steps of the QK^T product = ceil(head_dim / 16) <= DP / 16: the UNet's d = 40 heads ride in 64-wide tiles (the
// padded Q / K columns are zeros) but need only 3 of the tile's 4 k-steps -- 6 instead of 8 QK^T MFMAs per 64-key tile.
template <int DP, bool ONES, int KQ = DP / 16>
__global__ __launch_bounds__(256, (DP <= 96 ? 2 : 1)) void attn_flash_pipe_kernel(AttnArgs p) {
    using C = Cfg<DP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16_t* const Kb = reinterpret_cast<h16_t*>(smem);                  // 2 x [64][KS]
    h16_t* const Vb = Kb + 2 * 64 * C::KS;                              // 2 x [64][VS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h32 = lane >> 5, l32 = lane & 31;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs in launch order (x fastest), so with the plain
    // (q tile, head, batch) indices every XCD saw every head and pulled all of K / V through its own 4 MiB L2 (PMC: 89 MB fetched
    // for 16 MB of operands at the UNet's 64^2 self-attention). Remapped, an XCD owns a contiguous run of the logical grid: the
    // q tiles of one (batch, head) share an L2 and K / V come from HBM once.
    int qt, hq, b;
    {
        const int gx = gridDim.x, gy = gridDim.y;
        const int flat = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int L = p.xcd_order ? xcd_remap(flat, gx * gy * (int)gridDim.z) : flat;
        qt = L % gx;
        hq = (L / gx) % gy;
        b = L / (gx * gy);
    }
    const int hk = hq / (p.Hq / p.Hkv);
    const int q0 = qt * 128, lk_end = p.Lk;
    const int qi = q0 + wave * 32 + l32;
    const bool q_ok = qi < p.Lq;
    const h16_t* qb = p.q + b * p.q_bs + hq * p.q_hs;
    const h16_t* kb = p.k + b * p.k_bs + hk * p.k_hs;
    const h16_t* vb = p.v + b * p.v_bs + hk * p.v_hs;

    h16x8 qf[KQ];
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks) {
        const int dd = ks * 16 + h32 * 8;
        u32x4 t = {0u, 0u, 0u, 0u};
        if (q_ok && dd < p.d) t = *reinterpret_cast<const u32x4*>(qb + (long)qi * p.q_rs + dd);
        qf[ks] = __builtin_bit_cast(h16x8, t);
    }
    const int nt = lk_end / 64;

    f32x16 acc_o[DP / 32];
#pragma unroll
    for (int i = 0; i < DP / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    auto span = [&](long rs) { const long b_ = ((long)(lk_end - 1) * rs + p.d) * 2; return (uint32_t)(b_ < 0xFFFFFFFFl ? b_ : 0xFFFFFFFFl); };
    const __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(kb), 0, span(p.k_rs), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(vb), 0, span(p.v_rs), 0x00020000);
    u32x4 rk[C::NCH], rv[C::NCH];
    uint32_t ld_row[C::NCH], ld_cb[C::NCH], ld_inv[C::NCH];
#pragma unroll
    for (int i = 0; i < C::NCH; ++i) {
        const int c = tid + i * 256;
        ld_row[i] = (uint32_t)(c / C::CPR);
        ld_cb[i] = (uint32_t)(c % C::CPR) * 16u;
        ld_inv[i] = ((c < 64 * C::CPR) && (c % C::CPR) * 8 < p.d) ? 0u : 0xFFFFFFFFu;
    }
    // tiles past the end are requested with all-ones offsets (zeros): the loop body needs no "is there a next tile" branch
    auto load_k = [&](int t) {
        const uint32_t tinv = (uint32_t)((nt - 1 - t) >> 31);
#pragma unroll
        for (int i = 0; i < C::NCH; ++i)
            rk[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_k, ((uint32_t)(t * 64 + (int)ld_row[i]) * (uint32_t)p.k_rs * 2u + ld_cb[i]) | ld_inv[i] | tinv, 0, 0));
    };
    auto load_v = [&](int t) {
        const uint32_t tinv = (uint32_t)((nt - 1 - t) >> 31);
#pragma unroll
        for (int i = 0; i < C::NCH; ++i)
            rv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, ((uint32_t)(t * 64 + (int)ld_row[i]) * (uint32_t)p.v_rs * 2u + ld_cb[i]) | ld_inv[i] | tinv, 0, 0));
    };
    auto store_k = [&](int buf) {
        h16_t* Kd = Kb + buf * 64 * C::KS;
#pragma unroll
        for (int i = 0; i < C::NCH; ++i) {
            const int c = tid + i * 256;
            if (c < 64 * C::CPR) *reinterpret_cast<u32x4*>(Kd + (c / C::CPR) * C::KS + (c % C::CPR) * 8) = rk[i];
        }
    };
    auto store_v = [&](int buf) {
        h16_t* Vd = Vb + buf * 64 * C::VS;
#pragma unroll
        for (int i = 0; i < C::NCH; ++i) {
            const int c = tid + i * 256;
            if (c < 64 * C::CPR) {
                u32x4 vv = rv[i];
                if (ONES) vv.x = (ld_cb[i] == (uint32_t)p.d * 2u) ? H16_ONE : vv.x;       // V[key][d] = 1.0 (every key of a whole tile is valid)
                *reinterpret_cast<u32x4*>(Vd + (c / C::CPR) * C::VS + (c % C::CPR) * 8) = vv;
            }
        }
    };
    auto qk = [&](int buf, f32x16 (&s)[2]) {
        const h16_t* Ks = Kb + buf * 64 * C::KS;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
            const h16_t* krow = Ks + (kt * 32 + l32) * C::KS + h32 * 8;
#pragma unroll
            for (int ks = 0; ks < KQ; ++ks) {
                const h16x8 kf = *reinterpret_cast<const h16x8*>(krow + ks * 16);
                s[kt] = mfma_32x32x16_h16(kf, qf[ks], s[kt]);
            }
        }
    };

    // ---- prologue: K(0), V(0), K(1) staged; S_cur = scores of tile 0
    load_k(0); load_v(0);
    store_k(0); store_v(0);
    load_k(1);
    store_k(1);
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks) asm volatile("" ::"v"(qf[ks]));      // retire the Q loads before the loop (see above)
    __syncthreads();
    f32x16 sc[2], sn[2];
    qk(0, sc);
    // Iteration 0 re-stages K image 0 (store_k(0) -> K(2)) with no barrier of its own in front of it: every wave must have
    // finished the prologue's reads of K(0) first, or a wave running ahead overwrites rows a stalled sibling still scores against.
    __syncthreads();

    const int g16 = (lane >> 4) & 1, i16 = lane & 15;
    for (int t = 0; t < nt; ++t) {
        load_k(t + 2);
        load_v(t + 1);
        // ---- [B] online softmax of tile t (independent of [A]: the scheduler interleaves the two chains)
        float tm[4] = {sc[0][0], sc[0][1], sc[0][2], sc[0][3]};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) tm[r & 3] = fmaxf(tm[r & 3], sc[kt][r]);
        float tmax = fmaxf(fmaxf(tm[0], tm[1]), fmaxf(tm[2], tm[3]));
        tmax = fmaxf(tmax, partner32(tmax)) * p.scale_log2e;
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int i = 0; i < DP / 32; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[i][r] *= alpha;
        }
        // ---- [A] S_next = K(t+1) . Q^T   (tile nt reads a zero / stale image: its scores are never used)
        qk((t + 1) & 1, sn);
        float ps4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], p.scale_log2e, -m_new));
                sc[kt][r] = pv;
                if (!ONES) ps4[r & 3] += pv;
            }
        m_run = m_new;
        if (!ONES) {
            float psum = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
            psum += partner32(psum);
            l_run = l_run * alpha + psum;
        }
        h16x8 pf[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                u32x4 t4;
                t4.x = pack_h16x2(sc[kt][8 * sx + 0], sc[kt][8 * sx + 1]);
                t4.y = pack_h16x2(sc[kt][8 * sx + 2], sc[kt][8 * sx + 3]);
                t4.z = pack_h16x2(sc[kt][8 * sx + 4], sc[kt][8 * sx + 5]);
                t4.w = pack_h16x2(sc[kt][8 * sx + 6], sc[kt][8 * sx + 7]);
                pf[kt][sx] = __builtin_bit_cast(h16x8, t4);
            }
        // ---- [C] O^T += V(t)^T . P^T
        const h16_t* Vs = Vb + (t & 1) * 64 * C::VS;
#pragma unroll
        for (int db = 0; db < DP / 32; ++db) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    const int r0 = kt * 32 + 16 * sx + 4 * h32 + (i16 >> 2);
                    const int c0 = db * 32 + 16 * g16 + 4 * (i16 & 3);
                    const h16_t* a0 = Vs + r0 * C::VS + c0;
                    const h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4*)(a0));
                    const h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4*)(a0 + 8 * C::VS));
                    const h16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    acc_o[db] = mfma_32x32x16_h16(vf, pf[kt][sx], acc_o[db]);
                }
        }
        // ---- stage K(t+2) (its image held K(t), last read in iteration t-1) and V(t+1) (held V(t-1))
        store_k(t & 1);
        store_v((t + 1) & 1);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) sc[kt] = sn[kt];
        __syncthreads();
    }

    if (ONES) {
        float l = 0.f;
#pragma unroll
        for (int db = 0; db < DP / 32; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (db * 32 + 8 * g == p.d) l = acc_o[db][4 * g];
        l_run = __shfl(l, l32, 64);
    }
    if (q_ok) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        h16_t* ob = p.o + b * p.o_bs + hq * p.o_hs + (long)qi * p.o_rs;
#pragma unroll
        for (int db = 0; db < DP / 32; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dd = db * 32 + 8 * g + 4 * h32;
                if (dd < p.d) {
                    u32x2 o2;
                    o2.x = pack_h16x2(acc_o[db][4 * g + 0] * inv, acc_o[db][4 * g + 1] * inv);
                    o2.y = pack_h16x2(acc_o[db][4 * g + 2] * inv, acc_o[db][4 * g + 3] * inv);
                    *reinterpret_cast<u32x2*>(ob + dd) = o2;
                }
            }
    }
}

#ifndef SPIDER_F16
// ------------------------------------------------------------------------------------------------------------------
// Visible-key lists of the consistent self-attention (cal_attn_mask_xl, gradio_utils.py:241-287, as a list instead of a mask):
// query image `img0 + list` sees key j iff keep bit j is set or j lies in its own block [img*N, (img+1)*N). One block per list
// compacts the visible key indices in increasing order into key_idx[list * stride ...] and writes the query-tile records
// {q_start, q_len <= 128, k_start = list * stride, k_len = number of visible keys} that attn_flash_kernel<KIDX> walks.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void key_lists_kernel(const unsigned long long* __restrict__ keep, int n_keys, int N, int img0,
                                                         int q_img0, int stride, int* __restrict__ key_idx, int* __restrict__ tiles) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int list = blockIdx.x, img = img0 + list;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int* out = key_idx + (size_t)list * stride;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int j0 = 0; j0 < n_keys; j0 += 1024) {
        const int j = j0 + threadIdx.x;
        bool vis = false;
        if (j < n_keys) vis = ((keep[j >> 6] >> (j & 63)) & 1ull) != 0ull || (j / N == img);
        const unsigned long long bal = __ballot(vis);
        const int rank = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (vis) out[off + rank] = j;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            base_s += tot;
        }
        __syncthreads();
    }
    const int cnt = base_s;
    const int tpl = (N + 127) / 128;
    if ((int)threadIdx.x < tpl) {
        int* rec = tiles + ((size_t)list * tpl + threadIdx.x) * 4;
        const int qs = threadIdx.x * 128;
        rec[0] = (img - q_img0) * N + qs;
        rec[1] = min(128, N - qs);
        rec[2] = list * stride;
        rec[3] = cnt;
    }
}

#endif

// ------------------------------------------------------------------------------------------------------------------
// Short sequences (Lq, Lk <= 16; head_dim 64): the UNet3D's temporal attention -- every pixel of a sample is its own sequence
// of F = 16 frames (TransformerTemporalModel, custom_vd.py:25), 2880 x 5 sequence-heads per sample at 40 x 72. In the 128-row
// flash tiling above such a problem is one block of 256 threads per sequence-head with 16 of its rows in use; here ONE WAVE
// owns a sequence-head, with no block-level synchronisation:
//   S^T = K . Q^T  by two mfma_16x16x32 (A = K rows, B = Q rows: both fragments are 16-byte global loads, row lane & 15, head
//         dims 32 kb + 8 (lane >> 4) ..+7), so the lane holds S^T[key 4g + i][query lane & 15];
//   softmax over the keys = over the 4 registers and the lane groups g (two cross-lane steps);
//   O^T = V^T . P^T by four mfma_16x16x16 (one per 16 head dims): the B fragment IS the lane's 4 probabilities (keys 4g..4g+3 of
//         its query), the A fragment V[4g + j][16 t + (lane & 15)] comes from the wave's own 16 x 64 V tile in LDS through
//         ds_read_b64_tr_b16 (lane 4q + p of a group addresses row 4g + q, columns 16 t + 4p ..+3).
// The kernel is HBM-bound by construction (8 KiB in / 2 KiB out per sequence-head for ~6 MFMAs).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_short_kernel(AttnArgs a) {
    constexpr int VS = 72;                                   // V tile row stride (elements): 144 B, 8-byte aligned rows
    __shared__ __attribute__((aligned(16))) h16_t vs_all[4][16 * VS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long total = (long)a.B * a.Hq;
    long wid = (long)blockIdx.x * 4 + wave;
    const bool live = wid < total;
    if (!live) wid = total - 1;                              // keep EXEC full for the transposed reads: recompute, never store
    const int b = (int)(wid / a.Hq), h = (int)(wid % a.Hq);
    const int r = lane & 15, g = lane >> 4;
    const h16_t* qp = a.q + (long)b * a.q_bs + (long)h * a.q_hs + (long)min(r, a.Lq - 1) * a.q_rs + 8 * g;
    const h16_t* kp = a.k + (long)b * a.k_bs + (long)h * a.k_hs + (long)min(r, a.Lk - 1) * a.k_rs + 8 * g;
    const h16_t* vp = a.v + (long)b * a.v_bs + (long)h * a.v_hs + (long)min(r, a.Lk - 1) * a.v_rs + 8 * g;
    const h16x8 q0 = *reinterpret_cast<const h16x8*>(qp), q1 = *reinterpret_cast<const h16x8*>(qp + 32);
    const h16x8 k0 = *reinterpret_cast<const h16x8*>(kp), k1 = *reinterpret_cast<const h16x8*>(kp + 32);
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(vp), v1 = *reinterpret_cast<const u32x4*>(vp + 32);
    h16_t* vs = vs_all[wave];
    *reinterpret_cast<u32x4*>(vs + r * VS + 8 * g) = v0;
    *reinterpret_cast<u32x4*>(vs + r * VS + 32 + 8 * g) = v1;

    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    s = mfma_16x16x32_h16(k0, q0, s);
    s = mfma_16x16x32_h16(k1, q1, s);
    // lane: S^T[key 4g + i][query r]; softmax over the keys in the exp2 domain
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s[i] = (4 * g + i < a.Lk) ? s[i] * a.scale_log2e : -INFINITY;
        mx = fmaxf(mx, s[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float pr[4], l = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t pk = pack_h16x2(exp2f(s[i] - mx), 0.f);      // P is rounded to 16 bits, as the MFMA consumes it
        pr[i] = h16lo_to_f32(pk);
        l += pr[i];
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    u32x2 pp;
    pp.x = pack_h16x2(pr[0], pr[1]);
    pp.y = pack_h16x2(pr[2], pr[3]);
    const h16x4 pf = __builtin_bit_cast(h16x4, pp);

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own V tile is in LDS (no other wave touches it)
    const float inv = 1.f / l;
    const int i16 = lane & 15;
    h16_t* op = a.o + (long)b * a.o_bs + (long)h * a.o_hs + (long)r * a.o_rs + 4 * g;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const h16_t* va = vs + (4 * g + (i16 >> 2)) * VS + 16 * t + 4 * (i16 & 3);
        const h16x4 vf = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4*)(va));
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        o = mfma_16x16x16_h16(vf, pf, o);                     // O^T[dim 16 t + 4g + i][query r]
        if (live && r < a.Lq) {
            u32x2 w;
            w.x = pack_h16x2(o[0] * inv, o[1] * inv);
            w.y = pack_h16x2(o[2] * inv, o[3] * inv);
            *reinterpret_cast<u32x2*>(op + 16 * t) = w;
        }
    }
}

template <int DP>
int launch(const AttnArgs& a, void* stream) {
    using C = Cfg<DP>;
    dim3 grid(a.tiles ? a.n_tiles : (a.Lq + 127) / 128, a.Hq, a.B);
    const size_t smem = (size_t)2 * 64 * (C::KS + C::VS) * sizeof(h16_t);
    static const int xcd_env = [] { const char* e = getenv("SPIDER_ATTN_XCD"); return e ? atoi(e) : 1; }();
    const_cast<AttnArgs&>(a).xcd_order = xcd_env;
    if (a.key_idx) {
        if constexpr (DP == 64) {       // the StoryDiffusion / SDXL heads (d = 64; smaller heads ride in the same 64-wide tile)
            if (a.d < DP) attn_flash_kernel<DP, true, false, true><<<grid, 256, smem, (hipStream_t)stream>>>(a);
            else attn_flash_kernel<DP, false, false, true><<<grid, 256, smem, (hipStream_t)stream>>>(a);
            SPIDER_LAUNCH_OK();
            return 0;
        }
        spider_set_error("attn_keylist: head_dim must be <= 64");
        return -1;
    }
    const bool plain = !a.causal && !a.keep_bits && !a.kv_beg && !a.tiles;
    static const int short_env = [] { const char* e = getenv("SPIDER_ATTN_SHORT"); return e ? atoi(e) : 1; }();
    if constexpr (DP == 64) {
        if (plain && short_env && a.d == 64 && a.Lq <= 16 && a.Lk <= 16 && a.Hq == a.Hkv) {     // one wave per sequence-head
            const long total = (long)a.B * a.Hq;
            attn_short_kernel<<<(unsigned)((total + 3) / 4), 256, 0, (hipStream_t)stream>>>(a);
            SPIDER_LAUNCH_OK();
            return 0;
        }
    }
    static const int pipe_env = [] { const char* e = getenv("SPIDER_ATTN_PIPE"); return e ? atoi(e) : 1; }();
    if (plain && pipe_env && a.Lk % 64 == 0 && a.Lk >= 128 && DP <= 96) {      // software-pipelined dense form (whole 64-key tiles)
        if constexpr (DP == 64) {
            if (a.d <= 48) {       // 3 k-steps cover the head (d = 40: the SD-v1.5 64^2 level)
                attn_flash_pipe_kernel<DP, true, 3><<<grid, 256, smem, (hipStream_t)stream>>>(a);
                SPIDER_LAUNCH_OK();
                return 0;
            }
        }
        if (a.d < DP) attn_flash_pipe_kernel<DP, true><<<grid, 256, smem, (hipStream_t)stream>>>(a);
        else attn_flash_pipe_kernel<DP, false><<<grid, 256, smem, (hipStream_t)stream>>>(a);
        SPIDER_LAUNCH_OK();
        return 0;
    }
    if (a.d < DP) {
        if (plain) attn_flash_kernel<DP, true, true><<<grid, 256, smem, (hipStream_t)stream>>>(a);
        else attn_flash_kernel<DP, true, false><<<grid, 256, smem, (hipStream_t)stream>>>(a);
    } else {
        if (plain) attn_flash_kernel<DP, false, true><<<grid, 256, smem, (hipStream_t)stream>>>(a);
        else attn_flash_kernel<DP, false, false><<<grid, 256, smem, (hipStream_t)stream>>>(a);
    }
    SPIDER_LAUNCH_OK();
    return 0;
}

}  // namespace

extern "C" {

// Generic strided attention. Tensors are bf16; strides are in elements (batch, head, row), the head
// dimension is contiguous. d must be a multiple of 8 and <= 160. Output row = softmax(q k^T * scale + mask) v.
//   causal != 0 : key j visible to query i iff j <= i + kv_off
//   kv_beg      : optional device int[B]; keys below kv_beg[b] are masked
//   keep_bits   : optional device uint64[ceil(Lk/64)] column keep vector; a key is visible if its bit is
//                 set OR it lies in the query's own image block ((i + q_off) / blk == j / blk)
int SPIDER_FN(spider_attn)(const void* q, const void* k, const void* v, void* o,
                     long q_bs, long q_hs, long q_rs, long k_bs, long k_hs, long k_rs,
                     long v_bs, long v_hs, long v_rs, long o_bs, long o_hs, long o_rs,
                     int B, int Hq, int Hkv, int Lq, int Lk, int d, float scale, int causal, int kv_off,
                     const int* kv_beg, const void* keep_bits, int blk, int q_off, void* stream) {
    SPIDER_CHECK(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Lq > 0 && Lk > 0, "attn: bad shape");
    SPIDER_CHECK(d > 0 && d % 8 == 0 && d <= 160, "attn: head_dim must be a multiple of 8 and <= 160");
    SPIDER_CHECK(q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && o_rs % 4 == 0, "attn: row strides must keep 16-byte alignment");
    SPIDER_CHECK(q_hs % 8 == 0 && k_hs % 8 == 0 && v_hs % 8 == 0 && o_hs % 4 == 0, "attn: head strides must keep 16-byte alignment");
    SPIDER_CHECK(q_bs % 8 == 0 && k_bs % 8 == 0 && v_bs % 8 == 0 && o_bs % 4 == 0, "attn: batch strides must keep 16-byte alignment");
    SPIDER_CHECK(!keep_bits || blk > 0, "attn: keep_bits needs the image block length");
    SPIDER_CHECK((long)Lk * k_rs * 2 < (1L << 32) && (long)Lk * v_rs * 2 < (1L << 32), "attn: one (batch, head) K / V view must span < 4 GiB");
    AttnArgs a{};
    a.q = (const h16_t*)q; a.k = (const h16_t*)k; a.v = (const h16_t*)v; a.o = (h16_t*)o;
    a.q_bs = q_bs; a.q_hs = q_hs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_hs = k_hs; a.k_rs = k_rs;
    a.v_bs = v_bs; a.v_hs = v_hs; a.v_rs = v_rs; a.o_bs = o_bs; a.o_hs = o_hs; a.o_rs = o_rs;
    a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.Lq = Lq; a.Lk = Lk; a.d = d;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.causal = causal; a.kv_off = kv_off; a.kv_beg = kv_beg;
    a.keep_bits = (const unsigned long long*)keep_bits; a.blk = keep_bits ? blk : 0; a.q_off = q_off;
    if (d <= 64) return launch<64>(a, stream);
    if (d <= 96) return launch<96>(a, stream);
    if (d <= 128) return launch<128>(a, stream);
    return launch<160>(a, stream);
}

// Consistent self-attention through visible-key lists (see key_lists_kernel): same result as spider_attn_bf16 with keep_bits /
// blk / q_off -- masked keys contribute exactly zero there and are skipped here -- at the cost of the visible keys only.
//   key_idx [idx_len] int32, tiles [n_tiles][4] from spider_story_key_lists_i32; q [B, Lq, ...], k / v [B, Lk, ...] strided as in
//   spider_attn_bf16; head_dim 64 (the SDXL UNet of StoryDiffusion/Comic_Generation.py).
int SPIDER_FN(spider_attn_keylist)(const void* q, const void* k, const void* v, void* o,
                             long q_bs, long q_hs, long q_rs, long k_bs, long k_hs, long k_rs,
                             long v_bs, long v_hs, long v_rs, long o_bs, long o_hs, long o_rs,
                             int B, int Hq, int Hkv, int Lq, int Lk, int d, float scale,
                             const int* key_idx, int idx_len, const int* tiles, int n_tiles, void* stream) {
    SPIDER_CHECK(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Lq > 0 && Lk > 0, "attn_keylist: bad shape");
    SPIDER_CHECK(d > 0 && d % 8 == 0 && d <= 64, "attn_keylist: head_dim must be a multiple of 8 and <= 64");
    SPIDER_CHECK(key_idx && tiles && idx_len > 0 && n_tiles > 0, "attn_keylist: key lists and tile records required");
    SPIDER_CHECK(q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && o_rs % 4 == 0, "attn_keylist: row strides must keep 16-byte alignment");
    SPIDER_CHECK(q_hs % 8 == 0 && k_hs % 8 == 0 && v_hs % 8 == 0 && o_hs % 4 == 0, "attn_keylist: head strides must keep 16-byte alignment");
    SPIDER_CHECK(q_bs % 8 == 0 && k_bs % 8 == 0 && v_bs % 8 == 0 && o_bs % 4 == 0, "attn_keylist: batch strides must keep 16-byte alignment");
    SPIDER_CHECK((long)Lk * k_rs * 2 < (1L << 32) && (long)Lk * v_rs * 2 < (1L << 32), "attn_keylist: one (batch, head) K / V view must span < 4 GiB");
    AttnArgs a{};
    a.q = (const h16_t*)q; a.k = (const h16_t*)k; a.v = (const h16_t*)v; a.o = (h16_t*)o;
    a.q_bs = q_bs; a.q_hs = q_hs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_hs = k_hs; a.k_rs = k_rs;
    a.v_bs = v_bs; a.v_hs = v_hs; a.v_rs = v_rs; a.o_bs = o_bs; a.o_hs = o_hs; a.o_rs = o_rs;
    a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.Lq = Lq; a.Lk = Lk; a.d = d;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.tiles = tiles; a.n_tiles = n_tiles; a.key_idx = key_idx; a.idx_len = idx_len;
    return launch<64>(a, stream);
}

#ifndef SPIDER_F16   // dtype-free: built once (bf16 translation unit)
// Build the visible-key lists and query-tile records for spider_attn_keylist_bf16. keep_bits: uint64 words over n_keys keys
// (spider_pack_keep_bits_f32); N = tokens per image; list l serves query image img0 + l, whose rows start at
// (img0 + l - q_img0) * N in the q tensor (q_img0 = img0 when q holds only those images, 0 when it holds all of them).
// key_idx: n_lists * stride int32 (stride >= n_keys, multiple of 64); tiles: n_lists * ceil(N / 128) records of 4 int32.
int spider_story_key_lists_i32(const void* keep_bits, int n_keys, int N, int img0, int n_lists, int q_img0, int stride,
                               int* key_idx, int* tiles, void* stream) {
    SPIDER_CHECK(keep_bits && key_idx && tiles && n_keys > 0 && N > 0 && n_lists > 0, "key_lists: bad arguments");
    SPIDER_CHECK(stride >= n_keys && stride % 64 == 0, "key_lists: stride must cover the keys and be a multiple of 64");
    SPIDER_CHECK((N + 127) / 128 <= 1024, "key_lists: image block too long");
    key_lists_kernel<<<n_lists, 1024, 0, (hipStream_t)stream>>>((const unsigned long long*)keep_bits, n_keys, N, img0, q_img0, stride,
                                                               key_idx, tiles);
    SPIDER_LAUNCH_OK();
    return 0;
}

#endif

// Packed variable-length attention (no mask tensor): q/k/v/o are [total_rows, heads, d] views with the given row
// strides (head stride = d); `tiles` is a device int[n_tiles][4] = {q_start, q_len (1..128), k_start, k_len}: the
// block computes softmax(q[q_start : q_start+q_len] k[k_start : k_start+k_len]^T * scale) v[...]. The host cuts each
// cu_seqlens segment into <= 128-row query tiles that all see the segment's keys -- the per-segment SDPA loop of
// transformers' Qwen2_5OmniVisionAttention / Qwen2_5OmniAudioAttention (reached through Qwen2_5OmniModel.generate,
// qwen2.5omni_spider_web.py:468) without materialising a block-diagonal mask.
int SPIDER_FN(spider_attn_varlen)(const void* q, const void* k, const void* v, void* o, long q_rs, long k_rs, long v_rs, long o_rs,
                            int total_rows, int Hq, int Hkv, int d, float scale, const int* tiles, int n_tiles, void* stream) {
    SPIDER_CHECK(total_rows > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && n_tiles > 0 && tiles, "attn_varlen: bad shape");
    SPIDER_CHECK(d > 0 && d % 8 == 0 && d <= 160, "attn_varlen: head_dim must be a multiple of 8 and <= 160");
    SPIDER_CHECK(q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && o_rs % 4 == 0, "attn_varlen: row strides must keep 16-byte alignment");
    SPIDER_CHECK((long)total_rows * k_rs * 2 < (1L << 32) && (long)total_rows * v_rs * 2 < (1L << 32), "attn_varlen: K / V views must span < 4 GiB");
    AttnArgs a{};
    a.q = (const h16_t*)q; a.k = (const h16_t*)k; a.v = (const h16_t*)v; a.o = (h16_t*)o;
    a.q_hs = a.k_hs = a.v_hs = a.o_hs = d;
    a.q_rs = q_rs; a.k_rs = k_rs; a.v_rs = v_rs; a.o_rs = o_rs;
    a.B = 1; a.Hq = Hq; a.Hkv = Hkv; a.Lq = total_rows; a.Lk = total_rows; a.d = d;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.tiles = tiles; a.n_tiles = n_tiles;
    if (d <= 64) return launch<64>(a, stream);
    if (d <= 96) return launch<96>(a, stream);
    if (d <= 128) return launch<128>(a, stream);
    return launch<160>(a, stream);
}

}  // extern "C"
