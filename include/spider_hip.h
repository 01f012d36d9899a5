/* spider_hip.h -- C ABI of libspider_hip.so: the MI355X (gfx950) kernels behind Spider's any-to-many
 * generation hot path (LLM greedy decode + SD/StoryDiffusion UNet step).
 *
 * The reference (Layjins/Spider) has no FFI: every operator below replaces a PyTorch-level op sequence,
 * cited as <file>:<lines> relative to the reference tree. A maintainer binds these with ctypes (see
 * INTEGRATION.md); spider_amd/lib.py is that binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; bf16 tensors are raw uint16 bit patterns
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work, they never synchronise,
 *     allocate or free, so they can be captured into a hipGraph
 *   - return 0 on success, <0 on error; spider_last_error() (thread-local) holds the message
 *   - all reductions accumulate in fp32; outputs are rounded to bf16 where the reference's bf16 module
 *     would round (documented per function)
 */
#ifndef SPIDER_HIP_H
#define SPIDER_HIP_H
#define SPIDER_ABI_VERSION 4   /* 2: w_tiled argument of the GEMM / conv entry points; *_f16 instantiations
                                * 3: producer-side GroupNorm statistics (spider_conv_nhwc_gn, spider_groupnorm_stats / _apply, spider_gemm_gn_in)
                                * 4: w_tiled = 2 (fragment-major conv weights, in-launch split-K combine); the "precise" fp32-operand forms
                                *    (spider_gemm_a32, spider_gemm_gn_in_a32, spider_conv_nhwc_a32, spider_groupnorm_f32in_nhwc,
                                *    spider_conv2d_small_c{in,out}_f32in, spider_latent_to_nhwc_f32); act 9 = GEGLU rounded once */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library identity / errors ---- */
int spider_abi_version(void);
const char* spider_target_arch(void);
const char* spider_last_error(void);

/* ======================= LLM decode path (HBM-bound) ======================= */

/* nn.Embedding lookup: out[r,:] = table[ids[r],:]   (base_model.py:253-258 embed_tokens) */
int spider_embed_bf16(const void* table, const int* ids, void* out, int rows, int H, int V, void* stream);

/* LlamaRMSNorm (modeling_llama3.py:68-82; modeling_llama.py:57-74) with optional fused residual add
 * (decoder-layer wiring modeling_llama3.py:339-361): h = x (+ res); res_out = h; y = w * bf16(h*rsqrt(mean(h^2)+eps)) */
int spider_rmsnorm_bf16(const void* x, const void* res, const void* w, void* y, void* res_out, int rows, int H,
                        float eps, void* stream);

/* nn.Linear at decode (1..8 tokens): out[b,n] = sum_k xin[b,k] W[n,k] (+bias[n]) (+res[b,n]);
 * xin = RMSNorm(x)*norm_w when norm_w != NULL (fused prologue).  q/k/v/o_proj, down_proj:
 * modeling_llama3.py:186-199,240-313 */
int spider_gemv_bf16(const void* W, const void* x, void* out, const void* bias, const void* res, const void* norm_w,
                     float eps, int B, int N, int K, void* stream);

/* LlamaMLP gate/up + SiLU*mul (modeling_llama3.py:197-199): W_gate_up = [gate rows (I) | up rows (I)] x K */
int spider_gemv_swiglu_bf16(const void* W_gate_up, const void* x, void* out, const void* norm_w, float eps, int B,
                            int I, int K, void* stream);

/* final norm + lm_head + greedy argmax (modeling_llama3.py:619,870-871; HF greedy loop driven from
 * spider.py:1492-1508). logits (bf16 [B,V]) optional. ws_val/ws_idx: B*spider_lm_head_nparts(V) floats/ints. */
int spider_lm_head_nparts(int V);

/* End of one greedy decode step (the bookkeeping of GenerationMixin's loop, spider.py:1492-1508, kept on the device so that the
 * step is one hipGraph): cur_ids <- next_ids; pos, slot, kv_end += 1; hist[b][n_hist[b]++] = next_ids[b]. int32 [B] each; hist
 * [B, cap] / n_hist [B] optional (both NULL: cursors only); entries past cap are dropped, the counter still advances. */
int spider_decode_advance_i32(const int* next_ids, int* cur_ids, int* pos, int* slot, int* kv_end, int* hist, int* n_hist,
                              int cap, int B, void* stream);
int spider_lm_head_argmax_bf16(const void* W, const void* x, const void* norm_w, float eps, int* out_ids, void* logits,
                               void* ws_val, void* ws_idx, int B, int V, int K, void* stream);

/* apply_rotary_pos_emb (modeling_llama3.py:150-183; modeling_llama.py:116-123) on q,k of a fused QKV
 * projection + KV-cache append (modeling_llama.py:190-193). qkv [B*S,(n_q+2n_kv)*d]; cos_sin fp32
 * [max_pos, d] = [cos(d/2) | sin(d/2)]; q_out [B*S,n_q,d]; caches [B,n_kv,T_max,d]. */
int spider_rope_kv_append_bf16(const void* qkv, const int* pos, const int* slot, const float* cos_sin, void* q_out,
                               void* k_cache, void* v_cache, int B, int S, int n_q, int n_kv, int d, int T_max,
                               void* stream);
/* Multimodal RoPE of the Qwen2.5-Omni thinker (transformers apply_multimodal_rotary_pos_emb, reached from
 * Qwen2_5OmniModel.generate, qwen2.5omni_infer.py:3 / qwen2.5omni_spider_web.py:468): pos3 is [3, B*S]
 * (temporal, height, width); the first sec_t rotary pairs follow component 0, the next sec_h component 1, the rest
 * component 2 (mrope_section 16/24/24 at head_dim 128). sec_t = sec_h = 0 degenerates to spider_rope_kv_append_bf16. */
int spider_rope_kv_append_mrope_bf16(const void* qkv, const int* pos3, const int* slot, const float* cos_sin, void* q_out,
                                     void* k_cache, void* v_cache, int B, int S, int n_q, int n_kv, int d, int T_max,
                                     int sec_t, int sec_h, void* stream);

/* decode attention, one query token per sequence, GQA, fp32 online softmax, split over the KV length
 * (eager_attention_forward + repeat_kv, modeling_llama3.py:202-237). Valid cache slots per sequence:
 * [kv_beg[b], kv_end[b]) (kv_beg may be NULL = 0). ws_o: B*n_q*nsplit*d floats, ws_ml: B*n_q*nsplit*2.
 * head_dim d = 128; GQA group n_q / n_kv = 1 ... 8 (Llama-3-8B: 4, Qwen2.5-Omni-7B: 7); other shapes return -1 with a message. */
int spider_attn_decode_bf16(const void* q, const void* k_cache, const void* v_cache, const int* kv_beg,
                            const int* kv_end, void* out, void* ws_o, void* ws_ml, int B, int n_q, int n_kv, int d,
                            int T_max, float scale, int nsplit, void* stream);

/* The three calls above (RoPE, KV append, attention + combine) fused into ONE launch for decode: qkv [B,(n_q+2n_kv)d]
 * is the current token's fused projection; kv_end[b] INCLUDES the current token (slot kv_end[b]-1 is written here).
 * counters: int[B*n_kv], zeroed once at allocation (the kernel resets them). */
int spider_attn_decode_fused_bf16(const void* qkv, const int* pos, const float* cos_sin, void* k_cache, void* v_cache,
                                  const int* kv_beg, const int* kv_end, void* out, void* ws_o, void* ws_ml,
                                  int* counters, int B, int n_q, int n_kv, int d, int T_max, float scale, int nsplit,
                                  void* stream);
/* Tuning / test aid: 1 / 2 = the split-KV combine of spider_attn_decode_fused_bf16 runs inside the attention launch (last-arriving
 * block; 1: acq_rel ticket, 2: write-through partials + relaxed ticket + sc1 loads, no fence), 0 = in the separate combine launch
 * (SPIDER_ATTN_INLINE is read once, at the first call). Returns the previous setting. */
int spider_set_attn_inline(int on);
/* Tuning / test aid: split-K combine of the weight-stationary streaming conv (w_tiled = 2): 1 = inside the launch by the last-arriving
 * block of a strip (default), 0 = partial slabs + the reduce kernel (SPIDER_WS_INLAUNCH is read at the first such launch). Both forms sum
 * the slabs in split order: bit-identical outputs. The arrival counters live in the 4096 bytes behind the declared workspace: ONE stream
 * at a time may use a workspace (and its counter tail) -- concurrent streams need workspaces of their own. Returns the previous setting. */
int spider_set_ws_inlaunch(int on);

/* Batched-decode forms on a FRAGMENT-MAJOR weight copy (5..16 sequences share one weight stream; also valid for 1..4).
 * Wfm = the weight W [N, K] repacked once at load time so that every wave instruction of the kernel reads 1 KiB of contiguous
 * memory: [ceil(N/16) row groups][K/64 k blocks][2 sub-steps][64 lanes][8 bf16], lane 16 g + r of a piece holding
 * W[16 rg + r][64 kb + 32 sx + 8 g + 0..7] (the A fragment of mfma_f32_16x16x32_bf16); rows beyond N are zero. K % 64 == 0.
 * fold_rmsnorm = 0: x [B, K] is used as is. fold_rmsnorm = 1: RMSNorm folded into the projection -- x is the un-normalised
 * residual stream, Wfm was repacked from W * diag(norm_w) (bf16), and the kernel takes sum x^2 per sequence from the x fragments
 * it streams anyway and scales its accumulators by rsqrt(mean + eps): LlamaRMSNorm + nn.Linear in one launch
 * (modeling_llama3.py:77-82 + :186-199). Replaces the same nn.Linear calls as spider_gemv_bf16 /
 * spider_gemv_swiglu_bf16 / spider_lm_head_argmax_bf16 (modeling_llama3.py:186-199,240-313,870-871) at batch sizes where the
 * row-major gather (16 rows x 64 B per instruction) cannot reach the HBM stream rate. */
int spider_gemv_fm_bf16(const void* Wfm, const void* x, void* out, const void* bias, const void* res, int B, int N, int K,
                        int fold_rmsnorm, float eps, void* stream);
/* Wfm_gate_up: the repacked [gate rows (I) | up rows (I)] matrix, I % 16 == 0; out[b, i] = silu(g) * u */
int spider_gemv_swiglu_fm_bf16(const void* Wfm_gate_up, const void* x, void* out, int B, int I, int K, int fold_rmsnorm, float eps,
                               void* stream);
/* logits (optional) [B, V] bf16; ws_val / ws_idx >= B * spider_lm_head_nparts(V) entries; ties -> lowest token id */
int spider_lm_head_argmax_fm_bf16(const void* Wfm, const void* x, int* out_ids, void* logits, void* ws_val, void* ws_idx, int B,
                                  int V, int K, int fold_rmsnorm, float eps, void* stream);

/* ======================= MFMA GEMM / conv / attention ======================= */

/* C = act(A[M,K] . W[N,K]^T + bias[N] + rowbias[row/rows_per_group, N]) (+res) * out_scale.
 * Exactly one of C (bf16) / C32 (fp32). act: 0 none, 1 silu, 2 gelu(erf), 3 quick-gelu, 4 GEGLU (W = [value rows |
 * gate rows], N counts W rows, the output has N/2 columns: out = (A.Wv^T + bv) * gelu(A.Wg^T + bg), diffusers FeedForward).
 * ws/ws_bytes: optional fp32 split-K workspace (NULL = never split K); small-M / large-K problems use it
 * to fill the 256 CUs.
 * Prefill projections (modeling_llama3.py:186-313), diffusers Attention/FeedForward/proj linears. */
int spider_gemm_bf16(const void* A, const void* W, void* C, void* C32, const void* bias, const void* res,
                     const void* rowbias, int rows_per_group, int M, int N, int K, int lda, int ldc, int act,
                     float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes, void* stream);

/* C = LayerNorm(A; gamma, beta, eps) . W^T + bias (+res) in ONE launch (act 0), or its GEGLU form (act 4, as above).
 * Replaces BasicTransformerBlock.norm1 / norm2 / norm3 + the projection that consumes it (attn1 to_q/k/v, attn2 to_q,
 * ff.net.0.proj; diffusers-0.25 attention.py, reached from custom_sd.py:634-639). The normalisation is folded:
 *   Wf = W * diag(gamma) (bf16 [N,K]), colsum[n] = sum_k Wf[n,k] (fp32), colbias[n] = sum_k beta[k] W[n,k] + bias[n] (fp32)
 * are prepared once per layer by the caller; the kernel takes the row statistics of A while it stages the rows and applies
 * C = rstd * (A.Wf^T - mean * colsum) + colbias in the epilogue. Rows of A are K wide (lda = K).
 * ws (optional, >= ceil(M/256)*256*8 bytes): lets large problems run on the 256x256 LDS-DMA kernel, whose row statistics
 * come from a preceding one-wave-per-row pass instead of the staging loop. */
int spider_gemm_ln_bf16(const void* A, const void* Wf, void* C, const float* colsum, const float* colbias, const void* res,
                        int M, int N, int K, int ldc, int act, float eps, int w_tiled, void* ws, long ws_bytes, void* stream);

/* Fused cross-attention sub-block of BasicTransformerBlock (diffusers-0.25 attention.py, reached from custom_sd.py:634-639):
 *   out = x + to_out( softmax( to_q(LayerNorm(x)) K^T / sqrt(d) ) V )      for 8 heads and <= 80 text keys, ONE launch.
 * The prompt's K / V are constant over the denoising loop and are folded into the projections once per prompt:
 *   Mq[b] [8*80, C] = scale * K_h[b] . Wq_h . diag(gamma)   (key l of head h at row 80 h + l; rows of keys >= n_keys are zero)
 *   Mo[b] [C, 8*80] = Wo_h . V_h[b]^T
 *   colsum[b, r] = sum_c Mq[b][r, c],  colbias[b, r] = scale * K_h[b][l] . (Wq_h . beta)      (LayerNorm fold, fp32)
 * mq_fm / mo_fm are the fragment-major copies of Mq [B2 * 640, C] and Mo [B2 * C, 640] (layout of spider_gemv_fm_bf16's Wfm).
 * x, out [B2 * n_tok, C] bf16 (sample-major rows); C % 64 == 0, C <= 1280; n_tok % 16 == 0. */
int spider_xattn_fused_bf16(const void* x, const void* mq_fm, const void* mo_fm, const float* colsum, const float* colbias,
                            const void* bias_o, void* out, int B2, int n_tok, int C, int heads, int n_keys, float eps,
                            const float* x32, float* out32, void* stream);

/* conv2d NHWC as implicit GEMM (ResnetBlock2D / Downsample2D / Upsample2D convs reached from
 * custom_sd.py:634-639). w is OHWI [Cout,ks,ks,Cin]; ups=1 fuses the nearest-2x upsample. */
int spider_conv2d_nhwc_bf16(const void* x, const void* w, void* y, const void* bias, const void* res,
                            const void* rowbias, int B, int Hin, int Win, int Cin, int Cout, int ks, int stride,
                            int pad, int ups, float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes, void* stream);

/* General NHWC conv as implicit GEMM: rectangular / dilated kernels (w OHWI [Cout,kh,kw,Cin], Cin % 8 == 0), a fused
 * nearest upsample to an explicit size up_h x up_w in (in, 2*in] (diffusers Upsample2D with `upsample_size`, reached when
 * the AudioLDM latent height 125 is not a multiple of 8: custom_ad.py:490-504), fused activation (1 silu, 2 gelu,
 * 3 quick-gelu, 5 leaky-relu(act_param), 6 relu, 7 tanh). 1-D convs (HiFi-GAN vocoder behind
 * custom_ad.py:293-300) are Hin = kh = 1; the (3,1,1) temporal convs of UNet3D (custom_vd.py:671-676) are
 * Hin = frames, Win = H*W, kh = 3, kw = 1.
 * w_tiled: 0 = w as stored (OHWI); 1 = its tile-major copy (see spider_gemm_bf16); 2 = its FRAGMENT-MAJOR copy
 * [ceil(Cout/32)*2][(Cin/32)*9][64 lanes][8]: piece (rg, cb*9 + tap) holds at lane 16 g + r the 8 values
 * w[16 rg + r, tap, 32 cb + 8 g .. + 8] (Cout zero-padded to a multiple of 32) -- operand of the weight-stationary streaming
 * kernel that serves the weight-bound levels of the UNet (ResnetBlock2D convs / the 2x upsampler at <= 512 output pixels,
 * custom_sd.py:634-639): 3x3, stride 1, pad 1, dil 1, Cin % 32 == 0, no activation, B*Hout*Wout <= 512 (and <= 128 input
 * pixels when B*Hout*Wout <= 128); anything else with w_tiled = 2 is refused. With w_tiled = 2 the workspace must extend 4096
 * bytes beyond ws_bytes, zero-initialised once: the arrival counters of the in-launch split-K combine (the block that arrives
 * last at a strip's counter sums the partial slabs in split order and applies the epilogue; the counters are zero again when
 * the call has completed, so graph replays and later calls need no reset). */
int spider_conv_nhwc_ex_bf16(const void* x, const void* w, void* y, const void* bias, const void* res,
                             const void* rowbias, int B, int Hin, int Win, int Cin, int Cout, int kh, int kw, int stride,
                             int pad_h, int pad_w, int dil, int up_h, int up_w, int act, float act_param,
                             float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes, void* stream);

/* fused attention (prefill causal GQA: modeling_llama3.py:202-237; UNet self/cross attention:
 * StoryDiffusion/utils/gradio_utils.py:400-472; consistent self-attention with the column keep vector of
 * cal_attn_mask_xl: Comic_Generation.py:129-196, gradio_utils.py:241-287). Strides in elements. */
int spider_attn_bf16(const void* q, const void* k, const void* v, void* o,
                     long q_bs, long q_hs, long q_rs, long k_bs, long k_hs, long k_rs,
                     long v_bs, long v_hs, long v_rs, long o_bs, long o_hs, long o_rs,
                     int B, int Hq, int Hkv, int Lq, int Lk, int d, float scale, int causal, int kv_off,
                     const int* kv_beg, const void* keep_bits, int blk, int q_off, void* stream);

/* Consistent self-attention (StoryDiffusion SpatialAttnProcessor2_0 + cal_attn_mask_xl, Comic_Generation.py:129-196,
 * gradio_utils.py:241-287) through visible-key lists: query image i sees key j iff its keep bit is set or j lies in image i's
 * own block, so instead of scoring all keys and zeroing the masked ones (spider_attn_bf16 keep_bits / blk / q_off) the kernel
 * walks a per-image list of visible keys -- same softmax, ~40 % fewer keys at keep rate 0.5. head_dim <= 64 (SDXL: 64).
 * spider_story_key_lists_i32 builds, once per UNet step and resolution, key_idx [n_lists * stride] and the query-tile records
 * tiles [n_lists * ceil(N/128)][4]; spider_attn_keylist_bf16 consumes them (strides as in spider_attn_bf16). */
int spider_story_key_lists_i32(const void* keep_bits, int n_keys, int N, int img0, int n_lists, int q_img0, int stride,
                               int* key_idx, int* tiles, void* stream);
int spider_attn_keylist_bf16(const void* q, const void* k, const void* v, void* o,
                             long q_bs, long q_hs, long q_rs, long k_bs, long k_hs, long k_rs,
                             long v_bs, long v_hs, long v_rs, long o_bs, long o_hs, long o_rs,
                             int B, int Hq, int Hkv, int Lq, int Lk, int d, float scale,
                             const int* key_idx, int idx_len, const int* tiles, int n_tiles, void* stream);

/* packed variable-length attention: the per-segment SDPA loop over cu_seqlens of transformers'
 * Qwen2_5OmniVisionAttention / Qwen2_5OmniAudioAttention (window / full-frame / audio-chunk segments), reached from the
 * reference through Qwen2_5OmniModel.generate(**inputs) with images / audios (qwen2.5omni_spider_web.py:461-468).
 * q/k/v/o: [total_rows, heads, d] views (row strides in elements, head stride d). tiles: device int[n_tiles][4] =
 * {q_start, q_len (1..128), k_start, k_len}; the caller guarantees every range lies inside [0, total_rows). */
int spider_attn_varlen_bf16(const void* q, const void* k, const void* v, void* o, long q_rs, long k_rs, long v_rs, long o_rs,
                            int total_rows, int Hq, int Hkv, int d, float scale, const int* tiles, int n_tiles, void* stream);

/* in-place half-rotation RoPE with a per-row angle table: transformers apply_rotary_pos_emb_vision of the Qwen2.5-Omni
 * vision tower (same call path as above). x: bf16 rows of heads*d contiguous elements, row_stride apart;
 * cos_sin: fp32 [rows, d] = [cos(d/2) | sin(d/2)]. */
int spider_rope_rows_bf16(void* x, const float* cos_sin, long row_stride, int rows, int heads, int d, void* stream);

/* ======================= UNet elementwise / normalisation (HBM-bound) ======================= */

/* GroupNorm(+SiLU) on NHWC (ResnetBlock2D.norm1/2, Transformer2DModel.norm, conv_norm_out).
 * ws: B*spider_groupnorm_nchunk(HW)*G*2 floats. */
int spider_groupnorm_nchunk(int HW);
int spider_groupnorm_nhwc_bf16(const void* x, const void* gamma, const void* beta, void* y, void* ws, int B, int HW,
                               int C, int G, float eps, int silu, void* stream);
/* The same GroupNorm on the channel concatenation [x1 | x2] without materialising it first: UpBlock2D / UpBlock3D's
 * torch.cat([hidden_states, res_hidden_states], dim=1) followed by ResnetBlock2D.norm1 (diffusers unet_2d_blocks, called from
 * custom_sd.py:634-639). x1 [B,HW,C1], x2 [B,HW,C2] are read in place; y [B,HW,C1+C2] is the normalised (+SiLU) result and
 * cat [B,HW,C1+C2] receives the concatenated input (the resnet's 1x1 conv_shortcut reads it). C1, C2 multiples of 8. */
int spider_groupnorm_cat_nhwc_bf16(const void* x1, const void* x2, const void* gamma, const void* beta, void* y, void* cat,
                                   void* ws, int B, int HW, int C1, int C2, int G, float eps, int silu, void* stream);
/* GroupNorm with the statistics taken from the PRODUCER of its input (round 4; diffusers ResnetBlock2D: conv1 -> norm2, conv2 ->
 * the next block's norm; Transformer2DModel: norm -> proj_in). The reference runs torch.nn.GroupNorm as its own pass over the
 * tensor (custom_sd.py:634-639 -> UNet2DConditionModel.forward); here the conv that writes the tensor also writes, per chunk of
 * `chunk_rows` consecutive output pixels and per group, (sum, sum of squares) of the 16-bit values it stores -- the [B, nchunk, G, 2]
 * fp32 partial array every consumer reduces in a fixed order (no atomics: hipGraph replay stays bit-identical to eager launches).
 *   spider_conv_nhwc_gn_*: spider_conv_nhwc_ex_* + gn_part (room for B * HW / 16 * gn_groups * 2 floats) / gn_groups; *produced =
 *       rows per chunk of the statistics written, nchunk = HW / *produced: 64 (LDS-DMA conv epilogue), 16 (split-K reduce), or 0
 *       when this shape's kernel cannot write them (run spider_groupnorm_stats_* instead). HW must be a multiple of 16.
 *   spider_groupnorm_stats_nhwc_*: the statistics pass alone (any nchunk).
 *   spider_groupnorm_apply_nhwc_*: normalise (+ SiLU) with given partials.
 *   spider_gemm_gn_in_*: C = GroupNorm(A) W^T + bias with the normalisation applied to A on its way into LDS (norm + proj_in in
 *       one launch); c32d optional fp32 copy of C (fp32 residual stream). */
int spider_conv_nhwc_gn_bf16(const void* x, const void* w, void* y, const void* bias, const void* res, const void* rowbias,
                             int B, int Hin, int Win, int Cin, int Cout, int kh, int kw, int stride, int pad_h, int pad_w,
                             int dil, int up_h, int up_w, int act, float act_param, float out_scale, int w_tiled,
                             const float* res32, float* c32d, void* ws, long ws_bytes, void* stream,
                             float* gn_part, int gn_groups, int* produced);
int spider_groupnorm_stats_nhwc_bf16(const void* x, void* partial, int B, int HW, int C, int G, int nchunk, void* stream);
int spider_groupnorm_apply_nhwc_bf16(const void* x, const void* partial, int nchunk, const void* gamma, const void* beta, void* y,
                                     int B, int HW, int C, int G, float eps, int silu, void* stream);
int spider_gemm_gn_in_bf16(const void* A, const void* W, void* C, const void* bias, int M, int N, int K, int ldc, int w_tiled,
                           const float* gn_part, int nchunk, const void* gamma, const void* beta, int G, float eps, int HW,
                           float* c32d, void* stream);
int spider_layernorm_bf16(const void* x, const void* gamma, const void* beta, void* y, int rows, int C, float eps,
                          void* stream);
/* GEGLU (diffusers FeedForward): y[m,n] = x[m,n] * gelu(x[m,inner+n]) */
int spider_geglu_bf16(const void* x, void* y, int M, int inner, void* stream);
/* SwiGLU on a fused [gate|up] prefill projection (modeling_llama3.py:197-199) */
int spider_swiglu_bf16(const void* x, void* y, int M, int inner, void* stream);
int spider_concat_channels_bf16(const void* a, const void* b, void* y, long rows, int C1, int C2, void* stream);
int spider_act_bf16(const void* x, void* y, long n, int act, void* stream);
int spider_add_bf16(const void* a, const void* b, void* y, long n, void* stream);
/* act_ex: 1 silu, 2 gelu(erf), 3 quick-gelu, 5 leaky-relu(param), 6 relu, 7 tanh (HiFi-GAN / CLAP pooler+projection) */
int spider_act_ex_bf16(const void* x, void* y, long n, int act, float param, void* stream);
/* y = (a + b) * scale  (HiFi-GAN: mean of the residual-block branches) */
int spider_add_scaled_bf16(const void* a, const void* b, void* y, long n, float scale, void* stream);
/* y = alpha*a + beta*b: the 0.1 / 0.9 blend of projected LLM states with text-encoder embeddings (spider.py:420,432,444) */
int spider_axpby_bf16(const void* a, const void* b, void* y, long n, float alpha, float beta, void* stream);
/* TextFcLayerMoE router input: mean over tokens, x [B,T,C] -> [B,C] (spider/models/layers.py:254) */
int spider_mean_tokens_bf16(const void* x, void* y, int B, int T, int C, void* stream);
/* TextFcLayerMoE mixing: r = sigmoid(logits[b,:E]) / sum, y[b,...] = sum_e r[e] * x_e[b,...]; host_xs is a HOST array of E
 * device pointers [B, per_batch]; logits rows are ld wide (layers.py:255-267) */
int spider_moe_combine_bf16(const void* const* host_xs, int E, const void* logits, int ld, void* y, int B, long per_batch,
                            void* stream);
/* ConvTranspose1d = per-tap GEMM (spider_gemm_bf16 with fp32 output, cols [B,L_in,k,Cout]) + this overlap-add:
 * y[b,t,:] = bias + sum_{i*stride - pad + j == t} cols[b,i,j,:]; L_out = (L_in-1)*stride - 2*pad + k
 * (HiFi-GAN upsampler, SpeechT5HifiGan called from custom_ad.py:293-300). */
int spider_col2im1d_f32_bf16(const float* cols, const void* bias, void* y, int B, int L_in, int k, int stride, int pad,
                             int Cout, void* stream);
/* y[r,:] = x[r,:] / max(||x[r,:]||_2, eps)  (F.normalize of the CLAP text embedding, custom_ad.py:217-219,272-273) */
int spider_l2_normalize_rows_bf16(const void* x, void* y, int rows, int n, float eps, void* stream);
int spider_conv2d_small_cin_bf16(const void* x, const void* w, const void* bias, void* y, int B, int H, int W, int Cin,
                                 int Cout, int ks, void* stream);
int spider_conv2d_small_cout_bf16(const void* x, const void* w, const void* bias, void* y32, void* y16, int B, int H,
                                  int W, int Cin, int Cout, int ks, void* stream);

/* ---- latent plumbing of the denoising loop (custom_sd.py:631-647) ---- */
/* torch.cat([latents]*reps) + scale_model_input: fp32 NCHW -> bf16 NHWC */
int spider_latent_to_nhwc_bf16(const float* lat, void* out, int B, int C, int HW, int reps, float scale, void* stream);
/* noise_pred_uncond + g*(noise_pred_text - noise_pred_uncond): fp32 NHWC [2,B,HW,C] -> fp32 NCHW */
int spider_cfg_combine_f32(const float* eps2, float* out, int B, int C, int HW, float guidance, void* stream);
/* scheduler.step as a linear update: out = sum_j host_coefs[j] * ins[j]; host_ins is a HOST array of n device ptrs */
int spider_lincomb_f32(const float* const* host_ins, const float* host_coefs, int n, float* out, long total, void* stream);
/* row softmax of fp32 scores -> bf16 probabilities (VAE mid-block single-head attention, d = 512). Rows are n wide;
 * the first n_valid columns are normalised, the rest (padding up to a multiple of 8 for the P.V GEMM) written as 0. */
int spider_softmax_rows_f32_bf16(const float* x, void* y, int rows, int n, int n_valid, float scale, void* stream);
int spider_nhwc_to_nchw_f32(const float* x, float* y, int B, int C, int HW, float mul, float add, int clamp01, void* stream);

/* StoryDiffusion keep vector -> bit words for spider_attn_bf16's keep_bits (cal_attn_mask_xl, utils/gradio_utils.py:241-287, reduced to
 * the random per-key keep decision): bit j % 64 of word j / 64 = (u[j] < thr) for j < n_valid, 0 beyond. u [n] fp32 on the device. */
int spider_pack_keep_bits_f32(const float* u, void* words, int n, int n_valid, float thr, void* stream);


/* ---- "precise" operand forms (ABI v4; UNetEngine(precise=True)) ----
 * north_star asks for UNet latents within 1e-3 relative of the reference; with every MFMA operand rounded to 11 significand bits one
 * evaluation sits at 1.2e-3. The per-site attribution (scripts/exp/precision_sites.py) puts ~70 % of that error variance on the few
 * places where the fp32 MASTER of the residual stream exists but its 16-bit shadow is what a kernel reads: ResnetBlock2D.conv_shortcut,
 * the Down / Upsample2D convs, conv_in / conv_out, Transformer2DModel.norm -> proj_in, proj_out, and every GroupNorm of the stream
 * (diffusers 0.25 UNet2DConditionModel.forward; call site custom_sd.py:634-639). These entry points read the fp32 tensor instead:
 *   spider_gemm_a32_*, spider_conv_nhwc_a32_*, spider_gemm_gn_in_a32_*: A (or the NHWC image) is fp32 and is split, on its way into
 *       LDS, into hi = round16(x) and lo = round16(x - hi); every K step runs two MFMAs (W.hi + W.lo). Otherwise the contracts of
 *       spider_gemm_* (no activation / GEGLU), spider_conv_nhwc_ex_* (no activation; w_tiled 0 / 1) and spider_gemm_gn_in_*.
 *   spider_gemm_ln_a32_*: LayerNorm(A32) . W^T (+ GEGLU) on the fp32 rows with gamma applied to A and W kept exact (spider_gemm_ln's
 *       fold re-rounds W * gamma); colsum[n] = sum_k gamma[k] W[n,k], colbias[n] = sum_k beta[k] W[n,k] + bias[n], both fp32.
 *   spider_groupnorm_f32in_nhwc_*: GroupNorm (+ SiLU) of the fp32 tensor x32 [B, HW, C]; statistics from `partial` [B, nchunk, G, 2]
 *       or (partial NULL) from a pass over x32 into ws (>= B * spider_groupnorm_nchunk(HW) * G * 2 floats); the result is rounded ONCE
 *       into y (16-bit, optional) and / or stored unrounded into y32 (optional).
 *   spider_split_hilo_f32_*: the operand split as a pass of its own (hi = round16(x), lo = round16(x - hi)) for convs large enough to run
 *       on the LDS-DMA / 256^2 kernels twice: conv(hi) (c32d out), then conv(lo) with res32 = that result -- the same fp32 sum.
 *   spider_row_split_f32_* / spider_groupnorm_f32in_split_nhwc_*: "split once, doubled K" -- the operand v (x32 itself, LayerNorm(x32)
 *       * gamma + beta with fp32 two-pass row statistics, or GroupNorm(x32) (+ SiLU)) is written ONCE as y2 [M, 2K] = [round16(v) |
 *       round16(v - round16(v))]; spider_gemm_* / spider_conv_nhwc_ex_* over y2 against the weight repeated along K ([W | W]: channels
 *       [0, Cin) and [Cin, 2 Cin) of every tap for a conv) accumulate hi . W + lo . W in their fp32 accumulator -- the product of the
 *       a32 entry points above on the LDS-DMA / 256^2 tile kernels, with every epilogue of those (GEGLU, rowbias, res32, c32d, GroupNorm
 *       statistics). The host side takes this route above a flop threshold (spider_amd/ops.py: A32_DUP_MIN_FLOP).
 *   spider_conv2d_small_cin_f32in_* / spider_conv2d_small_cout_f32in_*: conv_in on the fp32 NHWC latents (y32: optional fp32 master of
 *       its output) and conv_out on the fp32 GroupNorm + SiLU output.  spider_latent_to_nhwc_f32: fp32 NCHW latents -> fp32 NHWC. */
int spider_gemm_a32_bf16(const float* A32, const void* W, void* C, const void* bias, const void* res, int M, int N, int K, int lda,
                         int ldc, float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes, void* stream);
int spider_gemm_ln_a32_bf16(const float* A32, const void* W, void* C, const void* gamma, const float* colsum, const float* colbias,
                            int M, int N, int K, int ldc, int act, float eps, int w_tiled, void* stream);
int spider_gemm_gn_in_a32_bf16(const float* A32, const void* W, void* C, const void* bias, int M, int N, int K, int ldc, int w_tiled,
                               const float* gn_part, int nchunk, const void* gamma, const void* beta, int G, float eps, int HW,
                               float* c32d, void* stream);
int spider_conv_nhwc_a32_bf16(const float* x32, const void* w, void* y, const void* bias, const void* res, const void* rowbias,
                              int B, int Hin, int Win, int Cin, int Cout, int kh, int kw, int stride, int pad_h, int pad_w, int dil,
                              int up_h, int up_w, float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes,
                              void* stream);
int spider_groupnorm_f32in_nhwc_bf16(const float* x32, const float* partial, int nchunk, const void* gamma, const void* beta, void* y,
                                     float* y32, float* ws, int B, int HW, int C, int G, float eps, int silu, void* stream);
int spider_split_hilo_f32_bf16(const float* x, void* hi, void* lo, long n, void* stream);
int spider_row_split_f32_bf16(const float* x32, const void* gamma, const void* beta, void* y2, long M, int K, float eps, void* stream);
int spider_groupnorm_f32in_split_nhwc_bf16(const float* x32, const float* partial, int nchunk, const void* gamma, const void* beta,
                                            void* y2, float* ws, int B, int HW, int C, int G, float eps, int silu, void* stream);
int spider_conv2d_small_cin_f32in_bf16(const float* x32, const void* w, const void* bias, void* y, float* y32, int B, int H, int W,
                                       int Cin, int Cout, int ks, void* stream);
int spider_conv2d_small_cout_f32in_bf16(const float* x32, const void* w, const void* bias, float* y32, int B, int H, int W, int Cin,
                                        int Cout, int ks, void* stream);
int spider_latent_to_nhwc_f32(const float* lat, float* out, int B, int C, int HW, int reps, float scale, void* stream);

/* ======================= IEEE-half (f16) instantiations of the diffusion-side operators =======================
 * The reference runs its diffusion decoders in torch.float16 (spider_decoder.py:109,114,130,136,153,159; base_model.py:211;
 * StoryDiffusion/Comic_Generation.py:313). Every operator above that the UNet / VAE / text-encoder engines use also exists with
 * f16 tensors (raw uint16 IEEE-half bit patterns) in place of bf16: same arguments, same semantics, fp32 accumulation,
 * mfma_f32_*_f16 at the bf16 rate; outputs beyond +-65504 round to +-inf as torch.float16 does. The LLM decode operators
 * exist in bf16 only (the reference's LLM dtype). */
int spider_gemm_f16(const void* A, const void* W, void* C, void* C32, const void* bias, const void* res,
                     const void* rowbias, int rows_per_group, int M, int N, int K, int lda, int ldc, int act,
                     float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes, void* stream);
int spider_gemm_ln_f16(const void* A, const void* Wf, void* C, const float* colsum, const float* colbias, const void* res,
                        int M, int N, int K, int ldc, int act, float eps, int w_tiled, void* ws, long ws_bytes, void* stream);
int spider_xattn_fused_f16(const void* x, const void* mq_fm, const void* mo_fm, const float* colsum, const float* colbias,
                            const void* bias_o, void* out, int B2, int n_tok, int C, int heads, int n_keys, float eps,
                            const float* x32, float* out32, void* stream);
int spider_conv2d_nhwc_f16(const void* x, const void* w, void* y, const void* bias, const void* res,
                            const void* rowbias, int B, int Hin, int Win, int Cin, int Cout, int ks, int stride,
                            int pad, int ups, float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes, void* stream);
int spider_conv_nhwc_ex_f16(const void* x, const void* w, void* y, const void* bias, const void* res,
                             const void* rowbias, int B, int Hin, int Win, int Cin, int Cout, int kh, int kw, int stride,
                             int pad_h, int pad_w, int dil, int up_h, int up_w, int act, float act_param,
                             float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes, void* stream);
int spider_attn_f16(const void* q, const void* k, const void* v, void* o,
                     long q_bs, long q_hs, long q_rs, long k_bs, long k_hs, long k_rs,
                     long v_bs, long v_hs, long v_rs, long o_bs, long o_hs, long o_rs,
                     int B, int Hq, int Hkv, int Lq, int Lk, int d, float scale, int causal, int kv_off,
                     const int* kv_beg, const void* keep_bits, int blk, int q_off, void* stream);
int spider_attn_keylist_f16(const void* q, const void* k, const void* v, void* o,
                             long q_bs, long q_hs, long q_rs, long k_bs, long k_hs, long k_rs,
                             long v_bs, long v_hs, long v_rs, long o_bs, long o_hs, long o_rs,
                             int B, int Hq, int Hkv, int Lq, int Lk, int d, float scale,
                             const int* key_idx, int idx_len, const int* tiles, int n_tiles, void* stream);
int spider_attn_varlen_f16(const void* q, const void* k, const void* v, void* o, long q_rs, long k_rs, long v_rs, long o_rs,
                            int total_rows, int Hq, int Hkv, int d, float scale, const int* tiles, int n_tiles, void* stream);
int spider_groupnorm_nhwc_f16(const void* x, const void* gamma, const void* beta, void* y, void* ws, int B, int HW,
                               int C, int G, float eps, int silu, void* stream);
int spider_conv_nhwc_gn_f16(const void* x, const void* w, void* y, const void* bias, const void* res, const void* rowbias,
                            int B, int Hin, int Win, int Cin, int Cout, int kh, int kw, int stride, int pad_h, int pad_w,
                            int dil, int up_h, int up_w, int act, float act_param, float out_scale, int w_tiled,
                            const float* res32, float* c32d, void* ws, long ws_bytes, void* stream,
                            float* gn_part, int gn_groups, int* produced);
int spider_groupnorm_stats_nhwc_f16(const void* x, void* partial, int B, int HW, int C, int G, int nchunk, void* stream);
int spider_groupnorm_apply_nhwc_f16(const void* x, const void* partial, int nchunk, const void* gamma, const void* beta, void* y,
                                    int B, int HW, int C, int G, float eps, int silu, void* stream);
int spider_gemm_gn_in_f16(const void* A, const void* W, void* C, const void* bias, int M, int N, int K, int ldc, int w_tiled,
                          const float* gn_part, int nchunk, const void* gamma, const void* beta, int G, float eps, int HW,
                          float* c32d, void* stream);
int spider_groupnorm_cat_nhwc_f16(const void* x1, const void* x2, const void* gamma, const void* beta, void* y, void* cat,
                                   void* ws, int B, int HW, int C1, int C2, int G, float eps, int silu, void* stream);
int spider_layernorm_f16(const void* x, const void* gamma, const void* beta, void* y, int rows, int C, float eps,
                          void* stream);
int spider_geglu_f16(const void* x, void* y, int M, int inner, void* stream);
int spider_swiglu_f16(const void* x, void* y, int M, int inner, void* stream);
int spider_concat_channels_f16(const void* a, const void* b, void* y, long rows, int C1, int C2, void* stream);
int spider_act_f16(const void* x, void* y, long n, int act, void* stream);
int spider_add_f16(const void* a, const void* b, void* y, long n, void* stream);
int spider_act_ex_f16(const void* x, void* y, long n, int act, float param, void* stream);
int spider_add_scaled_f16(const void* a, const void* b, void* y, long n, float scale, void* stream);
int spider_axpby_f16(const void* a, const void* b, void* y, long n, float alpha, float beta, void* stream);
int spider_mean_tokens_f16(const void* x, void* y, int B, int T, int C, void* stream);
int spider_moe_combine_f16(const void* const* host_xs, int E, const void* logits, int ld, void* y, int B, long per_batch,
                            void* stream);
int spider_col2im1d_f32_f16(const float* cols, const void* bias, void* y, int B, int L_in, int k, int stride, int pad,
                             int Cout, void* stream);
int spider_l2_normalize_rows_f16(const void* x, void* y, int rows, int n, float eps, void* stream);
int spider_conv2d_small_cin_f16(const void* x, const void* w, const void* bias, void* y, int B, int H, int W, int Cin,
                                 int Cout, int ks, void* stream);
int spider_conv2d_small_cout_f16(const void* x, const void* w, const void* bias, void* y32, void* y16, int B, int H,
                                  int W, int Cin, int Cout, int ks, void* stream);
int spider_latent_to_nhwc_f16(const float* lat, void* out, int B, int C, int HW, int reps, float scale, void* stream);
int spider_softmax_rows_f32_f16(const float* x, void* y, int rows, int n, int n_valid, float scale, void* stream);

int spider_gemm_a32_f16(const float* A32, const void* W, void* C, const void* bias, const void* res, int M, int N, int K, int lda,
                         int ldc, float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes, void* stream);
int spider_gemm_ln_a32_f16(const float* A32, const void* W, void* C, const void* gamma, const float* colsum, const float* colbias,
                            int M, int N, int K, int ldc, int act, float eps, int w_tiled, void* stream);
int spider_gemm_gn_in_a32_f16(const float* A32, const void* W, void* C, const void* bias, int M, int N, int K, int ldc, int w_tiled,
                               const float* gn_part, int nchunk, const void* gamma, const void* beta, int G, float eps, int HW,
                               float* c32d, void* stream);
int spider_conv_nhwc_a32_f16(const float* x32, const void* w, void* y, const void* bias, const void* res, const void* rowbias,
                              int B, int Hin, int Win, int Cin, int Cout, int kh, int kw, int stride, int pad_h, int pad_w, int dil,
                              int up_h, int up_w, float out_scale, int w_tiled, const float* res32, float* c32d, void* ws, long ws_bytes,
                              void* stream);
int spider_groupnorm_f32in_nhwc_f16(const float* x32, const float* partial, int nchunk, const void* gamma, const void* beta, void* y,
                                     float* y32, float* ws, int B, int HW, int C, int G, float eps, int silu, void* stream);
int spider_split_hilo_f32_f16(const float* x, void* hi, void* lo, long n, void* stream);
int spider_row_split_f32_f16(const float* x32, const void* gamma, const void* beta, void* y2, long M, int K, float eps, void* stream);
int spider_groupnorm_f32in_split_nhwc_f16(const float* x32, const float* partial, int nchunk, const void* gamma, const void* beta,
                                            void* y2, float* ws, int B, int HW, int C, int G, float eps, int silu, void* stream);
int spider_conv2d_small_cin_f32in_f16(const float* x32, const void* w, const void* bias, void* y, float* y32, int B, int H, int W,
                                       int Cin, int Cout, int ks, void* stream);
int spider_conv2d_small_cout_f32in_f16(const float* x32, const void* w, const void* bias, float* y32, int B, int H, int W, int Cin,
                                        int Cout, int ks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPIDER_HIP_H */
