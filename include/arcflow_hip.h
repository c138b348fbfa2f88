/*
 * arcflow_hip.h -- C ABI of libarcflow_hip.so, the MI355X (gfx950) engine for the ArcFlow
 * 2-NFE hot path.
 *
 * The reference (pnotp/ArcFlow) has no FFI: its hot path is Python calling torch/diffusers
 * modules.  This header is the boundary a maintainer binds instead (ctypes stub in
 * INTEGRATION.md); every entry point names the reference interface it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch types.
 *   - all data pointers are DEVICE pointers owned by the caller (e.g. torch tensors);
 *     the library owns only the opaque afx_ctx (weight table, workspace pointer, plan).
 *   - every call is asynchronous on the given hipStream_t (passed as void*; NULL = default
 *     stream); the library never synchronises, allocates or frees device memory.
 *   - return value: 0 on success, a negative AFX_E_* code otherwise; afx_last_error() returns
 *     a human readable message for the calling thread.  Nothing throws across the boundary.
 *   - a context is thread-compatible, not thread-safe.
 *   - bf16 tensors are raw uint16 bit patterns, row-major, innermost dimension contiguous.
 */
#ifndef ARCFLOW_HIP_H_
#define ARCFLOW_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFX_OK 0
#define AFX_MAX_MICRO_BATCH 4  /* samples per grouped launch: afx_mmdit_forward splits larger batches itself; staged calls take <= 4 */
#define AFX_E_INVALID (-1)     /* bad argument / shape */
#define AFX_E_MISSING (-2)     /* a required weight is not bound */
#define AFX_E_WORKSPACE (-3)   /* workspace missing or too small */
#define AFX_E_HIP (-4)         /* a HIP runtime call failed */
#define AFX_E_UNSUPPORTED (-5)

#define AFX_DT_BF16 1
#define AFX_DT_F32 2
#define AFX_DT_FP8 3   /* OCP e4m3 bytes (quantised weights of the fp8 linear mode) */

typedef struct afx_ctx afx_ctx;

/* ---- model description -------------------------------------------------------------------
 * Mirrors the constructor arguments of the reference denoisers
 *   lakonlab/models/architecture/arcflow/arcflux.py:27-39   (_ArcFluxTransformer2DModel)
 *   lakonlab/models/architecture/arcflow/arcqwen.py:25-36   (_ArcQwenImageTransformer2DModel)
 */
typedef struct afx_model_desc {
  int32_t family;             /* 0 = FLUX MMDiT (double + single blocks), 1 = Qwen-Image MMDiT */
  int32_t num_double;         /* num_layers: 19 (FLUX) / 60 (Qwen) */
  int32_t num_single;         /* num_single_layers: 38 (FLUX) / 0 (Qwen) */
  int32_t heads;              /* num_attention_heads (24) */
  int32_t head_dim;           /* attention_head_dim (must be 128) */
  int32_t in_channels;        /* 64 */
  int32_t joint_dim;          /* joint_attention_dim: 4096 / 3584 */
  int32_t pooled_dim;         /* pooled_projection_dim: 768 (FLUX) / 0 (Qwen) */
  int32_t guidance_embeds;    /* 1 for FLUX.1-dev */
  int32_t num_gaussians;      /* K = 16 */
  int32_t logweights_channels;/* 4 (= patch_size^2) */
  int32_t head_mode;          /* 0 = ArcFlow 3-head student, 1 = plain proj_out teacher head */
} afx_model_desc;

const char* afx_last_error(void);
const char* afx_version(void);

/* ---- context life cycle ------------------------------------------------------------------ */
int afx_create(const afx_model_desc* desc, afx_ctx** out);
int afx_destroy(afx_ctx* ctx);

/* Bind one packed weight (device pointer stays owned by the caller and must outlive the ctx).
 * Names are the engine's packed names (arcflow_amd/weights.py builds them from the diffusers
 * state-dict keys the reference loads, lakonlab/pipelines/arcflow_loader.py:241-263):
 *   x_in.{weight,bias}  ctx_in.{weight,bias}  txt_norm.weight
 *   temb.{t,g,p}.l{1,2}.{weight,bias}
 *   mod.{weight,bias}                       all AdaLN modulation linears stacked on N
 *   d<i>.{img,txt}_{qkv,out,mlp1,mlp2}.{weight,bias}   d<i>.qknorm   (rows k|v|q in *_qkv)
 *   s<i>.{fused,out}.{weight,bias}          s<i>.qknorm              (rows k|v|q|mlp in fused)
 *   head.{weight,bias}
 *   mod_final.{weight,bias}   optional: a separately owned norm_out.linear [2D, D] that overrides the last 2D rows
 *                             of mod.* (the distillation student trains it while the teacher keeps the frozen copy)
 */
int afx_bind_weight(afx_ctx* ctx, const char* name, const void* dptr, int32_t dtype,
                    int32_t ndim, const int64_t* shape);
/* Verify every weight the description requires is bound with the right shape. */
int afx_finalize(afx_ctx* ctx);

/* Scratch memory: size for the largest (batch, image tokens, text tokens) the ctx will see. */
int64_t afx_workspace_bytes(const afx_ctx* ctx, int32_t batch, int32_t n_img, int32_t n_txt);
int afx_set_workspace(afx_ctx* ctx, void* dptr, int64_t bytes);

/* ---- denoiser forward ---------------------------------------------------------------------
 * Replaces  transformer(hidden_states, timestep, guidance, pooled_projections,
 *                        encoder_hidden_states, txt_ids, img_ids)
 *   lakonlab/pipelines/arcflux_pipeline.py:469-479  ->  arcflux.py:134-257
 *   lakonlab/pipelines/arcqwen_pipeline.py:409-418  ->  arcqwen.py:106-174
 * x        [B, N, in_channels]      bf16  packed latents
 * ctx_emb  [B, T, joint_dim]        bf16  text-encoder states (Qwen: only the T real tokens)
 * pooled   [B, pooled_dim]          bf16  (FLUX) or NULL
 * t        [B] f32  sigma in [0,1] (the pipeline's timestep/1000);  g [B] f32 guidance or NULL.  As the reference's forward
 *          does (arcflux.py:160-162 `timestep.to(hidden_states.dtype) * 1000`, arcqwen.py:128), t and g are cast to bf16 in
 *          front of the sinusoid (FLUX: the x1000 product too -- sigma 0.76190 is embedded as 760, guidance 3.5 as 3504)
 * rope_cos/rope_sin [T+N, head_dim/2] f32 rotation tables of the joint [text; image] sequence
 *          (FluxPosEmbed / QwenEmbedRope angles; built by the host, see arcflow_amd/rope.py)
 * outputs (head_mode 0): means [B,N,K,in_channels], logw [B,N,K,lw] (log_softmax over K),
 *          logg [B,N,K-1,lw], all bf16  == ArcFlowModelOutput (arc_output.py:9-25)
 * outputs (head_mode 1): means receives the velocity [B,N,in_channels]; logw/logg unused.
 */
int afx_mmdit_forward(afx_ctx* ctx, const void* x, const void* ctx_emb, const void* pooled,
                      const float* t, const float* g, const float* rope_cos, const float* rope_sin,
                      int32_t batch, int32_t n_img, int32_t n_txt,
                      void* means, void* logw, void* logg, void* stream);

/* The conditioning of SEVERAL denoising steps in one pass.  The AdaLN modulation vectors of a forward depend only on (t, guidance,
 * pooled text) -- the reference recomputes them inside every transformer call (arcflux.py:134-257: time_text_embed, then each
 * block's norm1 / norm linear) -- and computing them means streaming the stacked modulation matrix (6.5 GB for FLUX, 1.3 ms)
 * once per forward.  A sampler knows all its timesteps up front (arcflux_pipeline.py:455-467), so:
 *   afx_mmdit_prepare_steps(ctx, pooled, t_steps [nsteps][batch] f32, g, batch, nsteps, stream)   batch <= 4, batch * nsteps <= 8
 * streams the matrix ONCE for all steps, and afx_mmdit_use_prepared_step(ctx, k) makes the NEXT afx_mmdit_forward (same batch)
 * take step k's vectors instead of recomputing them (one-shot; k = -1 cancels; the t / g / pooled arguments of that forward are
 * not looked at for the conditioning).  Results are bit-identical to the plain call.  The prepared vectors live in the workspace:
 * a new afx_set_workspace or a changed pooled / guidance needs a new prepare. */
int afx_mmdit_prepare_steps(afx_ctx* ctx, const void* pooled, const float* t_steps, const float* g, int32_t batch, int32_t nsteps,
                            void* stream);
int afx_mmdit_use_prepared_step(afx_ctx* ctx, int32_t k);

/* Gradient checkpointing (arcflux.py:181-189,315-316 checkpoint every block): when a buffer of
 * (num_double + num_single) x [B*(T+N), D] bf16 is set, afx_mmdit_forward stores every block's input token matrix
 * there; the training trunk recomputes one block at a time from it.  NULL switches it off. */
int afx_set_checkpoint_buffer(afx_ctx* ctx, void* dptr);

/* Optional instrumentation for bench.py's roofline line: when enabled, afx_mmdit_forward records a HIP
 * event pair on its stream around every GEMM launch (klass 0) and attention launch (klass 1).
 * afx_profile_read waits for the recorded events and returns the summed duration, the number of
 * launches and their algorithmic FLOPs since the last afx_profile_enable(ctx, 1).  on = N > 1 times ONE launch in N (a counter over both
 * classes; a forward has an odd number of launches, so over several forwards the sample covers every launch position): an event pair on a
 * dispatch costs ~4 us, 1.8 % of the FLUX forward when every launch carries one (round 4, profiles/r04a_no_profile_ab.txt). */
int afx_profile_enable(afx_ctx* ctx, int32_t on);
int afx_profile_read(afx_ctx* ctx, int32_t klass, double* total_ms, int64_t* launches, double* flops);

/* ---- analytic ArcFlow step, token layout --------------------------------------------------
 * Replaces _unpack_latents + _unpack_mp + ArcFlowPolicy + momentum_integration + _pack_latents
 *   lakonlab/pipelines/arcflux_pipeline.py:482-510 (:195-249), arcqwen_pipeline.py:421-451
 *   training twin: lakonlab/models/diffusions/arcflow.py:28-79
 * x_in/x_out [B,N,ch] f32 (may alias); means [B,N,K,ch]; logw [B,N,K,pp]; logg [B,N,K-1,pp]
 * (mix_dtype AFX_DT_BF16 or AFX_DT_F32); sigma_* per call scalars, or per-sample device arrays
 * sigma_vec[3*B] = {src,start,end} per sample when sigma_vec != NULL.
 *   x_out = x - D * sum_k softmax(logw)_k m_k d_k phi(g_k D),  D = sigma_start - sigma_end.
 */
int afx_arcflow_step(const float* x_in, const void* means, const void* logw, const void* logg,
                     int32_t mix_dtype, float sigma_src, float sigma_start, float sigma_end,
                     const float* sigma_vec, float eps, float* x_out,
                     int32_t batch, int32_t n_tok, int32_t K, int32_t ch, int32_t pp, void* stream);

/* u = sum_k softmax(logw)_k m_k exp(g_k (sigma_src - sigma_t))
 *   lakonlab/models/diffusions/policies/arcflow.py:52-76 */
int afx_arcflow_velocity(const void* means, const void* logw, const void* logg, int32_t mix_dtype,
                         float sigma_src, float sigma_t, const float* sigma_vec, float* u_out,
                         int32_t batch, int32_t n_tok, int32_t K, int32_t ch, int32_t pp, void* stream);

/* ---- distillation-step kernels (training side) ------------------------------------------------
 * The reference runs these as eager torch + autograd; shapes are the token layout of afx_arcflow_step. */

/* Analytic step with per-sample sigmas sigma_vec[3*B] = {src,start,end} and GM dropout: component k of
 * sample b is removed (log-weight -> -inf) when drop_mask[b*K+k] != 0 (policies/arcflow.py:96-106);
 * roll-outs of the detached policy in piid_segment_momentum (arcflow.py:141-144,176-178,201-205). */
int afx_arcflow_step_dropout(const float* x_in, const void* means, const void* logw, const void* logg,
                             int32_t mix_dtype, const float* sigma_vec, const uint8_t* drop_mask, float eps,
                             float* x_out, int32_t batch, int32_t n_tok, int32_t K, int32_t ch, int32_t pp,
                             void* stream);

/* Gradients of D = sum_k softmax(logw)_k m_k e_k w.r.t. (means, logw, logg) given gD = gscale * gscale_vec[b] * g:
 * step form (velocity_only = 0): e_0 = Dt, e_k = exp(g_k Dp) Dt phi(g_k Dt)  -- x_end = x - D
 * velocity form (1):             e_0 = 1,  e_k = exp(g_k Dp)                  -- u = D
 * = autograd of policy_average_u_momentum (arcflow.py:81-110).  accumulate != 0 adds into the outputs.
 * d_means [B,N,K,ch], d_logw [B,N,K,pp], d_logg [B,N,K-1,pp] fp32.  ch <= 64, pp power of two. */
int afx_arcflow_backward(const float* g, const void* means, const void* logw, const void* logg, int32_t mix_dtype,
                         float sigma_src, float sigma_start, float sigma_end, const float* sigma_vec,
                         const float* gscale_vec, float gscale, float eps, float* d_means, float* d_logw,
                         float* d_logg, int32_t batch, int32_t n_tok, int32_t K, int32_t ch, int32_t pp,
                         int32_t velocity_only, int32_t accumulate, void* stream);

/* *loss_accum += coef * 0.5 * sum (pred-target)^2 ; grad (optional) = coef * (pred-target)
 * (DiffusionMSELoss, losses/diffusion_loss.py:44-83; coef folds the x30 rescale, the mean and the segment size) */
int afx_mse_loss(const float* pred, const float* target, float coef, float* grad, float* loss_accum, int64_t n,
                 void* stream);
/* x_out = x_a + u * (sigma_b[b] - sigma_a[b])   teacher Euler roll (arcflow.py:189-192) */
int afx_euler_roll(const float* x_a, const float* u, const float* sigma_a, const float* sigma_b, float* out,
                   int32_t batch, int64_t per_sample, void* stream);
/* out = alpha[b] * a + beta[b] * b, per-sample scalars: mean velocity (x_a - x_e) / (sigma_a - sigma_e) and the
 * short/long roll-out select of policy_average_u_momentum (arcflow.py:92-110) */
int afx_axpby_rows(const float* a, const float* alpha, const float* b, const float* beta, float* out, int32_t batch,
                   int64_t per_sample, void* stream);
/* out = pos + (pos - neg) * (scale - 1)   teacher CFG (gaussian_flow.py:18-26, orthogonal=False) */
int afx_cfg_combine(const float* pos, const float* neg, float scale, float* out, int64_t n, void* stream);

/* Head-logit gradient rows dY = [d_means | log_softmax^T(d_logw) | d_logg | 0] as bf16 (arcflux.py:243-249 backward) */
int afx_head_grad(const float* d_means, const float* d_logw, const float* d_logg, const void* logw_out, void* dy,
                  int64_t ldy, int64_t rows, int32_t K, int32_t ch, int32_t lw, void* stream);
/* C(float)[M,N] (+)= A[M,K] . W[N,K]^T, bf16 operands, fp32 result: weight gradients dW = dY^T . X on transposed copies */
int afx_linear_bf16_f32out(const void* A, int64_t lda, const void* W, int64_t ldw, float* C, int64_t ldc, int32_t M,
                           int32_t N, int32_t K, int32_t accumulate, void* stream);
/* C(float)[N1,N2] (+)= X^T . Y with X [M,N1], Y [M,N2] TOKEN-major bf16 (row strides ldx, ldy): the contraction runs over the rows.  The LoRA weight
 * gradients dB += dy^T t, dA += dT^T dropout(x) of peft's adapted linears (reference arcflux.py:294-302) straight from the token-major activations --
 * no transposed copies (ds_read_b64_tr_b16 gathers the MFMA fragments out of LDS).  N1, N2 multiples of 8; deterministic (no atomics). */
int afx_linear_tn_f32out(const void* X, int64_t ldx, const void* Y, int64_t ldy, float* C, int64_t ldc, int32_t M, int32_t N1, int32_t N2,
                         int32_t accumulate, void* stream);
/* The same product with the TOKEN loop split over work-groups (round 6): a [3072, 256] gradient is 48 tiles for 512 work-group slots, so each tile's 4608
 * tokens are cut into afx_linear_tn_ws_bytes() / (4 N1 N2) runs whose fp32 partial tiles go to `ws` and are then added IN ORDER into C by a second launch --
 * deterministic (the summation tree depends on the shape only), no atomics.  ws: afx_linear_tn_ws_bytes(M, N1, N2) bytes, 16-byte aligned, private to the
 * call until it has run (0 bytes = this shape runs unsplit; ws may then be NULL).  Same reference as above (peft lora_A / lora_B gradients, arcflux.py:294-302). */
int64_t afx_linear_tn_ws_bytes(int32_t M, int32_t N1, int32_t N2);
int afx_linear_tn_f32out_ws(const void* X, int64_t ldx, const void* Y, int64_t ldy, float* C, int64_t ldc, int32_t M, int32_t N1, int32_t N2,
                            int32_t accumulate, void* ws, void* stream);
/* C = res + (A . W^T) . keep / (1 - p): the input gradient of peft's LoRA branch with lora_dropout, dx = dx0 + ((dy B) A) . mask, in ONE launch -- the mask
 * (the counter hash of afx_lora_dropout_bf16: seed, row0 + row, column) is applied to the fp32 product in the GEMM's epilogue; res may alias C.  bf16. */
int afx_linear_bf16_dropres(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                            const void* res, int64_t ldr, float p, uint32_t seed, int64_t row0, void* stream);
int afx_transpose_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols, void* stream);
int afx_colsum_bf16(const void* x, int64_t ldx, float* out_accum, int32_t rows, int32_t cols, void* stream);
/* AdaLayerNormContinuous backward w.r.t. (scale, shift): dmod_accum[B,2,D] += sum_rows (dxn * LN(x) | dxn).  Keeps a per-(device, stream)
 * scratch inside the library (grown with hipFree / hipMalloc when a larger shape shows up: not capturable into a graph on that call);
 * long row ranges are folded in 8 splits that meet in float atomics, so the result is reproducible to the rounding of that sum's order. */
int afx_normout_backward(const void* x, int64_t ldx, const void* dxn, int64_t ldd, float* dmod_accum, int32_t rows,
                         int32_t D, int32_t rows_per_batch, void* stream);
/* The same sums for ONE batch entry into two separate fp32 [D] accumulators (d_scale += sum_rows dxn * LN(x), d_shift += sum_rows dxn): the training trunk adds
 * every block's AdaLN modulation gradients straight into its per-sample [n_mod] vector -- no scratch pair, no add passes. */
int afx_normout_backward_split(const void* x, int64_t ldx, const void* dxn, int64_t ldd, float* d_scale_accum, float* d_shift_accum, int32_t rows,
                               int32_t D, void* stream);
/* Modulation gradients of the blocks (needed by the timestep-embedder LoRA pair, configs/flux/arcflux_2nfe_k16.py:46-47):
 * out_accum[c] += sum_r a[r,c] b[r,c] (d_gate = sum_tokens dX_out * branch output);  out = res + gate[c] * y (gated residual kept
 * apart from the GEMM so that y can be stored);  out_accum[b,k] += sum_n x[b,n] W[n,k] (d silu(temb) through the stacked
 * modulation matrix W [n_mod, D], read once; B <= 4) */
int afx_coldot_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, float* out_accum, int32_t rows, int32_t cols, void* stream);
int afx_gate_residual_bf16(const void* y, int64_t ldy, const float* gate, const void* res, int64_t ldr, void* out, int64_t ldo,
                           int64_t rows, int32_t cols, void* stream);
int afx_gemv_t_bf16(const float* x, int64_t ldx, const void* W, int64_t ldw, float* out_accum, int32_t B, int64_t N, int32_t K,
                    void* stream);
/* dW_accum[J,Kd] += sum_b dmod[b,J] x[b,Kd]   (norm_out.linear weight gradient, B <= 8) */
int afx_outer_accum(const float* dmod, const float* x, float* dW_accum, int32_t B, int32_t J, int32_t Kd, void* stream);
/* Staged forward for the training student with LoRA input dropout (lakonlab .../arcflux.py:294-302 lora_dropout): stage 1 runs
 * the conditioning + embedders only, the caller runs the blocks itself on the token matrix (afx_mmdit_export "x_tokens",
 * [B*(T+N), D] bf16, text rows first per sample), puts the result back with afx_mmdit_import_tokens and calls stage 2
 * (norm_out + velocity head).  stage 0 = afx_mmdit_forward. */
int afx_mmdit_forward_stage(afx_ctx* ctx, const void* x, const void* ctx_emb, const void* pooled, const float* t, const float* g,
                            const float* rope_cos, const float* rope_sin, int32_t B, int32_t N, int32_t T, void* means, void* logw,
                            void* logg, int32_t stage, void* stream);
int afx_mmdit_import_tokens(afx_ctx* ctx, const void* src, int32_t batch, int32_t n_img, int32_t n_txt, void* stream);
/* fp8 linear mode (BASELINE.json configs[4] "fp8 MFMA fwd"): every block linear (qkv / out / mlp of the double blocks, fused
 * projection and proj_out of the single blocks) runs on the fp8 MFMA with row-wise scales: activations are quantised per
 * token right before the GEMM, weights come pre-quantised as "<linear>.weight_q" (AFX_DT_FP8 [out, in]) + "<linear>.wscale"
 * (f32 [out]) next to the bf16 ones.  Embedders, modulation, head and attention stay bf16.  Set before afx_workspace_bytes. */
int afx_set_fp8_linear(afx_ctx* ctx, int32_t on);
/* Timestep-embedding override: when set (device pointer to [B, D] f32, NULL to clear) the next forwards use it in place of
 * timestep_embedder(sincos(1000 t)); the guidance / pooled-text embeddings are still added.  The training student computes it
 * host-side with the LoRA pair on the two tiny linears (and their input dropout). */
int afx_set_temb_override(afx_ctx* ctx, const float* temb_t);

/* Copy an activation of the LAST afx_mmdit_forward out of the workspace: "head_in" [B*N,D] bf16,
 * "x_final" [B*N,D] bf16, "silu_temb" [B,D] f32, "mod_final" [B,2D] f32 (scale|shift of norm_out),
 * "mod_all" [B, n_mod] f32 (every modulation vector: per double block img 6D | txt 6D, per single block 3D, final 2D). */
int afx_mmdit_export(afx_ctx* ctx, const char* what, void* dst, int32_t batch, int32_t n_img, int32_t n_txt,
                     void* stream);

/* Element-wise backward kernels of the trunk (gradient checkpointed block recompute + backward):
 * dx = dres + LN^T(dxn (1+scale));  out-of-place RMSNorm+RoPE forward (backward = 0) / backward (dy -> dx, written
 * to y);  GELU(tanh) forward (dh == NULL) / backward;  out = (a (+ b)) * gate[batch] residual-gradient plumbing. */
int afx_ln_modulate_backward(const void* x, int64_t ldx, const void* dxn, int64_t ldd, const float* scale, int64_t ldmod,
                             int32_t rows_per_batch, const void* dres, int64_t ldr, void* dx, int64_t ldo, int32_t rows,
                             int32_t D, void* stream);
int afx_qk_norm_rope_oop_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, const void* dy, int64_t lddy,
                              const float* w_txt, const float* w_img, const float* rope_cos, const float* rope_sin,
                              int32_t batch, int32_t S, int32_t n_txt, int32_t heads, int32_t backward, void* stream);
int afx_gelu_bf16(const void* pre, int64_t ldp, const void* dh, int64_t ldh, void* out, int64_t ldo, int64_t rows,
                  int32_t cols, void* stream);
int afx_add_scale_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, const float* gate, int64_t ldg,
                       int32_t rows_per_batch, void* out, int64_t ldo, int64_t rows, int32_t cols, void* stream);

/* Optimiser step of lakonlab/models/base.py:76-103 + ema_hook.py:86-124 on flat fp32 buffers:
 * global grad-norm (afx_sumsq accumulates sum of squares), AdamW with decoupled decay (grad_scale folds
 * 1/world and the clip factor; step >= 1 for bias correction), Karras EMA  ema = net + (ema - net) * beta. */
int afx_sumsq(const float* x, float* out_accum, int64_t n, void* stream);
int afx_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int32_t step, float grad_scale, int64_t n, void* stream);
/* The same update with block-wise 8-bit moments (the reference's optimizer is bitsandbytes AdamW8bit: _ddp_train.py:18-26,
 * optimizer/builder.py:11-24): state1 / state2 hold one code byte per value, absmax1 / absmax2 one f32 per block of 256 values
 * (ceil(n / 256) entries), qmap1 (signed) / qmap2 (unsigned) the 256-entry sorted code books (arcflow_amd/ops.py dynamic_map).
 * Zero-initialised absmax + any codes = zero moments.  The parameter is updated from the new moments BEFORE they are re-quantised. */
int afx_adamw8bit_step(float* param, const float* grad, void* state1, void* state2, float* absmax1, float* absmax2,
                       const float* qmap1, const float* qmap2, float lr, float beta1, float beta2, float eps, float weight_decay,
                       int32_t step, float grad_scale, int64_t n, void* stream);
int afx_ema_lerp(float* ema, const float* net, float beta, int64_t n, void* stream);
int afx_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream);

/* ---- VAE decoder (AutoencoderKL, the step after the loop: arcflux_pipeline.py:531-534) ---------------------------
 * Activations are NHWC bf16 on ZERO-BORDERED grids [(H+2)*(W+2), C]; a 3x3 convolution is an implicit GEMM on the MFMA
 * kernel (K-tile = (tap, 64-channel chunk) = the same pixel rows shifted by dy*(W+2)+dx, no im2col), the epilogue
 * re-zeroes the border.  x must be readable (W+3) rows before and after the grid.  w: [Cout][3][3][Cin] bf16. */
int afx_conv3x3_bf16(const void* x, const void* w, const void* bias, void* y, int32_t H, int32_t W, int32_t Cin,
                     int32_t Cout, const void* res, void* stream);
#define AFX_GN_SLOTS 64
/* Bytes of the stats_ws scratch afx_groupnorm_nhwc / afx_groupnorm_nhwc_from_stats need for (C, groups): (2 + 2*AFX_GN_SLOTS)*groups + C
 * doubles.  (The requirement GREW in round 3 -- it was 2*groups doubles + 2*C floats -- when the per-slot partial sums moved into it:
 * size the buffer with this call, not with a constant.)  Negative status on bad arguments. */
int64_t afx_groupnorm_ws_bytes(int32_t C, int32_t groups);
/* y = act(GroupNorm(x)) on the interior, 0 on the border; stats_ws: afx_groupnorm_ws_bytes(C, groups) of scratch; act 1 = SiLU;
 * C/8 must divide 256 (C = 64, 128, 256, 512, 1024, 2048) */
int afx_groupnorm_nhwc(const void* x, void* y, double* stats_ws, int32_t H, int32_t W, int32_t C, int32_t groups,
                       const float* gamma, const float* beta, float eps, int32_t act, void* stream);
/* The same convolution, and the GroupNorm sums of ITS OUTPUT accumulated by the GEMM epilogue from the fp32 accumulators BEFORE the bf16
 * rounding of the stored values (mean / variance differ from sums over the stored grid by that rounding: ~2^-9 relative, bounded in
 * tests/test_vae.py): gn_stats = AFX_GN_SLOTS x groups x 2 doubles (zeroed by the call; partial sums spread over the slots by tile), Cout <= 128 (the 256x128-tile kernel: the full-resolution stage, whose
 * grids are the largest), Cout / groups = 4, 8 or a multiple of 8.  afx_groupnorm_nhwc_from_stats
 * then normalises y without a statistics pass of its own (diffusers' ResnetBlock2D order norm -> act -> conv: every GroupNorm input of the
 * decoder except the attention output is a convolution output).  afx_conv_stats_available(): 0 when the GEMM kernel override in force has no
 * such epilogue (AFX_GEMM_IMPL=1). */
int afx_conv3x3_bf16_stats(const void* x, const void* w, const void* bias, void* y, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                           const void* res, double* gn_stats, int32_t groups, void* stream);
int afx_groupnorm_nhwc_from_stats(const void* x, void* y, const double* gn_stats, double* stats_ws, int32_t H, int32_t W, int32_t C,
                                  int32_t groups, const float* gamma, const float* beta, float eps, int32_t act, void* stream);
int afx_conv_stats_available(void);
/* y = conv3x3(nearest-2x upsample(x)) + bias with the upsample FOLDED into the convolution (diffusers Upsample2D: F.interpolate(nearest) then conv,
 * reached through vae.decode, arcflux_pipeline.py:531-534): x = the low-resolution zero-bordered grid [(H+2)*(W+2), Cin], y = [(2H+2)*(2W+2), Cout],
 * w4 = four 2x2 phase kernels [2 py + px][Cout][2][2][Cin] bf16 built from the 3x3 weight (arcflow_amd.vae.phase_weights): 44 % of the flops of the
 * convolution on the upsampled grid, which is never written.  Needs afx_conv_stats_available() (the one-wave-per-SIMD GEMM). */
int afx_upconv3x3_bf16(const void* x, const void* w4, const void* bias, void* y, int32_t H, int32_t W, int32_t Cin, int32_t Cout, void* stream);
int afx_upsample2x_nhwc(const void* x, void* y, int32_t H, int32_t W, int32_t C, void* stream);
/* scatter == 0: compact[H*W, C] = interior(padded);  != 0: interior(padded) = compact (+ interior(res_padded)) */
int afx_interior_nhwc(void* padded, void* compact, const void* res_padded, int32_t H, int32_t W, int32_t C, int32_t scatter,
                      void* stream);
/* P = softmax(scale * S) row-wise, S fp32 [rows, cols], P bf16 (mid-block single-head attention, head dim 512) */
int afx_softmax_rows_f32(const float* s, int64_t lds_, void* p, int64_t ldp, int32_t rows, int32_t cols, float scale,
                         void* stream);
/* packed latent tokens [hp*wp, 64] f32 -> padded NHWC [(2hp+2)*(2wp+2), Cpad] bf16 of lat/scaling + shift; and back to
 * an image [3, H, W] f32 from the first 3 channels of a padded NHWC grid */
int afx_latent_to_nhwc(const float* tokens, void* y, int32_t hp, int32_t wp, int32_t Cpad, float scaling_factor,
                       float shift_factor, void* stream);
int afx_nhwc_to_image(const void* x, float* img, int32_t H, int32_t W, int32_t C, void* stream);
/* AutoencoderKLQwenImage (lakonlab/pipelines/arcqwen_pipeline.py:470-481): the unpack applies v = A . lat + b on the 16
 * latent channels per pixel (A [16][16] row-major = post_quant_conv . diag(latents_std), b = post_quant_conv . mean + bias);
 * the norm is the per-pixel channel RMS norm of that VAE, y = x / max(|x|_2, 1e-12) * sqrt(Creal) * gamma, act 1 = SiLU,
 * Cpad <= 512 */
int afx_latent_to_nhwc_affine(const float* tokens, void* y, int32_t hp, int32_t wp, int32_t Cpad, const float* A, const float* b,
                              void* stream);
int afx_rmsnorm_nhwc(const void* x, void* y, int64_t rows, int32_t Cpad, int32_t Creal, const float* gamma, int32_t act,
                     void* stream);

/* ---- text encoders: T5-XXL + CLIP-L (FLUX), Qwen2.5-VL language model (Qwen-Image) -----------------------
 * Replaces the transformers modules behind lakonlab/models/architecture/diffusers/pretrained.py:152-238
 * (PretrainedFluxTextEncoder / PretrainedQwenImageTextEncoder -> pipeline.encode_prompt). */
/* out[s, :] = table[ids[s], :] (+ pos[s, :]) */
int afx_embed_rows_bf16(const void* table, const int32_t* ids, const void* pos, void* out, int32_t S, int32_t D, void* stream);
/* rms 0: LayerNorm(x) * w + b;  rms 1: x * rsqrt(mean(x^2) + eps) * w (b ignored).  fp32 statistics, D <= 8192 */
int afx_norm_rows_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t D, const float* w, const float* b,
                       float eps, int32_t rms, void* stream);
/* out[m, j] = act(x[m, j]) * (gate_off >= 0 ? x[m, gate_off + j] : 1);  act 0 none, 1 SiLU, 2 GELU(tanh), 3 quick-GELU */
int afx_act_mul_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t M, int32_t F, int32_t gate_off, int32_t act,
                     void* stream);
/* rotate-half RoPE in place on H heads side by side in a row; cos/sin [S, head_dim / 2] f32 */
int afx_rope_half_bf16(void* x, int64_t ldx, const float* cos_t, const float* sin_t, int32_t S, int32_t H, int32_t head_dim,
                       void* stream);
/* fp8 (OCP e4m3) linear on v_mfma_scale_f32_16x16x128_f8f6f4 (BASELINE.json configs[4]): row-wise
 * quantisation q = round(x / scale[r]), scale[r] = absmax(row) / 448, for activations (per token) and weights (per output
 * channel); C = epi(a_scale[m] w_scale[n] (Aq . Wq^T) + bias), bf16 out, same epilogues as afx_linear_bf16.  K % 128 == 0. */
int afx_quant_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int32_t rows, int32_t K, void* stream);
int afx_linear_fp8(const void* Aq, int64_t lda, const float* a_scale, const void* Wq, int64_t ldw, const float* w_scale, const void* bias,
                   void* C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t epi, int32_t gelu_col0, const float* gate, int64_t ldg,
                   int32_t rows_per_batch, const void* res, int64_t ldr, void* stream);
/* Block-scaled activations (the MX layout with 128-value blocks = one K-tile of the one-wave-per-SIMD fp8 kernel): mx[r][j] = E8M0 byte b,
 * scale 2^(b - 127) = the smallest power of two with absmax(x[r, 128 j .. 128 j + 127]) <= 448 * scale; q = round(x / scale), e4m3.
 * afx_linear_fp8_mx consumes it (the matrix instruction applies the block scales); a_scale[m] still multiplies row m (pass ones),
 * weights keep their per-output-channel fp32 scales.  K % 512 == 0, ld_mx % 4 == 0.  No reference counterpart (the reference has no
 * fp8 path): the format exists so that GEMM / attention / LayerNorm epilogues can quantise the 128 columns they hold. */
int afx_quant_rows_mx8(const void* x, int64_t ldx, void* q, int64_t ldq, void* mx, int64_t ld_mx, int32_t rows, int32_t K, void* stream);
int afx_linear_fp8_mx(const void* Aq, int64_t lda, const void* a_mx, int64_t ld_mx, const float* a_scale, const void* Wq, int64_t ldw,
                      const float* w_scale, const void* bias, void* C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t epi,
                      int32_t gelu_col0, const float* gate, int64_t ldg, int32_t rows_per_batch, const void* res, int64_t ldr, void* stream);
/* ... and the producer side: the fp8 GEMM whose epilogue writes the NEXT GEMM's block-scaled operand.  Columns [0, c8_col0) leave as bf16 in C
 * (bias only), columns [c8_col0, N) as e4m3 bytes in c8[m][n - c8_col0] with scale bytes c_mx[m][(n - c8_col0) / 128], after bias and
 * (gelu != 0) tanh-GELU -- the mlp hidden of a double block (c8_col0 = 0), the mlp part of a single block's k|v|q|mlp projection
 * (c8_col0 = 3 D).  a_mx may be NULL (per-row a_scale only). */
int afx_linear_fp8_to_mx8(const void* Aq, int64_t lda, const void* a_mx, int64_t ld_mx, const float* a_scale, const void* Wq, int64_t ldw,
                          const float* w_scale, const void* bias, void* C, int64_t ldc, void* c8, int64_t ldc8, void* c_mx, int64_t ld_cmx,
                          int32_t c8_col0, int32_t M, int32_t N, int32_t K, int32_t gelu, void* stream);
/* Split-K GEMM for few-row operands (M <= 1-2 tiles: the prompt encoders): the K range is cut into chunks, chunk c stores
 * its partial A . W^T (+ bias on chunk 0) into the f32 slab partials[c][M][N]; afx_finish_f32_bf16 sums the slabs into bf16
 * (+ residual).  afx_linear_splitk_chunks = number of slabs for (M, N, K, split_k); split_k 0: chosen to fill the chip. */
int afx_linear_splitk_chunks(int32_t M, int32_t N, int32_t K, int32_t split_k);
int afx_linear_bf16_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, float* partials, int32_t M,
                           int32_t N, int32_t K, int32_t split_k, void* stream);
int afx_finish_f32_bf16(const float* partials, int32_t nslab, const void* res, int64_t ldr, void* out, int64_t ldo, int64_t M, int32_t N,
                        void* stream);
/* softmax(scale * Q K^T + scale * bias [, causal]) V with H query heads over Hkv KV heads (H % Hkv == 0), head_dim 64 or 128;
 * bias: [H][2S-1] f32 indexed by key - query + S - 1, ALREADY DIVIDED by scale, or NULL; ws: afx_attention_ext_ws_bytes */
int64_t afx_attention_ext_ws_bytes(int32_t B, int32_t Hkv, int32_t S, int32_t head_dim);
int afx_attention_ext_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo,
                           void* ws, int32_t B, int32_t H, int32_t Hkv, int32_t S, int32_t head_dim, float scale, int32_t causal,
                           const float* bias, void* stream);

/* ---- building-block kernels (exported for the per-kernel parity tests and micro benches) -- */

/* C[M,N] = epi(A[M,K] . W[N,K]^T + bias)   bf16 in/out, fp32 accumulate (nn.Linear semantics).
 * epi 0: none; 1: GELU(tanh) on columns >= gelu_col0; 2: C = res + gate[m / rows_per_batch, n] * (.)  (gate NULL: C = res + (.))
 * K % 64 == 0, N % 8 == 0, lda/ldw/ldc/ldr % 8 == 0. */
int afx_linear_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias,
                    void* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                    int32_t epi, int32_t gelu_col0, const float* gate, int64_t ldg,
                    int32_t rows_per_batch, const void* res, int64_t ldr, void* stream);
/* same, with pre [M,N] bf16 added to A.W^T + bias BEFORE the activation / gate (the LoRA-dropout correction B A (x . delta)) */
int afx_linear_bf16_pre(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias,
                        void* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                        int32_t epi, int32_t gelu_col0, const float* gate, int64_t ldg,
                        int32_t rows_per_batch, const void* res, int64_t ldr, const void* pre, int64_t ldp, void* stream);
/* afx_linear_bf16 with a caller-lent workspace for the stream-K tail of the 8-phase GEMM (the MMDiT engine lends a region of its
 * own workspace to every block GEMM; this entry point exists for the parity tests and micro benches): sk_ws holds
 * afx_linear_sk_ws_bytes() bytes, 256-byte aligned, whose first 4096 bytes were zeroed ONCE by the caller (hand-off flags; the
 * kernel re-arms them).  When the launch's tile count does not leave an under-filled last round the call is a plain GEMM. */
int64_t afx_linear_sk_ws_bytes(void);
/* CUs per XCD over which the calling thread's last afx_linear_bf16_sk launch split its under-filled last round (0: it ran as a plain
 * GEMM) -- lets a test assert that the stream-K path really executed. */
int afx_linear_sk_last_split(void);
/* Tuning / test knob of every bf16 GEMM in the library (process-wide, not thread-safe against running launches): impl 3 (default) =
 * one-wave-per-SIMD kernel for the bf16 epilogue modes with the tile shape picked per launch (tile 0) or forced (1: 256x256,
 * 2: 288x192, 3: 320x192, 4: 128x128, 5: 256x224); impl 2 = 8-phase 256x256 kernel for everything; impl 1 = simple reference kernel.  Same meaning as the
 * AFX_GEMM_IMPL / AFX_GEMM_TILE environment variables, which it overrides.  Returns 0. */
int afx_gemm_set_mode(int32_t impl, int32_t tile);
/* 1 when afx_linear_bf16_dropres can run under the current kernel choice (its masked residual add lives in the one-wave-per-SIMD kernel's epilogue: kernel
 * mode 3, no stream-K request), 0 otherwise -- the caller then computes the product with afx_linear_bf16 and masks + adds it with afx_lora_dropout_bf16 mode 3
 * (arcflow_amd/ops.py linear_dropres does).  Host-side, no GPU needed. */
int afx_gemm_dropres_available(void);
/* Kernel choice of every joint attention launch (process-wide; same meaning as AFX_ATTN_IMPL, which it overrides): 0 (default) = the
 * one-wave-per-SIMD kernel (afx_attn3.hip: 64 queries per wave, any S > 64: ragged tails handled) where eligible, else the 4-wave kernel; 1 = 4-wave kernel
 * always; 2 = 8-wave ping-pong kernel (experimental); 3 = the one-wave-per-SIMD kernel on its plain grid (0 cuts the 256-query blocks of an under-filled
 * last round into one run of key tiles per CU and merges the partial results: afx_attn3.hip).  For A/B runs and the parity tests.  Returns 0. */
int afx_attn_set_impl(int32_t impl);
int afx_linear_bf16_sk(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias,
                       void* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                       int32_t epi, int32_t gelu_col0, const float* gate, int64_t ldg,
                       int32_t rows_per_batch, const void* res, int64_t ldr, void* sk_ws, void* stream);
/* LoRA input dropout masks from a counter-based hash of (seed, row0 + row, col), keep probability 1 - p, delta = keep/(1-p) - 1:
 * mode 0: dst = src * delta;  1: dst = src * (1 + delta) (= dropout(src));  2: dst += src * delta;  3: dst += src * (1 + delta) */
int afx_lora_dropout_bf16(const void* src, int64_t lds_, void* dst, int64_t ldd, int64_t M, int32_t N, int64_t row0, float p,
                          uint32_t seed, int32_t mode, void* stream);

/* Joint attention over S tokens, no mask: O = softmax(Q K^T / sqrt(128)) V per (batch, head).
 * q,k,v,o: row (b*S + s), head h at column h*128, row strides ld* (elements).  head_dim = 128.
 * vt_ws: scratch of afx_attention_ws_bytes() bytes (transposed V).  o may alias q. */
int64_t afx_attention_ws_bytes(int32_t batch, int32_t heads, int32_t S);
int afx_attention_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                       int64_t ldv, void* o, int64_t ldo, void* vt_ws,
                       int32_t batch, int32_t heads, int32_t S, void* stream);
/* The same attention with the output as the next fp8 GEMM's block-scaled operand (afx_quant_rows_mx8's layout; a head's 128 columns = one block):
 * o8 [batch * S, ldo8] e4m3 bytes, mx [batch * S, ld_mx] one E8M0 byte per token and head.  S > 64.  vt_ws as afx_attention_bf16. */
int afx_attention_to_mx8(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o8, int64_t ldo8,
                         void* mx, int64_t ld_mx, void* vt_ws, int32_t batch, int32_t heads, int32_t S, void* stream);

/* Training twins: forward that also returns lse [B, H, roundup(S,64)] f32 (log2-domain log-sum-exp of the scaled
 * scores; the caller pre-fills it with +inf so padded queries drop out), and the backward producing dq, dk, dv
 * (same row/head addressing as q, k, v) from o, dout and lse.  ws: afx_attention_bwd_ws_bytes() bytes. */
int afx_attention_fwd_lse_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                               int64_t ldo, float* lse, void* vt_ws, int32_t batch, int32_t heads, int32_t S, void* stream);
int64_t afx_attention_bwd_ws_bytes(int32_t batch, int32_t heads, int32_t S);
/* Kernel generation of afx_attention_bwd_bf16 (reference: the FlashAttention-2 backward SDPA runs under arcflux.py:181-189): 3 (default) = the generated
 * one-wave-per-SIMD dK / dV and dQ streams in one launch (csrc/afx_attn_bwd3.hip; S > 64, row strides multiples of 8), 4 = the same as two launches,
 * 1 = generated dK / dV + round-4 dQ, 2 = the round-4 kernels (also what S <= 64 runs).  For A/B runs and the parity tests.  Returns 0. */
int afx_attn_bwd_set_impl(int32_t impl);
int afx_attention_bwd_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                           int64_t ldo, const void* dout, int64_t lddo, const float* lse, void* dq, int64_t lddq, void* dk,
                           int64_t lddk, void* dv, int64_t lddv, void* ws, int32_t batch, int32_t heads, int32_t S,
                           void* stream);

/* out = LayerNorm(x, eps=1e-6, no affine) * (1 + scale[b]) + shift[b]   (AdaLN modulate), or with
 * rms != 0: out = x * rsqrt(mean(x^2) + 1e-6) * w  (scale = w as f32[D], shift ignored). */
int afx_norm_modulate_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int32_t rows,
                           int32_t D, const float* scale, const float* shift, int64_t ldmod,
                           int32_t rows_per_batch, int32_t rms, void* stream);

/* In-place per-head RMSNorm(eps 1e-6, weight) + interleaved-pair RoPE on q or k.
 * x: row (b*S+s), head h at column h*128.  rows s < n_txt use w_txt, others w_img (f32[128]). */
int afx_qk_norm_rope_bf16(void* x, int64_t ldx, const float* w_txt, const float* w_img,
                          const float* rope_cos, const float* rope_sin,
                          int32_t batch, int32_t S, int32_t n_txt, int32_t heads, void* stream);

/* y[b,n] (+)= act(sum_k x[b,k] W[n,k] + bias[n]); x f32 [B,K], W bf16 [N,K], y f32 [B,N];
 * act 0 none, 1 SiLU; accumulate != 0 adds into y. */
int afx_gemv_bf16(const float* x, const void* W, const void* bias, float* y, int32_t B, int32_t N,
                  int32_t K, int32_t act, int32_t accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ARCFLOW_HIP_H_ */
