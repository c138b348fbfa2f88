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
#define AFX_E_INVALID (-1)     /* bad argument / shape */
#define AFX_E_MISSING (-2)     /* a required weight is not bound */
#define AFX_E_WORKSPACE (-3)   /* workspace missing or too small */
#define AFX_E_HIP (-4)         /* a HIP runtime call failed */
#define AFX_E_UNSUPPORTED (-5)

#define AFX_DT_BF16 1
#define AFX_DT_F32 2

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
 * t        [B] f32  sigma in [0,1] (the pipeline's timestep/1000);  g [B] f32 guidance or NULL
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

/* Optional instrumentation for bench.py's roofline line: when enabled, afx_mmdit_forward records a HIP
 * event pair on its stream around every GEMM launch (klass 0) and attention launch (klass 1).
 * afx_profile_read waits for the recorded events and returns the summed duration, the number of
 * launches and their algorithmic FLOPs since the last afx_profile_enable(ctx, 1). */
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

/* ---- building-block kernels (exported for the per-kernel parity tests and micro benches) -- */

/* C[M,N] = epi(A[M,K] . W[N,K]^T + bias)   bf16 in/out, fp32 accumulate (nn.Linear semantics).
 * epi 0: none; 1: GELU(tanh) on columns >= gelu_col0; 2: C = res + gate[m / rows_per_batch, n] * (.)
 * K % 64 == 0, N % 8 == 0, lda/ldw/ldc/ldr % 8 == 0. */
int afx_linear_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias,
                    void* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                    int32_t epi, int32_t gelu_col0, const float* gate, int64_t ldg,
                    int32_t rows_per_batch, const void* res, int64_t ldr, void* stream);

/* Joint attention over S tokens, no mask: O = softmax(Q K^T / sqrt(128)) V per (batch, head).
 * q,k,v,o: row (b*S + s), head h at column h*128, row strides ld* (elements).  head_dim = 128.
 * vt_ws: scratch of afx_attention_ws_bytes() bytes (transposed V).  o may alias q. */
int64_t afx_attention_ws_bytes(int32_t batch, int32_t heads, int32_t S);
int afx_attention_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                       int64_t ldv, void* o, int64_t ldo, void* vt_ws,
                       int32_t batch, int32_t heads, int32_t S, void* stream);

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
