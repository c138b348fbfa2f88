// Internal launcher declarations shared by the C-ABI layer and the MMDiT engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace afx {

// ---- grouped bf16 GEMM  C = epi(A . W^T + bias) -------------------------------------------
enum { EPI_NONE = 0, EPI_GELU = 1, EPI_GATE_RES = 2 };
constexpr int GEMM_MAX_PROBLEMS = 8;

struct GemmProblem {
  const uint16_t* A;    // [M,K], row stride lda
  const uint16_t* W;    // [N,K], row stride ldw (nn.Linear weight)
  const uint16_t* bias; // [N] or nullptr
  uint16_t* C;          // [M,N], row stride ldc
  const float* gate;    // EPI_GATE_RES: [*, ldg] f32, row = m / rows_per_batch
  const uint16_t* res;  // EPI_GATE_RES: [M,N] residual, row stride ldr (may alias C)
  const uint16_t* pre;  // optional [M,N] bf16 added to A.W^T + bias BEFORE the activation / gate (LoRA-dropout correction term)
  int64_t lda, ldw, ldc, ldg, ldr, ldp;
  int32_t M, N, K;
  int32_t epi, gelu_col0, rows_per_batch;
  int32_t out_f32;      // 0: C is bf16; 1: C is float (ldc in floats); 2: C (float) += result; 3: split-K partial slabs (float)
  int32_t split_k;      // > 1 (out_f32 == 3 only): the K range is cut into split_k chunks, one work-group per (tile, chunk);
  int64_t split_stride; //   chunk c stores its partial tile at C + c * split_stride floats (summed by a finishing pass)
  int32_t conv_cin_tiles, conv_wp, conv_hp;   // > 0: implicit 3x3 conv on a padded [conv_hp][conv_wp] NHWC grid, K = 9 * 64 * conv_cin_tiles
  int32_t fp8;          // 1: A and W hold OCP e4m3 bytes (lda/ldw/K in elements = bytes, K % 128 == 0); the product is scaled by
  const float* a_scale; //    a_scale[m] * w_scale[n] (per-row activation scale, per-output-channel weight scale) before the epilogue
  const float* w_scale;
  // fp8 with BLOCK-scaled activations (the one-wave-per-SIMD fp8 kernel only; gemm_fp8_mx_ok()): a_mx[m][t] = the E8M0 byte (2^(b - 127)) of
  // row m's K-tile t = its k-values 128 t .. 128 t + 127 (row stride ld_mx bytes, a multiple of 4; K % 512 == 0).  a_scale still multiplies
  // the row (pass ones); afx_quant_rows_mx8 / the fused producers write this layout.
  const uint8_t* a_mx;
  int64_t ld_mx;
  // ... and the PRODUCER side of that layout (same kernel; epi EPI_NONE / EPI_GELU only): the columns from c8_col0 on (a multiple of 128) are
  // not stored to C as bf16 but to c8[m][n - c8_col0] as e4m3 bytes with one scale byte per row and 128 columns in c_mx[m][(n - c8_col0) / 128]
  // -- the next GEMM's A operand straight out of this one's epilogue (mlp hidden, the mlp part of the single blocks' [O | mlp] operand).
  uint8_t* c8;
  uint8_t* c_mx;
  int64_t ldc8, ld_cmx;
  int32_t c8_col0;
  // Fused q / k preparation of a k|v|q(|mlp) projection (one-wave-per-SIMD kernel, 256x256 tile only; gemm_qk_fusion_available()):
  // columns [0, qk_D) are keys, [2 qk_D, 3 qk_D) queries -- every 128-column head of those two ranges leaves the epilogue as
  // RoPE(RMSNorm_128(x + bias) * w) (diffusers FluxAttnProcessor order: norm_q / norm_k, then apply_rotary_emb on interleaved
  // pairs; reference call sites arcflux.py:63-83), computed on the fp32 accumulators.  Row r of the problem sits at joint
  // sequence position (rope_row0 + r) % rope_period of the [rope_rows, 64] f32 cos / sin tables.  qk_D == 0: off.
  const float* qk_wk;   // [128] RMSNorm weight of the keys
  const float* qk_wq;   // [128] ... of the queries
  const float* rope_cos;
  const float* rope_sin;
  int32_t qk_D, rope_row0, rope_period, rope_rows;
  // V^T straight out of the projection (one-wave-per-SIMD kernel): the product is computed transposed -- A = the V rows of the
  // weight, W = the token matrix -- so C [head_dim * heads, keys] IS the attention kernel's V^T operand; w_perm16 loads the W rows
  // (keys) of every 16-group in the order key_of_pos (afx_attn.hip) wants its columns, bias_rows adds bias[row] instead of bias[col].
  int32_t w_perm16, bias_rows;
  // Convolution launches (one-wave-per-SIMD kernel): GroupNorm statistics of the OUTPUT grid, accumulated by the epilogue so that the
  // normalisation that follows needs no pass of its own over the grid: per group g the sum and the sum of squares of the stored values
  // are added (fp64 atomics) to gn_stats[slot][g][2], slot = a hash of the tile in [0, GN_SLOTS) (spreads the atomics; the reader sums
  // the slots).  gn_gs = channels per group (4, 8 or a multiple of 8); N <= 128 only (the 256x128 tile).  nullptr: off.
  double* gn_stats;
  int32_t gn_gs, gn_groups;
  // Nearest-2x upsample folded into the 3x3 convolution that follows it (one-wave-per-SIMD kernel, afx_upconv3x3_bf16): output pixel
  // (2y + py, 2x + px) of conv3x3(upsample2x(X)) only sees the 2x2 neighbourhood rows {y - 1 + py, y + py} x cols {x - 1 + px, x + px} of X, with
  // the 3x3 taps that fall on the same source pixel summed -- four PHASE convolutions with 2x2 taps (K = 4 * Cin instead of 9 * Cin on 4x the
  // pixels: 44 % of the flops and no upsampled grid in memory).  up_phase = 1 + 2 py + px (0: plain convolution): the A rows are the LOW-resolution
  // padded grid [conv_hp][conv_wp], the taps are 2 x 2 starting at (py - 1, px - 1), and the epilogue scatters row (yy, xx) to pixel
  // (2 yy + py - 1, 2 xx + px - 1) of the [2 conv_hp - 2][2 conv_wp - 2] output grid (border pixels as zeros, positions outside dropped).
  int32_t up_phase;
  // LoRA input-dropout mask on the PRODUCT of a residual-add launch (EPI_GATE_RES without a gate; one-wave-per-SIMD kernel): C = res + (A . W^T) . keep / (1 - p)
  // with keep(row, col) = the counter hash of afx_lora_dropout_bf16 (seed, drop_row0 + row, col) -- the input gradient of the LoRA branch,
  // dx = dx0 + ((dy B) A) . mask, without the [M, in] product in memory and without the separate mask-and-add pass.  drop_on == 0: off.
  int32_t drop_on;
  uint32_t drop_thresh, drop_seed;
  float drop_inv_keep;
  int64_t drop_row0;
  int32_t tiles_m, tiles_n, tile_start;   // filled by the launcher
};

struct GemmBatch {
  int32_t nprob;
  int32_t total_tiles;
  int32_t group_m;     // tile-order super-row height (set by the launcher)
  // stream-K tail (afx_gemm.hip): the caller lends a workspace -- sk_flags: 1024 zero-initialised uint32 (re-armed by the
  // kernel), sk_slab: 256 x 256 KiB fp32 accumulator slabs; the launcher fills the three ints (sk_cus == 0: plain launch)
  float* sk_slab;
  uint32_t* sk_flags;
  int32_t sk_cus, sk_tiles_per_xcd, sk_full;
  int32_t sk_force;    // caller asks for the stream-K tail on every eligible launch (parity tests / micro benches), not only where it pays
  GemmProblem p[GEMM_MAX_PROBLEMS];
};

hipError_t launch_gemm(GemmBatch& batch, hipStream_t stream);
// C (fp32) [N1, N2] (+)= X^T Y with X [M, N1], Y [M, N2] token-major bf16 (afx_tn.hip: LDS transpose reads; the LoRA weight gradients without transposed copies)
hipError_t launch_gemm_tn_f32(const uint16_t* X, int64_t ldx, const uint16_t* Y, int64_t ldy, float* C, int64_t ldc, int M, int N1, int N2, int accumulate,
                              hipStream_t stream, float* ws = nullptr);      // ws: gemm_tn_ws_bytes() bytes (token split, round 6) or nullptr (one work-group per tile)
int gemm_tn_ksplit(int M, int N1, int N2);
int64_t gemm_tn_ws_bytes(int M, int N1, int N2);
bool gemm_conv_stats_available();           // GemmProblem::gn_stats is honoured (kernel mode 3, no forced tile shape)
bool gemm_qk_fusion_available();            // the launcher would take a problem with qk_D > 0 (kernel mode 3, no stream-K request)
bool gemm_dropres_available();              // ... a problem with drop_on (the LoRA branch's masked residual add in the epilogue)
void gemm_set_mode(int impl, int tile);      // kernel / tile-shape override of AFX_GEMM_IMPL / AFX_GEMM_TILE (see launch_gemm)
constexpr int GN_SLOTS = 64;
constexpr int64_t GEMM_SK_FLAG_BYTES = 4096;                       // 1024 flag words
constexpr int64_t GEMM_SK_SLAB_BYTES = 256ll * 256 * 256 * 4;      // 256 work-groups x one fp32 256x256 tile

// Optional per-launch timing without extra queue packets: when both are non-null, the NEXT launch_gemm / launch_attention issues
// its kernel with hipExtLaunchKernelGGL(start, stop), which stamps the kernel's own begin / end on the events (the engine's
// ProfScope sets and clears them; a plain hipEventRecord pair costs a barrier packet each: ~2 % of the forward).
struct LaunchTimer { hipEvent_t start = nullptr, stop = nullptr; };
LaunchTimer& launch_timer();
int& last_sk_cus();

// ---- attention ------------------------------------------------------------------------------
// vt: [B, H, 128, S_pad] transposed + key-permuted V (see afx_attn.hip); S_pad = roundup(S, 64)
hipError_t launch_v_transpose(const uint16_t* v, int64_t ldv, uint16_t* vt, int B, int H, int S,
                              hipStream_t stream);
// mx8 (fp8 engine path): the output leaves as the next fp8 GEMM's block-scaled operand -- e4m3 bytes in o8[token][head * 128 + .] (row stride ldo8) and
// one E8M0 byte per token and head in mx[token][head] -- INSTEAD of bf16 `o`, when the one-wave-per-SIMD kernel takes the launch (*fused = true);
// otherwise bf16 `o` is written as always and *fused = false (the caller then quantises it).
struct AttnMx8 { uint8_t* o8; int64_t ldo8; uint8_t* mx; int64_t ld_mx; };
hipError_t launch_attention(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
                            const uint16_t* vt, uint16_t* o, int64_t ldo, int B, int H, int S,
                            hipStream_t stream, float* lse = nullptr, const AttnMx8* mx8 = nullptr, bool* fused = nullptr);
// one-wave-per-SIMD kernel (afx_attn3.hip): any S > 64 (ragged tails handled); launch_attention dispatches to it (AFX_ATTN_IMPL=1 / attn_set_impl(1): 4-wave kernel)
bool attention_v3_eligible(int S);
hipError_t launch_attention_v3(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* vt, uint16_t* o,
                               int64_t ldo, int B, int H, int S, hipStream_t stream, float* lse, const AttnMx8* mx8 = nullptr, bool split = true);
void attn_set_impl(int impl);               // 0 = default (v3 where eligible, under-filled last round KV-split), 1 = 4-wave kernel, 2 = 8-wave ping-pong (experimental), 3 = v3 on the plain grid
// K, Q <- RoPE(RMSNorm(.) w) in place (same row stride) and V -> V^T (key-permuted), one launch
hipError_t launch_kv_prep(uint16_t* k, uint16_t* q, int64_t ldk, const float* wk_txt, const float* wk_img, const float* wq_txt,
                          const float* wq_img, const float* cos_t, const float* sin_t, int n_txt, const uint16_t* v, int64_t ldv,
                          uint16_t* vt, int B, int H, int S, hipStream_t stream);
// text encoders: runtime scale, causal mask, additive bias table [H][2S-1] (pre-divided by scale), grouped KV heads, d 64|128
hipError_t launch_attention_ext(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv,
                                uint16_t* vt_ws, uint16_t* o, int64_t ldo, int B, int H, int Hkv, int S, int head_dim, float scale,
                                int causal, const float* bias, hipStream_t stream);
// backward: see afx_attn_bwd.hip / afx_attn_bwd3.hip.  ws layout: Kt | Qt | dOt (each B*H*128*S_pad bf16) | delta (B*H*S_pad f32) | stats (2*B*H*S_pad f32)
// (Qt / dOt are used by the round-4 dK / dV kernel only: AFX_ATTN_BWD_IMPL=2 and S <= 64)
int64_t attn_bwd_ws_bytes(int B, int H, int S);
void attn_bwd_set_impl(int impl);           // 3 (default) / 4 / 1 / 2: see afx_attn_bwd.hip (A/B runs, parity tests)
int64_t attn_bwd3_stats_bytes(int B, int H, int S);
// the generated streams take the call (S > 64, row strides multiples of 8 elements, 16-byte aligned bases); otherwise the round-4 kernels do
bool attn_bwd3_eligible(const void* q, const void* k, const void* v, const void* dout, int64_t ldq, int64_t ldk, int64_t ldv, int64_t lddo, int S);
hipError_t launch_attn_bwd_stats(const uint16_t* o, int64_t ldo, const uint16_t* dout, int64_t lddo, const float* lse, float* stats, float* delta_old,
                                 int B, int H, int S, hipStream_t stream);
hipError_t launch_attn_bwd_dkv3(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv, const uint16_t* dout,
                                int64_t lddo, const float* stats, uint16_t* dk, int64_t lddk, uint16_t* dv, int64_t lddv, int B, int H, int S,
                                hipStream_t stream);
hipError_t launch_attn_bwd_dq3(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv, const uint16_t* dout,
                               int64_t lddo, const float* stats, uint16_t* dq, int64_t lddq, int B, int H, int S, hipStream_t stream);
hipError_t launch_attn_bwd_fused3(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv, const uint16_t* dout,
                                  int64_t lddo, const float* stats, uint16_t* dq, int64_t lddq, uint16_t* dk, int64_t lddk, uint16_t* dv, int64_t lddv, int B,
                                  int H, int S, hipStream_t stream);
hipError_t launch_attention_backward(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v,
                                     int64_t ldv, const uint16_t* o, int64_t ldo, const uint16_t* dout, int64_t lddo,
                                     const float* lse, uint16_t* dq, int64_t lddq, uint16_t* dk, int64_t lddk,
                                     uint16_t* dv, int64_t lddv, void* ws, int B, int H, int S, hipStream_t stream);
inline int64_t attn_spad(int S) { return ((int64_t)S + 63) / 64 * 64; }

// row-wise OCP e4m3 quantisation (afx_text.hip): q = round(x / scale[r]), scale[r] = absmax(row) / 448
hipError_t launch_quant_rows_fp8(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int rows, int K, hipStream_t stream);
hipError_t launch_quant_rows_mx8(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* mx, int64_t ld_mx, int rows, int K, hipStream_t stream);
bool gemm_fp8_mx_ok(int64_t rows_total, int N, int K);      // would an fp8 launch of this size take the block-scaled kernel?

// ---- element-wise / reductions -----------------------------------------------------------------
hipError_t launch_norm_modulate(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int rows,
                                int D, const float* scale, const float* shift, int64_t ldmod,
                                int rows_per_batch, int rms, hipStream_t stream);
hipError_t launch_norm_modulate_mx8(const uint16_t* x, int64_t ldx, uint8_t* q8, int64_t ldq, uint8_t* mx, int64_t ld_mx, int rows, int D,
                                    const float* scale, const float* shift, const float* scale_txt, const float* shift_txt, int64_t ldmod, int S,
                                    int n_txt, hipStream_t stream, bool* fused, float* rowscale = nullptr);
hipError_t launch_norm_modulate_joint(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int rows, int D, const float* scale,
                                      const float* shift, const float* scale_txt, const float* shift_txt, int64_t ldmod, int S,
                                      int n_txt, hipStream_t stream);
hipError_t launch_qk_norm_rope(uint16_t* x, int64_t ldx, const float* w_txt, const float* w_img,
                               const float* cos_t, const float* sin_t, int B, int S, int n_txt, int H,
                               hipStream_t stream);
hipError_t launch_qk_norm_rope2(uint16_t* xk, uint16_t* xq, int64_t ldx, const float* wk_txt, const float* wk_img, const float* wq_txt,
                                const float* wq_img, const float* cos_t, const float* sin_t, int B, int S, int n_txt, int H,
                                hipStream_t stream);
hipError_t launch_gemv(const float* x, const uint16_t* W, const uint16_t* bias, float* y, int B, int N,
                       int K, int act, int accumulate, hipStream_t stream, int64_t ldy = 0);
hipError_t launch_sincos(const float* t, float scale, float* out, int B, int cast_mode, hipStream_t stream);
hipError_t launch_silu(const float* x, float* y, int64_t n, hipStream_t stream);
hipError_t launch_bf16_to_f32(const uint16_t* x, float* y, int64_t n, hipStream_t stream);
hipError_t launch_head_split(const uint16_t* head, int64_t ldh, uint16_t* means, uint16_t* logw,
                             uint16_t* logg, int64_t rows, int K, int ch, int lw, hipStream_t stream);
hipError_t launch_copy_rows(const uint16_t* src, int64_t lds_, uint16_t* dst, int64_t ldd, int64_t rows,
                            int cols, hipStream_t stream);

// ---- ArcFlow policy math --------------------------------------------------------------------------
hipError_t launch_arcflow_step(const float* x_in, const void* means, const void* logw, const void* logg,
                               int mix_bf16, float s_src, float s_start, float s_end,
                               const float* sigma_vec, float eps, float* x_out, int B, int n_tok, int K,
                               int ch, int pp, int velocity_only, const uint8_t* drop, hipStream_t stream);
hipError_t launch_arcflow_bwd(const float* g, const void* means, const void* logw, const void* logg, int mix_bf16,
                              float s_src, float s_start, float s_end, const float* sigma_vec,
                              const float* gscale_vec, float gscale, float eps, float* d_means, float* d_logw,
                              float* d_logg, int B, int n_tok, int K, int ch, int pp, int velocity_only,
                              int accumulate, hipStream_t stream);

}  // namespace afx
