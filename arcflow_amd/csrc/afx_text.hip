// Text-encoder glue kernels (the prompt-embedding step in front of the denoiser -- T5-XXL + CLIP-L for FLUX,
// Qwen2.5-VL's language model for Qwen-Image; reference call sites lakonlab/models/architecture/diffusers/pretrained.py:152-238,
// the encoders themselves are transformers models).  GEMMs run on the grouped MFMA kernel (afx_gemm.hip), attention on the
// EXT instantiations of the flash kernel (afx_attn.hip); what is here is HBM-bound: embedding gather, LayerNorm / RMSNorm
// rows, activation (* gate) for the plain and gated MLPs, and the rotate-half RoPE of the Qwen2.5 language model.
#include "afx_api_util.h"
#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {

// out[s, :] = table[ids[s], :] (+ pos[s, :])
__global__ __launch_bounds__(256) void embed_rows_kernel(const bf16_t* __restrict__ table, const int32_t* __restrict__ ids,
                                                         const bf16_t* __restrict__ pos, bf16_t* __restrict__ out, int S, int D) {
  const int cpr = D >> 3;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (int64_t)S * cpr) return;
  const int s = (int)(g / cpr), c = (int)(g % cpr);
  u32x4_t w = *reinterpret_cast<const u32x4_t*>(table + (int64_t)ids[s] * D + c * 8);
  if (pos != nullptr) {
    float a[8], b[8];
    unpack8(w, a);
    unpack8(*reinterpret_cast<const u32x4_t*>(pos + (int64_t)s * D + c * 8), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += b[e];
    w = pack8(a);
  }
  *reinterpret_cast<u32x4_t*>(out + (int64_t)s * D + c * 8) = w;
}

// One wave per row.  rms = 0: LayerNorm (x - mean) * rstd * w + b;  rms = 1: x * rsqrt(mean(x^2) + eps) * w  (T5LayerNorm,
// Qwen2RMSNorm: statistics in fp32).  D <= 8192.
__global__ __launch_bounds__(256) void norm_rows_kernel(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ y, int64_t ldy,
                                                        int rows, int D, const float* __restrict__ w, const float* __restrict__ b,
                                                        float eps, int rms) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (int64_t)row * ldx;
  const int nch = D >> 3;
  float v[16][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ch = i * 64 + lane;
    if (ch < nch) {
      unpack8(*reinterpret_cast<const u32x4_t*>(xr + ch * 8), v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1 += v[i][e]; s2 += v[i][e] * v[i][e]; }
    }
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const float mean = rms ? 0.f : s1 / D;
  const float var = rms ? s2 / D : s2 / D - mean * mean;
  const float rstd = rsqrtf(var + eps);
  bf16_t* yr = y + (int64_t)row * ldy;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ch = i * 64 + lane;
    if (ch < nch) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int cc = ch * 8 + e;
        o[e] = (v[i][e] - mean) * rstd * w[cc] + (b != nullptr ? b[cc] : 0.f);
      }
      *reinterpret_cast<u32x4_t*>(yr + ch * 8) = pack8(o);
    }
  }
}

AFX_DEV float act_fn(float x, int act) {
  if (act == 1) return silu(x);
  if (act == 2) return gelu_tanh(x);
  if (act == 3) return x / (1.0f + __expf(-1.702f * x));          // quick_gelu (CLIP)
  return x;
}

// out[m, j] = act(x[m, j]) * (gate_off >= 0 ? x[m, gate_off + j] : 1)     -- plain and gated MLP activations
__global__ __launch_bounds__(256) void act_mul_kernel(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ out, int64_t ldo,
                                                      int64_t M, int F, int gate_off, int act) {
  const int cpr = F >> 3;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= M * cpr) return;
  const int64_t m = g / cpr;
  const int c = (int)(g % cpr);
  float a[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(x + m * ldx + c * 8), a);
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = act_fn(a[e], act);
  if (gate_off >= 0) {
    float b[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(x + m * ldx + gate_off + c * 8), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] *= b[e];
  }
  *reinterpret_cast<u32x4_t*>(out + m * ldo + c * 8) = pack8(a);
}

// rotate-half RoPE in place on H heads of dimension d stored side by side in a row: for i < d/2
//   x[i] <- x[i] cos[s,i] - x[i + d/2] sin[s,i];   x[i + d/2] <- x[i + d/2] cos[s,i] + x[i] sin[s,i]
__global__ __launch_bounds__(256) void rope_half_kernel(bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ cs,
                                                        const float* __restrict__ sn, int S, int H, int d) {
  const int half = d >> 1, cph = half >> 3;                         // 16-byte chunks per half head
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (int64_t)S * H * cph) return;
  const int c = (int)(g % cph);
  const int h = (int)((g / cph) % H);
  const int s = (int)(g / ((int64_t)cph * H));
  bf16_t* p = x + (int64_t)s * ldx + h * d + c * 8;
  float a[8], b[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(p), a);
  unpack8(*reinterpret_cast<const u32x4_t*>(p + half), b);
  float oa[8], ob[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float co = cs[(int64_t)s * half + c * 8 + e], si = sn[(int64_t)s * half + c * 8 + e];
    oa[e] = a[e] * co - b[e] * si;
    ob[e] = b[e] * co + a[e] * si;
  }
  *reinterpret_cast<u32x4_t*>(p) = pack8(oa);
  *reinterpret_cast<u32x4_t*>(p + half) = pack8(ob);
}

// Row-wise fp8 quantisation (OCP e4m3): scale[r] = absmax(row) / 448, q = round(x / scale[r]).  One wave per row, K <= 16384.
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q, int64_t ldq,
                                                             float* __restrict__ scale, int rows, int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (int64_t)row * ldx;
  const int nch = K >> 3;
  float amax = 0.f;
  for (int ch = lane; ch < nch; ch += 64) {
    float v[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(xr + ch * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const float sc = fmaxf(amax, 1e-12f) / 448.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  uint8_t* qr = q + (int64_t)row * ldq;
  for (int ch = lane; ch < nch; ch += 64) {
    float v[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(xr + ch * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e] * inv, 448.0f, -448.0f);
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w1, true);
    *reinterpret_cast<u32x2_t*>(qr + ch * 8) = (u32x2_t){(uint32_t)w0, (uint32_t)w1};
  }
}

// out (bf16) = sum of the split-K partial slabs (f32, [nslab][M][N]) [+ res (bf16)]
__global__ __launch_bounds__(256) void finish_f32_kernel(const float* __restrict__ src, int nslab, const bf16_t* __restrict__ res,
                                                         int64_t ldr, bf16_t* __restrict__ out, int64_t ldo, int64_t M, int N) {
  const int cpr = N >> 3;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= M * cpr) return;
  const int64_t m = g / cpr;
  const int c = (int)(g % cpr);
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < nslab; ++s) {
    const float* p = src + ((int64_t)s * M + m) * N + c * 8;
    const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(p);
    const f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(p + 4);
    v[0] += a0[0]; v[1] += a0[1]; v[2] += a0[2]; v[3] += a0[3]; v[4] += a1[0]; v[5] += a1[1]; v[6] += a1[2]; v[7] += a1[3];
  }
  if (res != nullptr) {
    float r[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(res + m * ldr + c * 8), r);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += r[e];
  }
  *reinterpret_cast<u32x4_t*>(out + m * ldo + c * 8) = pack8(v);
}

// Block-scaled fp8 (OCP e4m3 values, one E8M0 scale byte per row and 128 columns -- the MX layout with the block = one K-tile of the fp8
// GEMM): byte b = 127 + ceil(log2(absmax(block) / 448)), q = round(x * 2^(127 - b)).  A block never sees another block's outlier, and a
// producer can emit the format from the 128 columns it holds (afx_common.h mx_exp / mx_inv: the same arithmetic in the fused epilogues).
// Thread = 8 columns, 16 consecutive lanes = one block.
__global__ __launch_bounds__(256) void quant_rows_mx8_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q, int64_t ldq,
                                                             uint8_t* __restrict__ mx, int64_t ld_mx, int64_t rows, int K) {
  const int cpr = K >> 3;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = g < rows * cpr;                 // (whole 16-lane groups: cpr % 16 == 0)
  const int64_t row = live ? g / cpr : 0;
  const int c = live ? (int)(g - row * cpr) : 0;
  float v[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(x + row * ldx + c * 8), v);
  float amax = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const int eb = mx_exp(amax);
  const float inv = mx_inv(eb);
  if (!live) return;
  uint32_t w0, w1;
  mx_pack8(v, inv, w0, w1);
  *reinterpret_cast<u32x2_t*>(q + row * ldq + c * 8) = (u32x2_t){w0, w1};
  if ((c & 15) == 0) mx[row * ld_mx + (c >> 4)] = (uint8_t)eb;
}

hipError_t launch_quant_rows_mx8(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* mx, int64_t ld_mx, int rows, int K, hipStream_t stream) {
  const int64_t n = (int64_t)rows * (K >> 3);
  hipLaunchKernelGGL(quant_rows_mx8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, ldx, q, ldq, mx, ld_mx, (int64_t)rows, K);
  return hipGetLastError();
}

// The same quantiser for K = 512 NCH with the row held in registers (raw bf16: 4 NCH VGPRs): ONE read of the row instead of two and no second trip through
// memory between the maximum and the conversion -- 27 -> ~10 us for 4608 x 3072 (the training trunk's per-linear passes, AFX_FP8_MX=0).  Bit-identical.
template <int NCH>
__global__ __launch_bounds__(256) void quant_rows_fp8_reg_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q, int64_t ldq,
                                                                 float* __restrict__ scale, int rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (int64_t)row * ldx;
  u32x4_t raw[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) raw[i] = *reinterpret_cast<const u32x4_t*>(xr + (lane + i * 64) * 8);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    float v[8];
    unpack8(raw[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const float sc = fmaxf(amax, 1e-12f) / 448.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  uint8_t* qr = q + (int64_t)row * ldq;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    float v[8];
    unpack8(raw[i], v);
    uint32_t w0, w1;
    mx_pack8(v, inv, w0, w1);
    *reinterpret_cast<u32x2_t*>(qr + (lane + i * 64) * 8) = (u32x2_t){w0, w1};
  }
}

hipError_t launch_quant_rows_fp8(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int rows, int K, hipStream_t stream) {
  const dim3 grid((rows + 3) / 4);
  if (K == 3072) hipLaunchKernelGGL((quant_rows_fp8_reg_kernel<6>), grid, dim3(256), 0, stream, x, ldx, q, ldq, scale, rows);
  else if (K == 12288) hipLaunchKernelGGL((quant_rows_fp8_reg_kernel<24>), grid, dim3(256), 0, stream, x, ldx, q, ldq, scale, rows);
  else if (K == 15360) hipLaunchKernelGGL((quant_rows_fp8_reg_kernel<30>), grid, dim3(256), 0, stream, x, ldx, q, ldq, scale, rows);
  else hipLaunchKernelGGL(quant_rows_fp8_kernel, grid, dim3(256), 0, stream, x, ldx, q, ldq, scale, rows, K);
  return hipGetLastError();
}

}  // namespace afx

using namespace afx;

static inline unsigned tblocks(int64_t n) { return (unsigned)((n + 255) / 256); }

extern "C" {

int afx_embed_rows_bf16(const void* table, const int32_t* ids, const void* pos, void* out, int32_t S, int32_t D, void* stream) {
  if (!table || !ids || !out || S < 1 || D % 8) return fail(AFX_E_INVALID, "bad argument to afx_embed_rows_bf16");
  hipLaunchKernelGGL(embed_rows_kernel, dim3(tblocks((int64_t)S * (D >> 3))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)table,
                     ids, (const bf16_t*)pos, (bf16_t*)out, S, D);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_norm_rows_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t D, const float* w, const float* b,
                       float eps, int32_t rms, void* stream) {
  if (!x || !y || !w || rows < 1 || D % 8 || D > 8192 || ldx % 8 || ldy % 8)
    return fail(AFX_E_INVALID, "afx_norm_rows_bf16: need D %% 8 == 0, D <= 8192, strides %% 8 == 0");
  hipLaunchKernelGGL(norm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy,
                     rows, D, w, b, eps, rms);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_act_mul_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t M, int32_t F, int32_t gate_off, int32_t act,
                     void* stream) {
  if (!x || !out || M < 1 || F % 8 || ldx % 8 || ldo % 8 || (gate_off >= 0 && gate_off % 8) || act < 0 || act > 3)
    return fail(AFX_E_INVALID, "bad argument to afx_act_mul_bf16");
  hipLaunchKernelGGL(act_mul_kernel, dim3(tblocks(M * (F >> 3))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                     (bf16_t*)out, ldo, M, F, gate_off, act);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_rope_half_bf16(void* x, int64_t ldx, const float* cos_t, const float* sin_t, int32_t S, int32_t H, int32_t head_dim,
                       void* stream) {
  if (!x || !cos_t || !sin_t || S < 1 || H < 1 || head_dim % 16 || ldx % 8) return fail(AFX_E_INVALID, "bad argument to afx_rope_half_bf16");
  hipLaunchKernelGGL(rope_half_kernel, dim3(tblocks((int64_t)S * H * (head_dim >> 4))), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x,
                     ldx, cos_t, sin_t, S, H, head_dim);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_quant_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int32_t rows, int32_t K, void* stream) {
  if (!x || !q || !scale || rows < 1 || K % 8 || ldx % 8 || ldq % 8) return fail(AFX_E_INVALID, "bad argument to afx_quant_rows_fp8");
  HIP_TRY(launch_quant_rows_fp8((const uint16_t*)x, ldx, (uint8_t*)q, ldq, scale, rows, K, (hipStream_t)stream));
  return AFX_OK;
}

int afx_linear_fp8(const void* Aq, int64_t lda, const float* a_scale, const void* Wq, int64_t ldw, const float* w_scale, const void* bias,
                   void* C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t epi, int32_t gelu_col0, const float* gate, int64_t ldg,
                   int32_t rows_per_batch, const void* res, int64_t ldr, void* stream) {
  if (!Aq || !Wq || !C || !a_scale || !w_scale) return fail(AFX_E_INVALID, "null argument to afx_linear_fp8");
  if (M < 0 || N < 0 || K <= 0 || K % 128 || N % 8 || lda % 16 || ldw % 16 || ldc % 8 || epi < 0 || epi > 2)
    return fail(AFX_E_INVALID, "afx_linear_fp8: need K%%128==0, N%%8==0, lda/ldw%%16==0, ldc%%8==0");
  if (epi == EPI_GATE_RES && (!res || ldr % 8 || (gate && rows_per_batch < 1))) return fail(AFX_E_INVALID, "gated residual epilogue needs res");
  GemmBatch gb{};
  gb.nprob = 1;
  GemmProblem& p = gb.p[0];
  p = GemmProblem{};
  p.A = (const uint16_t*)Aq; p.lda = lda; p.W = (const uint16_t*)Wq; p.ldw = ldw; p.bias = (const uint16_t*)bias;
  p.C = (uint16_t*)C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.epi = epi; p.gelu_col0 = gelu_col0;
  p.gate = gate; p.ldg = ldg; p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; p.res = (const uint16_t*)res; p.ldr = ldr;
  p.fp8 = 1; p.a_scale = a_scale; p.w_scale = w_scale;
  HIP_TRY(launch_gemm(gb, (hipStream_t)stream));
  return AFX_OK;
}

int afx_quant_rows_mx8(const void* x, int64_t ldx, void* q, int64_t ldq, void* mx, int64_t ld_mx, int32_t rows, int32_t K, void* stream) {
  if (!x || !q || !mx || rows < 1 || K < 128 || K % 128 || ldx % 8 || ldq % 8 || ld_mx < K / 128)
    return fail(AFX_E_INVALID, "afx_quant_rows_mx8: need K %% 128 == 0, ldx / ldq %% 8 == 0, ld_mx >= K / 128");
  HIP_TRY(launch_quant_rows_mx8((const uint16_t*)x, ldx, (uint8_t*)q, ldq, (uint8_t*)mx, ld_mx, rows, K, (hipStream_t)stream));
  return AFX_OK;
}

int afx_linear_fp8_mx(const void* Aq, int64_t lda, const void* a_mx, int64_t ld_mx, const float* a_scale, const void* Wq, int64_t ldw,
                      const float* w_scale, const void* bias, void* C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t epi,
                      int32_t gelu_col0, const float* gate, int64_t ldg, int32_t rows_per_batch, const void* res, int64_t ldr, void* stream) {
  if (!Aq || !a_mx || !Wq || !C || !a_scale || !w_scale) return fail(AFX_E_INVALID, "null argument to afx_linear_fp8_mx");
  if (M < 0 || N < 0 || K < 512 || K % 512 || N % 8 || lda % 16 || ldw % 16 || ldc % 8 || ld_mx % 4 || ld_mx < K / 128 || epi < 0 || epi > 2)
    return fail(AFX_E_INVALID, "afx_linear_fp8_mx: need K%%512==0, N%%8==0, lda/ldw%%16==0, ldc%%8==0, ld_mx%%4==0");
  if (epi == EPI_GATE_RES && (!res || ldr % 8 || (gate && rows_per_batch < 1))) return fail(AFX_E_INVALID, "gated residual epilogue needs res");
  if (!gemm_fp8_mx_ok(M, N, K)) return fail(AFX_E_INVALID, "afx_linear_fp8_mx: the block-scaled kernel is switched off (AFX_FP8_V3 / AFX_GEMM_IMPL / AFX_GEMM_SK)");
  GemmBatch gb{};
  gb.nprob = 1;
  GemmProblem& p = gb.p[0];
  p = GemmProblem{};
  p.A = (const uint16_t*)Aq; p.lda = lda; p.W = (const uint16_t*)Wq; p.ldw = ldw; p.bias = (const uint16_t*)bias;
  p.C = (uint16_t*)C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.epi = epi; p.gelu_col0 = gelu_col0;
  p.gate = gate; p.ldg = ldg; p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; p.res = (const uint16_t*)res; p.ldr = ldr;
  p.fp8 = 1; p.a_scale = a_scale; p.w_scale = w_scale; p.a_mx = (const uint8_t*)a_mx; p.ld_mx = ld_mx;
  HIP_TRY(launch_gemm(gb, (hipStream_t)stream));
  return AFX_OK;
}

int afx_linear_fp8_to_mx8(const void* Aq, int64_t lda, const void* a_mx, int64_t ld_mx, const float* a_scale, const void* Wq, int64_t ldw,
                          const float* w_scale, const void* bias, void* C, int64_t ldc, void* c8, int64_t ldc8, void* c_mx, int64_t ld_cmx,
                          int32_t c8_col0, int32_t M, int32_t N, int32_t K, int32_t gelu, void* stream) {
  if (!Aq || !Wq || !c8 || !c_mx || !a_scale || !w_scale || (c8_col0 > 0 && !C)) return fail(AFX_E_INVALID, "null argument to afx_linear_fp8_to_mx8");
  if (M < 0 || N < 0 || K < 256 || K % 128 || N % 8 || lda % 16 || ldw % 16 || ldc % 8 || ldc8 % 8 || c8_col0 < 0 || c8_col0 % 128 || c8_col0 >= N ||
      ld_cmx < (N - c8_col0 + 127) / 128 || (a_mx && (K % 512 || ld_mx % 4 || ld_mx < K / 128)))
    return fail(AFX_E_INVALID, "afx_linear_fp8_to_mx8: need K%%128==0 (K%%512==0 with block-scaled A), c8_col0%%128==0, ldc8%%8==0");
  // (the producer epilogue lives in the one-wave-per-SIMD fp8 kernel: the same switches as the block-scaled consumer; K % 512 only binds with a_mx)
  if (!gemm_fp8_mx_ok(M, N, a_mx ? K : 512)) return fail(AFX_E_INVALID, "afx_linear_fp8_to_mx8: the one-wave-per-SIMD fp8 kernel is switched off");
  GemmBatch gb{};
  gb.nprob = 1;
  GemmProblem& p = gb.p[0];
  p = GemmProblem{};
  p.A = (const uint16_t*)Aq; p.lda = lda; p.W = (const uint16_t*)Wq; p.ldw = ldw; p.bias = (const uint16_t*)bias;
  p.C = (uint16_t*)C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.epi = gelu ? EPI_GELU : EPI_NONE; p.gelu_col0 = gelu ? c8_col0 : 0;
  p.rows_per_batch = 1;
  p.fp8 = 1; p.a_scale = a_scale; p.w_scale = w_scale; p.a_mx = (const uint8_t*)a_mx; p.ld_mx = ld_mx;
  p.c8 = (uint8_t*)c8; p.ldc8 = ldc8; p.c_mx = (uint8_t*)c_mx; p.ld_cmx = ld_cmx; p.c8_col0 = c8_col0;
  HIP_TRY(launch_gemm(gb, (hipStream_t)stream));
  return AFX_OK;
}

int afx_finish_f32_bf16(const float* partials, int32_t nslab, const void* res, int64_t ldr, void* out, int64_t ldo, int64_t M, int32_t N,
                        void* stream) {
  if (!partials || !out || M < 1 || N % 8 || nslab < 1 || ldo % 8 || (res && ldr % 8)) return fail(AFX_E_INVALID, "bad argument to afx_finish_f32_bf16");
  hipLaunchKernelGGL(finish_f32_kernel, dim3(tblocks(M * (N >> 3))), dim3(256), 0, (hipStream_t)stream, partials, nslab, (const bf16_t*)res,
                     ldr, (bf16_t*)out, ldo, M, N);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int64_t afx_attention_ext_ws_bytes(int32_t B, int32_t Hkv, int32_t S, int32_t head_dim) {
  return (int64_t)B * Hkv * head_dim * attn_spad(S) * 2;
}

int afx_attention_ext_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo,
                           void* ws, int32_t B, int32_t H, int32_t Hkv, int32_t S, int32_t head_dim, float scale, int32_t causal,
                           const float* bias, void* stream) {
  if (!q || !k || !v || !o || !ws || B < 1 || S < 1 || H < 1 || Hkv < 1 || H % Hkv || (head_dim != 64 && head_dim != 128) ||
      ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8 || !(scale > 0.f))
    return fail(AFX_E_INVALID, "afx_attention_ext_bf16: head_dim 64 or 128, H %% Hkv == 0, strides %% 8 == 0, scale > 0");
  HIP_TRY(launch_attention_ext((const uint16_t*)q, ldq, (const uint16_t*)k, ldk, (const uint16_t*)v, ldv, (uint16_t*)ws, (uint16_t*)o,
                               ldo, B, H, Hkv, S, head_dim, scale, causal, bias, (hipStream_t)stream));
  return AFX_OK;
}

}  // extern "C"
