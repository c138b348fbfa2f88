// VAE decoder kernels (FLUX AutoencoderKL decoder: the step right after the denoising loop,
// reference lakonlab/pipelines/arcflux_pipeline.py:531-534; diffusers-owned architecture).
//
// Layout: every activation is NHWC bf16 on a ZERO-BORDERED grid [H+2][W+2][C] (flat row = pixel, C contiguous).
// A 3x3 convolution is then an implicit GEMM on the MFMA kernel of afx_gemm.hip with NO im2col: K-tile
// (tap, 64-channel chunk) reads the same pixel rows shifted by dy*(W+2)+dx (see GemmProblem::conv_*), the
// epilogue re-zeroes the border, so each layer's output is directly the next layer's padded input.
// The kernels here are the HBM-bound glue: GroupNorm(32) statistics + apply(+SiLU), nearest 2x upsample,
// interior gather / scatter for the single-head mid-block attention, fp32 row softmax, latent / image layout.
#include <algorithm>

#include "afx_api_util.h"
#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {

// ---- GroupNorm statistics: stats[g] = {sum, sum of squares} in fp64 (border rows are zero and do not count) ----
// Every thread owns one 16-byte channel chunk and strides over the rows (the grid is sized to ~4 blocks per CU, so the
// fp64 atomics on the 2 * groups result words stay in the low thousands).
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, int64_t rows, int C, int groups,
                                                       double* __restrict__ stats) {
  __shared__ float part[64][2];
  const int cpr = C >> 3;                       // 16-byte chunks per row
  const int gs = C / groups;                    // channels per group (>= 4)
  if (threadIdx.x < 64) part[threadIdx.x][0] = part[threadIdx.x][1] = 0.f;
  __syncthreads();
  const int rstep = 256 / cpr;
  const int c = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  if (rl < rstep) {
    const int64_t stride = (int64_t)gridDim.x * rstep;
    int64_t r = (int64_t)blockIdx.x * rstep + rl;
    for (; r + 3 * stride < rows; r += 4 * stride) {        // four independent 16-byte loads in flight per lane
      u32x4_t w4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w4[u] = *reinterpret_cast<const u32x4_t*>(x + (r + u * stride) * C + c * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8];
        unpack8(w4[u], v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s0 += v[e]; q0 += v[e] * v[e];
          s1 += v[4 + e]; q1 += v[4 + e] * v[4 + e];
        }
      }
    }
    for (; r < rows; r += stride) {
      float v[8];
      unpack8(*reinterpret_cast<const u32x4_t*>(x + r * C + c * 8), v);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s0 += v[e]; q0 += v[e] * v[e];
        s1 += v[4 + e]; q1 += v[4 + e] * v[4 + e];
      }
    }
    const int g0 = (c * 8) / gs, g1 = (c * 8 + 4) / gs;
    atomicAdd(&part[g0][0], s0); atomicAdd(&part[g0][1], q0);
    atomicAdd(&part[g1][0], s1); atomicAdd(&part[g1][1], q1);
  }
  __syncthreads();
  // 2048 blocks adding to the same 2 * groups doubles serialise in the L2's atomic unit: 93 us for a 17 MB grid.  One of GN_SLOTS copies per
  // block (gn_coeff_kernel folds them): 32 atomics per address instead of 2048.
  if (threadIdx.x < groups) {
    double* dst = stats + ((int64_t)(blockIdx.x % GN_SLOTS) * groups + threadIdx.x) * 2;
    atomicAdd(dst, (double)part[threadIdx.x][0]);
    atomicAdd(dst + 1, (double)part[threadIdx.x][1]);
  }
}

// stats: [GN_SLOTS][groups][2] partial sums (the statistics kernel's or a convolution epilogue's)
__global__ void gn_coeff_kernel(const double* __restrict__ stats, double count, int C, int groups, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float eps, float* __restrict__ coef) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= C) return;
  const int gi = ch / (C / groups);
  double s0 = 0.0, s1 = 0.0;                                     // fold the GN_SLOTS partial sums of this channel's group
  for (int sl = 0; sl < GN_SLOTS; ++sl) {
    s0 += stats[((int64_t)sl * groups + gi) * 2];
    s1 += stats[((int64_t)sl * groups + gi) * 2 + 1];
  }
  const double mean = s0 / count;
  const double var = s1 / count - mean * mean;
  const float a = rsqrtf((float)var + eps) * gamma[ch];
  coef[ch] = a;
  coef[C + ch] = beta[ch] - (float)mean * a;
}

// y = act(x * a_c + b_c), border pixels forced to 0.  act: 0 none, 1 SiLU.  One 16-byte chunk column per thread.
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t rows, int C,
                                                       const float* __restrict__ coef, int act, int hp, int wp) {
  const int cpr = C >> 3;
  const int rstep = 256 / cpr;
  const int c = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  if (rl >= rstep) return;
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = coef[c * 8 + e]; b[e] = coef[C + c * 8 + e]; }
  for (int64_t r = (int64_t)blockIdx.x * rstep + rl; r < rows; r += (int64_t)gridDim.x * rstep) {
    const int yy = (int)(r / wp), xx = (int)(r - (int64_t)yy * wp);
    float o[8];
    if (yy == 0 || yy == hp - 1 || xx == 0 || xx == wp - 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
    } else {
      float v[8];
      unpack8(*reinterpret_cast<const u32x4_t*>(x + r * C + c * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = v[e] * a[e] + b[e];
        o[e] = act ? silu(t) : t;
      }
    }
    *reinterpret_cast<u32x4_t*>(y + r * C + c * 8) = pack8(o);
  }
}

// nearest 2x upsample between zero-bordered NHWC grids: out[(2h+a+1), (2w+b+1)] = in[h+1, w+1]
__global__ __launch_bounds__(256) void upsample2x_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int H, int W, int C) {
  const int cpr = C >> 3;
  const int Ho = 2 * H + 2, Wo = 2 * W + 2;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (int64_t)Ho * Wo * cpr) return;
  const int c = (int)(g % cpr);
  const int64_t p = g / cpr;
  const int yo = (int)(p / Wo), xo = (int)(p % Wo);
  u32x4_t v = (u32x4_t){0u, 0u, 0u, 0u};
  if (yo >= 1 && yo <= 2 * H && xo >= 1 && xo <= 2 * W) {
    const int yi = (yo - 1) / 2 + 1, xi = (xo - 1) / 2 + 1;
    v = *reinterpret_cast<const u32x4_t*>(x + ((int64_t)yi * (W + 2) + xi) * C + c * 8);
  }
  *reinterpret_cast<u32x4_t*>(y + p * C + c * 8) = v;
}

// interior <-> compact: dir 0: compact[h*W+w] = padded[(h+1),(w+1)];  dir 1: padded interior = compact (+ res interior)
__global__ __launch_bounds__(256) void interior_kernel(bf16_t* __restrict__ padded, bf16_t* __restrict__ compact,
                                                       const bf16_t* __restrict__ res, int H, int W, int C, int dir) {
  const int cpr = C >> 3;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (int64_t)H * W * cpr) return;
  const int c = (int)(g % cpr);
  const int64_t p = g / cpr;
  const int h = (int)(p / W), w = (int)(p % W);
  const int64_t pp = ((int64_t)(h + 1) * (W + 2) + (w + 1)) * C + c * 8;
  if (dir == 0) {
    *reinterpret_cast<u32x4_t*>(compact + p * C + c * 8) = *reinterpret_cast<const u32x4_t*>(padded + pp);
  } else {
    float a[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(compact + p * C + c * 8), a);
    if (res != nullptr) {
      float b[8];
      unpack8(*reinterpret_cast<const u32x4_t*>(res + pp), b);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += b[e];
    }
    *reinterpret_cast<u32x4_t*>(padded + pp) = pack8(a);
  }
}

// P[r, :] = softmax(scale * S[r, :])   fp32 in, bf16 out; one block per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int64_t lds_, bf16_t* __restrict__ p,
                                                           int64_t ldp, int cols, float scale) {
  __shared__ float red[4];
  const float* sr = s + (int64_t)blockIdx.x * lds_;
  bf16_t* pr = p + (int64_t)blockIdx.x * ldp;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < cols; i += 256) mx = fmaxf(mx, sr[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int i = threadIdx.x; i < cols; i += 256) sum += __expf((sr[i] - mx) * scale);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int i = threadIdx.x; i < cols; i += 256) pr[i] = f32_to_bf16(__expf((sr[i] - mx) * scale) * inv);
}

// The same for rows of at most 16384 columns (cols % 4 == 0): the row is read ONCE, 16 bytes per lane and load, and stays in registers
// between the maximum, the exponentials and the store (the three-pass kernel above reads it three times with 4-byte loads and takes
// exp twice: 685 us for the 16384 x 16384 scores of the 1024^2 decode's mid-block attention).
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const float* __restrict__ s, int64_t lds_, bf16_t* __restrict__ p, int64_t ldp,
                                                               int cols, float scale) {
  constexpr int NV = 16;
  __shared__ float red[4];
  const f32x4_t* sr = reinterpret_cast<const f32x4_t*>(s + (int64_t)blockIdx.x * lds_);
  bf16_t* pr = p + (int64_t)blockIdx.x * ldp;
  const int nv = cols >> 2;
  f32x4_t v[NV];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = threadIdx.x + 256 * j;
    v[j] = i < nv ? sr[i] : (f32x4_t){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) mx = fmaxf(fmaxf(fmaxf(v[j][0], v[j][1]), fmaxf(v[j][2], v[j][3])), mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[j][e] = __expf((v[j][e] - mx) * scale);                  // (-inf padding -> 0)
      sum += v[j][e];
    }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = threadIdx.x + 256 * j;
    if (i < nv) {
      const u32x2_t o = {pack_bf16x2(v[j][0] * inv, v[j][1] * inv), pack_bf16x2(v[j][2] * inv, v[j][3] * inv)};
      *reinterpret_cast<u32x2_t*>(pr + 4 * i) = o;
    }
  }
}

// packed latent tokens [N = hp*wp, 64] f32 (channel c*4 + ph*2 + pw) -> zero-bordered NHWC [2hp+2][2wp+2][Cpad] bf16,
// value = lat / scaling + shift  (arcflux_pipeline.py:531-532), channels >= 16 zero.
__global__ __launch_bounds__(256) void latent_to_nhwc_kernel(const float* __restrict__ tok, bf16_t* __restrict__ y, int hp, int wp,
                                                             int Cpad, float inv_scale, float shift) {
  const int H = 2 * hp, W = 2 * wp;
  const int64_t total = (int64_t)(H + 2) * (W + 2) * Cpad;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= total) return;
  const int c = (int)(g % Cpad);
  const int64_t p = g / Cpad;
  const int yy = (int)(p / (W + 2)), xx = (int)(p % (W + 2));
  float v = 0.f;
  if (c < 16 && yy >= 1 && yy <= H && xx >= 1 && xx <= W) {
    const int h = yy - 1, w = xx - 1;
    v = tok[((int64_t)(h >> 1) * wp + (w >> 1)) * 64 + c * 4 + (h & 1) * 2 + (w & 1)] * inv_scale + shift;
  }
  y[g] = f32_to_bf16(v);
}

// Same unpack with a per-pixel affine map on the 16 latent channels: v = A . lat + b (A row-major [16][16]).  The Qwen-Image
// pipeline's per-channel un-normalisation (lat * std + mean, arcqwen_pipeline.py:470-479) and the VAE's 1x1x1
// post_quant_conv are both of this form, so their product is applied here; border and channels >= 16 stay zero.
__global__ __launch_bounds__(256) void latent_to_nhwc_affine_kernel(const float* __restrict__ tok, bf16_t* __restrict__ y, int hp,
                                                                    int wp, int Cpad, const float* __restrict__ A,
                                                                    const float* __restrict__ b) {
  const int H = 2 * hp, W = 2 * wp;
  const int64_t total = (int64_t)(H + 2) * (W + 2) * Cpad;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= total) return;
  const int c = (int)(g % Cpad);
  const int64_t p = g / Cpad;
  const int yy = (int)(p / (W + 2)), xx = (int)(p % (W + 2));
  float v = 0.f;
  if (c < 16 && yy >= 1 && yy <= H && xx >= 1 && xx <= W) {
    const int h = yy - 1, w = xx - 1;
    const float* t = tok + ((int64_t)(h >> 1) * wp + (w >> 1)) * 64 + (h & 1) * 2 + (w & 1);
    v = b[c];
#pragma unroll
    for (int k = 0; k < 16; ++k) v += A[c * 16 + k] * t[k * 4];
  }
  y[g] = f32_to_bf16(v);
}

// Per-pixel RMS norm over the channels of an NHWC row (QwenImageRMS_norm: F.normalize(x, dim=C) * sqrt(C) * gamma), optional
// SiLU.  Creal channels carry data, channels up to Cpad are zero padding (gamma is zero there).  A row of Cpad <= 512
// channels is <= 64 16-byte chunks: LP = next power of two lanes share a row, 64 / LP rows per wave-load, the sum of squares
// is reduced inside the lane group; every lane keeps its gamma chunk in registers and strides over the rows.
template <int LP>
__global__ __launch_bounds__(256) void rmsnorm_nhwc_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t rows, int Cpad,
                                                           int Creal, const float* __restrict__ gamma, int act) {
  constexpr int RPW = 64 / LP;                  // rows per wave-load
  const int lane = threadIdx.x & 63;
  const int sub = lane % LP, rsel = lane / LP;
  const int nch = Cpad >> 3;
  const bool live = sub < nch;
  float gm[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) gm[e] = live ? gamma[sub * 8 + e] : 0.f;
  const float sc = sqrtf((float)Creal);
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwave = (int64_t)gridDim.x * 4;
  for (int64_t r0 = wave * RPW; r0 < rows; r0 += nwave * RPW) {
    const int64_t row = r0 + rsel;
    const bool ok = live && row < rows;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (ok) unpack8(*reinterpret_cast<const u32x4_t*>(x + row * Cpad + sub * 8), v);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
#pragma unroll
    for (int o = LP / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = sc / fmaxf(sqrtf(ss), 1e-12f);
    if (ok) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = v[e] * inv * gm[e];
        o[e] = act ? silu(t) : t;
      }
      *reinterpret_cast<u32x4_t*>(y + row * Cpad + sub * 8) = pack8(o);
    }
  }
}

// zero-bordered NHWC [H+2][W+2][C] bf16 -> image [3][H][W] f32
__global__ __launch_bounds__(256) void nhwc_to_image_kernel(const bf16_t* __restrict__ x, float* __restrict__ img, int H, int W, int C) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (int64_t)3 * H * W) return;
  const int ch = (int)(g / ((int64_t)H * W));
  const int64_t p = g % ((int64_t)H * W);
  const int h = (int)(p / W), w = (int)(p % W);
  img[g] = bf16_to_f32(x[((int64_t)(h + 1) * (W + 2) + (w + 1)) * C + ch]);
}

}  // namespace afx

using namespace afx;

static inline unsigned vblocks(int64_t n) { return (unsigned)((n + 255) / 256); }

extern "C" {

static int conv3x3_impl(const void* x, const void* w, const void* bias, void* y, int32_t H, int32_t W, int32_t Cin,
                        int32_t Cout, const void* res, double* gn_stats, int32_t groups, void* stream) {
  // x, y, res: zero-bordered NHWC grids [(H+2)*(W+2), C]; x must have >= (W+3) readable rows before and after
  // (guard band of the shifted taps).  w: [Cout][3][3][Cin] bf16 (tap-major K).  Cin % 64 == 0, Cout % 8 == 0.
  if (!x || !w || !y || H < 1 || W < 1 || Cin < 64 || Cin % 64 || Cout < 8 || Cout % 8)
    return fail(AFX_E_INVALID, "afx_conv3x3_bf16: need Cin %% 64 == 0, Cout %% 8 == 0");
  GemmBatch gb{};
  gb.nprob = 1;
  GemmProblem& p = gb.p[0];
  p = GemmProblem{};
  p.A = (const uint16_t*)x; p.lda = Cin; p.W = (const uint16_t*)w; p.ldw = 9 * (int64_t)Cin; p.bias = (const uint16_t*)bias;
  p.C = (uint16_t*)y; p.ldc = Cout; p.M = (H + 2) * (W + 2); p.N = Cout; p.K = 9 * Cin;
  p.rows_per_batch = 1 << 30;
  p.conv_cin_tiles = Cin / 64; p.conv_wp = W + 2; p.conv_hp = H + 2;
  if (res != nullptr) {          // y = res + conv(x): residual epilogue with a unit gate
    p.epi = EPI_GATE_RES; p.gate = nullptr; p.ldg = 0; p.res = (const uint16_t*)res; p.ldr = Cout;
  }
  if (gn_stats != nullptr) {   // GroupNorm sums of y from the epilogue (one-wave-per-SIMD kernel only): [GN_SLOTS][groups][2] doubles, zeroed here
    const int gs = groups > 0 && Cout % groups == 0 ? Cout / groups : 0;
    if (groups < 1 || groups > 64 || gs < 4 || gs % 4 || (gs > 8 && gs % 8) || Cout > 128 || !gemm_conv_stats_available())
      return fail(AFX_E_INVALID, "afx_conv3x3_bf16_stats: need Cout <= 128 (the 256x128-tile kernel), Cout / groups = 4, 8 or a multiple of 8, and the one-wave-per-SIMD GEMM");
    HIP_TRY(hipMemsetAsync(gn_stats, 0, sizeof(double) * 2 * groups * GN_SLOTS, (hipStream_t)stream));
    p.gn_stats = gn_stats; p.gn_gs = Cout / groups; p.gn_groups = groups;
  }
  HIP_TRY(launch_gemm(gb, (hipStream_t)stream));
  return AFX_OK;
}

int afx_conv3x3_bf16(const void* x, const void* w, const void* bias, void* y, int32_t H, int32_t W, int32_t Cin,
                     int32_t Cout, const void* res, void* stream) {
  return conv3x3_impl(x, w, bias, y, H, W, Cin, Cout, res, nullptr, 0, stream);
}

int afx_conv3x3_bf16_stats(const void* x, const void* w, const void* bias, void* y, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                           const void* res, double* gn_stats, int32_t groups, void* stream) {
  if (!gn_stats) return fail(AFX_E_INVALID, "afx_conv3x3_bf16_stats: null statistics buffer");
  return conv3x3_impl(x, w, bias, y, H, W, Cin, Cout, res, gn_stats, groups, stream);
}

int afx_upconv3x3_bf16(const void* x, const void* w4, const void* bias, void* y, int32_t H, int32_t W, int32_t Cin, int32_t Cout, void* stream) {
  // y = conv3x3(upsample_nearest_2x(x)) + bias without the upsampled grid: x is the LOW-resolution zero-bordered grid [(H+2)*(W+2), Cin] (guard rows
  // as for afx_conv3x3_bf16), y the [(2H+2)*(2W+2), Cout] grid, w4 = the four phase kernels [2 py + px][Cout][2][2][Cin] bf16 (the 3x3 taps that fall
  // on the same source pixel summed: arcflow_amd.vae.phase_weights).  One launch of four problems on the one-wave-per-SIMD kernel.
  if (!x || !w4 || !y || H < 1 || W < 1 || Cin < 64 || Cin % 64 || Cout < 8 || Cout % 8)
    return fail(AFX_E_INVALID, "afx_upconv3x3_bf16: need Cin %% 64 == 0, Cout %% 8 == 0");
  if (!gemm_conv_stats_available()) return fail(AFX_E_INVALID, "afx_upconv3x3_bf16 needs the one-wave-per-SIMD GEMM (AFX_GEMM_IMPL=3, no forced tile)");
  if ((int64_t)(2 * H + 2) * (2 * W + 2) * Cout * 2 >= (1ll << 31)) return fail(AFX_E_INVALID, "afx_upconv3x3_bf16: output grid >= 2 GiB");
  GemmBatch gb{};
  gb.nprob = 4;
  for (int ph = 0; ph < 4; ++ph) {
    GemmProblem& p = gb.p[ph];
    p = GemmProblem{};
    p.A = (const uint16_t*)x; p.lda = Cin; p.W = (const uint16_t*)w4 + (int64_t)ph * Cout * 4 * Cin; p.ldw = 4 * (int64_t)Cin; p.bias = (const uint16_t*)bias;
    p.C = (uint16_t*)y; p.ldc = Cout; p.M = (H + 2) * (W + 2); p.N = Cout; p.K = 4 * Cin;
    p.rows_per_batch = 1 << 30;
    p.conv_cin_tiles = Cin / 64; p.conv_wp = W + 2; p.conv_hp = H + 2;
    p.up_phase = 1 + ph;
  }
  HIP_TRY(launch_gemm(gb, (hipStream_t)stream));
  return AFX_OK;
}

int afx_conv_stats_available(void) { return gemm_conv_stats_available() ? 1 : 0; }


int64_t afx_groupnorm_ws_bytes(int32_t C, int32_t groups) {
  if (C < 8 || C > 2048 || groups < 1 || groups > 64) return AFX_E_INVALID;
  return (int64_t)sizeof(double) * ((2 + 2 * GN_SLOTS) * (int64_t)groups + C);
}

int afx_groupnorm_nhwc(const void* x, void* y, double* stats_ws, int32_t H, int32_t W, int32_t C, int32_t groups,
                       const float* gamma, const float* beta, float eps, int32_t act, void* stream) {
  // stats_ws layout (doubles): [0, 2 groups) the sums | [2 groups, 2 groups + C) the 2 C float coefficients | then AFX_GN_SLOTS x 2 groups slot partials
  if (!x || !y || !stats_ws || !gamma || !beta || C % 8 || groups < 1 || groups > 64 || C % groups || (C / groups) % 4 || 256 % (C >> 3))
    return fail(AFX_E_INVALID, "bad argument to afx_groupnorm_nhwc");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)(H + 2) * (W + 2);
  if (C > 2048) return fail(AFX_E_INVALID, "afx_groupnorm_nhwc: C <= 2048");
  double* slots = stats_ws + 2 * groups + C;
  HIP_TRY(hipMemsetAsync(slots, 0, sizeof(double) * 2 * groups * GN_SLOTS, st));
  const int rstep = 256 / (C >> 3);
  const unsigned nblk = (unsigned)std::min<int64_t>(2048, (rows + rstep - 1) / rstep);
  float* coef = reinterpret_cast<float*>(stats_ws + 2 * groups);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nblk), dim3(256), 0, st, (const bf16_t*)x, rows, C, groups, slots);
  const double count = (double)H * W * (C / groups);
  hipLaunchKernelGGL(gn_coeff_kernel, dim3((C + 255) / 256), dim3(256), 0, st, slots, count, C, groups, gamma, beta, eps, coef);
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)std::min<int64_t>(4096, (rows + rstep - 1) / rstep)), dim3(256), 0, st,
                     (const bf16_t*)x, (bf16_t*)y, rows, C, coef, act, H + 2, W + 2);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_groupnorm_nhwc_from_stats(const void* x, void* y, const double* gn_stats, double* stats_ws, int32_t H, int32_t W, int32_t C,
                                  int32_t groups, const float* gamma, const float* beta, float eps, int32_t act, void* stream) {
  // gn_stats: what afx_conv3x3_bf16_stats left for the grid x ([GN_SLOTS][groups][2] partial sums); stats_ws as in afx_groupnorm_nhwc
  if (!x || !y || !gn_stats || !stats_ws || !gamma || !beta || C % 8 || groups < 1 || groups > 64 || C % groups || 256 % (C >> 3) || C > 2048)
    return fail(AFX_E_INVALID, "bad argument to afx_groupnorm_nhwc_from_stats");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)(H + 2) * (W + 2);
  const int rstep = 256 / (C >> 3);
  float* coef = reinterpret_cast<float*>(stats_ws + 2 * groups);
  const double count = (double)H * W * (C / groups);
  hipLaunchKernelGGL(gn_coeff_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gn_stats, count, C, groups, gamma, beta, eps, coef);
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)std::min<int64_t>(4096, (rows + rstep - 1) / rstep)), dim3(256), 0, st,
                     (const bf16_t*)x, (bf16_t*)y, rows, C, coef, act, H + 2, W + 2);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_upsample2x_nhwc(const void* x, void* y, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!x || !y || C % 8) return fail(AFX_E_INVALID, "bad argument to afx_upsample2x_nhwc");
  hipLaunchKernelGGL(upsample2x_kernel, dim3(vblocks((int64_t)(2 * H + 2) * (2 * W + 2) * (C >> 3))), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, H, W, C);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_interior_nhwc(void* padded, void* compact, const void* res_padded, int32_t H, int32_t W, int32_t C, int32_t scatter,
                      void* stream) {
  if (!padded || !compact || C % 8) return fail(AFX_E_INVALID, "bad argument to afx_interior_nhwc");
  hipLaunchKernelGGL(interior_kernel, dim3(vblocks((int64_t)H * W * (C >> 3))), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)padded, (bf16_t*)compact, (const bf16_t*)res_padded, H, W, C, scatter);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_softmax_rows_f32(const float* s, int64_t lds_, void* p, int64_t ldp, int32_t rows, int32_t cols, float scale,
                         void* stream) {
  if (!s || !p || rows < 1 || cols < 1) return fail(AFX_E_INVALID, "bad argument to afx_softmax_rows_f32");
  if (cols % 4 == 0 && cols <= 16384 && lds_ % 4 == 0 && ldp % 4 == 0 && ((uintptr_t)s & 15) == 0 && ((uintptr_t)p & 7) == 0)
    hipLaunchKernelGGL(softmax_rows_reg_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, s, lds_, (bf16_t*)p, ldp, cols, scale);
  else
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, s, lds_, (bf16_t*)p, ldp, cols, scale);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_latent_to_nhwc(const float* tokens, void* y, int32_t hp, int32_t wp, int32_t Cpad, float scaling_factor,
                       float shift_factor, void* stream) {
  if (!tokens || !y || Cpad < 16 || Cpad % 8) return fail(AFX_E_INVALID, "bad argument to afx_latent_to_nhwc");
  const int64_t total = (int64_t)(2 * hp + 2) * (2 * wp + 2) * Cpad;
  hipLaunchKernelGGL(latent_to_nhwc_kernel, dim3(vblocks(total)), dim3(256), 0, (hipStream_t)stream, tokens, (bf16_t*)y, hp, wp,
                     Cpad, 1.0f / scaling_factor, shift_factor);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_latent_to_nhwc_affine(const float* tokens, void* y, int32_t hp, int32_t wp, int32_t Cpad, const float* A, const float* b,
                              void* stream) {
  if (!tokens || !y || !A || !b || Cpad < 16 || Cpad % 8) return fail(AFX_E_INVALID, "bad argument to afx_latent_to_nhwc_affine");
  const int64_t total = (int64_t)(2 * hp + 2) * (2 * wp + 2) * Cpad;
  hipLaunchKernelGGL(latent_to_nhwc_affine_kernel, dim3(vblocks(total)), dim3(256), 0, (hipStream_t)stream, tokens, (bf16_t*)y, hp,
                     wp, Cpad, A, b);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_rmsnorm_nhwc(const void* x, void* y, int64_t rows, int32_t Cpad, int32_t Creal, const float* gamma, int32_t act,
                     void* stream) {
  if (!x || !y || !gamma || rows < 1 || Cpad % 8 || Cpad > 512 || Creal < 1 || Creal > Cpad)
    return fail(AFX_E_INVALID, "afx_rmsnorm_nhwc: need Cpad %% 8 == 0, Cpad <= 512, 1 <= Creal <= Cpad");
  const int nch = Cpad >> 3;
  const int lp = nch <= 8 ? 8 : nch <= 16 ? 16 : nch <= 32 ? 32 : 64;
  const int64_t loads = (rows + (64 / lp) - 1) / (64 / lp);                 // wave-loads of 64 / lp rows
  const dim3 grid((unsigned)std::min<int64_t>(4096, (loads + 3) / 4)), blk(256);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* xi = (const bf16_t*)x;
  bf16_t* yo = (bf16_t*)y;
  if (lp == 8) hipLaunchKernelGGL(rmsnorm_nhwc_kernel<8>, grid, blk, 0, st, xi, yo, rows, Cpad, Creal, gamma, act);
  else if (lp == 16) hipLaunchKernelGGL(rmsnorm_nhwc_kernel<16>, grid, blk, 0, st, xi, yo, rows, Cpad, Creal, gamma, act);
  else if (lp == 32) hipLaunchKernelGGL(rmsnorm_nhwc_kernel<32>, grid, blk, 0, st, xi, yo, rows, Cpad, Creal, gamma, act);
  else hipLaunchKernelGGL(rmsnorm_nhwc_kernel<64>, grid, blk, 0, st, xi, yo, rows, Cpad, Creal, gamma, act);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_nhwc_to_image(const void* x, float* img, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!x || !img || C < 3) return fail(AFX_E_INVALID, "bad argument to afx_nhwc_to_image");
  hipLaunchKernelGGL(nhwc_to_image_kernel, dim3(vblocks((int64_t)3 * H * W)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, img, H, W, C);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

}  // extern "C"
