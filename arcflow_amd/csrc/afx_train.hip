// Kernels of the distillation step that the reference runs as eager torch autograd:
//   * backward of the analytic ArcFlow transport / policy velocity w.r.t. the mixture parameters
//     (lakonlab/models/diffusions/arcflow.py:81-110 under autograd; policies/arcflow.py:52-76)
//   * flow MSE loss + gradient (lakonlab/models/losses/diffusion_loss.py:44-83)
//   * head / norm_out gradient helpers (log-softmax backward, column reductions, transposes)
//   * fused AdamW, global grad-norm, clip, Karras-EMA lerp (lakonlab/models/base.py:76-103,
//     lakonlab/runner/hooks/ema_hook.py:86-124)
// All HBM-bound streaming kernels: 16 B per lane where the layout allows, wave64 shuffle reductions.
#include <algorithm>

#include "afx_api_util.h"
#include <map>
#include <mutex>
#include <utility>

#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {

// ------------------------------------------------------------------------------------------------
// d/d(means, logw, logg) of   D[c] = sum_k softmax(logw)_k * m[k,c] * e_k
//   step mode:      e_0 = Dt,  e_k = exp(g_k Dp) * Dt * phi(g_k Dt)       (D = displacement, x_end = x - D)
//   velocity mode:  e_0 = 1,   e_k = exp(g_k Dp)                           (D = u)
// given the upstream gradient gD = gscale[b] * g[c].  One wave per token, lane = packed channel (ch <= 64,
// pp a power of two dividing 64): the reductions over the channels that share a sub-pixel q = c % pp are
// xor-shuffles over lane bits >= log2(pp).
constexpr int TR_MAXK = 32;

template <typename MixT> AFX_DEV float tr_load(const MixT* p, int64_t i);
template <> AFX_DEV float tr_load<float>(const float* p, int64_t i) { return p[i]; }
template <> AFX_DEV float tr_load<bf16_t>(const bf16_t* p, int64_t i) { return bf16_to_f32(p[i]); }

template <typename MixT>
__global__ __launch_bounds__(256) void arcflow_bwd_kernel(
    const float* __restrict__ g, const MixT* __restrict__ means, const MixT* __restrict__ logw,
    const MixT* __restrict__ logg, float s_src, float s_start, float s_end, const float* __restrict__ sigma_vec,
    const float* __restrict__ gscale_vec, float gscale, float eps, float* __restrict__ d_means,
    float* __restrict__ d_logw, float* __restrict__ d_logg, int64_t tokens, int n_tok, int K, int ch, int pp,
    int velocity_only, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= tokens) return;
  const int64_t b = tok / n_tok;
  if (sigma_vec != nullptr) {
    s_src = sigma_vec[3 * b];
    s_start = sigma_vec[3 * b + 1];
    s_end = sigma_vec[3 * b + 2];
  }
  if (gscale_vec != nullptr) gscale *= gscale_vec[b];
  const float d_past = s_src - s_start;
  const float d_step = s_start - s_end;
  const bool active = lane < ch;
  const int c = active ? lane : 0;
  const int q = c % pp;
  const MixT* mt = means + tok * (int64_t)K * ch;
  const MixT* wt = logw + tok * (int64_t)K * pp;
  const MixT* gt = logg + tok * (int64_t)(K - 1) * pp;
  const float gD = active ? gscale * g[tok * ch + c] : 0.f;

  float w[TR_MAXK];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < TR_MAXK; ++k)
    if (k < K) {
      w[k] = tr_load<MixT>(wt, k * pp + q);
      mx = fmaxf(mx, w[k]);
    }
  float den = 0.f;
#pragma unroll
  for (int k = 0; k < TR_MAXK; ++k)
    if (k < K) {
      w[k] = expf(w[k] - mx);
      den += w[k];
    }
  const float inv = 1.0f / den;

  float* dm = d_means + tok * (int64_t)K * ch;
  float* dw = d_logw + tok * (int64_t)K * pp;
  float* dg = d_logg + tok * (int64_t)(K - 1) * pp;
  float a[TR_MAXK];       // a_k = e_k * sum_{c in q} gD_c m_{k,c}
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < TR_MAXK; ++k)
    if (k < K) {
      w[k] *= inv;
      const float m = tr_load<MixT>(mt, (int64_t)k * ch + c);
      float e, de;
      if (k == 0) {
        e = velocity_only ? 1.0f : d_step;
        de = 0.f;
      } else {
        const float gam = tr_load<MixT>(gt, (k - 1) * pp + q);
        const float dec = expf(gam * d_past);
        if (velocity_only) {
          e = dec;
          de = d_past * dec;
        } else {
          const float z = gam * d_step;
          const bool clamped = fabsf(z) < eps;
          const float zs = (z < 0.f ? -1.0f : 1.0f) * fmaxf(fabsf(z), eps);
          const float em1 = expm1f(zs);
          const float phi = em1 / zs;
          const float dphi = clamped ? 0.f : (em1 + 1.0f - phi) / zs;   // phi'(z), 0 inside the clamp
          e = dec * d_step * phi;
          de = d_past * e + dec * d_step * dphi * d_step;
        }
      }
      // gradient w.r.t. the mean of this lane's channel
      if (active) {
        const float v = gD * w[k] * e;
        dm[(int64_t)k * ch + c] = accumulate ? dm[(int64_t)k * ch + c] + v : v;
      }
      // r_k = sum over the channels of sub-pixel q of gD_c m_{k,c}
      float r = gD * m;
      for (int o = pp; o < 64; o <<= 1) r += __shfl_xor(r, o, 64);
      a[k] = e * r;
      s += w[k] * a[k];
      if (k > 0 && lane < pp) {
        const float v = w[k] * de * r;
        dg[(k - 1) * pp + q] = accumulate ? dg[(k - 1) * pp + q] + v : v;
      }
    }
  if (lane < pp) {
#pragma unroll
    for (int k = 0; k < TR_MAXK; ++k)
      if (k < K) {
        const float v = w[k] * (a[k] - s);
        dw[k * pp + q] = accumulate ? dw[k * pp + q] + v : v;
      }
  }
}

hipError_t launch_arcflow_bwd(const float* g, const void* means, const void* logw, const void* logg, int mix_bf16,
                              float s_src, float s_start, float s_end, const float* sigma_vec,
                              const float* gscale_vec, float gscale, float eps, float* d_means, float* d_logw,
                              float* d_logg, int B, int n_tok, int K, int ch, int pp, int velocity_only,
                              int accumulate, hipStream_t stream) {
  if (K < 1 || K > TR_MAXK || ch < 1 || ch > 64 || pp < 1 || (pp & (pp - 1)) || 64 % pp || ch % pp)
    return hipErrorInvalidValue;
  const int64_t tokens = (int64_t)B * n_tok;
  if (tokens == 0) return hipSuccess;
  dim3 grid((unsigned)((tokens + 3) / 4)), block(256);
  if (mix_bf16)
    hipLaunchKernelGGL(arcflow_bwd_kernel<bf16_t>, grid, block, 0, stream, g, (const bf16_t*)means, (const bf16_t*)logw,
                       (const bf16_t*)logg, s_src, s_start, s_end, sigma_vec, gscale_vec, gscale, eps, d_means, d_logw,
                       d_logg, tokens, n_tok, K, ch, pp, velocity_only, accumulate);
  else
    hipLaunchKernelGGL(arcflow_bwd_kernel<float>, grid, block, 0, stream, g, (const float*)means, (const float*)logw,
                       (const float*)logg, s_src, s_start, s_end, sigma_vec, gscale_vec, gscale, eps, d_means, d_logw,
                       d_logg, tokens, n_tok, K, ch, pp, velocity_only, accumulate);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// loss += coef * 0.5 * sum (p - t)^2 ;  grad = coef * (p - t)       (coef folds scale / numel / segment size)
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                  float coef, float* __restrict__ grad, float* __restrict__ loss,
                                                  int64_t n) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = p[i] - t[i];
    acc += d * d;
    if (grad) grad[i] = coef * d;
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, 0.5f * coef * (part[0] + part[1] + part[2] + part[3]));
}

// sum of squares of a flat fp32 buffer -> atomicAdd into out[0]  (global grad norm)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += x[i] * x[i];
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// AdamW (decoupled weight decay), bias-corrected, fp32 state; gscale folds 1/world and the clip factor.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float bc2, float gscale,
                                                    int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  float pi = p[i] * (1.0f - lr * wd);
  pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
  p[i] = pi;
}

// AdamW with block-wise 8-bit moments (the reference trains with bitsandbytes AdamW8bit: lakonlab/configs/flux/_ddp_train.py:18-26,
// lakonlab/runner/optimizer/builder.py:11-24).  bitsandbytes is absent from /root/reference (unpinned third-party dependency):
// this restates its published block-wise scheme -- each block of 256 values stores one fp32 absmax and 256 one-byte codes into a
// 256-entry "dynamic" code book (signed for exp_avg, unsigned for exp_avg_sq; arcflow_amd/ops.py dynamic_map); a step dequantises,
// applies the fp32 AdamW recurrences above, updates the parameter from the UNquantised new moments, and stores the moments
// re-quantised to the nearest code against the block's new absmax.  One work-group = one block, one value per lane.
AFX_DEV int nearest_code(const float* __restrict__ qmap, float x) {
  // qmap is sorted ascending over 256 entries: binary search for the insertion point, then the nearer neighbour
  int lo = 0, hi = 255;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int mid = (lo + hi) >> 1;
    if (qmap[mid] < x) lo = mid + 1; else hi = mid;
  }
  if (lo > 0 && x - qmap[lo - 1] <= qmap[lo] - x) --lo;
  return lo;
}

__global__ __launch_bounds__(256) void adamw8bit_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        uint8_t* __restrict__ c1, uint8_t* __restrict__ c2,
                                                        float* __restrict__ absmax1, float* __restrict__ absmax2,
                                                        const float* __restrict__ qmap1, const float* __restrict__ qmap2, float lr,
                                                        float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                        float gscale, int64_t n) {
  __shared__ float q1[256], q2[256], red[2][4];
  const int tid = threadIdx.x;
  q1[tid] = qmap1[tid];
  q2[tid] = qmap2[tid];
  __syncthreads();
  const int64_t blk = blockIdx.x, i = blk * 256 + tid;
  const bool on = i < n;
  float mi = 0.f, vi = 0.f, gi = 0.f;
  if (on) {
    gi = g[i] * gscale;
    mi = b1 * (q1[c1[i]] * absmax1[blk]) + (1.0f - b1) * gi;
    vi = b2 * (q2[c2[i]] * absmax2[blk]) + (1.0f - b2) * gi * gi;
  }
  float a1 = fabsf(mi), a2 = vi;                     // block maxima (v >= 0)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a1 = fmaxf(a1, __shfl_xor(a1, o, 64));
    a2 = fmaxf(a2, __shfl_xor(a2, o, 64));
  }
  if ((tid & 63) == 0) { red[0][tid >> 6] = a1; red[1][tid >> 6] = a2; }
  __syncthreads();
  a1 = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  a2 = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  if (tid == 0) { absmax1[blk] = a1; absmax2[blk] = a2; }
  if (!on) return;
  c1[i] = (uint8_t)nearest_code(q1, a1 > 0.f ? mi / a1 : 0.f);
  c2[i] = (uint8_t)nearest_code(q2, a2 > 0.f ? vi / a2 : 0.f);
  float pi = p[i] * (1.0f - lr * wd);
  pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
  p[i] = pi;
}

// ema = net + (ema - net) * beta        (mmgen lerp; lakonlab/runner/hooks/ema_hook.py:118-124)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ ema, const float* __restrict__ net, float beta,
                                                  int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) ema[i] = net[i] + (ema[i] - net[i]) * beta;
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i + 1 < n) {
    *reinterpret_cast<uint32_t*>(y + i) = pack_bf16x2(x[i], x[i + 1]);
  } else if (i < n) {
    y[i] = f32_to_bf16(x[i]);
  }
}

// ------------------------------------------------------------------------------------------------
// Head-logit gradient row:  dY = [ d_means | log_softmax-backward(d_logw) | d_logg | 0-pad ]   (bf16)
//   d raw[k,q] = d_lw[k,q] - exp(logw_out[k,q]) * sum_k d_lw[k,q]          (arcflux.py:246-247 backward)
__global__ __launch_bounds__(256) void head_grad_kernel(const float* __restrict__ d_means, const float* __restrict__ d_logw,
                                                        const float* __restrict__ d_logg, const bf16_t* __restrict__ logw_out,
                                                        bf16_t* __restrict__ dy, int64_t ldy, int K, int ch, int lw) {
  const int64_t row = blockIdx.x;
  const int nm = K * ch, nw = K * lw, ng = (K - 1) * lw;
  bf16_t* out = dy + row * ldy;
  for (int i = threadIdx.x; i < nm; i += 256) out[i] = f32_to_bf16(d_means[row * nm + i]);
  for (int i = threadIdx.x; i < ng; i += 256) out[nm + nw + i] = f32_to_bf16(d_logg[row * ng + i]);
  for (int i = nm + nw + ng + threadIdx.x; i < ldy; i += 256) out[i] = 0;
  if (threadIdx.x < lw) {
    const int q = threadIdx.x;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += d_logw[row * nw + k * lw + q];
    for (int k = 0; k < K; ++k) {
      const float p = expf(bf16_to_f32(logw_out[row * nw + k * lw + q]));
      out[nm + k * lw + q] = f32_to_bf16(d_logw[row * nw + k * lw + q] - p * s);
    }
  }
}

// [R, C] bf16 -> [C, R] bf16 (64x64 LDS tiles; R, C multiples of 8)
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ y,
                                                        int64_t ldy, int R, int C) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? x[(int64_t)(r0 + r) * ldx + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < C && r0 + r < R) y[(int64_t)(c0 + c) * ldy + r0 + r] = tile[r][c];
  }
}

// column sums of a bf16 [R, C] matrix into fp32 out[C] (accumulating): bias gradients
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int64_t ldx, float* __restrict__ out, int R,
                                                     int C, int rows_per_block) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(R, r0 + rows_per_block);
  float acc = 0.f;
  for (int r = r0; r < r1; ++r) acc += bf16_to_f32(x[(int64_t)r * ldx + c]);
  atomicAdd(out + c, acc);
}

// out[c] += sum_r a[r, c] * b[r, c]: gradient of an AdaLN gate, d_gate = sum_tokens dX_out * branch_output.
// One 16-byte column chunk per thread, a slab of rows per block.
__global__ __launch_bounds__(256) void coldot_kernel(const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ b, int64_t ldb,
                                                     float* __restrict__ out, int R, int C, int rows_per_block) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= (C >> 3)) return;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(R, r0 + rows_per_block);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = r0; r < r1; ++r) {
    float x[8], y[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(a + (int64_t)r * lda + c * 8), x);
    unpack8(*reinterpret_cast<const u32x4_t*>(b + (int64_t)r * ldb + c * 8), y);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += x[e] * y[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) atomicAdd(out + c * 8 + e, acc[e]);
}

// out[r, c] = res[r, c] + gate[c] * y[r, c]   (the gated residual, kept apart from the GEMM when y itself is needed later)
__global__ __launch_bounds__(256) void gate_residual_kernel(const bf16_t* __restrict__ y, int64_t ldy, const float* __restrict__ gate,
                                                            const bf16_t* __restrict__ res, int64_t ldr, bf16_t* __restrict__ out,
                                                            int64_t ldo, int64_t R, int C) {
  const int cpr = C >> 3;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= R * cpr) return;
  const int64_t r = g / cpr;
  const int c = (int)(g % cpr);
  float a[8], b[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(y + r * ldy + c * 8), a);
  unpack8(*reinterpret_cast<const u32x4_t*>(res + r * ldr + c * 8), b);
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = b[e] + gate[c * 8 + e] * a[e];
  *reinterpret_cast<u32x4_t*>(out + r * ldo + c * 8) = pack8(a);
}

// out[b, k] += sum_n x[b, n] * W[n, k]: the transposed weight-streaming product (gradient of silu(temb) through the stacked
// AdaLN modulation matrix W [n_mod, D]).  Every block owns a slab of rows n, every thread 8 consecutive columns k; W is read
// once, coalesced; B <= 4 partial rows live in registers and are added atomically.
__global__ __launch_bounds__(256) void gemv_t_kernel(const float* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ W, int64_t ldw,
                                                     float* __restrict__ out, int B, int64_t N, int K, int rows_per_block) {
  const int c = blockIdx.x * 256 + threadIdx.x;          // 16-byte column chunk
  if (c >= (K >> 3)) return;
  const int64_t n0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t n1 = n0 + rows_per_block < N ? n0 + rows_per_block : N;
  float acc[4][8];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[b][e] = 0.f;
  for (int64_t n = n0; n < n1; ++n) {
    float w[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(W + n * ldw + c * 8), w);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b < B) {
        const float xv = x[(int64_t)b * ldx + n];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[b][e] += xv * w[e];
      }
    }
  }
#pragma unroll
  for (int b = 0; b < 4; ++b)
    if (b < B)
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(out + (int64_t)b * K + c * 8 + e, acc[b][e]);
}

// AdaLayerNormContinuous backward w.r.t. its modulation:  xn = LN(x) (1 + scale) + shift
//   d_scale[b, c] += sum_rows dxn * LN(x),   d_shift[b, c] += sum_rows dxn     (rows of batch b)
// A wave computes the row statistics of its rows_per_wave rows and keeps its 8-column chunks' sums in registers; the per-wave [2][D]
// partials are folded into the [B, 2, D] fp32 result by normout_reduce_kernel (scale first, like the reference's chunk order).
__global__ __launch_bounds__(256) void normout_bwd_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                          const bf16_t* __restrict__ dxn, int64_t ldd, float* __restrict__ part,
                                                          int rows, int D, int rows_per_batch, int rows_per_wave) {
  const int lane = threadIdx.x & 63;
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int row0 = wv * rows_per_wave;
  const bool active = row0 < rows;
  const int nchunk = D >> 3;
  float ds[8][8], dh[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) ds[i][e] = dh[i][e] = 0.f;
  // the next row's x and dxn chunks are requested before the current row is reduced (two dependent wave reductions per row)
  const int row_end = active ? min(rows, row0 + rows_per_wave) : row0;
  u32x4_t xc[8], dc[8], xn[8], dn[8];
  auto fetch = [&](int row, u32x4_t (&xa)[8], u32x4_t (&da)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        xa[i] = *reinterpret_cast<const u32x4_t*>(x + (int64_t)row * ldx + c * 8);
        da[i] = *reinterpret_cast<const u32x4_t*>(dxn + (int64_t)row * ldd + c * 8);
      }
    }
  };
  if (row0 < row_end) fetch(row0, xc, dc);
  for (int row = row0; row < row_end; ++row) {
    if (row + 1 < row_end) fetch(row + 1, xn, dn);
    float v[8][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        unpack8(xc[i], v[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float qv = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[i][e] - mean;
          qv += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(qv) / (float)D + 1e-6f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        float gr[8];
        unpack8(dc[i], gr);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          ds[i][e] += gr[e] * (v[i][e] - mean) * rstd;
          dh[i][e] += gr[e];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { xc[i] = xn[i]; dc[i] = dn[i]; }
  }
  // every wave stores its [2][D] partial sums (plain, coalesced stores); normout_reduce_kernel folds the waves of a batch entry.  (The first
  // version combined the work-group's waves with LDS float atomics and added one atomic set per work-group to the result: 176 us for a
  // 4608 x 3072 call with 32 rows per wave, 60-86 us with 4-8 -- device-scope float atomics are performed at the memory side.)
  if (active) {
    float* dst = part + (int64_t)wv * 2 * D;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        *reinterpret_cast<f32x4_t*>(dst + c * 8) = (f32x4_t){ds[i][0], ds[i][1], ds[i][2], ds[i][3]};
        *reinterpret_cast<f32x4_t*>(dst + c * 8 + 4) = (f32x4_t){ds[i][4], ds[i][5], ds[i][6], ds[i][7]};
        *reinterpret_cast<f32x4_t*>(dst + D + c * 8) = (f32x4_t){dh[i][0], dh[i][1], dh[i][2], dh[i][3]};
        *reinterpret_cast<f32x4_t*>(dst + D + c * 8 + 4) = (f32x4_t){dh[i][4], dh[i][5], dh[i][6], dh[i][7]};
      }
    }
  }
}

// dmod[b][j] += sum over the waves w of batch entry b of part[w][j]; blockIdx.y splits a batch entry's waves, one atomic per (split, column)
__global__ __launch_bounds__(256) void normout_reduce_kernel(const float* __restrict__ part, float* __restrict__ d_scale, float* __restrict__ d_shift,
                                                             int64_t batch_stride, int waves_per_batch, int twoD, int splits) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= twoD) return;
  const int b = blockIdx.y / splits, sp = blockIdx.y % splits;
  const int per = (waves_per_batch + splits - 1) / splits;
  const int w0 = sp * per, w1 = min(waves_per_batch, w0 + per);
  const float* p = part + ((int64_t)b * waves_per_batch + w0) * twoD + j;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int w = w0;
  for (; w + 3 < w1; w += 4, p += 4 * (int64_t)twoD) {
    a0 += p[0]; a1 += p[twoD]; a2 += p[2 * (int64_t)twoD]; a3 += p[3 * (int64_t)twoD];
  }
  for (; w < w1; ++w, p += twoD) a0 += p[0];
  const int D_ = twoD >> 1;         // column j < D: d_scale, else d_shift (afx_normout_backward: one [B, 2, D] block; afx_normout_backward_split: two vectors)
  atomicAdd((j < D_ ? d_scale + j : d_shift + (j - D_)) + (int64_t)b * batch_stride, (a0 + a1) + (a2 + a3));
}

// dW[j, k] += sum_b dmod[b, j] * x[b, k]   (rank-B update of the norm_out.linear weight, B <= 8)
__global__ __launch_bounds__(256) void outer_accum_kernel(const float* __restrict__ dmod, const float* __restrict__ x,
                                                          float* __restrict__ dW, int B, int J, int Kd) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)J * Kd) return;
  const int j = (int)(i / Kd), k = (int)(i % Kd);
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc += dmod[(int64_t)b * J + j] * x[(int64_t)b * Kd + k];
  dW[i] += acc;
}

// x_out = x_a + u * (sb[b] - sa[b])     teacher Euler roll of the scheduled trajectory mixing (arcflow.py:189-192)
__global__ __launch_bounds__(256) void euler_kernel(const float* __restrict__ xa, const float* __restrict__ u,
                                                    const float* __restrict__ sa, const float* __restrict__ sb,
                                                    float* __restrict__ out, int64_t per_sample, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t b = i / per_sample;
  out[i] = xa[i] + u[i] * (sb[b] - sa[b]);
}

// out = alpha[b] * a + beta[b] * b   (per-sample blend: mean velocity (x_a - x_e)/(sigma_a - sigma_e), short/long select)
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ a, const float* __restrict__ alpha,
                                                    const float* __restrict__ bb, const float* __restrict__ beta,
                                                    float* __restrict__ out, int64_t per_sample, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t s = i / per_sample;
  out[i] = alpha[s] * a[i] + beta[s] * bb[i];
}

// classifier-free guidance of the teacher:  u = pos + (pos - neg) * (scale - 1)    (gaussian_flow.py:18-26, non-orthogonal)
__global__ __launch_bounds__(256) void cfg_kernel(const float* __restrict__ pos, const float* __restrict__ neg, float scale,
                                                  float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = pos[i] + (pos[i] - neg[i]) * (scale - 1.0f);
}

// LoRA input dropout (peft: y = W x + B A dropout(x); the engine folds B A into W, so what remains is the zero-mean
// correction B A (x . delta), delta = keep / (1 - p) - 1).  keep(row, col) is a counter-based hash of (seed, global row, col):
// the batched forward and the per-sample recompute / backward regenerate identical masks from the seed.
//   mode 0: dst = src . delta      mode 1: dst = src . (1 + delta) = dropout(src)      mode 2: dst += src . delta
//   mode 3: dst += src . (1 + delta)      (the input gradient of the LoRA branch: dx += ((dy B) A) . keep/(1-p))
AFX_DEV uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}

__global__ __launch_bounds__(256) void lora_dropout_kernel(const bf16_t* __restrict__ src, int64_t lds_, bf16_t* __restrict__ dst, int64_t ldd,
                                                           int64_t M, int N, int64_t row0, uint32_t thresh, float inv_keep, uint32_t seed,
                                                           int mode) {
  const int cpr = N >> 3;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= M * cpr) return;
  const int64_t m = g / cpr;
  const int c = (int)(g % cpr);
  float a[8], o[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(src + m * lds_ + c * 8), a);
  if (mode >= 2) unpack8(*reinterpret_cast<const u32x4_t*>(dst + m * ldd + c * 8), o);
  const uint32_t hr = mix32(seed ^ (uint32_t)((row0 + m) * 0x9e3779b1u));
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool keep = mix32(hr + (uint32_t)(c * 8 + e) * 0x85ebca77u) >= thresh;
    const float delta = keep ? inv_keep - 1.0f : -1.0f;
    if (mode == 0) o[e] = a[e] * delta;
    else if (mode == 1) o[e] = a[e] * (1.0f + delta);
    else if (mode == 2) o[e] += a[e] * delta;
    else o[e] += a[e] * (1.0f + delta);
  }
  *reinterpret_cast<u32x4_t*>(dst + m * ldd + c * 8) = pack8(o);
}

}  // namespace afx

using namespace afx;

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

extern "C" {

int afx_arcflow_backward(const float* g, const void* means, const void* logw, const void* logg, int32_t mix_dtype,
                         float sigma_src, float sigma_start, float sigma_end, const float* sigma_vec,
                         const float* gscale_vec, float gscale, float eps, float* d_means, float* d_logw,
                         float* d_logg, int32_t batch, int32_t n_tok, int32_t K, int32_t ch, int32_t pp,
                         int32_t velocity_only, int32_t accumulate, void* stream) {
  if (!g || !means || !logw || !logg || !d_means || !d_logw || !d_logg)
    return fail(AFX_E_INVALID, "null argument to afx_arcflow_backward");
  if (mix_dtype != AFX_DT_BF16 && mix_dtype != AFX_DT_F32) return fail(AFX_E_INVALID, "bad mix_dtype");
  if (batch < 0 || n_tok < 0 || K < 1 || K > 32 || ch < 1 || ch > 64 || pp < 1 || (pp & (pp - 1)) || 64 % pp || ch % pp)
    return fail(AFX_E_INVALID, "afx_arcflow_backward: need K<=32, ch<=64, pp a power of two dividing ch");
  HIP_TRY(launch_arcflow_bwd(g, means, logw, logg, mix_dtype == AFX_DT_BF16, sigma_src, sigma_start, sigma_end, sigma_vec,
                             gscale_vec, gscale, eps, d_means, d_logw, d_logg, batch, n_tok, K, ch, pp, velocity_only,
                             accumulate, (hipStream_t)stream));
  return AFX_OK;
}

int afx_mse_loss(const float* pred, const float* target, float coef, float* grad, float* loss_accum, int64_t n,
                 void* stream) {
  if (!pred || !target || !loss_accum || n < 0) return fail(AFX_E_INVALID, "bad argument to afx_mse_loss");
  if (n == 0) return AFX_OK;
  const unsigned nb = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, pred, target, coef, grad, loss_accum, n);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_sumsq(const float* x, float* out_accum, int64_t n, void* stream) {
  if (!x || !out_accum || n < 0) return fail(AFX_E_INVALID, "bad argument to afx_sumsq");
  if (n == 0) return AFX_OK;
  const unsigned nb = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, out_accum, n);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int32_t step, float grad_scale, int64_t n, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return fail(AFX_E_INVALID, "bad argument to afx_adamw_step");
  if (n == 0) return AFX_OK;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, n);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_adamw8bit_step(float* param, const float* grad, void* state1, void* state2, float* absmax1, float* absmax2,
                       const float* qmap1, const float* qmap2, float lr, float beta1, float beta2, float eps, float weight_decay,
                       int32_t step, float grad_scale, int64_t n, void* stream) {
  if (!param || !grad || !state1 || !state2 || !absmax1 || !absmax2 || !qmap1 || !qmap2 || n < 0 || step < 1)
    return fail(AFX_E_INVALID, "bad argument to afx_adamw8bit_step");
  if (n == 0) return AFX_OK;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw8bit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, (uint8_t*)state1,
                     (uint8_t*)state2, absmax1, absmax2, qmap1, qmap2, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, n);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_ema_lerp(float* ema, const float* net, float beta, int64_t n, void* stream) {
  if (!ema || !net || n < 0) return fail(AFX_E_INVALID, "bad argument to afx_ema_lerp");
  if (n == 0) return AFX_OK;
  hipLaunchKernelGGL(ema_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, ema, net, beta, n);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream) {
  if (!x || !y || n < 0) return fail(AFX_E_INVALID, "bad argument to afx_cast_f32_bf16");
  if (n == 0) return AFX_OK;
  hipLaunchKernelGGL(cast_kernel, dim3(blocks_for((n + 1) / 2)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_head_grad(const float* d_means, const float* d_logw, const float* d_logg, const void* logw_out, void* dy,
                  int64_t ldy, int64_t rows, int32_t K, int32_t ch, int32_t lw, void* stream) {
  if (!d_means || !d_logw || !d_logg || !logw_out || !dy || rows < 0 || lw < 1 || lw > 256 ||
      ldy < (int64_t)K * ch + K * lw + (K - 1) * lw)
    return fail(AFX_E_INVALID, "bad argument to afx_head_grad");
  if (rows == 0) return AFX_OK;
  hipLaunchKernelGGL(head_grad_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, d_means, d_logw, d_logg,
                     (const bf16_t*)logw_out, (bf16_t*)dy, ldy, K, ch, lw);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_transpose_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols, void* stream) {
  if (!x || !y || rows < 1 || cols < 1) return fail(AFX_E_INVALID, "bad argument to afx_transpose_bf16");
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, (bf16_t*)y, ldy, rows, cols);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_colsum_bf16(const void* x, int64_t ldx, float* out_accum, int32_t rows, int32_t cols, void* stream) {
  if (!x || !out_accum || rows < 1 || cols < 1) return fail(AFX_E_INVALID, "bad argument to afx_colsum_bf16");
  const int rpb = 256;
  hipLaunchKernelGGL(colsum_kernel, dim3((cols + 255) / 256, (rows + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, out_accum, rows, cols, rpb);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_coldot_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, float* out_accum, int32_t rows, int32_t cols, void* stream) {
  if (!a || !b || !out_accum || rows < 1 || cols < 8 || cols % 8 || lda % 8 || ldb % 8) return fail(AFX_E_INVALID, "bad argument to afx_coldot_bf16");
  const int rpb = 32;
  hipLaunchKernelGGL(coldot_kernel, dim3(((cols >> 3) + 255) / 256, (rows + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)a, lda, (const bf16_t*)b, ldb, out_accum, rows, cols, rpb);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_gate_residual_bf16(const void* y, int64_t ldy, const float* gate, const void* res, int64_t ldr, void* out, int64_t ldo,
                           int64_t rows, int32_t cols, void* stream) {
  if (!y || !gate || !res || !out || rows < 1 || cols % 8 || ldy % 8 || ldr % 8 || ldo % 8)
    return fail(AFX_E_INVALID, "bad argument to afx_gate_residual_bf16");
  hipLaunchKernelGGL(gate_residual_kernel, dim3(blocks_for(rows * (cols >> 3))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, ldy,
                     gate, (const bf16_t*)res, ldr, (bf16_t*)out, ldo, rows, cols);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_gemv_t_bf16(const float* x, int64_t ldx, const void* W, int64_t ldw, float* out_accum, int32_t B, int64_t N, int32_t K,
                    void* stream) {
  if (!x || !W || !out_accum || B < 1 || B > 4 || N < 1 || K % 8 || ldw % 8) return fail(AFX_E_INVALID, "afx_gemv_t_bf16: 1 <= B <= 4, K %% 8 == 0");
  const int rpb = 512;
  hipLaunchKernelGGL(gemv_t_kernel, dim3(((K >> 3) + 255) / 256, (unsigned)((N + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     (const bf16_t*)W, ldw, out_accum, B, N, K, rpb);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

static int normout_backward_impl(const void* x, int64_t ldx, const void* dxn, int64_t ldd, float* d_scale, float* d_shift, int64_t batch_stride, int32_t rows,
                                 int32_t D, int32_t rows_per_batch, void* stream);

int afx_normout_backward(const void* x, int64_t ldx, const void* dxn, int64_t ldd, float* dmod_accum, int32_t rows,
                         int32_t D, int32_t rows_per_batch, void* stream) {
  if (!dmod_accum) return fail(AFX_E_INVALID, "bad argument to afx_normout_backward");
  return normout_backward_impl(x, ldx, dxn, ldd, dmod_accum, dmod_accum + D, 2 * (int64_t)D, rows, D, rows_per_batch, stream);
}

int afx_normout_backward_split(const void* x, int64_t ldx, const void* dxn, int64_t ldd, float* d_scale_accum, float* d_shift_accum, int32_t rows,
                               int32_t D, void* stream) {
  if (!d_scale_accum || !d_shift_accum) return fail(AFX_E_INVALID, "bad argument to afx_normout_backward_split");
  return normout_backward_impl(x, ldx, dxn, ldd, d_scale_accum, d_shift_accum, 0, rows, D, rows, stream);
}

static int normout_backward_impl(const void* x, int64_t ldx, const void* dxn, int64_t ldd, float* d_scale, float* d_shift, int64_t batch_stride, int32_t rows,
                                 int32_t D, int32_t rows_per_batch, void* stream) {
  if (!x || !dxn || rows < 1 || D < 8 || D % 8 || D > 4096 || rows_per_batch < 1 || rows % rows_per_batch)
    return fail(AFX_E_INVALID, "bad argument to afx_normout_backward");
  // rows per wave: 8 (a 4608-row call = 576 waves on 256 CUs; the first version's 32 left it with 36 work-groups).  AFX_NORMOUT_RPW overrides.
  static int rpw_env = -1;
  if (rpw_env < 0) {
    const char* e = getenv("AFX_NORMOUT_RPW");
    rpw_env = e ? atoi(e) : 0;
  }
  int rpw = rpw_env > 0 ? rpw_env : 8;
  while (rows_per_batch % rpw) rpw >>= 1;
  const int waves = rows / rpw, wpb = rows_per_batch / rpw, nb = rows / rows_per_batch;
  hipStream_t st = (hipStream_t)stream;
  // [waves][2][D] fp32 scratch, kept per stream and grown on demand (hipMallocAsync / hipFreeAsync per call cost ~20 us of the 30 us a 512-row
  // call took); calls on one stream are ordered, so the buffer is free again when the next call's first kernel starts
  // keyed by (device, stream): the NULL stream and recycled stream handles exist on every device of a process, and a buffer of device 0
  // must never serve a call on device 1.  The buffers live until the library is unloaded (NormoutScratch's destructor).
  // Growing one calls hipFree / hipMalloc -- device-synchronising, illegal under stream capture -- so it only happens when a LARGER
  // shape shows up on that (device, stream); the split reduction below ends in float atomics over 8 splits: d_mod is reproducible to
  // the rounding of that sum's order, not bit for bit (documented in include/arcflow_hip.h).
  struct NormoutScratch {
    std::mutex mu;
    std::map<std::pair<int, hipStream_t>, std::pair<float*, size_t>> map;
    ~NormoutScratch() {
      for (auto& kv : map)
        if (kv.second.first != nullptr) (void)hipFree(kv.second.first);
    }
  };
  static NormoutScratch ws;
  int dev_id = 0;
  HIP_TRY(hipGetDevice(&dev_id));
  float* part = nullptr;
  {
    std::lock_guard<std::mutex> lk(ws.mu);
    auto& e = ws.map[std::make_pair(dev_id, st)];
    const size_t need = (size_t)waves * 2 * D * sizeof(float);
    if (e.second < need) {
      if (e.first != nullptr) HIP_TRY(hipFree(e.first));          // (synchronises: only when a larger shape shows up)
      e.first = nullptr; e.second = 0;
      HIP_TRY(hipMalloc((void**)&e.first, need));
      e.second = need;
    }
    part = e.first;
  }
  hipLaunchKernelGGL(normout_bwd_kernel, dim3((waves + 3) / 4), dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dxn, ldd, part, rows, D,
                     rows_per_batch, rpw);
  const int splits = wpb >= 64 ? 8 : 1;
  hipLaunchKernelGGL(normout_reduce_kernel, dim3((2 * D + 255) / 256, nb * splits), dim3(256), 0, st, part, d_scale, d_shift, batch_stride, wpb, 2 * D, splits);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_outer_accum(const float* dmod, const float* x, float* dW_accum, int32_t B, int32_t J, int32_t Kd, void* stream) {
  if (!dmod || !x || !dW_accum || B < 1 || J < 1 || Kd < 1) return fail(AFX_E_INVALID, "bad argument to afx_outer_accum");
  hipLaunchKernelGGL(outer_accum_kernel, dim3(blocks_for((int64_t)J * Kd)), dim3(256), 0, (hipStream_t)stream, dmod, x,
                     dW_accum, B, J, Kd);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_euler_roll(const float* x_a, const float* u, const float* sigma_a, const float* sigma_b, float* out,
                   int32_t batch, int64_t per_sample, void* stream) {
  if (!x_a || !u || !sigma_a || !sigma_b || !out || batch < 0 || per_sample < 1) return fail(AFX_E_INVALID, "bad argument to afx_euler_roll");
  const int64_t n = (int64_t)batch * per_sample;
  if (n == 0) return AFX_OK;
  hipLaunchKernelGGL(euler_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x_a, u, sigma_a, sigma_b, out,
                     per_sample, n);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_axpby_rows(const float* a, const float* alpha, const float* b, const float* beta, float* out, int32_t batch,
                   int64_t per_sample, void* stream) {
  if (!a || !alpha || !b || !beta || !out || batch < 0 || per_sample < 1) return fail(AFX_E_INVALID, "bad argument to afx_axpby_rows");
  const int64_t n = (int64_t)batch * per_sample;
  if (n == 0) return AFX_OK;
  hipLaunchKernelGGL(axpby_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, a, alpha, b, beta, out, per_sample, n);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_cfg_combine(const float* pos, const float* neg, float scale, float* out, int64_t n, void* stream) {
  if (!pos || !neg || !out || n < 0) return fail(AFX_E_INVALID, "bad argument to afx_cfg_combine");
  if (n == 0) return AFX_OK;
  hipLaunchKernelGGL(cfg_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, pos, neg, scale, out, n);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

}  // extern "C"

// ====================================================================================================
// Element-wise backward kernels of the MMDiT trunk (used by arcflow_amd/train/trunk.py)
namespace afx {

// dx = dres + LN-backward(dxn * (1 + scale[b]))   for xn = LN(x) (1 + scale) + shift, LN without affine, eps 1e-6
//   y = (x - mu) rstd ; g = dxn (1 + scale) ; dx = rstd (g - mean(g) - y mean(g y))
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                         const bf16_t* __restrict__ dxn, int64_t ldd,
                                                         const float* __restrict__ scale, int64_t ldmod, int rows_per_batch,
                                                         const bf16_t* __restrict__ dres, int64_t ldr,
                                                         bf16_t* __restrict__ dx, int64_t ldo, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = D >> 3;
  const float* sc = scale + (int64_t)(row / rows_per_batch) * ldmod;
  float v[8][8], gq[8][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      unpack8(*reinterpret_cast<const u32x4_t*>(x + (int64_t)row * ldx + c * 8), v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] -= mean;
        q += v[i][e] * v[i][e];
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-6f);
  float sg = 0.f, sgy = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      float d[8];
      unpack8(*reinterpret_cast<const u32x4_t*>(dxn + (int64_t)row * ldd + c * 8), d);
      const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(sc + c * 8);
      const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(sc + c * 8 + 4);
      const float sv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] *= rstd;                       // y
        gq[i][e] = d[e] * (1.0f + sv[e]);      // g
        sg += gq[i][e];
        sgy += gq[i][e] * v[i][e];
      }
    }
  }
  const float mg = wave_sum(sg) / (float)D, mgy = wave_sum(sgy) / (float)D;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      float r[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = rstd * (gq[i][e] - mg - v[i][e] * mgy);
      if (dres != nullptr) {
        float a[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(dres + (int64_t)row * ldr + c * 8), a);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] += a[e];
      }
      *reinterpret_cast<u32x4_t*>(dx + (int64_t)row * ldo + c * 8) = pack8(r);
    }
  }
}

// Out-of-place per-head RMSNorm + RoPE (forward, keeps the pre-norm projection for the backward) and its backward:
//   y = R(s) (x rstd w);   dz = R(s)^T dy;  xh = x rstd;  g = dz w;  dx = rstd (g - xh mean(g xh))
__global__ __launch_bounds__(256) void qk_norm_rope_oop_kernel(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ y,
                                                               int64_t ldy, const bf16_t* __restrict__ dy, int64_t lddy,
                                                               const float* __restrict__ w_txt, const float* __restrict__ w_img,
                                                               const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                               int S, int n_txt, int H, int64_t total, int backward) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= total) return;
  const int c = (int)(g & 15);
  const int64_t th = g >> 4;
  const int h = (int)(th % H);
  const int64_t row = th / H;
  const int s = (int)(row % S);
  float v[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(x + row * ldx + h * 128 + c * 8), v);
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float rstd = rsqrtf(ss * (1.0f / 128.0f) + 1e-6f);
  const float* w = (s < n_txt ? w_txt : w_img) + c * 8;
  const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(w);
  const f32x4_t w1 = *reinterpret_cast<const f32x4_t*>(w + 4);
  const float wv[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
  const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(cos_t + (int64_t)s * 64 + c * 4);
  const f32x4_t sn = *reinterpret_cast<const f32x4_t*>(sin_t + (int64_t)s * 64 + c * 4);
  float r[8];
  if (!backward) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = v[2 * i] * rstd * wv[2 * i];
      const float b = v[2 * i + 1] * rstd * wv[2 * i + 1];
      r[2 * i] = a * cs[i] - b * sn[i];
      r[2 * i + 1] = a * sn[i] + b * cs[i];
    }
  } else {
    float d[8], gg[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(dy + row * lddy + h * 128 + c * 8), d);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float ze = d[2 * i] * cs[i] + d[2 * i + 1] * sn[i];
      const float zo = -d[2 * i] * sn[i] + d[2 * i + 1] * cs[i];
      gg[2 * i] = ze * wv[2 * i];
      gg[2 * i + 1] = zo * wv[2 * i + 1];
      dot += gg[2 * i] * v[2 * i] * rstd + gg[2 * i + 1] * v[2 * i + 1] * rstd;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    dot *= (1.0f / 128.0f);
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = rstd * (gg[e] - v[e] * rstd * dot);
  }
  *reinterpret_cast<u32x4_t*>(y + row * ldy + h * 128 + c * 8) = pack8(r);
}

// h = gelu_tanh(pre)   /   dpre = dh * gelu_tanh'(pre)      (bf16, strided rows)
__global__ __launch_bounds__(256) void gelu_kernel(const bf16_t* __restrict__ pre, int64_t ldp, const bf16_t* __restrict__ dh,
                                                   int64_t ldh, bf16_t* __restrict__ out, int64_t ldo, int64_t rows, int cols) {
  const int cpr = cols >> 3;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= rows * cpr) return;
  const int64_t r = g / cpr;
  const int c = (int)(g % cpr);
  float x[8], o[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(pre + r * ldp + c * 8), x);
  if (dh == nullptr) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = gelu_tanh(x[e]);
  } else {
    float d[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(dh + r * ldh + c * 8), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float u = 0.7978845608028654f * (x[e] + 0.044715f * x[e] * x[e] * x[e]);
      const float t = 1.0f - 2.0f / (__expf(2.0f * u) + 1.0f);
      const float du = 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x[e] * x[e]);
      o[e] = d[e] * (0.5f * (1.0f + t) + 0.5f * x[e] * (1.0f - t * t) * du);
    }
  }
  *reinterpret_cast<u32x4_t*>(out + r * ldo + c * 8) = pack8(o);
}

// out[r, c] = (a[r, c] (+ b[r, c])) * gate[batch(r), c]     gate == nullptr -> 1       (bf16 rows; residual-grad plumbing)
__global__ __launch_bounds__(256) void addscale_kernel(const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ b,
                                                       int64_t ldb, const float* __restrict__ gate, int64_t ldg,
                                                       int rows_per_batch, bf16_t* __restrict__ out, int64_t ldo,
                                                       int64_t rows, int cols) {
  const int cpr = cols >> 3;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= rows * cpr) return;
  const int64_t r = g / cpr;
  const int c = (int)(g % cpr);
  float x[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(a + r * lda + c * 8), x);
  if (b != nullptr) {
    float y[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(b + r * ldb + c * 8), y);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] += y[e];
  }
  if (gate != nullptr) {
    const float* gp = gate + (r / rows_per_batch) * ldg + c * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] *= gp[e];
  }
  *reinterpret_cast<u32x4_t*>(out + r * ldo + c * 8) = pack8(x);
}

}  // namespace afx

extern "C" {

int afx_ln_modulate_backward(const void* x, int64_t ldx, const void* dxn, int64_t ldd, const float* scale, int64_t ldmod,
                             int32_t rows_per_batch, const void* dres, int64_t ldr, void* dx, int64_t ldo, int32_t rows,
                             int32_t D, void* stream) {
  if (!x || !dxn || !scale || !dx || rows < 0 || D < 8 || D % 8 || D > 4096 || rows_per_batch < 1)
    return fail(AFX_E_INVALID, "bad argument to afx_ln_modulate_backward");
  if (rows == 0) return AFX_OK;
  hipLaunchKernelGGL(ln_mod_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                     (const bf16_t*)dxn, ldd, scale, ldmod, rows_per_batch, (const bf16_t*)dres, ldr, (bf16_t*)dx, ldo, rows, D);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_qk_norm_rope_oop_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, const void* dy, int64_t lddy,
                              const float* w_txt, const float* w_img, const float* rope_cos, const float* rope_sin,
                              int32_t batch, int32_t S, int32_t n_txt, int32_t heads, int32_t backward, void* stream) {
  if (!x || !y || !w_txt || !w_img || !rope_cos || !rope_sin || (backward && !dy))
    return fail(AFX_E_INVALID, "null argument to afx_qk_norm_rope_oop_bf16");
  const int64_t total = (int64_t)batch * S * heads * 16;
  if (total <= 0) return fail(AFX_E_INVALID, "bad qk_norm_rope shape");
  hipLaunchKernelGGL(qk_norm_rope_oop_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, (bf16_t*)y, ldy, (const bf16_t*)dy, lddy, w_txt, w_img, rope_cos, rope_sin, S,
                     n_txt, heads, total, backward);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_gelu_bf16(const void* pre, int64_t ldp, const void* dh, int64_t ldh, void* out, int64_t ldo, int64_t rows,
                  int32_t cols, void* stream) {
  if (!pre || !out || rows < 0 || cols < 8 || cols % 8) return fail(AFX_E_INVALID, "bad argument to afx_gelu_bf16");
  const int64_t n = rows * (cols >> 3);
  if (n == 0) return AFX_OK;
  hipLaunchKernelGGL(gelu_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pre, ldp,
                     (const bf16_t*)dh, ldh, (bf16_t*)out, ldo, rows, cols);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_add_scale_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, const float* gate, int64_t ldg,
                       int32_t rows_per_batch, void* out, int64_t ldo, int64_t rows, int32_t cols, void* stream) {
  if (!a || !out || rows < 0 || cols < 8 || cols % 8 || (gate && rows_per_batch < 1))
    return fail(AFX_E_INVALID, "bad argument to afx_add_scale_bf16");
  const int64_t n = rows * (cols >> 3);
  if (n == 0) return AFX_OK;
  hipLaunchKernelGGL(addscale_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda,
                     (const bf16_t*)b, ldb, gate, ldg, rows_per_batch > 0 ? rows_per_batch : 1, (bf16_t*)out, ldo, rows, cols);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

int afx_lora_dropout_bf16(const void* src, int64_t lds_, void* dst, int64_t ldd, int64_t M, int32_t N, int64_t row0, float p,
                          uint32_t seed, int32_t mode, void* stream) {
  if (!src || !dst || M < 1 || N % 8 || lds_ % 8 || ldd % 8 || !(p >= 0.f && p < 1.f) || mode < 0 || mode > 3)
    return fail(AFX_E_INVALID, "bad argument to afx_lora_dropout_bf16");
  const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
  hipLaunchKernelGGL(lora_dropout_kernel, dim3((unsigned)((M * (N >> 3) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                     lds_, (bf16_t*)dst, ldd, M, N, row0, thresh, 1.0f / (1.0f - p), seed, mode);
  HIP_TRY(hipGetLastError());
  return AFX_OK;
}

}  // extern "C"
