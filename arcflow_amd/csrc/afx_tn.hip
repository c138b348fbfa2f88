// TN product for the LoRA weight gradients (round 5):  C[N1, N2] (+)= X^T Y  with X [M, N1] and Y [M, N2] TOKEN-major bf16 -- the contraction runs
// over the tokens, the operands' slow index.  Reference: peft's lora_A / lora_B gradients of the adapted linears (arcflux.py:294-302):
//     dB [out, r] += dy^T t          dA [r, in] += dT^T dropout(x)
// Until round 4 both operands were first transposed ([M, N] -> [N, M]: two passes over dy, t, dT, x~ per adapted linear, 10.9 k launches per two
// iterations) so that the K-contiguous NT kernel could take them.  Here the token-major tiles go to LDS as they lie in memory (LDS-DMA, 256-byte rows) and
// the MFMA fragments -- 8 consecutive tokens of one column per lane -- are gathered by gfx950's LDS transpose read:
//     ds_read_b64_tr_b16: a 16-lane group reads a [4 rows][16 columns] block; lane i SUPPLIES the address of row i / 4, columns 4 (i % 4) .. + 3 (8 bytes)
//     and RECEIVES column i of the block, rows 0 .. 3 (tools/tr_probe.hip)
// Two reads (rows 8 g + 0..3 and 8 g + 4..7 for lane group g) are one 16x16x32 fragment.  The product is issued as C^T = Y^T X (A = the Y fragment, B = the
// X fragment): a lane then holds 4 consecutive n2 of one n1 row -- 16-byte fp32 accesses to C.
// Tile 128 (n1) x 128 (n2) x 64 tokens, 4 waves (2 x 2) of 64 x 64, two LDS stages of 16 KB X + 16 KB Y (64 KB: two work-groups per CU).  16-byte chunk c of
// token row r sits at physical chunk c ^ f(r), f(r) = 2 (r % 4) + 8 ((r / 8) % 2): the 8 rows x 2 chunks a half-wave's transpose read touches are 16 distinct
// chunks = all 64 banks once.  Deterministic: a tile is owned by one work-group, the token loop runs in order (no split over tokens, no atomics).
#include <algorithm>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {
namespace tn {

constexpr int BM = 128, BN = 128, BK = 64, THREADS = 256;
constexpr int TILE_BYTES = BK * 256;                  // one operand tile: 64 token rows x 128 columns x 2 bytes
constexpr int LDS_BYTES = 4 * TILE_BYTES;             // (X, Y) x 2 stages

AFX_DEV int swz(int row) { return 2 * (row & 3) + 8 * ((row >> 3) & 1); }

// Token split (round 6): a [3072, 256] gradient is 48 tiles for 256 CUs x 2 resident work-groups, and the launch takes as long as ONE work-group's walk over
// all 4608 tokens.  With ksplit > 1 the grid is tiles x ksplit: work-group (tile, s) contracts K-steps [s per, (s + 1) per) and stores its fp32 partial tile
// into slab s of a workspace ([ksplit][N1][N2], no accumulate); tn_reduce_kernel then adds the slabs IN ORDER s = 0, 1, ... into C (deterministic: the
// summation tree is fixed by the shape, no atomics).
__global__ __launch_bounds__(THREADS, 2) void gemm_tn_f32_kernel(const bf16_t* __restrict__ X, int64_t ldx, const bf16_t* __restrict__ Y, int64_t ldy,
                                                                 float* __restrict__ C, int64_t ldc, int M, int N1, int N2, int tiles2, int accumulate,
                                                                 int ksplit, int per, int64_t slab_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;             // this wave's 64 n1 columns (wr) x 64 n2 columns (wc)
  const int tile = blockIdx.x / ksplit, split = blockIdx.x - tile * ksplit;
  const int n1_0 = (tile / tiles2) * BM, n2_0 = (tile % tiles2) * BN;
  const int nk_all = (M + BK - 1) / BK;
  const int t_lo = split * per, nk = min(nk_all, t_lo + per);      // this work-group's K-steps [t_lo, nk)
  C += (int64_t)split * slab_stride;                                // (slab_stride = 0 without a split)

  // ---- LDS-DMA sources: instruction j of a wave moves rows 16 wave + 4 j .. + 3 (1 KB); lane l -> row + l / 16, physical chunk l % 16 = logical chunk (l % 16) ^ f(row)
  uint32_t xoff[4], yoff[4];        // byte offsets of this lane's 16 bytes relative to token row k0 (rows clamped per K-step through the base pointer: see stage())
  int drow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 16 * wave + 4 * j + (lane >> 4);
    const int lc = (lane & 15) ^ swz(row);
    drow[j] = row;
    // columns past N1 / N2: clamp the chunk so that the access stays inside the row (those output columns are never stored)
    const int c1 = min(n1_0 + lc * 8, max(N1 - 8, 0)), c2 = min(n2_0 + lc * 8, max(N2 - 8, 0));
    xoff[j] = (uint32_t)(c1 * 2);
    yoff[j] = (uint32_t)(c2 * 2);
  }
  auto stage = [&](int t) {
    char* xb = smem + (t & 1) * 2 * TILE_BYTES;
    char* yb = xb + TILE_BYTES;
    const int k0 = t * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = min(k0 + drow[j], M - 1);          // (token rows past M: a clamped copy, zeroed in LDS before use)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(X) + (int64_t)r * ldx * 2 + xoff[j]),
                                       (__attribute__((address_space(3))) void*)(xb + (16 * wave + 4 * j) * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(Y) + (int64_t)r * ldy * 2 + yoff[j]),
                                       (__attribute__((address_space(3))) void*)(yb + (16 * wave + 4 * j) * 256), 16, 0, 0);
    }
  };

  // ---- fragment addresses of the transpose reads: lane (i = lane % 16, g = lane / 16), read h: row 8 g + 4 h + i / 4 (+ 32 per MFMA k-step), column
  // tile 16 q + 4 (i % 4) of the wave's 64 columns -> logical chunk 8 w + 2 q + (i % 4) / 2, byte (i % 4) % 2 * 8
  const int fi = lane & 15, fg = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  auto frag_addr = [&](int buf_off, int ks, int h, int wcol, int q) -> uint32_t {
    const int row = 32 * ks + 8 * fg + 4 * h + (fi >> 2);
    const int lc = 8 * wcol + 2 * q + ((fi & 3) >> 1);
    return lds0 + buf_off + row * 256 + ((lc ^ swz(row)) << 4) + ((fi & 1) << 3);
  };

  f32x4_t acc[4][4];       // [n1 tile][n2 tile] of C^T blocks: lane holds C[n1 = 16 p + lane % 16][n2 = 16 q + 4 (lane / 16) + 0..3]
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[p][q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  stage(t_lo);
#pragma unroll 1
  for (int t = t_lo; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile t has landed (this wave's pieces; the barrier makes it everybody's)
    __syncthreads();                                       // ... and every wave is done reading the other stage
    if (t + 1 < nk) stage(t + 1);
    const int boff = (t & 1) * 2 * TILE_BYTES;
    const int valid = M - t * BK;
    if (valid < BK) {                                      // (uniform) last, ragged K-step: token rows past M are zeros
      for (int idx = tid; idx < 2 * BK * 16; idx += THREADS) {
        const int row = (idx >> 4) & (BK - 1);
        if (row >= valid) *reinterpret_cast<u32x4_t*>(smem + boff + (idx >> 10) * TILE_BYTES + row * 256 + (idx & 15) * 16) = (u32x4_t){0u, 0u, 0u, 0u};
      }
      __syncthreads();
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x2_t xf[4][2], yf[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xf[q][h]) : "v"(frag_addr(boff, ks, h, wr, q)) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(yf[q][h]) : "v"(frag_addr(boff + TILE_BYTES, ks, h, wc, q)) : "memory");
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const bf16x8_t b = __builtin_bit_cast(bf16x8_t, (u32x4_t){xf[p][0][0], xf[p][0][1], xf[p][1][0], xf[p][1][1]});     // B operand: X, column n1 = lane % 16
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bf16x8_t a = __builtin_bit_cast(bf16x8_t, (u32x4_t){yf[q][0][0], yf[q][0][1], yf[q][1][0], yf[q][1][1]});   // A operand: Y^T, row n2 = lane % 16
          acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[p][q], 0, 0, 0);
        }
      }
    }
  }
  // ---- store: D = C^T block: lane holds D[n2 = 4 (lane / 16) + e][n1 = lane % 16] = C[n1][n2 + e]: four consecutive n2 of one n1 row
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int n1 = n1_0 + 64 * wr + 16 * p + fi;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n2 = n2_0 + 64 * wc + 16 * q + 4 * fg;
      if (n1 < N1 && n2 < N2) {
        float* dst = C + (int64_t)n1 * ldc + n2;
        if (n2 + 3 < N2) {
          f32x4_t v = acc[p][q];
          if (accumulate) v += *reinterpret_cast<const f32x4_t*>(dst);
          *reinterpret_cast<f32x4_t*>(dst) = v;
        } else {
          for (int e = 0; e < 4 && n2 + e < N2; ++e) dst[e] = acc[p][q][e] + (accumulate ? dst[e] : 0.f);
        }
      }
    }
  }
}

// C[n1][n2] (+)= slab[0][n1][n2] + slab[1][n1][n2] + ... in this order; 4 floats per thread
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ slabs, int64_t slab_stride, int ksplit, float* __restrict__ C, int64_t ldc, int N1,
                                                        int N2, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int q = N2 >> 2;
  if (idx >= (int64_t)N1 * q) return;
  const int n1 = (int)(idx / q), n2 = (int)(idx - (int64_t)n1 * q) * 4;
  const float* src = slabs + (int64_t)n1 * N2 + n2;
  f32x4_t v = *reinterpret_cast<const f32x4_t*>(src);
  for (int s_ = 1; s_ < ksplit; ++s_) v += *reinterpret_cast<const f32x4_t*>(src + (int64_t)s_ * slab_stride);
  float* dst = C + (int64_t)n1 * ldc + n2;
  if (accumulate) v += *reinterpret_cast<const f32x4_t*>(dst);
  *reinterpret_cast<f32x4_t*>(dst) = v;
}

}  // namespace tn

// How many ways the token loop of a [N1, N2] product over M tokens is cut (1: no split, no workspace): fill the 2 x CUs work-group slots, at least 4 K-steps each.
int gemm_tn_ksplit(int M, int N1, int N2) {
  static const int forced = [] { const char* e = getenv("AFX_TN_SPLIT"); return e ? atoi(e) : -1; }();      // 0 / 1: never split (A/B), n: force n ways
  const int tiles = ((N1 + tn::BM - 1) / tn::BM) * ((N2 + tn::BN - 1) / tn::BN), nk = (M + tn::BK - 1) / tn::BK;
  if (forced == 0 || forced == 1 || N2 % 4 || tiles <= 0 || nk < 8) return 1;
  int want = forced > 1 ? forced : 512 / tiles;
  want = std::min(want, nk / 4);
  if (want < 2) return 1;
  const int per = (nk + want - 1) / want;
  return (nk + per - 1) / per;
}
int64_t gemm_tn_ws_bytes(int M, int N1, int N2) {
  const int ks = gemm_tn_ksplit(M, N1, N2);
  return ks > 1 ? (int64_t)ks * N1 * N2 * 4 : 0;
}

hipError_t launch_gemm_tn_f32(const uint16_t* X, int64_t ldx, const uint16_t* Y, int64_t ldy, float* C, int64_t ldc, int M, int N1, int N2, int accumulate,
                              hipStream_t stream, float* ws) {
  if (M <= 0 || N1 <= 0 || N2 <= 0) return hipSuccess;
  if (N1 % 8 || N2 % 8 || ldx % 8 || ldy % 8 || ldc % 4 || (reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(Y) & 15) ||
      (reinterpret_cast<uintptr_t>(C) & 15))
    return hipErrorInvalidValue;
  static bool attr = false;
  if (!attr) {
    hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(tn::gemm_tn_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tn::LDS_BYTES);
    if (r != hipSuccess) return r;
    attr = true;
  }
  const int tiles1 = (N1 + tn::BM - 1) / tn::BM, tiles2 = (N2 + tn::BN - 1) / tn::BN, nk = (M + tn::BK - 1) / tn::BK;
  const int ks = ws != nullptr ? gemm_tn_ksplit(M, N1, N2) : 1;
  if (ks <= 1) {
    hipLaunchKernelGGL(tn::gemm_tn_f32_kernel, dim3(tiles1 * tiles2), dim3(tn::THREADS), tn::LDS_BYTES, stream, X, ldx, Y, ldy, C, ldc, M, N1, N2, tiles2, accumulate, 1,
                       nk, (int64_t)0);
    return hipGetLastError();
  }
  if (reinterpret_cast<uintptr_t>(ws) & 15) return hipErrorInvalidValue;
  const int per = (nk + ks - 1) / ks;
  const int64_t slab = (int64_t)N1 * N2;
  hipLaunchKernelGGL(tn::gemm_tn_f32_kernel, dim3(tiles1 * tiles2 * ks), dim3(tn::THREADS), tn::LDS_BYTES, stream, X, ldx, Y, ldy, ws, (int64_t)N2, M, N1, N2, tiles2, 0, ks,
                     per, slab);
  hipLaunchKernelGGL(tn::tn_reduce_kernel, dim3((unsigned)((slab / 4 + 255) / 256)), dim3(256), 0, stream, ws, slab, ks, C, ldc, N1, N2, accumulate);
  return hipGetLastError();
}

}  // namespace afx
