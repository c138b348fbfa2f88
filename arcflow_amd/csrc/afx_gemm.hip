// Grouped bf16 GEMM for gfx950:  C = epi(A . W^T + bias),  A [M,K] and W [N,K] both K-contiguous
// (the nn.Linear layout), fp32 accumulation on v_mfma_f32_16x16x32_bf16.
//
// Structure (wave64, 8 waves = 2(M) x 4(N), 256x256x64 tile, one work-group per CU):
//   * HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane): no VGPR round trip.  The DMA
//     writes LDS lane-linearly, so the bank-conflict swizzle is applied to the per-lane SOURCE
//     address and again on the ds_read_b128 side (same involution): 16-byte chunk c of tile row r
//     lives at chunk c ^ ((r >> 1) & 7) of its 128-byte LDS row -> every 16-lane ds_read_b128
//     group hits 16 distinct 16-byte slots of the 256-byte bank row.
//   * two LDS stages (2 x 64 KiB): tile t+1 streams in while tile t feeds 64 MFMAs per wave.
//   * epilogue: accumulators are transposed through LDS (per-wave 64x64 fp32 patches) so that
//     bias / GELU / gate*x+residual run on row-contiguous data and C is stored 16 B per lane.
//   * several problems (image + text stream, or per-sample slices) share one launch; the 1-D
//     grid is remapped so every XCD owns a contiguous run of tiles (private-L2 reuse of A/W panels).
#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int GEMM_THREADS = 512;
constexpr int TILE_BYTES = BM * BK * 2;                 // 32 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;             // A + W
constexpr int EPI_LD = 68;                              // fp32 row stride of the epilogue patch
constexpr int EPI_WAVE_BYTES = 64 * EPI_LD * 4;         // 17408
constexpr int GEMM_LDS_BYTES = 8 * EPI_WAVE_BYTES;      // 139264 >= 2 * STAGE_BYTES (131072)
constexpr int GROUP_M = 6;                              // super-row height of the tile order

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// Stage one 256x64 bf16 tile (rows row0.., columns k0..k0+63) into LDS by DMA.
AFX_DEV void stage_tile(const bf16_t* __restrict__ base, int64_t ld, int row0, int nrows, int k0,
                        char* lds_tile, int tid, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = i * GEMM_THREADS + tid;       // 16-byte chunk index inside the tile
    const int r = p >> 3;                       // tile row
    const int c = (p & 7) ^ ((r >> 1) & 7);     // logical chunk stored at physical chunk p&7
    int gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;           // clamp: rows past the edge are never stored
    const bf16_t* src = base + (int64_t)gr * ld + k0 + c * 8;
    char* dst = lds_tile + (i * GEMM_THREADS + wave * 64) * 16;   // wave-uniform base
    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
  }
}

AFX_DEV bf16x8_t lds_frag(const char* tile, int row, int chunk) {
  const int phys = chunk ^ ((row >> 1) & 7);
  return *reinterpret_cast<const bf16x8_t*>(tile + row * 128 + phys * 16);
}

__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16_kernel(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- which tile of which problem --------------------------------------------------------
  int wg = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
    if (i < batch.nprob && wg >= batch.p[i].tile_start) pi = i;
  const GemmProblem& P = batch.p[pi];
  wg -= P.tile_start;
  const int per_group = GROUP_M * P.tiles_n;
  const int grp = wg / per_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(P.tiles_m - first_m, GROUP_M);
  const int in_grp = wg - grp * per_group;
  const int tm = first_m + in_grp % gsz;
  const int tn = in_grp / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  const bf16_t* __restrict__ A = P.A;
  const bf16_t* __restrict__ W = P.W;
  const int nk = P.K / BK;

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- main loop -------------------------------------------------------------------------
  stage_tile(A, P.lda, m0, P.M, 0, smem, tid, wave);
  stage_tile(W, P.ldw, n0, P.N, 0, smem + TILE_BYTES, tid, wave);
  __syncthreads();      // drains the DMA (vmcnt(0)) and releases the work-group

  const int frow = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    char* sa = smem + cur * STAGE_BYTES;
    char* sw = sa + TILE_BYTES;
    if (kt + 1 < nk) {
      char* na = smem + (cur ^ 1) * STAGE_BYTES;
      stage_tile(A, P.lda, m0, P.M, (kt + 1) * BK, na, tid, wave);
      stage_tile(W, P.ldw, n0, P.N, (kt + 1) * BK, na + TILE_BYTES, tid, wave);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t bfr[4], afr[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = lds_frag(sw, wn * 64 + j * 16 + frow, kk * 4 + fq);
#pragma unroll
      for (int i = 0; i < 8; ++i) afr[i] = lds_frag(sa, wm * 128 + i * 16 + frow, kk * 4 + fq);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();    // next stage landed (vmcnt(0)) and every wave is done with this one
  }

  // ---- epilogue: transpose through LDS, fused bias / activation / gated residual ------------
  float* patch = reinterpret_cast<float*>(smem + wave * EPI_WAVE_BYTES);
  const int er = lane >> 3;          // row within an 8-row pass
  const int ec = (lane & 7) * 8;     // first of 8 consecutive columns
  const int gcol = n0 + wn * 64 + ec;
  const bool col_ok = gcol < P.N;

  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = 0.f;
  if (P.bias != nullptr && col_ok) {
    const u32x4_t bw = *reinterpret_cast<const u32x4_t*>(P.bias + gcol);
    unpack8(bw, bias);
  }

#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          patch[(i * 16 + fq * 4 + r) * EPI_LD + j * 16 + frow] = acc[h * 4 + i][j][r];
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int lr = ps * 8 + er;
      const int grow = m0 + wm * 128 + h * 64 + lr;
      if (grow < P.M && col_ok) {
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(patch + lr * EPI_LD + ec);
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(patch + lr * EPI_LD + ec + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bias[e];
        if (P.epi == EPI_GELU) {
          if (gcol >= P.gelu_col0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
          }
        } else if (P.epi == EPI_GATE_RES) {
          const float* gp = P.gate + (int64_t)(grow / P.rows_per_batch) * P.ldg + gcol;
          const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(gp);
          const f32x4_t g1 = *reinterpret_cast<const f32x4_t*>(gp + 4);
          const u32x4_t rw = *reinterpret_cast<const u32x4_t*>(P.res + (int64_t)grow * P.ldr + gcol);
          float rr[8];
          unpack8(rw, rr);
          const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = rr[e] + g[e] * v[e];
        }
        *reinterpret_cast<u32x4_t*>(P.C + (int64_t)grow * P.ldc + gcol) = pack8(v);
      }
    }
  }
}

hipError_t launch_gemm(GemmBatch& batch, hipStream_t stream) {
  int total = 0;
  for (int i = 0; i < batch.nprob; ++i) {
    GemmProblem& p = batch.p[i];
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.tile_start = total;
    total += p.tiles_m * p.tiles_n;
  }
  batch.total_tiles = total;
  if (total == 0) return hipSuccess;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_bf16_kernel, dim3(total), dim3(GEMM_THREADS), GEMM_LDS_BYTES, stream, batch);
  return hipGetLastError();
}

}  // namespace afx
