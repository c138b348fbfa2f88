// Flash-attention backward for gfx950 (head_dim 128, bf16 in/out, no mask), the gradient twin of afx_attn.hip.
//
// Given Q, K, V, O, dO and the forward's log2-domain row statistic L (P = exp2(s c - L)):
//     delta_q = sum_d dO[q,d] O[q,d]
//     dV = P^T dO        dP = dO V^T        dS = P o (dP - delta) / sqrt(d)        dQ = dS K        dK = dS^T Q
// Two kernels, no atomics, both on v_mfma_f32_32x32x16_bf16 with the same "reduction index lives in the
// registers of ONE lane" trick as the forward:
//   * dQ kernel   (query-stationary, lane = query):  S^T = K Q^T, dP^T = V dO^T  -> dS^T is already the B operand
//                 of dQ^T += K^T dS^T  (K^T tile pre-transposed with the 16-group key permutation of afx_attn.hip)
//   * dKdV kernel (key-stationary,   lane = key):    S = Q K^T,  dP = dO V^T   -> P and dS are already the B
//                 operands of dV^T += dO^T P and dK^T += Q^T dS (Q^T / dO^T tiles pre-transposed, queries permuted)
// S and dP are recomputed in both kernels (7 matmuls instead of 5) in exchange for determinism and zero
// inter-work-group traffic.  Tiles stream HBM -> LDS by LDS-DMA into 2-stage XOR-swizzled rings.
#include <cstdlib>

#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {

namespace {
constexpr int HD = 128;
constexpr int TB = 64;                    // tile of the streamed index (keys in dQ, queries in dKdV)
constexpr int ROWMAJ_BYTES = TB * HD * 2; // [64][128] bf16 tile, 256-byte rows, chunk ^ (row & 15) swizzle
constexpr int TRANS_BYTES = HD * TB * 2;  // [128][64] bf16 tile, 128-byte rows, chunk ^ ((row >> 1) & 7) swizzle
constexpr float SCALE = 0.08838834764831845f;                 // 1/sqrt(128)
constexpr float C_LOG2 = 0.08838834764831845f * 1.4426950408889634f;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// one 1 KiB piece of a row-major [64][128] tile: chunk index p -> (row, physical chunk), source swizzled
template <int nthreads>
AFX_DEV void dma_rowmajor(const bf16_t* base, int64_t ld, int row0, int nrows, char* dst, int tid, int wave_u) {
  constexpr int pieces = (TB * 16) / nthreads;
#pragma unroll
  for (int i = 0; i < pieces; ++i) {
    int seed = i * nthreads + tid;
    const int r = seed >> 4, cp = seed & 15;
    const int gr = min(row0 + r, nrows - 1);
    const bf16_t* src = base + (int64_t)gr * ld + ((cp ^ (r & 15)) << 3);
    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dst + (i * nthreads + wave_u * 64) * 16), 16, 0, 0);
  }
}
// one piece of a transposed [128][64] tile out of X^T[128][S_pad]
template <int nthreads>
AFX_DEV void dma_trans(const bf16_t* base, int64_t S_pad, int col0, char* dst, int tid, int wave_u) {
  constexpr int pieces = (HD * 8) / nthreads;
#pragma unroll
  for (int i = 0; i < pieces; ++i) {
    int seed = i * nthreads + tid;
    const int d = seed >> 3, vp = seed & 7;
    const bf16_t* src = base + (int64_t)d * S_pad + col0 + ((vp ^ ((d >> 1) & 7)) << 3);
    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dst + (i * nthreads + wave_u * 64) * 16), 16, 0, 0);
  }
}
AFX_DEV bf16x8_t frag_rowmaj(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8_t*>(tile + row * 256 + ((chunk ^ (row & 15)) << 4));
}
AFX_DEV bf16x8_t frag_trans(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8_t*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
}  // namespace

// delta[b,h,q] = sum_d dO O   (one 16-lane group per (row, head); pad rows get 0)
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, int64_t ldo,
                                                         const bf16_t* __restrict__ dout, int64_t lddo,
                                                         float* __restrict__ delta, int H, int S, int S_pad, int64_t total) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= total) return;
  const int c = (int)(g & 15);
  const int64_t th = g >> 4;
  const int h = (int)(th % H);
  const int64_t row = th / H;               // b * S + s
  float a[8], b[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(o + row * ldo + h * HD + c * 8), a);
  unpack8(*reinterpret_cast<const u32x4_t*>(dout + row * lddo + h * HD + c * 8), b);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += a[e] * b[e];
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (c == 0) {
    const int64_t bb = row / S, ss = row % S;
    delta[(bb * H + h) * S_pad + ss] = s;
  }
}

// ------------------------------------------------------------------------------------------------ dQ
constexpr int DQ_WAVES = 8;
constexpr int DQ_THREADS = DQ_WAVES * 64;
constexpr int DQ_STAGE = 2 * ROWMAJ_BYTES + TRANS_BYTES;      // K | V | K^T = 48 KiB

__global__ __launch_bounds__(DQ_THREADS, 2) void attn_bwd_dq_kernel(
    const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k, int64_t ldk,
    const bf16_t* __restrict__ v, int64_t ldv, const bf16_t* __restrict__ kt, const bf16_t* __restrict__ dout,
    int64_t lddo, const float* __restrict__ lse, const float* __restrict__ delta, bf16_t* __restrict__ dq,
    int64_t lddq, int H, int S, int S_pad, int nq, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int ql = lane & 31, hi = lane >> 5;
  const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
  const int per_head = nq * B;
  const int h = xcd + 8 * (slot_id / per_head);
  if (h >= H) return;
  const int rem = slot_id % per_head;
  const int b = rem / nq;
  const int q0 = (rem % nq) * (DQ_WAVES * 32) + wave * 32;
  const bool q_ok = q0 + ql < S;
  const int qrow = min(q0 + ql, S - 1);
  const int ntiles = S_pad / TB;

  const bf16_t* kbase = k + (int64_t)b * S * ldk + h * HD;
  const bf16_t* vbase = v + (int64_t)b * S * ldv + h * HD;
  const bf16_t* ktbase = kt + ((int64_t)(b * H + h) * HD) * S_pad;
  const bf16_t* qp = q + ((int64_t)b * S + qrow) * ldq + h * HD;
  const bf16_t* dop = dout + ((int64_t)b * S + qrow) * lddo + h * HD;
  bf16x8_t qf[8], dof[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    qf[s] = *reinterpret_cast<const bf16x8_t*>(qp + s * 16 + hi * 8);
    dof[s] = *reinterpret_cast<const bf16x8_t*>(dop + s * 16 + hi * 8);
  }
  const int64_t stat = ((int64_t)b * H + h) * S_pad + qrow;
  const float Lq = q_ok ? lse[stat] : INFINITY;          // rows past S: P = 0
  const float Dq = q_ok ? delta[stat] : 0.f;

  auto stage = [&](int t, int buf) {
    char* kd = smem + buf * DQ_STAGE;
    dma_rowmajor<DQ_THREADS>(kbase, ldk, t * TB, S, kd, tid, wave_u);
    dma_rowmajor<DQ_THREADS>(vbase, ldv, t * TB, S, kd + ROWMAJ_BYTES, tid, wave_u);
    dma_trans<DQ_THREADS>(ktbase, S_pad, t * TB, kd + 2 * ROWMAJ_BYTES, tid, wave_u);
  };
  f32x16_t acc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  stage(0, 0);
  AFX_SYNC_DMA();
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    asm volatile("" : "+v"(qf[s]));
    asm volatile("" : "+v"(dof[s]));
  }
  // (Measured in round 4, same box, S = 4608 / H = 24: this compiler-scheduled form with two waves per SIMD 434-450 us; fragments requested
  // as one batch per 16-MFMA group behind sched_barrier fences 531 us; one wave per SIMD with the dK / dV kernel's five regions 628 us.
  // With two waves the partner's MFMAs already cover the ds_read -> mfma latency and the fences only remove overlap.)
  for (int t = 0; t < ntiles; ++t) {
    const char* ks = smem + (t & 1) * DQ_STAGE;
    const char* vs = ks + ROWMAJ_BYTES;
    const char* kts = ks + 2 * ROWMAJ_BYTES;
    if (t + 1 < ntiles) stage(t + 1, (t + 1) & 1);
    bf16x8_t dsf[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16_t sa, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] = dp[r] = 0.f;
      const int krow = kb * 32 + ql;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rowmaj(ks, krow, s * 2 + hi), qf[s], sa, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rowmaj(vs, krow, s * 2 + hi), dof[s], dp, 0, 0, 0);
      }
      float dsv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // no key mask: rows past S of the K / V tiles are clamped copies of the last key (finite scores, p <= 1) and the matching columns
        // of the zero-padded K^T tile are zeros, so their dS never reaches dQ.  1 / sqrt(d) is applied once, in the epilogue.
        const float p = __builtin_amdgcn_exp2f(sa[r] * C_LOG2 - Lq);
        dsv[r] = p * (dp[r] - Dq);
      }
#pragma unroll
      for (int ksub = 0; ksub < 2; ++ksub) {
        float tmp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tmp[j] = dsv[ksub * 8 + j];
        const u32x4_t w = pack8(tmp);
        dsf[kb * 2 + ksub] = __builtin_bit_cast(bf16x8_t, w);
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trans(kts, d * 32 + ql, g * 2 + hi), dsf[g], acc[d], 0, 0, 0);
    AFX_SYNC_DMA();
  }
  if (q_ok) {
    bf16_t* op = dq + ((int64_t)b * S + q0 + ql) * lddq + h * HD;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t w;
        w[0] = pack_bf16x2(acc[d][4 * g + 0] * SCALE, acc[d][4 * g + 1] * SCALE);
        w[1] = pack_bf16x2(acc[d][4 * g + 2] * SCALE, acc[d][4 * g + 3] * SCALE);
        *reinterpret_cast<u32x2_t*>(op + d * 32 + g * 8 + hi * 4) = w;
      }
  }
}

// ------------------------------------------------------------------------------------------------ dK, dV
#ifdef AFX_BWD_TRACE        // tools/attn_bwd_trace.py: cycle stamps of one query tile (t = 36) of two work-groups, per wave
__device__ unsigned g_bwd_trace[2 * 4 * 12];
#define BWD_TR(i) if (t == 36) tr[i] = (unsigned)__builtin_readcyclecounter();
#else
#define BWD_TR(i)
#endif
constexpr int DKV_WAVES = 4;
constexpr int DKV_THREADS = DKV_WAVES * 64;
constexpr int DKV_STAT = 2 * TB * 4;                                       // L | delta of the 64 queries
constexpr int DKV_STAGE = 2 * ROWMAJ_BYTES + 2 * TRANS_BYTES + DKV_STAT;   // Q | dO | Q^T | dO^T | stats

__global__ __launch_bounds__(DKV_THREADS, 1) void attn_bwd_dkv_kernel(
    const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k, int64_t ldk,
    const bf16_t* __restrict__ v, int64_t ldv, const bf16_t* __restrict__ qt, const bf16_t* __restrict__ dout,
    int64_t lddo, const bf16_t* __restrict__ dot, const float* __restrict__ lse, const float* __restrict__ delta,
    bf16_t* __restrict__ dk, int64_t lddk, bf16_t* __restrict__ dv, int64_t lddv, int H, int S, int S_pad, int nk,
    int B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int kl = lane & 31, hi = lane >> 5;
  const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
  const int per_head = nk * B;
  const int h = xcd + 8 * (slot_id / per_head);
  if (h >= H) return;
  const int rem = slot_id % per_head;
  const int b = rem / nk;
  const int k0 = (rem % nk) * (DKV_WAVES * 32) + wave * 32;
  const bool k_ok = k0 + kl < S;
  const int krow = min(k0 + kl, S - 1);
  const int ntiles = S_pad / TB;

  const bf16_t* qbase = q + (int64_t)b * S * ldq + h * HD;
  const bf16_t* dobase = dout + (int64_t)b * S * lddo + h * HD;
  const bf16_t* qtbase = qt + ((int64_t)(b * H + h) * HD) * S_pad;
  const bf16_t* dotbase = dot + ((int64_t)(b * H + h) * HD) * S_pad;
  const float* lbase = lse + ((int64_t)b * H + h) * S_pad;
  const float* dbase = delta + ((int64_t)b * H + h) * S_pad;
  const bf16_t* kp = k + ((int64_t)b * S + krow) * ldk + h * HD;
  const bf16_t* vp = v + ((int64_t)b * S + krow) * ldv + h * HD;
  bf16x8_t kf[8], vf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    kf[s] = *reinterpret_cast<const bf16x8_t*>(kp + s * 16 + hi * 8);
    vf[s] = *reinterpret_cast<const bf16x8_t*>(vp + s * 16 + hi * 8);
  }

  auto stage = [&](int t, int buf) {
    char* base = smem + buf * DKV_STAGE;
    dma_rowmajor<DKV_THREADS>(qbase, ldq, t * TB, S, base, tid, wave_u);
    dma_rowmajor<DKV_THREADS>(dobase, lddo, t * TB, S, base + ROWMAJ_BYTES, tid, wave_u);
    dma_trans<DKV_THREADS>(qtbase, S_pad, t * TB, base + 2 * ROWMAJ_BYTES, tid, wave_u);
    dma_trans<DKV_THREADS>(dotbase, S_pad, t * TB, base + 2 * ROWMAJ_BYTES + TRANS_BYTES, tid, wave_u);
    char* st = base + 2 * ROWMAJ_BYTES + 2 * TRANS_BYTES;
    if (wave_u == 0)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(lbase + t * TB + lane), (lds_void_t*)st, 4, 0, 0);
    else if (wave_u == 1)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(dbase + t * TB + lane), (lds_void_t*)(st + TB * 4), 4, 0, 0);
  };
  // Full tiles (all but a ragged last one) are staged PIECE BY PIECE inside the MFMA regions below (one LDS-DMA per four MFMAs): issued as a
  // burst at the top of a tile the 17 pieces cost ~1600 of a tile's 5700 cycles (tools/attn_bwd_trace.py) -- a 64-lane x 16-byte request
  // occupies the CU's address unit for ~90 cycles and a lone wave has nothing to overlap it with.  Per-lane 32-bit offsets are computed
  // once; the tile base is wave-uniform.
  uint32_t o_sw[4], o_t[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int seed = i * DKV_THREADS + tid;
    const int r = seed >> 4, cp = seed & 15;
    o_sw[i] = (uint32_t)(((cp ^ (r & 15)) << 3) * 2);                      // swizzled chunk inside the row (bytes)
    const int d = seed >> 3, vp = seed & 7;
    o_t[i] = (uint32_t)(((int64_t)d * S_pad + ((vp ^ ((d >> 1) & 7)) << 3)) * 2);
  }
  // which: 0 Q rows, 1 dO rows, 2 Q^T cols, 3 dO^T cols (+ the L / delta rows).  UNCONDITIONAL (a branch would fence the pieces off from
  // the region's MFMAs): past the last tile the caller passes the last tile again, into the buffer nobody reads any more.  Rows past S of a
  // ragged last tile are clamped to S - 1 per piece (the transposed tensors are zero-padded to S_pad).
  auto stage_piece = [&](int which, int t, int buf) {
    char* base = smem + buf * DKV_STAGE;
    const int rmax = S - 1 - t * TB;                                        // (uniform) last existing row of this tile
    const char* src = which == 0 ? (const char*)qbase + (int64_t)t * TB * ldq * 2
                    : which == 1 ? (const char*)dobase + (int64_t)t * TB * lddo * 2
                    : which == 2 ? (const char*)qtbase + (int64_t)t * TB * 2 : (const char*)dotbase + (int64_t)t * TB * 2;
    char* dst = base + (which == 0 ? 0 : which == 1 ? ROWMAJ_BYTES : which == 2 ? 2 * ROWMAJ_BYTES : 2 * ROWMAJ_BYTES + TRANS_BYTES);
    const int ld2 = (int)((which == 0 ? ldq : lddo) * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t off;
      if (which < 2) off = (uint32_t)(min((i * DKV_THREADS + tid) >> 4, rmax) * ld2) + o_sw[i];
      else off = o_t[i];
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + off), (lds_void_t*)(dst + (i * DKV_THREADS + wave_u * 64) * 16), 16, 0, 0);
    }
    if (which == 3) {
      char* st = base + 2 * ROWMAJ_BYTES + 2 * TRANS_BYTES;
      if (wave_u == 0)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(lbase + t * TB + lane), (lds_void_t*)st, 4, 0, 0);
      else if (wave_u == 1)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(dbase + t * TB + lane), (lds_void_t*)(st + TB * 4), 4, 0, 0);
    }
  };
  f32x16_t dva[4], dka[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dva[d][r] = dka[d][r] = 0.f;

#ifdef AFX_BWD_TRACE
  unsigned tr[12];
  for (int i = 0; i < 12; ++i) tr[i] = 0;
  tr[7] = (unsigned)__builtin_readcyclecounter();
  tr[10] = (unsigned)__builtin_amdgcn_s_memrealtime();
  tr[11] = (unsigned)ntiles;
#endif
  stage(0, 0);
  AFX_SYNC_DMA();
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    asm volatile("" : "+v"(kf[s]));
    asm volatile("" : "+v"(vf[s]));
  }
  // One KV-stationary wave per SIMD: the loop is laid out in five scheduling regions per query tile so that every LDS fragment is
  // requested a whole MFMA group (16 x 32 cycles) before its first use -- left alone hipcc emits ds_read / s_waitcnt lgkmcnt(0) / mfma
  // triples (the full LDS latency in front of every one of the 64 MFMAs of a tile: 7.7 k cycles per tile for 2 k of matrix work) --
  // and so that the exp / dS arithmetic of one 32-query half sits in the same region as the other half's MFMAs:
  //   R0  read F1 (Q, dO rows of half 0)
  //   R1  read F3 (half 1)                  | S0 = Q0 K^T, dP0 = dO0 V^T
  //   R2  read F2 (dO^T, Q^T cols of half 0) | S1, dP1                       | P0 = exp2(S0 c - L), dS0 = P0 (dP0 - delta) / sqrt(d)
  //   R3  read F4 (half 1)                  | dV^T += dO0^T P0, dK^T += Q0^T dS0 | P1, dS1
  //   R4                                     | dV^T += dO1^T P1, dK^T += Q1^T dS1
  auto load_rows = [&](bf16x8_t (&f)[16], f32x4_t (&l4)[4], f32x4_t (&d4)[4], const char* qs, const char* dos, const float* lst,
                       const float* dst, int qb) {
    const int qrow = qb * 32 + kl;                         // A-operand row of this lane = a query of the tile
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      f[s] = frag_rowmaj(qs, qrow, s * 2 + hi);
      f[8 + s] = frag_rowmaj(dos, qrow, s * 2 + hi);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {                          // L / delta of the half's queries this lane's registers hold (broadcast reads)
      l4[g] = *reinterpret_cast<const f32x4_t*>(lst + qb * 32 + 8 * g + 4 * hi);
      d4[g] = *reinterpret_cast<const f32x4_t*>(dst + qb * 32 + 8 * g + 4 * hi);
    }
  };
  auto load_cols = [&](bf16x8_t (&f)[16], const char* dots, const char* qts, int qb) {
#pragma unroll
    for (int ksub = 0; ksub < 2; ++ksub) {
      const int chunk = (qb * 2 + ksub) * 2 + hi;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        f[ksub * 8 + d] = frag_trans(dots, d * 32 + kl, chunk);
        f[ksub * 8 + 4 + d] = frag_trans(qts, d * 32 + kl, chunk);
      }
    }
  };
  auto sdp = [&](const bf16x8_t (&f)[16], f32x16_t& sa, f32x16_t& dp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[r] = dp[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[s], kf[s], sa, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[8 + s], vf[s], dp, 0, 0, 0);
    }
  };
  // lane = key (column), register r = query 32 qb + (r&3) + 8 (r>>2) + 4 hi
  auto softmax_grad = [&](const f32x16_t& sa, const f32x16_t& dp, const f32x4_t (&l4)[4], const f32x4_t (&d4)[4], bf16x8_t (&pf)[2],
                          bf16x8_t (&df)[2]) {
    float pv[16], dsv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        // (no key mask: a lane past S holds a clamped copy of the last key and its dK / dV column is never stored; 1 / sqrt(d) once, in the epilogue)
        const float p = __builtin_amdgcn_exp2f(sa[r] * C_LOG2 - l4[g][e]);
        pv[r] = p;
        dsv[r] = p * (dp[r] - d4[g][e]);
      }
    }
#pragma unroll
    for (int ksub = 0; ksub < 2; ++ksub) {
      float t0[8], t1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        t0[j] = pv[ksub * 8 + j];
        t1[j] = dsv[ksub * 8 + j];
      }
      pf[ksub] = __builtin_bit_cast(bf16x8_t, pack8(t0));
      df[ksub] = __builtin_bit_cast(bf16x8_t, pack8(t1));
    }
  };
  auto dvdk = [&](const bf16x8_t (&f)[16], const bf16x8_t (&pf)[2], const bf16x8_t (&df)[2]) {
#pragma unroll
    for (int ksub = 0; ksub < 2; ++ksub)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        dva[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ksub * 8 + d], pf[ksub], dva[d], 0, 0, 0);
        dka[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ksub * 8 + 4 + d], df[ksub], dka[d], 0, 0, 0);
      }
  };
  // (MFMA, LDS read, VALU) triples: one matrix instruction, one fragment read for the next group and a slice of the other half's
  // arithmetic per 32-cycle MFMA slot
// Tuning knobs of the regions (build variants with -D...): VALU per MFMA slot in R2 / R3, LDS reads and DMA spacing in R1, no group barriers at all.
// Swept in round 4 on one box (BWD_NV 4 / 7 / 10 / 13, BWD_R1_NR 1 / 2, BWD_R1_DMA 2 / 4, BWD_NO_GROUPS): 561-571 TF over the 5 matmuls for every
// setting -- the order INSIDE a region is not what limits this kernel.
#ifndef BWD_NV
#define BWD_NV 7
#endif
#ifndef BWD_R1_NR
#define BWD_R1_NR 2
#endif
#ifndef BWD_R1_DMA
#define BWD_R1_DMA 2
#endif
#ifdef BWD_NO_GROUPS
#define AFX_BWD_INTERLEAVE(NR, NV, NDMA)
#else
#define AFX_BWD_INTERLEAVE(NR, NV, NDMA)                         \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {            \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           \
    if (NR) __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);  \
    if ((i_ % (NDMA)) == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   /* one LDS-DMA piece per NDMA MFMAs */ \
    if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);  \
  }
#endif
  // (Measured and not taken, round 4: the loop ROTATED so that R4 runs behind the barrier in one region with the next tile's R0 requests --
  // the barrier then comes one region earlier and waits ~1100 cycles for the DMA pieces instead of ~100: the same time per call.)
  for (int t = 0; t < ntiles; ++t) {
    const char* qs = smem + (t & 1) * DKV_STAGE;
    const char* dos = qs + ROWMAJ_BYTES;
    const char* qts = qs + 2 * ROWMAJ_BYTES;
    const char* dots = qts + TRANS_BYTES;
    const float* lst = reinterpret_cast<const float*>(dots + TRANS_BYTES);
    const float* dst = lst + TB;
    const int tn = min(t + 1, ntiles - 1), bn = (t + 1) & 1;  // next tile and its buffer (past the end: the last tile again, into the free buffer)
    bf16x8_t fa[16], fb[16], pf0[2], df0[2], pf1[2], df1[2];
    f32x16_t sa0, dp0, sa1, dp1;
    f32x4_t l0[4], d0[4], l1[4], d1[4];
    BWD_TR(0)
    load_rows(fa, l0, d0, qs, dos, lst, dst, 0);             // R0
    __builtin_amdgcn_sched_barrier(0);
    BWD_TR(1)
    load_rows(fb, l1, d1, qs, dos, lst, dst, 1);             // R1
    sdp(fa, sa0, dp0);
    stage_piece(0, tn, bn);                                  // (every piece is issued >= 2 regions ahead of the barrier that retires it)
    stage_piece(3, tn, bn);
    AFX_BWD_INTERLEAVE(BWD_R1_NR, 0, BWD_R1_DMA)
    __builtin_amdgcn_sched_barrier(0);
    BWD_TR(2)
    load_cols(fa, dots, qts, 0);                             // R2
    sdp(fb, sa1, dp1);
    softmax_grad(sa0, dp0, l0, d0, pf0, df0);
    stage_piece(1, tn, bn);
    AFX_BWD_INTERLEAVE(1, BWD_NV, 4)
    __builtin_amdgcn_sched_barrier(0);
    BWD_TR(3)
    load_cols(fb, dots, qts, 1);                             // R3
    dvdk(fa, pf0, df0);
    softmax_grad(sa1, dp1, l1, d1, pf1, df1);
    stage_piece(2, tn, bn);
    AFX_BWD_INTERLEAVE(1, BWD_NV, 4)
    __builtin_amdgcn_sched_barrier(0);
    BWD_TR(4)
    dvdk(fb, pf1, df1);                                      // R4
    __builtin_amdgcn_sched_barrier(0);
    BWD_TR(5)
    AFX_SYNC_DMA();
    BWD_TR(6)
  }
#ifdef AFX_BWD_TRACE
  tr[8] = (unsigned)__builtin_readcyclecounter() - tr[7];
  tr[9] = (unsigned)__builtin_amdgcn_s_memrealtime() - tr[10];
  if ((blockIdx.x == 0 || blockIdx.x == 1000) && lane == 0)
    for (int i = 0; i < 12; ++i) g_bwd_trace[((blockIdx.x ? 1 : 0) * 4 + wave) * 12 + i] = tr[i];
#endif
#undef AFX_BWD_INTERLEAVE
  if (k_ok) {
    bf16_t* kp_o = dk + ((int64_t)b * S + k0 + kl) * lddk + h * HD;
    bf16_t* vp_o = dv + ((int64_t)b * S + k0 + kl) * lddv + h * HD;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t w;
        w[0] = pack_bf16x2(dka[d][4 * g + 0] * SCALE, dka[d][4 * g + 1] * SCALE);
        w[1] = pack_bf16x2(dka[d][4 * g + 2] * SCALE, dka[d][4 * g + 3] * SCALE);
        *reinterpret_cast<u32x2_t*>(kp_o + d * 32 + g * 8 + hi * 4) = w;
        w[0] = pack_bf16x2(dva[d][4 * g + 0], dva[d][4 * g + 1]);
        w[1] = pack_bf16x2(dva[d][4 * g + 2], dva[d][4 * g + 3]);
        *reinterpret_cast<u32x2_t*>(vp_o + d * 32 + g * 8 + hi * 4) = w;
      }
  }
}

int64_t attn_bwd_ws_bytes(int B, int H, int S) {
  const int64_t S_pad = attn_spad(S);
  return 3 * (int64_t)B * H * HD * S_pad * 2 + (int64_t)B * H * S_pad * 4 + attn_bwd3_stats_bytes(B, H, S);
}

// 3 (default): the generated one-wave-per-SIMD streams of afx_attn_bwd3.hip in ONE launch (no transposed copies at all); 4: the same as two launches;
// 2: the round-4 kernels; 1: round-4 dQ + generated dK / dV   (A/B: AFX_ATTN_BWD_IMPL)
static int& attn_bwd_impl() {
  static int impl = [] {
    const char* e = getenv("AFX_ATTN_BWD_IMPL");
    return e != nullptr ? atoi(e) : 3;
  }();
  return impl;
}
void attn_bwd_set_impl(int impl) { attn_bwd_impl() = (impl >= 1 && impl <= 4) ? impl : 3; }

hipError_t launch_attention_backward(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v,
                                     int64_t ldv, const uint16_t* o, int64_t ldo, const uint16_t* dout, int64_t lddo,
                                     const float* lse, uint16_t* dq, int64_t lddq, uint16_t* dk, int64_t lddk,
                                     uint16_t* dv, int64_t lddv, void* ws, int B, int H, int S, hipStream_t stream) {
  const int S_pad = (int)attn_spad(S);
  const int64_t tsz = (int64_t)B * H * HD * S_pad;
  uint16_t* kt = (uint16_t*)ws;
  uint16_t* qt = kt + tsz;
  uint16_t* dot = qt + tsz;
  float* delta = (float*)(dot + tsz);
  float* stats = delta + (int64_t)B * H * S_pad;
  const int impl = attn_bwd3_eligible(q, k, v, dout, ldq, ldk, ldv, lddo, S) ? attn_bwd_impl() : 2;
  const bool dkv3 = impl != 2, dq3 = impl == 3 || impl == 4;
  hipError_t e;
  if (dq3) {           // L | delta side array, then the two generated kernels: nothing else
    if ((e = launch_attn_bwd_stats(o, ldo, dout, lddo, lse, stats, nullptr, B, H, S, stream)) != hipSuccess) return e;
    if (impl == 3) return launch_attn_bwd_fused3(q, ldq, k, ldk, v, ldv, dout, lddo, stats, dq, lddq, dk, lddk, dv, lddv, B, H, S, stream);
    if ((e = launch_attn_bwd_dq3(q, ldq, k, ldk, v, ldv, dout, lddo, stats, dq, lddq, B, H, S, stream)) != hipSuccess) return e;
    return launch_attn_bwd_dkv3(q, ldq, k, ldk, v, ldv, dout, lddo, stats, dk, lddk, dv, lddv, B, H, S, stream);
  }
  if ((e = launch_v_transpose(k, ldk, kt, B, H, S, stream)) != hipSuccess) return e;
  if (dkv3) {
    if ((e = launch_attn_bwd_stats(o, ldo, dout, lddo, lse, stats, delta, B, H, S, stream)) != hipSuccess) return e;
  } else {
    if ((e = hipMemsetAsync(delta, 0, (size_t)B * H * S_pad * 4, stream)) != hipSuccess) return e;
    if ((e = launch_v_transpose(q, ldq, qt, B, H, S, stream)) != hipSuccess) return e;
    if ((e = launch_v_transpose(dout, lddo, dot, B, H, S, stream)) != hipSuccess) return e;
    const int64_t total = (int64_t)B * S * H * 16;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, o, ldo, dout, lddo,
                       delta, H, S, S_pad, total);
  }
  static bool attr = false;
  if (!attr) {
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 2 * DQ_STAGE)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 2 * DKV_STAGE)) != hipSuccess) return e;
    attr = true;
  }
  const int hpx = (H + 7) / 8;
  const int nq = (S + DQ_WAVES * 32 - 1) / (DQ_WAVES * 32);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(8 * hpx * nq * B), dim3(DQ_THREADS), 2 * DQ_STAGE, stream, q, ldq, k, ldk, v,
                     ldv, kt, dout, lddo, lse, delta, dq, lddq, H, S, S_pad, nq, B);
  if (dkv3) return launch_attn_bwd_dkv3(q, ldq, k, ldk, v, ldv, dout, lddo, stats, dk, lddk, dv, lddv, B, H, S, stream);
  const int nk = (S + DKV_WAVES * 32 - 1) / (DKV_WAVES * 32);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(8 * hpx * nk * B), dim3(DKV_THREADS), 2 * DKV_STAGE, stream, q, ldq, k, ldk,
                     v, ldv, qt, dout, lddo, dot, lse, delta, dk, lddk, dv, lddv, H, S, S_pad, nk, B);
  return hipGetLastError();
}

}  // namespace afx

extern "C" int afx_debug_bwd_trace(unsigned* host_out) {      // [2 blocks][4 waves][12]: region stamps of tile 36, whole-kernel cycles / 100 MHz ticks / tiles
#ifdef AFX_BWD_TRACE
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(afx::g_bwd_trace), 2 * 4 * 12 * sizeof(unsigned)) == hipSuccess ? 0 : -1;
#else
  (void)host_out;
  return -1;
#endif
}
