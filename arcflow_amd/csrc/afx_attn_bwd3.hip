// Flash-attention backward, dK / dV part, for gfx950 -- ONE wave per SIMD, hand-placed instruction stream (round 5).
// head_dim 128, bf16 in / out, no mask.  Replaces attn_bwd_dkv_kernel (afx_attn_bwd.hip) for S > 64; the mathematics are unchanged:
//     S = Q K^T        dP = dO V^T        P = exp2(S c - L)        dS = P o (dP - delta)        dV^T += dO^T P        dK^T += Q^T dS      (x 1/sqrt(d) at the end)
// key-stationary, lane = key on v_mfma_f32_32x32x16_bf16, so P and dS leave the softmax gradient already as the B operands of the two
// accumulating products.  Reference semantics: the FlashAttention-2 backward torch's SDPA runs under the reference's student trunk
// (arcflux.py:181-189 recompute path).
//
// What is different from the round-4 kernel (compiler-scheduled regions, 0.23 of the peak with the dQ kernel):
//   * NO transposed copies of Q and dO.  The row-major tiles go to LDS as they lie in memory and the A operands of dV^T += dO^T P / dK^T += Q^T dS --
//     8 consecutive queries of one head-dim column per lane -- are gathered by ds_read_b64_tr_b16 (pinned by tools/tr_probe.hip; afx_tn.hip uses the
//     same read).  Half the LDS-DMA pieces and half the L2 traffic per tile, and two transpose passes per call are gone.
//   * the stream is GENERATED (tools/gen_attn_bwd3.py -> gen/b3_*.inc, committed): every instruction its own asm statement, every wide operand
//     asm-owned (register map in the generator's header), hipcc confined to v[0:63] by amdgpu_num_vgpr(64), ISA audited by arcflow_amd/build.py.
//     Per 32 MFMAs a wave issues 80 VALU instructions (the round-4 kernel: ~245, two thirds of them address arithmetic, accumulator-file moves and
//     register shuffles the compiler added), 56 LDS reads and 5 DMA pieces.
//   * software pipeline over HALVES of 32 queries (phase p: S / dP of half p | softmax gradient of half p - 1 | dV / dK of half p - 2), an 8-slot LDS
//     ring of half tiles (Q | dO | L | delta = 16640 bytes) fed three phases ahead, one counted `s_waitcnt vmcnt(15)` + barrier per phase.
//   * L and delta come from ONE padded side array (attn_bwd_stats_kernel below: rows past S carry L = +inf, delta = 0, so that the clamped copies of the
//     last row that fill a ragged tile contribute exactly zero: no mask in the loop).
#include <hip/hip_runtime.h>

#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {
namespace b3 {

constexpr int HD = 128;
constexpr int SLOT = 16640;
constexpr int NSLOT = 8;
constexpr int LDS_BYTES = SLOT * NSLOT;                  // 133120
constexpr int THREADS = 256;
constexpr float SCALE = 0.08838834764831845f;            // 1/sqrt(128)
constexpr float C_LOG2 = 0.08838834764831845f * 1.4426950408889634f;

AFX_DEV uint64_t uniform_u64(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

#ifndef B3_GEN
#define B3_GEN gen
#endif
#define B3_XSTR(x) #x
#define B3_STR(x) B3_XSTR(x)
#define B3_INC(name) B3_STR(B3_GEN/name)
#include B3_INC(b3_readout.inc)

#ifdef AFX_BWD3_TRACE
__device__ unsigned g_bwd3_trace[2 * 4 * 16];
#endif

// stats[(b H + h)][half u][64] = L of the half's 32 queries | their delta = sum_d dO O; rows past S: +inf | 0.  One 16-lane group per (row, head).
// delta_old (may be null): the [B, H, S_pad] layout the dQ kernel reads.
__global__ __launch_bounds__(256) void attn_bwd_stats_kernel(const bf16_t* __restrict__ o, int64_t ldo, const bf16_t* __restrict__ dout, int64_t lddo,
                                                            const float* __restrict__ lse, float* __restrict__ stats, float* __restrict__ delta_old,
                                                            int H, int S, int S_pad, int64_t total) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= total) return;
  const int c = (int)(g & 15);
  const int64_t th = g >> 4;
  const int h = (int)(th % H);
  const int64_t prow = th / H;               // b * S_pad + s
  const int64_t bb = prow / S_pad;
  const int ss = (int)(prow % S_pad);
  float* dst = stats + (((bb * H + h) * (S_pad / 32) + (ss >> 5)) << 6) + (ss & 31);
  if (ss >= S) {                             // (uniform per 16-lane group)
    if (c == 0) {
      dst[0] = INFINITY;
      dst[32] = 0.f;
      if (delta_old != nullptr) delta_old[(bb * H + h) * S_pad + ss] = 0.f;
    }
    return;
  }
  const int64_t row = bb * S + ss;
  float a[8], b[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(o + row * ldo + h * HD + c * 8), a);
  unpack8(*reinterpret_cast<const u32x4_t*>(dout + row * lddo + h * HD + c * 8), b);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += a[e] * b[e];
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (c == 0) {
    dst[0] = lse[(bb * H + h) * S_pad + ss];
    dst[32] = s;
    if (delta_old != nullptr) delta_old[(bb * H + h) * S_pad + ss] = s;
  }
}

__global__ __launch_bounds__(THREADS, 1) __attribute__((amdgpu_num_vgpr(64))) void attn_bwd_dkv3_kernel(
    const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v, int64_t ldv,
    const bf16_t* __restrict__ dout, int64_t lddo, const float* __restrict__ stats, bf16_t* __restrict__ dk, int64_t lddk,
    bf16_t* __restrict__ dv, int64_t lddv, int H, int S, int S_pad, int nk, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kl = lane & 31, hi = lane >> 5;
  // head -> XCD affinity as in the forward: XCD x owns heads x, x + 8, ... (the work-groups of a head stream the same Q / dO through one L2)
  const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
  const int per_head = nk * B;
  const int h = xcd + 8 * (slot_id / per_head);
  if (h >= H) return;
  const int rem = slot_id % per_head;
  const int b = rem / nk;
  const int k0 = (rem % nk) * 128 + wave * 32;
  const int NH = S_pad / 32;                                  // halves of 32 queries (even, >= 4: launcher)
  const float c = C_LOG2;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

  // ---- lane constants ---------------------------------------------------------------------------------------------------------------------
  // row fragments (A of S / dP): row kl of the half, logical 16-byte chunk 2 (s & 1) + hi of d-block s >> 1, at physical chunk ^ ((row >> 2) & 3)
  const uint32_t rsw = (uint32_t)((kl >> 2) & 3);
  const uint32_t raddr0l = lds0 + kl * 64 + (((uint32_t)hi ^ rsw) << 4);
  const uint32_t raddr1l = lds0 + kl * 64 + (((uint32_t)(2 + hi) ^ rsw) << 4);
  const uint32_t raddr0h = raddr0l + 4 * SLOT, raddr1h = raddr1l + 4 * SLOT;
  // transpose reads (A of dV^T / dK^T): 16-lane group (b4 = d columns 0-15 / 16-31 of the block, hi = rows +4), lane i supplies row i / 4, columns 4 (i % 4) ..+3
  const int ti = lane & 15, b4 = (lane >> 4) & 1;
  const uint32_t trow = (uint32_t)(4 * hi + (ti >> 2));
  const uint32_t tch = (uint32_t)(2 * b4 + ((ti & 3) >> 1));
  const uint32_t taddr0l = lds0 + trow * 64 + ((tch ^ (uint32_t)hi) << 4) + ((ti & 1) << 3);          // rd = 0: (row >> 2) & 3 = hi
  const uint32_t taddr1l = lds0 + (trow + 8) * 64 + ((tch ^ (uint32_t)(2 + hi)) << 4) + ((ti & 1) << 3);    // rd = 1: rows + 8
  const uint32_t taddr0h = taddr0l + 4 * SLOT, taddr1h = taddr1l + 4 * SLOT;
  const uint32_t saddrl = lds0 + 16384 + hi * 16, saddrh = saddrl + 4 * SLOT;     // L | delta block behind the two 8 KiB tensors
  // LDS-DMA: waves 0, 1 move Q rows 0-15 / 16-31 of a half, waves 2, 3 dO; piece i = d-block i (1 KiB: 16 rows x 64 bytes), lane -> row lane / 4,
  // physical chunk lane % 4 = logical chunk (lane % 4) ^ ((row >> 2) & 3) = (lane % 4) ^ (lane / 16)
  const int tensor = wave >> 1, rg = wave & 1;
  const uint32_t wave_lds = lds0 + tensor * 8192 + rg * 1024;
  const int64_t ldx = tensor ? lddo : ldq;
  const char* xbase = reinterpret_cast<const char*>((tensor ? dout : q) + (int64_t)b * S * ldx + h * HD);
  const int drow = 16 * rg + (lane >> 2);
  const int dcol = 8 * ((lane & 3) ^ (lane >> 4));
  const int64_t half_bytes = 32 * ldx * 2;
  // normal halves: offsets relative to the half's first row; the last tile's two halves (rows clamped to S - 1): relative to half NH - 2
  const int rA = min(32 * (NH - 2) + drow, S - 1) - 32 * (NH - 2), rB = min(32 * (NH - 1) + drow, S - 1) - 32 * (NH - 2);
#define B3_XOFF(R, I) ((uint32_t)(((int64_t)(R) * ldx + 32 * (I) + dcol) * 2))
  const uint32_t xoff0 = B3_XOFF(drow, 0), xoff1 = B3_XOFF(drow, 1), xoff2 = B3_XOFF(drow, 2), xoff3 = B3_XOFF(drow, 3);
  const uint32_t xoffA0 = B3_XOFF(rA, 0), xoffA1 = B3_XOFF(rA, 1), xoffA2 = B3_XOFF(rA, 2), xoffA3 = B3_XOFF(rA, 3);
  const uint32_t xoffB0 = B3_XOFF(rB, 0), xoffB1 = B3_XOFF(rB, 1), xoffB2 = B3_XOFF(rB, 2), xoffB3 = B3_XOFF(rB, 3);
  const uint32_t sofs = (uint32_t)lane * 4;
  const char* sbase = reinterpret_cast<const char*>(stats + ((int64_t)(b * H + h) * NH << 6));
  // sources of half U (halves past the end re-fetch the last one into a free slot: every phase issues 5 pieces)
#define B3_SRC(U)                                                                                                              \
  const int uu_ = min((U), NH - 1);                                                                                            \
  const uint64_t xsrc = uniform_u64((uint64_t)(uintptr_t)(xbase + (int64_t)min(uu_, NH - 2) * half_bytes));                    \
  const uint64_t ssrc = uniform_u64((uint64_t)(uintptr_t)(sbase + (int64_t)uu_ * 256));                                       \
  const uint32_t xofs0 = uu_ >= NH - 1 ? xoffB0 : (uu_ == NH - 2 ? xoffA0 : xoff0), xofs1 = uu_ >= NH - 1 ? xoffB1 : (uu_ == NH - 2 ? xoffA1 : xoff1), \
                 xofs2 = uu_ >= NH - 1 ? xoffB2 : (uu_ == NH - 2 ? xoffA2 : xoff2), xofs3 = uu_ >= NH - 1 ? xoffB3 : (uu_ == NH - 2 ? xoffA3 : xoff3);

#ifdef AFX_BWD3_TRACE
  unsigned tr[16];
  for (int i = 0; i < 16; ++i) tr[i] = 0;
  const unsigned tr_c0 = (unsigned)__builtin_readcyclecounter();
#define B3_TR(i)                                                                        \
  if (p == 72 + (((i) + 6) & 7)) {                                                       \
    uint64_t st_;                                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_)::"memory");        \
    tr[i] = (unsigned)st_;                                                              \
  }
#else
#define B3_TR(i)
#endif

  // ---- prologue: dV^T = dK^T = 0, the wave's K / V rows (accumulator file), halves 0..4 ----------------------------------------------------
#include B3_INC(b3_init.inc)
  {
    const int krow = min(k0 + kl, S - 1);                       // a lane past S holds a clamped copy of the last key; its columns are never stored
    const bf16_t* kptr = k + ((int64_t)b * S + krow) * ldk + h * HD + hi * 8;
    const bf16_t* vptr = v + ((int64_t)b * S + krow) * ldv + h * HD + hi * 8;
#include B3_INC(b3_kvload.inc)
  }
#include B3_INC(b3_prologue_dma.inc)
#include B3_INC(b3_first_rows.inc)

  int p = 0;
  {
    B3_SRC(p + 5)
#include B3_INC(b3_p0.inc)
  }
  p = 1;
  {
    B3_SRC(p + 5)
#include B3_INC(b3_p1.inc)
  }
  p = 2;
  if (NH > 2) {
#pragma unroll 1
    for (;;) {
      {
        B3_SRC(p + 5)
#include B3_INC(b3_body2.inc)
      }
      ++p;
      {
        B3_SRC(p + 5)
#include B3_INC(b3_body3.inc)
      }
      if (++p == NH) break;
      {
        B3_SRC(p + 5)
#include B3_INC(b3_body4.inc)
      }
      ++p;
      {
        B3_SRC(p + 5)
#include B3_INC(b3_body5.inc)
      }
      if (++p == NH) break;
      {
        B3_SRC(p + 5)
#include B3_INC(b3_body6.inc)
      }
      ++p;
      {
        B3_SRC(p + 5)
#include B3_INC(b3_body7.inc)
      }
      if (++p == NH) break;
      {
        B3_SRC(p + 5)
#include B3_INC(b3_body0.inc)
      }
      ++p;
      {
        B3_SRC(p + 5)
#include B3_INC(b3_body1.inc)
      }
      if (++p == NH) break;
    }
  }
  // ---- the pipeline drains: SM(NH - 1), DV(NH - 2) | DV(NH - 1) ----------------------------------------------------------------------------
  {
    const uint32_t ts = (uint32_t)((NH - 2) & 7) * SLOT;
#include B3_INC(b3_tail0.inc)
  }
  {
    const uint32_t ts = (uint32_t)((NH - 1) & 7) * SLOT;
#include B3_INC(b3_tail1.inc)
  }
  // every DMA piece must have landed before this work-group's LDS can be handed to another one; MFMA -> accumulator-read wait states
  B3_DRAIN
#ifdef AFX_BWD3_TRACE
  tr[8] = (unsigned)__builtin_readcyclecounter() - tr_c0;
  tr[9] = (unsigned)NH;
  if ((blockIdx.x == 0 || blockIdx.x == 1000) && (threadIdx.x & 63) == 0)
    for (int i = 0; i < 16; ++i) g_bwd3_trace[((blockIdx.x ? 1 : 0) * 4 + wave) * 16 + i] = tr[i];
#endif

  // ---- epilogue: lane (key, hi) holds X^T[32 d + 8 g + 4 hi + 0..3][key] in ox[4 g + 0..3] --------------------------------------------------
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const int kl2 = tid2 & 31, hi2 = (tid2 >> 5) & 1;
  const int key = (rem % nk) * 128 + __builtin_amdgcn_readfirstlane(tid2 >> 6) * 32 + kl2;
  if (key < S) {
    bf16_t* kp_o = dk + ((int64_t)b * S + key) * lddk + h * HD + hi2 * 4;
    bf16_t* vp_o = dv + ((int64_t)b * S + key) * lddv + h * HD + hi2 * 4;
    float ox[16];
#define B3_STORE(PTR, D, SC)                                                            \
  _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                     \
    u32x2_t w_;                                                                          \
    w_[0] = pack_bf16x2(ox[4 * g_ + 0] * (SC), ox[4 * g_ + 1] * (SC));                   \
    w_[1] = pack_bf16x2(ox[4 * g_ + 2] * (SC), ox[4 * g_ + 3] * (SC));                   \
    *reinterpret_cast<u32x2_t*>((PTR) + (D) * 32 + g_ * 8) = w_;                         \
  }
    B3_READ_V_0 B3_STORE(vp_o, 0, 1.0f)
    B3_READ_V_1 B3_STORE(vp_o, 1, 1.0f)
    B3_READ_V_2 B3_STORE(vp_o, 2, 1.0f)
    B3_READ_V_3 B3_STORE(vp_o, 3, 1.0f)
    B3_READ_K_0 B3_STORE(kp_o, 0, SCALE)
    B3_READ_K_1 B3_STORE(kp_o, 1, SCALE)
    B3_READ_K_2 B3_STORE(kp_o, 2, SCALE)
    B3_READ_K_3 B3_STORE(kp_o, 3, SCALE)
  }
}

}  // namespace b3

int64_t attn_bwd3_stats_bytes(int B, int H, int S) { return (int64_t)B * H * attn_spad(S) * 2 * 4; }

// stats + (optionally) the dQ kernel's delta array
hipError_t launch_attn_bwd_stats(const uint16_t* o, int64_t ldo, const uint16_t* dout, int64_t lddo, const float* lse, float* stats, float* delta_old,
                                 int B, int H, int S, hipStream_t stream) {
  const int S_pad = (int)attn_spad(S);
  const int64_t total = (int64_t)B * S_pad * H * 16;
  hipLaunchKernelGGL(b3::attn_bwd_stats_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, o, ldo, dout, lddo, lse, stats, delta_old, H, S,
                     S_pad, total);
  return hipGetLastError();
}

hipError_t launch_attn_bwd_dkv3(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv, const uint16_t* dout,
                                int64_t lddo, const float* stats, uint16_t* dk, int64_t lddk, uint16_t* dv, int64_t lddv, int B, int H, int S,
                                hipStream_t stream) {
  const int S_pad = (int)attn_spad(S);
  if (S_pad / 32 < 4) return hipErrorInvalidValue;            // (the caller keeps the round-4 kernel for S <= 64)
  // 16-byte LDS-DMA / global loads: row strides and base pointers
  if ((ldq | ldk | ldv | lddo) % 8 || ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
                                        reinterpret_cast<uintptr_t>(dout)) & 15))
    return hipErrorInvalidValue;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(b3::attn_bwd_dkv3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, b3::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr = true;
  }
  const int hpx = (H + 7) / 8;
  const int nk = (S + 127) / 128;
  hipLaunchKernelGGL(b3::attn_bwd_dkv3_kernel, dim3(8 * hpx * nk * B), dim3(b3::THREADS), b3::LDS_BYTES, stream, q, ldq, k, ldk, v, ldv, dout, lddo, stats, dk,
                     lddk, dv, lddv, H, S, S_pad, nk, B);
  return hipGetLastError();
}

}  // namespace afx

extern "C" int afx_debug_bwd3_trace(unsigned* host_out) {      // [2 blocks][4 waves][16]: stamps of eight consecutive phases, whole-kernel cycles, NH
#ifdef AFX_BWD3_TRACE
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(afx::b3::g_bwd3_trace), 2 * 4 * 16 * sizeof(unsigned)) == hipSuccess ? 0 : -1;
#else
  (void)host_out;
  return -1;
#endif
}
