// Flash-attention backward for gfx950 -- ONE wave per SIMD, hand-placed instruction streams (round 5): the dK / dV kernel and the dQ kernel.
// head_dim 128, bf16 in / out, no mask.  They replace attn_bwd_dkv_kernel / attn_bwd_dq_kernel (afx_attn_bwd.hip) for S > 64; the mathematics are unchanged:
//     S = Q K^T        dP = dO V^T        P = exp2(S c - L)        dS = P o (dP - delta)
//     dV^T += dO^T P        dK^T += Q^T dS        dQ^T += K^T dS^T        (dK, dQ x 1/sqrt(d) at the end)
// Reference semantics: the FlashAttention-2 backward torch's SDPA runs under the reference's student trunk (arcflux.py:181-189 recompute path).
// Two kernels, no atomics, deterministic: S and dP are recomputed in both (7 matmuls for 5).
//   * dK / dV: key-stationary, lane = key on v_mfma_f32_32x32x16_bf16, so P and dS leave the softmax gradient already as the B operands of the two accumulating products;
//   * dQ: query-stationary, lane = query: the SAME stream with the tensors' roles swapped (S^T = K Q^T, dP^T = V dO^T, dS^T is the B operand of dQ^T += K^T dS^T).
//
// What is different from the round-4 kernels (compiler-scheduled, 0.23 of the peak together):
//   * NO transposed copies of Q, dO, K.  The row-major tiles go to LDS as they lie in memory and the A operands of the accumulating products --
//     8 consecutive rows of one head-dim column per lane -- are gathered by ds_read_b64_tr_b16 (pinned by tools/tr_probe.hip; afx_tn.hip uses the
//     same read).  Half the LDS-DMA pieces and half the L2 traffic per tile, and the three transpose passes per call are gone.
//   * the streams are GENERATED (tools/gen_attn_bwd3.py -> gen/b3_*.inc, gen/q3_*.inc, committed): every instruction its own asm statement, every wide
//     operand asm-owned (register map in the generator's header), hipcc confined to v[0:59] by amdgpu_num_vgpr(60), ISA audited by arcflow_amd/build.py.
//     Per 32 MFMAs the dK / dV wave issues 64 VALU instructions (the round-4 kernel: ~245, two thirds of them address arithmetic, accumulator-file moves and
//     register shuffles the compiler added), 56 LDS reads and 5 DMA pieces; the dQ wave 56 / 32 / 4 per 24 MFMAs.
//   * software pipeline over HALVES of 32 streamed rows (phase p: S / dP of half p | softmax gradient of half p - 1 | accumulating products of half p - 2),
//     an 8-slot LDS ring of half tiles fed three phases ahead, one counted `s_waitcnt vmcnt` + barrier per phase.
//   * L and delta come from ONE padded side array (attn_bwd_stats_kernel below: rows past S carry L = +inf, delta = 0, so that the clamped copies of the
//     last row that fill a ragged query tile contribute exactly zero: no mask in the dK / dV loop; the dQ kernel masks the last tile's keys on a cold path).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include <hip/hip_runtime.h>

#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {
namespace b3 {

constexpr int HD = 128;
constexpr int NSLOT = 8;
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
#include B3_INC(q3_readout.inc)
#include B3_INC(q3_mask.inc)

#ifdef AFX_BWD3_TRACE
__device__ unsigned g_bwd3_trace[2 * 2 * 4 * 32];
#endif

// stats[(b H + h)][half u][64] = L of the half's 32 queries | MINUS their delta = -sum_d dO O (the dP MFMA chains start from it); rows past S: +inf | 0.  One 16-lane group per (row, head).
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
    dst[32] = -s;
    if (delta_old != nullptr) delta_old[(bb * H + h) * S_pad + ss] = s;
  }
}

// the two streams: one body (afx_attn_bwd3_kernel.inc), the generated instruction streams differ
#define B3_DQ 0
#define B3_BODY dkv3_body
#define B3_G(name) B3_INC(b3_##name)
#include "afx_attn_bwd3_kernel.inc"
#undef B3_DQ
#undef B3_BODY
#undef B3_G
#define B3_DQ 1
#define B3_BODY dq3_body
#define B3_G(name) B3_INC(q3_##name)
#include "afx_attn_bwd3_kernel.inc"
#undef B3_DQ
#undef B3_BODY
#undef B3_G

#define B3_KATTR __global__ __launch_bounds__(THREADS, 1) __attribute__((amdgpu_num_vgpr(60)))
B3_KATTR void attn_bwd_dkv3_kernel(const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v, int64_t ldv,
                                   const bf16_t* __restrict__ dout, int64_t lddo, const float* __restrict__ stats, bf16_t* __restrict__ dk, int64_t lddk,
                                   bf16_t* __restrict__ dv, int64_t lddv, int H, int S, int S_pad, int nblk, int B) {
  dkv3_body(k, ldk, v, ldv, q, ldq, dout, lddo, stats, dk, lddk, dv, lddv, H, S, S_pad, nblk, B, (int)blockIdx.x);
}
B3_KATTR void attn_bwd_dq3_kernel(const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v, int64_t ldv,
                                  const bf16_t* __restrict__ dout, int64_t lddo, const float* __restrict__ stats, bf16_t* __restrict__ dq, int64_t lddq, int H,
                                  int S, int S_pad, int nblk, int B) {
  dq3_body(q, ldq, dout, lddo, k, ldk, v, ldv, stats, dq, lddq, nullptr, 0, H, S, S_pad, nblk, B, (int)blockIdx.x);
}
// ONE launch for both (the default): the two grids are 3.375 rounds of one work-group per CU each at S = 4608 / H = 24 and a launch costs ceil(rounds)
// (DESIGN 4.0); in one in-order grid the dQ work-groups start on the compute units the dK / dV grid's last round leaves idle (6.75 rounds -> 7 instead
// of 4 + 4).  Order inside an XCD's queue (slot = blockIdx / 8): ord_a dK / dV work-groups, ord_mid dQ ones, the other dK / dV ones, the other dQ ones --
// launch_attn_bwd_fused3 picks (ord_a, ord_mid) by simulating the in-order dispatch with the two streams' measured cost ratio.
B3_KATTR void attn_bwd_fused3_kernel(const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v, int64_t ldv,
                                     const bf16_t* __restrict__ dout, int64_t lddo, const float* __restrict__ stats, bf16_t* __restrict__ dq, int64_t lddq,
                                     bf16_t* __restrict__ dk, int64_t lddk, bf16_t* __restrict__ dv, int64_t lddv, int H, int S, int S_pad, int nblk, int B,
                                     int per_xcd, int ord_a, int ord_mid) {
  const int xcd = (int)blockIdx.x & 7, s = (int)blockIdx.x >> 3;
  int idx;
  bool is_dq;
  if (s < ord_a) { is_dq = false; idx = s; }
  else if (s < ord_a + ord_mid) { is_dq = true; idx = s - ord_a; }
  else if (s < per_xcd + ord_mid) { is_dq = false; idx = s - ord_mid; }
  else { is_dq = true; idx = s - per_xcd; }
  const int bid = (idx << 3) | xcd;
  if (!is_dq) dkv3_body(k, ldk, v, ldv, q, ldq, dout, lddo, stats, dk, lddk, dv, lddv, H, S, S_pad, nblk, B, bid);
  else dq3_body(q, ldq, dout, lddo, k, ldk, v, ldv, stats, dq, lddq, nullptr, 0, H, S, S_pad, nblk, B, bid);
}
#undef B3_KATTR

}  // namespace b3

int64_t attn_bwd3_stats_bytes(int B, int H, int S) { return (int64_t)B * H * attn_spad(S) * 2 * 4; }

// stats + (optionally) the round-4 dQ kernel's delta array
hipError_t launch_attn_bwd_stats(const uint16_t* o, int64_t ldo, const uint16_t* dout, int64_t lddo, const float* lse, float* stats, float* delta_old,
                                 int B, int H, int S, hipStream_t stream) {
  const int S_pad = (int)attn_spad(S);
  const int64_t total = (int64_t)B * S_pad * H * 16;
  hipLaunchKernelGGL(b3::attn_bwd_stats_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, o, ldo, dout, lddo, lse, stats, delta_old, H, S,
                     S_pad, total);
  return hipGetLastError();
}

bool attn_bwd3_eligible(const void* q, const void* k, const void* v, const void* dout, int64_t ldq, int64_t ldk, int64_t ldv, int64_t lddo, int S) {
  if (attn_spad(S) / 32 < 4) return false;                   // S <= 64: the round-4 kernels
  if ((ldq | ldk | ldv | lddo) % 8) return false;             // 16-byte LDS-DMA pieces / fragment loads: row strides and base pointers
  return ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(dout)) & 15) == 0;
}

static bool bwd3_args_ok(const void* a, const void* b, const void* c, const void* d, int64_t l0, int64_t l1, int64_t l2, int64_t l3, int S) {
  if (attn_spad(S) / 32 < 4) return false;                   // (the caller keeps the round-4 kernels for S <= 64)
  // 16-byte LDS-DMA / global loads: row strides and base pointers
  if ((l0 | l1 | l2 | l3) % 8) return false;
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}

hipError_t launch_attn_bwd_dkv3(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv, const uint16_t* dout,
                                int64_t lddo, const float* stats, uint16_t* dk, int64_t lddk, uint16_t* dv, int64_t lddv, int B, int H, int S,
                                hipStream_t stream) {
  if (!bwd3_args_ok(q, k, v, dout, ldq, ldk, ldv, lddo, S)) return hipErrorInvalidValue;
  const int S_pad = (int)attn_spad(S);
  constexpr int lds = 16640 * b3::NSLOT;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(b3::attn_bwd_dkv3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr = true;
  }
  const int hpx = (H + 7) / 8;
  const int nk = (S + 127) / 128;
  hipLaunchKernelGGL(b3::attn_bwd_dkv3_kernel, dim3(8 * hpx * nk * B), dim3(b3::THREADS), lds, stream, q, ldq, k, ldk, v, ldv, dout, lddo, stats, dk, lddk, dv, lddv,
                     H, S, S_pad, nk, B);
  return hipGetLastError();
}

hipError_t launch_attn_bwd_dq3(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv, const uint16_t* dout,
                               int64_t lddo, const float* stats, uint16_t* dq, int64_t lddq, int B, int H, int S, hipStream_t stream) {
  if (!bwd3_args_ok(q, k, v, dout, ldq, ldk, ldv, lddo, S)) return hipErrorInvalidValue;
  const int S_pad = (int)attn_spad(S);
  constexpr int lds = 16384 * b3::NSLOT;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(b3::attn_bwd_dq3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr = true;
  }
  const int hpx = (H + 7) / 8;
  const int nq = (S + 127) / 128;
  hipLaunchKernelGGL(b3::attn_bwd_dq3_kernel, dim3(8 * hpx * nq * B), dim3(b3::THREADS), lds, stream, q, ldq, k, ldk, v, ldv, dout, lddo, stats, dq, lddq, H, S, S_pad,
                     nq, B);
  return hipGetLastError();
}

hipError_t launch_attn_bwd_fused3(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv, const uint16_t* dout,
                                  int64_t lddo, const float* stats, uint16_t* dq, int64_t lddq, uint16_t* dk, int64_t lddk, uint16_t* dv, int64_t lddv, int B,
                                  int H, int S, hipStream_t stream) {
  if (!bwd3_args_ok(q, k, v, dout, ldq, ldk, ldv, lddo, S)) return hipErrorInvalidValue;
  const int S_pad = (int)attn_spad(S);
  constexpr int lds = 16640 * b3::NSLOT;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(b3::attn_bwd_fused3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr = true;
  }
  const int hpx = (H + 7) / 8;
  const int nb = (S + 127) / 128;
  const int per_xcd = hpx * nb * B;                          // work-groups of ONE stream per XCD
  // order of an XCD's queue: [a dK/dV][mid dQ][per - a dK/dV][per - mid dQ].  The dispatcher hands the next work-group to the first free CU (32 per XCD), so the
  // finish time of the grid is a list-scheduling makespan: search (a, mid) on a grid of 4 with the streams' cost ratio (dK/dV : dQ = 1.29 at 16 / 8 accumulating
  // MFMAs per phase); a = mid = 0 is "all dK/dV first".  AFX_ATTN_BWD_ORDER="a,mid" overrides (A/B runs).
  // Memoised per per_xcd under a mutex (variable prompt lengths alternate shapes; the search is host work inside the launch path); the makespan of
  // one candidate is a heap walk, O(per_xcd log 32).
  static std::mutex order_mu;
  static std::map<int, std::pair<int, int>> order_cache;
  int cache_a = 0, cache_mid = 0;
  {
    std::lock_guard<std::mutex> lock(order_mu);
    auto hit = order_cache.find(per_xcd);
    if (hit == order_cache.end()) {
      int best_a = 0, best_mid = 0;
      if (const char* e = getenv("AFX_ATTN_BWD_ORDER")) {
        if (sscanf(e, "%d,%d", &best_a, &best_mid) != 2) best_a = best_mid = 0;
        best_a = std::min(std::max(best_a, 0), per_xcd);
        best_mid = std::min(std::max(best_mid, 0), per_xcd);
      } else {
        const double cl = 1.29, cs = 1.0;
        double best = 1e30;
        std::vector<double> cu(32);
        for (int a = 0; a <= per_xcd; a += 4)
          for (int mid = 0; mid <= per_xcd; mid += 4) {
            std::fill(cu.begin(), cu.end(), 0.0);                     // (all equal: a valid min-heap under std::greater)
            double m = 0.0;
            for (int i = 0; i < 2 * per_xcd; ++i) {
              const bool is_dq = (i >= a && i < a + mid) || i >= per_xcd + mid;
              std::pop_heap(cu.begin(), cu.end(), std::greater<double>());      // the first free CU
              cu.back() += is_dq ? cs : cl;
              m = std::max(m, cu.back());
              std::push_heap(cu.begin(), cu.end(), std::greater<double>());
            }
            if (m < best - 1e-9) { best = m; best_a = a; best_mid = mid; }
          }
      }
      if (order_cache.size() >= 64) order_cache.clear();
      hit = order_cache.emplace(per_xcd, std::make_pair(best_a, best_mid)).first;
    }
    cache_a = hit->second.first; cache_mid = hit->second.second;
  }
  hipLaunchKernelGGL(b3::attn_bwd_fused3_kernel, dim3(16 * per_xcd), dim3(b3::THREADS), lds, stream, q, ldq, k, ldk, v, ldv, dout, lddo, stats, dq, lddq, dk, lddk, dv,
                     lddv, H, S, S_pad, nb, B, per_xcd, cache_a, cache_mid);
  return hipGetLastError();
}

}  // namespace afx

extern "C" int afx_debug_bwd3_trace(unsigned* host_out) {      // [dkv | dq][2 blocks][4 waves][32]: 3 stamps of eight consecutive phases, whole-kernel cycles, NH
#ifdef AFX_BWD3_TRACE
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(afx::b3::g_bwd3_trace), 2 * 2 * 4 * 32 * sizeof(unsigned)) == hipSuccess ? 0 : -1;
#else
  (void)host_out;
  return -1;
#endif
}
