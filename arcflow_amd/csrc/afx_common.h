// Shared device helpers for the ArcFlow gfx950 kernels (wave64, bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace afx {

typedef uint16_t bf16_t;                                             // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;         // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;           // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;         // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;        // 16-byte memory word
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define AFX_DEV __device__ __forceinline__

// Work-group barrier that ALSO retires this wave's LDS-DMA (global_load_lds) transfers.  A bare __syncthreads() is not enough:
// hipcc only puts the s_waitcnt vmcnt(0) in front of the s_barrier when its own bookkeeping still sees the DMA as pending,
// and after loop unswitching one copy of the attention main loop came out WITHOUT it (ROCm 7.2) -- a wave then read a K / V
// tile whose last pieces were still in flight (found with tools/race_probe_attn.py: 5-20 % of the launches at S = 4608 had a
// few 32-query slabs off by one tile's weight).  The explicit wait costs nothing where the compiler would have emitted it.
#define AFX_SYNC_DMA()                                   \
  do {                                                   \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     \
    __syncthreads();                                     \
  } while (0)

AFX_DEV float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16 on the gfx950 conversion unit (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN quieted --
// the same result as torch .to(bfloat16)).  A hand-rolled integer rounding costs ~10 VALU + a divergent
// NaN branch per element; in the attention loop that was the largest single VALU item.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

AFX_DEV uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

AFX_DEV bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

AFX_DEV void unpack8(const u32x4_t& w, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}

AFX_DEV u32x4_t pack8(const float (&f)[8]) {
  u32x4_t w;
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
  return w;
}

// ---- block-scaled fp8 (MX layout, E8M0 scale bytes; afx_text.hip quant_rows_mx8_kernel and the fused producers) -------------------------
// biased exponent b of the power-of-two scale of a block with absolute maximum amax: the smallest 2^(b - 127) with amax <= 448 * 2^(b - 127)
AFX_DEV int mx_exp(float amax) {
  const uint32_t u = __float_as_uint(amax * (1.0f / 448.0f));
  const int b = (int)((u >> 23) & 0xffu) + ((u & 0x7fffffu) != 0u ? 1 : 0);
  return min(max(b, 1), 254);
}
AFX_DEV float mx_inv(int b) { return __uint_as_float((uint32_t)(254 - b) << 23); }      // 2^(127 - b)
AFX_DEV void mx_pack8(const float (&v)[8], float inv, uint32_t& w0, uint32_t& w1) {     // 8 values -> 8 e4m3 bytes (saturating)
  float t[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = __builtin_amdgcn_fmed3f(v[e] * inv, 448.0f, -448.0f);
  int a = 0, b = 0;
  a = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], a, false);
  a = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], a, true);
  b = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], b, false);
  b = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], b, true);
  w0 = (uint32_t)a; w1 = (uint32_t)b;
}

AFX_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

AFX_DEV float gelu_tanh(float x) {
  // 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3)   ==   x sigmoid(2u) = x / (1 + exp2(-2 log2(e) u)):
  // 3 multiply-adds + v_exp_f32 + v_rcp_f32 (the tanh form with an IEEE division is ~25 instructions per value, and the GEMM
  // epilogue evaluates 256 values per lane: 12 % of a K = 3072 tile)
  const float z = x * (-2.3022081981f - 0.1029432397f * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
}

AFX_DEV float silu(float x) { return x / (1.0f + __expf(-x)); }

// XCD-aware bijective remap of a 1-D grid: block b runs on XCD b%8; give every XCD a contiguous
// chunk of logical work-group ids so neighbouring tiles share that XCD's L2.
AFX_DEV int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  const int q = nwg / nx, r = nwg % nx, x = bid % nx;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + bid / nx;
}

}  // namespace afx
