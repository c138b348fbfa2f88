// Joint (text+image) flash attention forward for gfx950, head_dim = 128, bf16 in/out, no mask.
//
// Both contractions are computed TRANSPOSED so that the softmax row is lane-local:
//     S^T = K . Q^T          (A operand = K rows,   B operand = Q rows)   D[key  ][query]
//     O^T = V^T . P^T        (A operand = V^T rows, B operand = P^T)      D[d    ][query]
// With v_mfma_f32_32x32x16_bf16 the D layout is  col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5),
// so lane (q, hi) owns query q of the wave's 32-query slab in BOTH products: the row max / sum, the
// rescale of O and the final 1/l are plain per-lane VALU work, no cross-lane traffic except one
// xor-32 exchange of the tile max.  The exponentiated scores of lane (q, hi) are exactly the B
// operand of the second product if the key order inside every 16-key group is permuted
//     position p = hi*8 + j   <->   key  4*hi + (j & 3) + 8*(j >> 2),
// and the sum over keys does not care about order -- so V is pre-transposed ONCE per layer into
// V^T[b][h][d][S_pad] with that permutation applied (launch_v_transpose), and the main kernel needs
// no transpose reads, no LDS round trip for P and no permlane traffic.
//
// Work-group = 4 waves x 32 queries; K tile [64 keys][128 d] and V^T tile [128 d][64 keys] stream
// HBM -> LDS by LDS-DMA into a 2-stage ring with an XOR swizzle (conflict-free ds_read_b128 reads).
//
// The kernel is a template <head dim, EXT>.  <128, false> is the MMDiT product path above.  EXT = true adds what the
// text encoders need (afx_text.hip): a runtime softmax scale, causal masking with tile skipping, an additive
// relative-position bias table (T5), grouped KV heads (Qwen2.5), and head dim 64 (T5, CLIP).
#include <cstdlib>

#include <hip/hip_ext.h>

#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {

constexpr int KVB = 64;            // keys per tile
constexpr int QW = 32;             // queries per wave
constexpr int ATT_WAVES = 4;
constexpr int ATT_THREADS = ATT_WAVES * 64;
constexpr int QB = ATT_WAVES * QW; // queries per work-group
constexpr float RESCALE_LOG2 = 5.0f;   // defer the O rescale until a row max grew by > 2^5

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// position p (0..15) inside a 16-key group -> key offset inside the group
AFX_DEV int key_of_pos(int p) { return 4 * (p >> 3) + (p & 3) + 8 * ((p >> 2) & 1); }

// ---------------------------------------------------------------------------------------------
// V [B*S rows, ldv] (head h at column h*128)  ->  Vt [B][H][128][S_pad], keys permuted per 16-group,
// keys >= S zero-filled.
template <int HD>
__global__ __launch_bounds__(256) void v_transpose_kernel(const bf16_t* __restrict__ v, int64_t ldv,
                                                          bf16_t* __restrict__ vt, int H, int S, int S_pad) {
  __shared__ bf16_t tile[KVB][HD + 2];
  const int kv0 = blockIdx.x * KVB, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  constexpr int CPR = HD / 8;                         // 16-byte chunks per row
#pragma unroll
  for (int i = 0; i < KVB * CPR / 256; ++i) {
    const int p = i * 256 + tid;
    const int r = p / CPR, c = p % CPR;
    u32x4_t w = (u32x4_t){0u, 0u, 0u, 0u};
    if (kv0 + r < S) w = *reinterpret_cast<const u32x4_t*>(v + ((int64_t)b * S + kv0 + r) * ldv + h * HD + c * 8);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&tile[r][c * 8]);   // (HD+2)*2 = 260 B rows: 4-byte aligned
    dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2]; dst[3] = w[3];
  }
  __syncthreads();
  bf16_t* out = vt + ((int64_t)(b * H + h) * HD) * S_pad + kv0;
#pragma unroll
  for (int i = 0; i < HD / 32; ++i) {
    const int d = i * 32 + (tid >> 3);
    const int c = tid & 7;                          // 8 consecutive storage positions c*8 .. c*8+7
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p0 = c * 8 + 2 * e, p1 = p0 + 1;
      const int k0 = (p0 >> 4) * 16 + key_of_pos(p0 & 15);
      const int k1 = (p1 >> 4) * 16 + key_of_pos(p1 & 15);
      w[e] = (uint32_t)tile[k0][d] | ((uint32_t)tile[k1][d] << 16);
    }
    *reinterpret_cast<u32x4_t*>(out + (int64_t)d * S_pad + c * 8) = (u32x4_t){w[0], w[1], w[2], w[3]};
  }
}

hipError_t launch_v_transpose(const uint16_t* v, int64_t ldv, uint16_t* vt, int B, int H, int S,
                              hipStream_t stream) {
  const int S_pad = (int)attn_spad(S);
  hipLaunchKernelGGL(v_transpose_kernel<128>, dim3(S_pad / KVB, H, B), dim3(256), 0, stream, v, ldv, vt, H, S, S_pad);
  return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// Operand preparation of one attention call in ONE launch (was three: k/q RMSNorm+RoPE, V transpose):
//   blocks [0, 2 nblk_k):  X <- RoPE(RMSNorm_128(X) * w) in place for X = K (first nblk_k blocks) and X = Q -- 16 lanes x 8
//                          elements = one (token, head), the 4 rotation pairs of a lane stay inside its own 16-byte chunk;
//   blocks [2 nblk_k, ..): V^T tiles as v_transpose_kernel<128>.
// Measured and dropped: the q part inside the attention prologue (QFuse, r02c): +23 us per attention launch (prologue VALU +
// 64 KB of cos/sin per work-group in front of the first MFMA, 1.69 rounds deep) against the 11 us it saves here.
struct PrepX { bf16_t* x[2]; const float* w_txt[2]; const float* w_img[2]; };

__global__ __launch_bounds__(256) void kv_prep_kernel(const PrepX px, int64_t ldk, const float* __restrict__ cos_t,
                                                      const float* __restrict__ sin_t, int n_txt, const bf16_t* __restrict__ v,
                                                      int64_t ldv, bf16_t* __restrict__ vt, int B, int H, int S, int S_pad,
                                                      int nblk_k) {
  constexpr int HD = 128;
  __shared__ bf16_t tile[KVB][HD + 2];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < 2 * nblk_k) {
    const int which = (int)blockIdx.x >= nblk_k;
    bf16_t* __restrict__ k = px.x[which];
    const float* __restrict__ w_txt = px.w_txt[which];
    const float* __restrict__ w_img = px.w_img[which];
    const int64_t g = (int64_t)((int)blockIdx.x - which * nblk_k) * 256 + tid;            // (row, head, chunk)
    const int64_t total = (int64_t)B * S * H * 16;
    if (g >= total) return;
    const int c = (int)(g & 15);
    const int64_t th = g >> 4;
    const int h = (int)(th % H);
    const int64_t row = th / H;
    const int s = (int)(row % S);
    bf16_t* p = k + row * ldk + h * HD + c * 8;
    float x[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(p), x);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rstd = rsqrtf(ss * (1.0f / 128.0f) + 1e-6f);
    const float* w = (s < n_txt ? w_txt : w_img) + c * 8;
    const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(w), w1 = *reinterpret_cast<const f32x4_t*>(w + 4);
    const float wv[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
    const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(cos_t + (int64_t)s * 64 + c * 4);
    const f32x4_t sn = *reinterpret_cast<const f32x4_t*>(sin_t + (int64_t)s * 64 + c * 4);
    float r[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = x[2 * i] * rstd * wv[2 * i];
      const float b = x[2 * i + 1] * rstd * wv[2 * i + 1];
      r[2 * i] = a * cs[i] - b * sn[i];
      r[2 * i + 1] = a * sn[i] + b * cs[i];
    }
    *reinterpret_cast<u32x4_t*>(p) = pack8(r);
    return;
  }
  const int vb = blockIdx.x - 2 * nblk_k;                          // (kv tile, head, batch)
  const int ntile = S_pad / KVB;
  const int kv0 = (vb % ntile) * KVB, h = (vb / ntile) % H, b = vb / (ntile * H);
  constexpr int CPR = HD / 8;
#pragma unroll
  for (int i = 0; i < KVB * CPR / 256; ++i) {
    const int p = i * 256 + tid;
    const int r = p / CPR, c = p % CPR;
    u32x4_t w = (u32x4_t){0u, 0u, 0u, 0u};
    if (kv0 + r < S) w = *reinterpret_cast<const u32x4_t*>(v + ((int64_t)b * S + kv0 + r) * ldv + h * HD + c * 8);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&tile[r][c * 8]);
    dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2]; dst[3] = w[3];
  }
  __syncthreads();
  bf16_t* out = vt + ((int64_t)(b * H + h) * HD) * S_pad + kv0;
#pragma unroll
  for (int i = 0; i < HD / 32; ++i) {
    const int d = i * 32 + (tid >> 3);
    const int c = tid & 7;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p0 = c * 8 + 2 * e, p1 = p0 + 1;
      const int k0 = (p0 >> 4) * 16 + key_of_pos(p0 & 15);
      const int k1 = (p1 >> 4) * 16 + key_of_pos(p1 & 15);
      w[e] = (uint32_t)tile[k0][d] | ((uint32_t)tile[k1][d] << 16);
    }
    *reinterpret_cast<u32x4_t*>(out + (int64_t)d * S_pad + c * 8) = (u32x4_t){w[0], w[1], w[2], w[3]};
  }
}

hipError_t launch_kv_prep(uint16_t* k, uint16_t* q, int64_t ldk, const float* wk_txt, const float* wk_img, const float* wq_txt,
                          const float* wq_img, const float* cos_t, const float* sin_t, int n_txt, const uint16_t* v, int64_t ldv,
                          uint16_t* vt, int B, int H, int S, hipStream_t stream) {
  const int S_pad = (int)attn_spad(S);
  const int nblk_k = (int)(((int64_t)B * S * H * 16 + 255) / 256);
  const int nblk_v = (S_pad / KVB) * H * B;
  PrepX px{};
  px.x[0] = k; px.w_txt[0] = wk_txt; px.w_img[0] = wk_img;
  px.x[1] = q; px.w_txt[1] = wq_txt; px.w_img[1] = wq_img;
  hipLaunchKernelGGL(kv_prep_kernel, dim3(2 * nblk_k + nblk_v), dim3(256), 0, stream, px, ldk, cos_t, sin_t, n_txt, v, ldv, vt, B, H, S,
                     S_pad, nblk_k);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
struct AttnExt {            // EXT = true only
  float scale;              // softmax scale (scores are multiplied by it)
  int causal;               // keys > query masked out
  const float* bias;        // [H][2 S - 1] additive bias, indexed by key - query + S - 1, already divided by scale; or nullptr
  int kv_group;             // query heads per KV head (1 = MHA)
};

#ifdef AFX_ATTN_TRACE
__device__ unsigned g_attn_trace4[2 * 4 * 4];
#endif
AFX_DEV uint64_t att_uniform_u64(uint64_t v) {      // a wave-uniform 64-bit value, in an SGPR pair
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
// LDS-DMA, 16 bytes per lane: M0 = LDS byte address of the wave's 1 KiB destination, global address = SGPR pair + 32-bit lane offset
#define ATT_DMA(BASE_U64, VOFF, LDS_PTR)                                                                                            \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"((uint32_t)(uintptr_t)(LDS_PTR)), "v"(VOFF), \
               "s"(BASE_U64)                                                                                                       \
               : "memory", "m0")     /* m0 is compiler-reserved: the entry only draws a warning, but it records the write for the */ \
                                     /* builtin LDS-DMA of the ragged tail (never leave the M0 write invisible)       */
template <int HD, bool EXT>
__global__ __launch_bounds__(ATT_THREADS, 2) void attention_kernel(
    const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k, int64_t ldk,
    const bf16_t* __restrict__ vt, bf16_t* __restrict__ o, int64_t ldo, int H, int S, int S_pad, int nq, int B,
    float* __restrict__ lse, const AttnExt ext) {
  constexpr int STAGE_BYTES = 2 * KVB * HD * 2;
  constexpr int KS = HD / 16;          // k-steps of the score product
  constexpr int DT = HD / 32;          // 32-row tiles of O^T
  constexpr int KROW = HD * 2;         // bytes per K row in LDS
  // two stages of { K tile [64][HD] | V^T tile [HD][64] }
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hi = lane >> 5;
  // 1-D grid, head -> XCD affinity: block id x runs on XCD x % 8 (observed dispatch order; speed only),
  // so XCD x owns heads x, x+8, x+16, ...: the K / V^T of the 1-2 heads an XCD works on at a time
  // (2.4 MB per head at S = 4608) stay resident in that XCD's private 4 MiB L2 while its q-tiles sweep them.
  const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
  const int per_head = nq * B;
  const int h = xcd + 8 * (slot_id / per_head);
  if (h >= H) return;
  const int rem = slot_id % per_head;
  const int b = rem / nq;
  const int q0 = (rem % nq) * QB + wave * QW;
  const int qrow = min(q0 + ql, S - 1);
  int ntiles = S_pad / KVB;
  int hk = h, Hk = H;
#ifdef AFX_ATTN_TRACE
  const unsigned tr_c0 = (unsigned)__builtin_readcyclecounter(), tr_r0 = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
  if (EXT) {
    hk = h / ext.kv_group;
    Hk = H / ext.kv_group;
    if (ext.causal) ntiles = min(ntiles, ((rem % nq) * QB + QB - 1) / KVB + 1);     // tiles right of the diagonal are skipped
  }

  const bf16_t* qp = q + ((int64_t)b * S + qrow) * ldq + h * HD;
  const bf16_t* kbase = k + (int64_t)b * S * ldk + hk * HD;
  const bf16_t* vbase = vt + ((int64_t)(b * Hk + hk) * HD) * S_pad;

  // Q^T fragments (B operand): query ql, d = 16*step + 8*hi .. +7
  bf16x8_t qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(qp + s * 16 + hi * 8);

  f32x16_t oacc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = EXT ? ext.scale * 1.4426950408889634f : 0.08838834764831845f * 1.4426950408889634f;   // 1/sqrt(128) * log2(e)

  // K / V^T tiles go HBM -> LDS by LDS-DMA (16 B per lane, lane-linear destination), double
  // buffered; the XOR swizzle is applied to the per-lane SOURCE chunk and undone by the readers.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // Per-lane byte offsets of the DT pieces of a K tile and a V^T tile relative to UNIFORM per-tile base pointers (SGPR base +
  // 32-bit VGPR offset: one v_add per DMA instead of 64-bit address arithmetic -- that arithmetic was a quarter of the loop's
  // VALU work).  K rows are KROW bytes: 256-byte rows swizzle chunk ^ (row & 15), 128-byte rows chunk ^ ((row >> 1) & 7).
  constexpr int KCPR = HD / 8;                       // 16-byte chunks per K row
  const int r0 = tid / KCPR, cp = tid % KCPR;        // K: row r0 + (256 / KCPR) i, physical chunk cp
  const int kswz = (HD == 128 ? (cp ^ (r0 & 15)) : (cp ^ ((r0 >> 1) & 7))) << 3;
  const int d0 = tid >> 3, vp = tid & 7;             // V^T: row d0 + 32 i, physical chunk vp
  const int vswz = (vp ^ ((d0 >> 1) & 7)) << 3;
  uint32_t koff[DT], voff[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) {
    koff[i] = (uint32_t)(((int64_t)(r0 + (256 / KCPR) * i) * ldk + kswz) * 2);
    voff[i] = (uint32_t)(((int64_t)(d0 + 32 * i) * S_pad + vswz) * 2);
  }
  const bool ragged = S_pad != S;
  auto stage_tile = [&](int t, int buf) {
    const int kv0 = t * KVB;
    char* kd = smem + buf * STAGE_BYTES;
    char* vd = kd + KVB * HD * 2;
    const char* kt = reinterpret_cast<const char*>(kbase) + (int64_t)kv0 * ldk * 2;     // uniform
    const char* vt_ = reinterpret_cast<const char*>(vbase) + (int64_t)kv0 * 2;
    if (ragged && t == ntiles - 1) {                 // keys past the end: clamp the row (never read unmasked)
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        const int kr = min(kv0 + r0 + (256 / KCPR) * i, S - 1);
        const bf16_t* ksrc = kbase + (int64_t)kr * ldk + kswz;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)ksrc, (lds_void_t*)(kd + (i * ATT_THREADS + wave_u * 64) * 16), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(vt_ + voff[i]), (lds_void_t*)(vd + (i * ATT_THREADS + wave_u * 64) * 16), 16, 0, 0);
      }
    } else {
      // SGPR base + 32-bit lane offset, from inline asm: inside the loop hipcc widens the builtin's address to a 64-bit VGPR pair
      // and a v_lshl_add_u64 per issue (8 VALU + 16 offset registers per tile in a loop bound by the issue port)
      const uint64_t kt_u = att_uniform_u64((uint64_t)(uintptr_t)kt), vt_u = att_uniform_u64((uint64_t)(uintptr_t)vt_);
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        ATT_DMA(kt_u, koff[i], kd + (i * ATT_THREADS + wave_u * 64) * 16);
        ATT_DMA(vt_u, voff[i], vd + (i * ATT_THREADS + wave_u * 64) * 16);
      }
    }
  };

  stage_tile(0, 0);
  if (ntiles > 1) stage_tile(1, 1);
  AFX_SYNC_DMA();
  // Pin the Q fragments as landed HERE: otherwise their pending global loads reach the loop header and
  // hipcc's conservative merge turns the first in-loop wait into vmcnt(0).
#pragma unroll
  for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qf[s]));

  // Software pipeline per KV tile t (buffer t&1):
  //   A  ds_read all 16 K fragments of tile t
  //   B  16 MFMAs  S^T = K Q^T
  //   C  ds_read all 16 V^T fragments of tile t                  [latency hides under the softmax VALU]
  //   D  online softmax in registers -> P^T fragments
  //   E  explicit s_waitcnt vmcnt(0) + barrier: tile t+1 landed, nobody reads buffer t&1 any more (AFX_SYNC_DMA: the
  //      compiler's own wait is NOT reliable here).  Measured alternatives (r02, S = 4608, H = 24): a second raw barrier +
  //      counted vmcnt(8) so that a tile stays in flight across one barrier: 374-380 us vs 335 us for this form -- with two
  //      independent work-groups per CU every extra barrier costs more in lost phase slack than the DMA wait it hides.
  //   F  LDS-DMA tile t+2 into buffer t&1                              [hides under G and the next B..D]
  //   G  16 MFMAs  O^T += V^T P^T
  bf16x8_t kf0[KS], kf1[KS], vf[DT][4];

  for (int t = 0; t < ntiles; ++t) {
    const char* ks = smem + (t & 1) * STAGE_BYTES;
    const char* vs = ks + KVB * HD * 2;
    // ---- A ----
    {
      const int krow = 32 + ql;
      const int sw0 = HD == 128 ? (ql & 15) : ((ql >> 1) & 7), sw1 = HD == 128 ? (krow & 15) : ((krow >> 1) & 7);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        kf0[s] = *reinterpret_cast<const bf16x8_t*>(ks + ql * KROW + (((s * 2 + hi) ^ sw0) << 4));
        kf1[s] = *reinterpret_cast<const bf16x8_t*>(ks + krow * KROW + (((s * 2 + hi) ^ sw1) << 4));
      }
    }
    // ---- B ----
    f32x16_t sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[s], qf[s], sacc[0], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[s], qf[s], sacc[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- C ----
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const int vrow = d * 32 + ql;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        vf[d][g] = *reinterpret_cast<const bf16x8_t*>(vs + vrow * 128 + (((g * 2 + hi) ^ ((vrow >> 1) & 7)) << 4));
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- D ---- mask keys past the end of the sequence (last tile only)
    if (t == ntiles - 1 && ragged) {
      asm volatile("" ::: "memory");       // keep this a wave-uniform BRANCH: as selects it costs 31 v_cndmask on every tile
      const int kv0 = t * KVB;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= S) sacc[kb][r] = -INFINITY;
        }
    }
    if (EXT) {               // relative-position bias and the causal mask, per score
      const int kv0 = t * KVB;
      const int qi = q0 + ql;
      const float* brow = ext.bias ? ext.bias + (int64_t)h * (2 * S - 1) + (S - 1) - qi : nullptr;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (brow != nullptr && key < S && qi < S) sacc[kb][r] += brow[key];
          if (ext.causal && key > qi) sacc[kb][r] = -INFINITY;
        }
    }
    // The softmax runs at raised issue priority: the loop is bound by the SIMD's issue port and this VALU chain is its critical
    // path; the other wave of the SIMD (another work-group) slips its MFMAs / reads into the gaps.  Measured in the forward (three
    // builds interleaved on one box): none 945 TF, softmax raised 958 TF, everything but the MFMA phases raised 903, MFMA phases
    // raised 870.
    __builtin_amdgcn_s_setprio(2);
    float mt = sacc[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sacc[kb][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    // Deferred rescale: the running max is only advanced (and O, l rescaled: 64 VALU per lane) when some
    // row of this wave grew by more than 2^RESCALE_LOG2; otherwise P is exponentiated against the stale max
    // (values <= 2^RESCALE_LOG2, harmless in bf16/fp32) and O needs no touch.  Exact algebra either way.
    const float m_new = fmaxf(m_run, mt);
    if (__any((m_new - m_run) * c > RESCALE_LOG2)) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
    }
    const float mc = m_run * c;
    // exponent arguments and row sums on value PAIRS (v_pk_fma_f32 / v_pk_add_f32: one issue slot per two values -- the loop is
    // bound by the SIMD's issue port, ~1700 issue cycles per wave and tile against 1024 matrix-pipe cycles)
    const f32x2_t c2 = {c, c}, nmc2 = {-mc, -mc};
    f32x2_t ps2 = {0.f, 0.f};
    bf16x8_t pf[4];          // [16-key group g = kb*2 + ksub]
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const f32x2_t s2 = {sacc[g >> 1][(g & 1) * 8 + j], sacc[g >> 1][(g & 1) * 8 + j + 1]};
        const f32x2_t a2 = __builtin_elementwise_fma(s2, c2, nmc2);
        const f32x2_t e2 = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
        ps2 += e2;
        pv[j] = e2[0];
        pv[j + 1] = e2[1];
      }
      const u32x4_t w = pack8(pv);
      pf[g] = __builtin_bit_cast(bf16x8_t, w);
    }
    l_run += ps2[0] + ps2[1];
    __builtin_amdgcn_s_setprio(0);
    // ---- E ----
    AFX_SYNC_DMA();
    // ---- F ----
    if (t + 2 < ntiles) stage_tile(t + 2, t & 1);
    // ---- G ----
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int d = 0; d < DT; ++d)
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[d][g], pf[g], oacc[d], 0, 0, 0);
  }

  // ---- normalise and store: lane (q, hi) holds O[q][32*d + 8*g + 4*hi + 0..3] -------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  // log2-domain log-sum-exp of the scaled scores: P = exp2(s * c - lse); consumed by the backward kernels
  if (lse != nullptr && hi == 0 && q0 + ql < S) lse[((int64_t)b * H + h) * S_pad + q0 + ql] = m_run * c + __log2f(l_tot);
  if (q0 + ql < S) {
    bf16_t* op = o + ((int64_t)b * S + q0 + ql) * ldo + h * HD;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t w;
        w[0] = pack_bf16x2(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv);
        w[1] = pack_bf16x2(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
        *reinterpret_cast<u32x2_t*>(op + d * 32 + g * 8 + hi * 4) = w;
      }
  }
#ifdef AFX_ATTN_TRACE
  if (!EXT && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 700)) {      // whole-work-group stamps: shader cycles, 100 MHz ticks, KV tiles
    unsigned* t4 = g_attn_trace4 + ((blockIdx.x ? 1 : 0) * 4 + wave) * 4;
    t4[0] = (unsigned)__builtin_readcyclecounter() - tr_c0;
    t4[1] = (unsigned)__builtin_amdgcn_s_memrealtime() - tr_r0;
    t4[2] = (unsigned)ntiles;
    t4[3] = tr_c0;
  }
#endif
}


// =================================================================================================================================
// 8-wave "ping-pong" flash attention (head dim 128, no mask) -- EXPERIMENTAL, opt-in with AFX_ATTN_IMPL=2.
// Status (r02, S = 4608, H = 24, tools/attn_trace.py stamps of one iteration): correct and race-free (same tests as the 4-wave
// kernel), 372 us vs 335 us for the 4-wave kernel.  Per wave and tile: 16 + 16 MFMAs issue in 770 + 690 cycles (45 per MFMA, not
// 32: the partner's v_exp / VALU issues block the shared issue port), the softmax takes 1300-1470 cycles beside the partner's
// MFMAs, the V^T reads + DMA issue of the group that reaches the barrier last another 700-900: a half step lasts ~2030 cycles
// where the matrix pipe needs 1024.  What it would take: hand-placed VALU fillers inside the MFMA gaps (<= 5 issues per gap,
// MI355X_MICROARCH "one wave per SIMD") instead of two compiler-scheduled streams arbitrating for one issue port.
//
// Same per-wave mathematics and operand layouts as attention_kernel above (S^T = K Q^T and O^T = V^T P^T on the 32x32x16 MFMA,
// lane-local softmax rows, key-permuted V^T), but ONE work-group of 8 waves x 32 queries per CU instead of two independent
// 4-wave work-groups: the two wave groups of a SIMD run the SAME loop half a KV tile apart, so that one issues its 32 MFMAs of a
// tile (O^T += V^T P^T of tile t-1, then S^T of tile t: 1024 matrix-pipe cycles) exactly while its partner does the online softmax
// of its own tile (~1000 VALU cycles, 32 v_exp) -- the matrix pipe and the VALU of every SIMD are both busy all the time instead
// of whenever two unsynchronised work-groups happen to be out of phase (measured: the 4-wave kernel keeps the matrix pipe 50 %
// busy at best).  One raw s_barrier per half step keeps the stagger; group 1 enters the loop one barrier late.
//   M(t):  ds_read K(t) fragments | 16 MFMA  O^T += V^T(t-1) P^T(t-1) | 16 MFMA  S^T = K(t) Q^T | ds_read V^T(t) fragments
//   S(t):  LDS-DMA of tile t+3 | mask / max / exp2 / row sum -> P^T(t) | counted wait
// K / V^T tiles are shared by all 8 waves (half the L2 -> LDS traffic per query of the 4-wave kernel) and live in a 4-slot LDS
// ring (4 x 32 KiB): the DMA of tile t+3 is issued in S(t) and must have landed before M(t+3) -- 2.5 tile times later, against
// 1.0 with the two-slot ring.  Waits are counted and explicit (never __syncthreads(), see AFX_SYNC_DMA): at the end of S(t)
// s_waitcnt vmcnt(4) leaves only tile t+3 in flight, so when group 0 enters M(t+1) group 1 (whose last check was S(t-1)) has
// confirmed tiles <= t+1 and group 0 tiles <= t+2.  A slot is refilled one barrier after its last reader's ds_reads retired
// (lgkmcnt(0) before every barrier).  Past the last tile the DMA re-fetches tile ntiles-1 into a free slot: the count per
// iteration stays 4 ops per lane.
constexpr int PP_WAVES = 8;
constexpr int PP_THREADS = PP_WAVES * 64;
constexpr int PP_QB = PP_WAVES * QW;                 // 256 queries per work-group
constexpr int PP_SLOT_BYTES = 2 * KVB * 128 * 2;     // K tile [64][128] | V^T tile [128][64]
constexpr int PP_SLOTS = 4;
constexpr int PP_LDS_BYTES = PP_SLOTS * PP_SLOT_BYTES;

#ifndef AFX_PP_PRIO
#define AFX_PP_PRIO 1
#endif
#if AFX_PP_PRIO
#define PP_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define PP_PRIO(x)
#endif
__global__ __launch_bounds__(PP_THREADS, 2) void attention_pp_kernel(
    const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k, int64_t ldk,
    const bf16_t* __restrict__ vt, bf16_t* __restrict__ o, int64_t ldo, int H, int S, int S_pad, int nq, int B,
    float* __restrict__ lse, unsigned* __restrict__ trace) {
  constexpr int HD = 128, KS = 8, DT = 4, KROW = 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef AFX_ATTN_TRACE
  unsigned tr[12];
#define PP_TR(i) if (t == 36) tr[i] = (unsigned)__builtin_readcyclecounter();
#else
#define PP_TR(i)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave_u >> 2;
  const int ql = lane & 31, hi = lane >> 5;
  const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
  const int per_head = nq * B;
  const int h = xcd + 8 * (slot_id / per_head);
  if (h >= H) return;
  const int rem = slot_id % per_head;
  const int b = rem / nq;
  const int q0 = (rem % nq) * PP_QB + wave_u * QW;
  const int qrow = min(q0 + ql, S - 1);
  const int ntiles = S_pad / KVB;

  const bf16_t* qp = q + ((int64_t)b * S + qrow) * ldq + h * HD;
  const bf16_t* kbase = k + (int64_t)b * S * ldk + h * HD;
  const bf16_t* vbase = vt + ((int64_t)(b * H + h) * HD) * S_pad;

  bf16x8_t qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(qp + s * 16 + hi * 8);

  // per-lane DMA offsets: K chunk p = i*512 + tid -> row tid/16 + 32 i, physical chunk tid%16 (swizzle chunk ^ (row & 15));
  //                       V^T chunk p = i*512 + tid -> row tid/8 + 64 i, physical chunk tid%8 (swizzle chunk ^ ((row >> 1) & 7))
  const int r0 = tid >> 4, cp = tid & 15;
  const int kswz = (cp ^ (r0 & 15)) << 3;
  const int d0 = tid >> 3, vp = tid & 7;
  const int vswz = (vp ^ ((d0 >> 1) & 7)) << 3;
  uint32_t koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    koff[i] = (uint32_t)(((int64_t)(r0 + 32 * i) * ldk + kswz) * 2);
    voff[i] = (uint32_t)(((int64_t)(d0 + 64 * i) * S_pad + vswz) * 2);
  }
  const bool ragged = S_pad != S;
  auto stage_tile = [&](int t, int slot) {
    t = t < ntiles ? t : ntiles - 1;
    const int kv0 = t * KVB;
    char* kd = smem + slot * PP_SLOT_BYTES;
    char* vd = kd + KVB * HD * 2;
    const char* kt = reinterpret_cast<const char*>(kbase) + (int64_t)kv0 * ldk * 2;     // uniform
    const char* vt_ = reinterpret_cast<const char*>(vbase) + (int64_t)kv0 * 2;
    if (ragged && t == ntiles - 1) {                 // keys past the end: clamp the row (never read unmasked)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kr = min(kv0 + r0 + 32 * i, S - 1);
        const bf16_t* ksrc = kbase + (int64_t)kr * ldk + kswz;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)ksrc, (lds_void_t*)(kd + (i * PP_THREADS + wave_u * 64) * 16), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(vt_ + voff[i]), (lds_void_t*)(vd + (i * PP_THREADS + wave_u * 64) * 16), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(kt + koff[i]), (lds_void_t*)(kd + (i * PP_THREADS + wave_u * 64) * 16), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(vt_ + voff[i]), (lds_void_t*)(vd + (i * PP_THREADS + wave_u * 64) * 16), 16, 0, 0);
      }
    }
  };

  stage_tile(0, 0);
  stage_tile(1, 1);
  stage_tile(2, 2);
  // the Q fragments (8 oldest VMEM ops) are pinned as landed here; the three tiles stay in flight
#pragma unroll
  for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qf[s]));
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // tiles 0 and 1 (this wave's pieces)
  __builtin_amdgcn_s_barrier();                            // ... and everybody else's
  if (grp == 1) __builtin_amdgcn_s_barrier();              // group 1 runs half a tile behind group 0

  f32x16_t oacc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = 0.08838834764831845f * 1.4426950408889634f;   // 1/sqrt(128) * log2(e)
  bf16x8_t kf0[KS], kf1[KS], vf[DT][4], pf[4];
  {
    const u32x4_t z = (u32x4_t){0u, 0u, 0u, 0u};           // M(0) multiplies a zero P^T with a zero V^T: no branch in the loop
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      pf[g] = __builtin_bit_cast(bf16x8_t, z);
#pragma unroll
      for (int d = 0; d < DT; ++d) vf[d][g] = __builtin_bit_cast(bf16x8_t, z);
    }
  }
  const int krow1 = 32 + ql;
  const int sw0 = ql & 15, sw1 = krow1 & 15;

  for (int t = 0; t < ntiles; ++t) {
    const char* ks = smem + (t & 3) * PP_SLOT_BYTES;
    const char* vs = ks + KVB * HD * 2;
    // ================================ M(t) ================================
    PP_TR(0)
#pragma unroll
    for (int s = 0; s < KS; ++s) kf0[s] = *reinterpret_cast<const bf16x8_t*>(ks + ql * KROW + (((s * 2 + hi) ^ sw0) << 4));
    __builtin_amdgcn_sched_barrier(0);
    PP_PRIO(1);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int d = 0; d < DT; ++d)
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[d][g], pf[g], oacc[d], 0, 0, 0);
    PP_PRIO(0);
    __builtin_amdgcn_sched_barrier(0);
    PP_TR(1)
#pragma unroll
    for (int s = 0; s < KS; ++s) kf1[s] = *reinterpret_cast<const bf16x8_t*>(ks + krow1 * KROW + (((s * 2 + hi) ^ sw1) << 4));
    f32x16_t sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
    PP_PRIO(1);
#pragma unroll
    for (int s = 0; s < KS; ++s) {      // the two accumulator chains interleaved: a dependent 32x32 MFMA issues every ~44 cycles, an independent one every 32
      sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[s], qf[s], sacc[0], 0, 0, 0);
      sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[s], qf[s], sacc[1], 0, 0, 0);
    }
    PP_PRIO(0);
    __builtin_amdgcn_sched_barrier(0);
    PP_TR(2)
    // V^T(t) fragments for the NEXT M step: issued here, behind the MFMAs (the matrix pipe drains its queue meanwhile), they
    // complete under this wave's softmax; the registers were released by O^T += V^T P^T(t-1) at the top of this step
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const int vrow = d * 32 + ql;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        vf[d][g] = *reinterpret_cast<const bf16x8_t*>(vs + vrow * 128 + (((g * 2 + hi) ^ ((vrow >> 1) & 7)) << 4));
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PP_TR(3)
    // ================================ S(t) ================================
    stage_tile(t + 3, (t + 3) & 3);                        // slot of tile t-1: its last readers retired their ds_reads before the barrier above
    __builtin_amdgcn_sched_barrier(0);
    PP_TR(4)
    if (t == ntiles - 1 && ragged) {
      asm volatile("" ::: "memory");       // keep this a wave-uniform BRANCH
      const int kv0 = t * KVB;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= S) sacc[kb][r] = -INFINITY;
        }
    }
    float mt = sacc[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sacc[kb][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    if (__any((m_new - m_run) * c > RESCALE_LOG2)) {       // deferred rescale, as in attention_kernel
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
    }
    const float mc = m_run * c;
    float psum = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pv[j] = __builtin_amdgcn_exp2f(sacc[g >> 1][(g & 1) * 8 + j] * c - mc);
        psum += pv[j];
      }
      const u32x4_t w = pack8(pv);
      pf[g] = __builtin_bit_cast(bf16x8_t, w);
    }
    l_run += psum;
    // Pin P^T and the row sum as COMPUTED here: they are only consumed by the next M step, and without an opaque use the
    // compiler sinks all 32 v_exp + the packing below the barrier -- into the partner's VALU half step (measured: the M
    // half step then lasts 2190 cycles instead of 1330 and the ping-pong degenerates).
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(pf[g]));
    asm volatile("" : "+v"(l_run));
    __builtin_amdgcn_sched_barrier(0);
    PP_TR(5)
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");   // only tile t+3 may still be in flight; the V^T reads retired
    PP_TR(6)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PP_TR(7)
  }
#ifdef AFX_ATTN_TRACE
  if (trace != nullptr && (blockIdx.x == 0 || blockIdx.x == 1000) && lane == 0)
    for (int i = 0; i < 8; ++i) trace[((blockIdx.x ? 1 : 0) * 8 + wave_u) * 8 + i] = tr[i];
#endif
  if (grp == 0) __builtin_amdgcn_s_barrier();              // pairs with group 1's extra barrier in front of the loop
  // ---- the last tile's O^T += V^T P^T, drain the re-fetches of the last tile ----
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int d = 0; d < DT; ++d)
      oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[d][g], pf[g], oacc[d], 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (lse != nullptr && hi == 0 && q0 + ql < S) lse[((int64_t)b * H + h) * S_pad + q0 + ql] = m_run * c + __log2f(l_tot);
  if (q0 + ql < S) {
    bf16_t* op = o + ((int64_t)b * S + q0 + ql) * ldo + h * HD;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t w;
        w[0] = pack_bf16x2(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv);
        w[1] = pack_bf16x2(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
        *reinterpret_cast<u32x2_t*>(op + d * 32 + g * 8 + hi * 4) = w;
      }
  }
}

// cycle stamps of one iteration of two work-groups (builds with -DAFX_ATTN_TRACE only: tools/attn_trace.py)
static unsigned* pp_trace_buffer() {
#ifdef AFX_ATTN_TRACE
  static unsigned* buf = nullptr;
  if (!buf && hipMalloc(&buf, 2 * 8 * 8 * sizeof(unsigned)) != hipSuccess) buf = nullptr;
  return buf;
#else
  return nullptr;
#endif
}
extern "C" int afx_debug_attn_trace4(unsigned* host_out) {      // 4-wave kernel: [2 blocks][4 waves][cycles, 100 MHz ticks, tiles, start]
#ifdef AFX_ATTN_TRACE
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_trace4), 2 * 4 * 4 * sizeof(unsigned)) == hipSuccess ? 0 : -1;
#else
  (void)host_out;
  return -1;
#endif
}
extern "C" int afx_debug_attn_trace(unsigned* host_out) {
#ifdef AFX_ATTN_TRACE
  unsigned* b = pp_trace_buffer();
  if (!b) return -1;
  return hipMemcpy(host_out, b, 2 * 8 * 8 * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
#else
  (void)host_out;
  return -1;
#endif
}

static int& attn_impl() {
  static int impl = -1;
  return impl;
}
void attn_set_impl(int impl) { attn_impl() = (impl >= 0 && impl <= 3) ? impl : 0; }

hipError_t launch_attention(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
                            const uint16_t* vt, uint16_t* o, int64_t ldo, int B, int H, int S,
                            hipStream_t stream, float* lse, const AttnMx8* mx8, bool* fused) {
  if (fused) *fused = false;
  const int S_pad = (int)attn_spad(S);
  const int heads_per_xcd = (H + 7) / 8;
  int& impl = attn_impl();
  static bool attr = false;
  if (impl < 0) {
    // AFX_ATTN_IMPL / attn_set_impl: 0 (default) = the one-wave-per-SIMD kernel (afx_attn3.hip) where eligible, else the 4-wave kernel;
    // 1 = 4-wave kernel always; 2 = the 8-wave ping-pong kernel (experimental, see its header); 3 = as 0 on the plain grid (no KV-split of the last round)
    const char* e = getenv("AFX_ATTN_IMPL");
    impl = (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 0;
  }
  if (!attr) {
    hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
    if (r != hipSuccess) return r;
    attr = true;
  }
  if ((impl == 0 || impl == 3) && attention_v3_eligible(S)) {
    const bool f = mx8 != nullptr && fused != nullptr && mx8->o8 != nullptr && mx8->mx != nullptr;
    if (f) *fused = true;
    return launch_attention_v3(q, ldq, k, ldk, vt, o, ldo, B, H, S, stream, lse, f ? mx8 : nullptr, impl == 0);
  }
  if (impl == 2 && S >= 2 * PP_QB) {
    const int nq8 = (S + PP_QB - 1) / PP_QB;
    dim3 grid8(8 * heads_per_xcd * nq8 * B);
    if (launch_timer().start != nullptr && launch_timer().stop != nullptr)
      hipExtLaunchKernelGGL(attention_pp_kernel, grid8, dim3(PP_THREADS), PP_LDS_BYTES, stream, launch_timer().start, launch_timer().stop, 0, q,
                            ldq, k, ldk, vt, o, ldo, H, S, S_pad, nq8, B, lse, (unsigned*)nullptr);
    else
      hipLaunchKernelGGL(attention_pp_kernel, grid8, dim3(PP_THREADS), PP_LDS_BYTES, stream, q, ldq, k, ldk, vt, o, ldo, H, S, S_pad, nq8, B, lse,
                         pp_trace_buffer());
    return hipGetLastError();
  }
  const int nq = (S + QB - 1) / QB;
  dim3 grid(8 * heads_per_xcd * nq * B);
  if (launch_timer().start != nullptr && launch_timer().stop != nullptr)
    hipExtLaunchKernelGGL((attention_kernel<128, false>), grid, dim3(ATT_THREADS), 0, stream, launch_timer().start, launch_timer().stop,
                          0, q, ldq, k, ldk, vt, o, ldo, H, S, S_pad, nq, B, lse, AttnExt{});
  else
    hipLaunchKernelGGL((attention_kernel<128, false>), grid, dim3(ATT_THREADS), 0, stream, q, ldq, k, ldk, vt, o, ldo, H, S, S_pad,
                       nq, B, lse, AttnExt{});
  return hipGetLastError();
}

// Text-encoder attention (afx_text.hip): V [B*S, ldv] with Hkv heads is transposed into vt_ws [B][Hkv][HD][S_pad] first.
hipError_t launch_attention_ext(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv,
                                uint16_t* vt_ws, uint16_t* o, int64_t ldo, int B, int H, int Hkv, int S, int head_dim, float scale,
                                int causal, const float* bias, hipStream_t stream) {
  const int S_pad = (int)attn_spad(S);
  const int nq = (S + QB - 1) / QB;
  dim3 grid(8 * ((H + 7) / 8) * nq * B);
  const AttnExt ext{scale, causal, bias, H / Hkv};
  if (head_dim == 128) {
    hipLaunchKernelGGL(v_transpose_kernel<128>, dim3(S_pad / KVB, Hkv, B), dim3(256), 0, stream, v, ldv, vt_ws, Hkv, S, S_pad);
    hipLaunchKernelGGL((attention_kernel<128, true>), grid, dim3(ATT_THREADS), 0, stream, q, ldq, k, ldk, vt_ws, o, ldo, H, S, S_pad,
                       nq, B, (float*)nullptr, ext);
  } else {
    hipLaunchKernelGGL(v_transpose_kernel<64>, dim3(S_pad / KVB, Hkv, B), dim3(256), 0, stream, v, ldv, vt_ws, Hkv, S, S_pad);
    hipLaunchKernelGGL((attention_kernel<64, true>), grid, dim3(ATT_THREADS), 0, stream, q, ldq, k, ldk, vt_ws, o, ldo, H, S, S_pad,
                       nq, B, (float*)nullptr, ext);
  }
  return hipGetLastError();
}

}  // namespace afx
