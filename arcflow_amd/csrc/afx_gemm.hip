// Grouped bf16 GEMM for gfx950:  C = epi(A . W^T + bias),  A [M,K] and W [N,K] both K-contiguous
// (the nn.Linear layout), fp32 accumulation on v_mfma_f32_16x16x32_bf16.
//
// Two kernels share the tile geometry (the 8-phase v2 is the product path; the plain 2-stage kernel stays as the
// simple reference for A/B runs, AFX_GEMM_IMPL=1).  Variants that were measured and dropped: the v2 schedule on
// v_mfma_f32_32x32x16_bf16 (-12 %: the 2-deep accumulator chains stall), 2 phases of 32 MFMAs per K-tile (+3 % on
// cache-resident operands but -7 % in the real forward, where the halved DMA lead time meets HBM latency), LDS reads
// levelled 8/4/8/4 over the phases (0 %), one barrier per K-tile with every wave interleaving its own ds_reads / DMA with
// its MFMAs (-21 %: the vmcnt(0) in front of the barrier drains the DMA queue), k-quarter phases of 8 independent
// v_mfma_f32_32x32x16_bf16 with 32-byte-row DMA pieces (-36 %: 4x the L2 requests per byte).
// Cycle budget of one 256x256 tile, K = 3072 (tools/gemm_trace.hip, s_memtime): prologue 3.2 k, main loop 118 k
// (2464 per K-tile; 2048 = 128 MFMAs x 16 cycles is the floor), epilogue 7.9 k.
// Structure (wave64, 8 waves = 2(M) x 4(N), 256x256x64 tile, one work-group per CU):
//   * HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane): no VGPR round trip.  The DMA
//     writes LDS lane-linearly, so the bank-conflict swizzle is applied to the per-lane SOURCE
//     address and again on the ds_read_b128 side (same involution): 16-byte chunk c of tile row r
//     lives at chunk c ^ ((r >> 1) & 7) of its 128-byte LDS row -> every 16-lane ds_read_b128
//     group hits 16 distinct 16-byte slots of the 256-byte bank row.
//   * two LDS stages (2 x 64 KiB): tile t+1 streams in while tile t feeds 64 MFMAs per wave.
//   * epilogue: the 8-phase kernel accumulates C^T (operands swapped) and pairs lanes with v_permlane16_swap, so bias /
//     GELU / gate*x+residual and the C store are 16 B per lane straight from registers (12.1 k -> 7.9 k cycles against
//     the LDS transpose the simple kernel still uses).
//   * several problems (image + text stream, or per-sample slices) share one launch; the 1-D
//     grid is remapped so every XCD owns a contiguous run of tiles (private-L2 reuse of A/W panels).
#include <cstdlib>

#include <hip/hip_ext.h>

#include <type_traits>
#include <utility>

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

// Row-contiguous part of the epilogue, shared by every kernel variant: each lane owns 8 consecutive
// columns of 8 rows of the wave's 64x64 fp32 patch; bias / GELU / gate*x+residual are applied here and C is
// stored 16 B per lane (bf16) or 2 x 16 B (fp32 output, optionally accumulated -- weight gradients).
AFX_DEV void epi_store_rows(const GemmProblem& P, const float* patch, int row0, int gcol, bool col_ok, int er, int ec,
                            const float (&bias)[8]) {
#pragma unroll
  for (int ps = 0; ps < 8; ++ps) {
    const int lr = ps * 8 + er;
    const int grow = row0 + lr;
    if (grow < P.M && col_ok) {
      const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(patch + lr * EPI_LD + ec);
      const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(patch + lr * EPI_LD + ec + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bias[e];
      if (P.pre != nullptr) {
        float pr[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(P.pre + (int64_t)grow * P.ldp + gcol), pr);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += pr[e];
      }
      if (P.epi == EPI_GELU) {
        if (gcol >= P.gelu_col0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
        }
      } else if (P.epi == EPI_GATE_RES) {
        float g[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};      // gate == nullptr: plain residual add
        if (P.gate != nullptr) {
          const float* gp = P.gate + (int64_t)(grow / P.rows_per_batch) * P.ldg + gcol;
          const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(gp);
          const f32x4_t g1 = *reinterpret_cast<const f32x4_t*>(gp + 4);
          g[0] = g0[0]; g[1] = g0[1]; g[2] = g0[2]; g[3] = g0[3]; g[4] = g1[0]; g[5] = g1[1]; g[6] = g1[2]; g[7] = g1[3];
        }
        const u32x4_t rw = *reinterpret_cast<const u32x4_t*>(P.res + (int64_t)grow * P.ldr + gcol);
        float rr[8];
        unpack8(rw, rr);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rr[e] + g[e] * v[e];
      }
      if (P.conv_wp > 0) {        // convolution on the padded grid: keep the 1-pixel border zero for the next layer
        const int yy = grow / P.conv_wp, xx = grow - yy * P.conv_wp;
        if (yy == 0 || yy == P.conv_hp - 1 || xx == 0 || xx == P.conv_wp - 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
      }
      if (P.out_f32 == 0) {
        *reinterpret_cast<u32x4_t*>(P.C + (int64_t)grow * P.ldc + gcol) = pack8(v);
      } else {
        float* cp = reinterpret_cast<float*>(P.C) + (int64_t)grow * P.ldc + gcol;
        f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        if (P.out_f32 == 2) {
          o0 += *reinterpret_cast<const f32x4_t*>(cp);
          o1 += *reinterpret_cast<const f32x4_t*>(cp + 4);
        }
        *reinterpret_cast<f32x4_t*>(cp) = o0;
        *reinterpret_cast<f32x4_t*>(cp + 4) = o1;
      }
    }
  }
}

// LDS-free epilogue of the 8-phase kernel.  Its MFMAs are issued with the operands swapped (W fragment as "A"), so the
// accumulator tile is C^T: lane (frow, fq) of acc[ii][jj] holds C[row ii*16 + frow][cols jj*16 + fq*4 .. +3].  One
// v_permlane16_swap per register between the column tiles jj, jj+1 pairs the 16-lane groups fq, fq^1: afterwards an even-fq
// lane owns 8 consecutive columns of tile jj, an odd-fq lane 8 of tile jj+1 -> every bias / gate / residual access and the
// store are 16 bytes per lane (64 contiguous bytes per row and instruction), no LDS round trip, no barrier.
//
// epi_store_fast<EPI>: the bf16-output modes of the forward (bias, bias + GELU, gate * x + residual) as straight-line code.  The
// generic body below re-reads its GemmProblem fields from the kernel-argument segment inside every one of its 16 unrolled
// (row tile, column pair) steps (hipcc rematerialises the s_load instead of holding ~40 SGPRs) and waits lgkmcnt(0) behind each:
// ~12.7 k cycles per tile, 9.5 % of a K = 3072 tile (tools/gemm_trace.hip).  Here every uniform field is read ONCE, the batch
// index of a row is a float multiply + fix-up instead of an integer division, and the residual / gate loads of row tile ii + 1
// are issued before row tile ii is converted and stored (the residual may alias C: a lane reads exactly the 16 bytes it writes).
// A raw buffer descriptor over [p, p + bytes) built from values forced into SGPRs: if hipcc cannot prove the descriptor
// wave-uniform it wraps every buffer access in a readfirstlane "waterfall" loop (12 instructions + a branch per store).
template <typename T>
AFX_DEV __amdgpu_buffer_rsrc_t uniform_rsrc(T* p, int bytes) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)p >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T*>(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// SWAP: pair the column tiles with v_permlane16_swap so that a lane owns 8 consecutive columns (16-byte accesses, 64 contiguous
// bytes per row and instruction) -- or not: 4 consecutive columns per lane straight from the C^T accumulator, 8-byte accesses,
// twice the memory instructions but no cross-lane exchange in the dependency chain (the one-wave-per-SIMD kernel has no partner
// wave to hide that chain's latency behind).
// FP8: the product is scaled by a_scale[row] * w_scale[col] first (row-wise activation, per-output-channel weight scales);
// PRE: a bf16 [M, N] term (GemmProblem::pre, the LoRA-dropout correction) is added before the activation / gate.
// CONV: the rows are pixels of a zero-bordered [conv_hp][conv_wp] grid (implicit 3x3 convolution): border pixels are stored as zero.
constexpr int EPI_GELU_ALL = 3;      // (internal) EPI_GELU with every column of the wave at or past gelu_col0: no per-lane test
AFX_DEV uint32_t drop_mix32(uint32_t h) {          // (= mix32 of afx_train.hip's lora_dropout_kernel: the masks must be the same bits)
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
// Compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) -- the index is a constant expression inside the body
// (the epilogues name accumulator tiles by it: AccLit below needs the register NUMBER in the instruction text).
template <class F, int... I>
AFX_DEV void static_for_(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
AFX_DEV void static_for(F&& f) { static_for_(f, std::make_integer_sequence<int, N>{}); }

// Where a wave's fp32 accumulator tiles live, as the epilogues see them: tile (II, JJ) = 4 floats per lane.
//   AccArr: a C++ array hipcc allocates (the 8-phase kernel, the fp8 kernel);
//   AccLit: ASM-OWNED accumulator registers, tile (II, JJ) = a[BASE + 4 (II NJ + JJ) : + 3] by literal name (gemm_kernel_v3, round 6): hipcc does not know these
//           registers exist, so a read is an `asm volatile` (ordered with the kernel's MFMA statements, which are volatile too) and the MFMA -> read wait
//           states are the caller's business.
template <int MI, int NJ>
struct AccArr {
  f32x4_t (&a)[MI][NJ];
  template <int II, int JJ>
  AFX_DEV void tile(float (&c)[4]) const {
#pragma unroll
    for (int e = 0; e < 4; ++e) c[e] = a[II][JJ][e];
  }
};
template <int NJ, int BASE = 0>
struct AccLit {
  template <int II, int JJ>
  AFX_DEV void tile(float (&c)[4]) const {
    constexpr int R = BASE + 4 * (II * NJ + JJ);
    asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"
                 : "=v"(c[0]), "=v"(c[1]), "=v"(c[2]), "=v"(c[3])
                 : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));
  }
};
#define AFX_INL __attribute__((always_inline))

// Tail hook (round 6, gemm_kernel_v3's V3_TAIL): the LAST K-tile's MFMAs are issued from inside the epilogue, row tile by row tile -- hk(-1) behind the
// epilogue's preamble (row tile 0, under the latency of the bias / residual / gate requests), then ONE MFMA of row tile ii + 1 at each of the 4 NS "points" of
// row tile ii's steps (behind the exchange, behind the bias add, behind the activation, behind the store), so the matrix pipe works under the epilogue's VALU
// and store issue instead of in front of it.  Point code = ii * (4 NS) + 4 st + q.  hk.all() = every remaining MFMA at once (epilogues without points).
struct NoHook {
  template <int CODE>
  AFX_DEV void at() const {}
  AFX_DEV void all() const {}
};
template <class F>
struct TailHook {
  F f;                                             // a generic lambda taking std::integral_constant<int, CODE>
  template <int CODE>
  AFX_DEV void at() const { f(std::integral_constant<int, CODE>{}); }
  AFX_DEV void all() const { f(std::integral_constant<int, -2>{}); }
};
template <int EPI, int MI, int NJ, bool SWAP, bool FP8, bool PRE, bool ROWB, bool CONV, bool GN, bool DROP, class HK, class ACC>
AFX_DEV void epi_store_fast_acc(const GemmProblem& P, const ACC& acc, int row_base, int col_base, int frow, int fq, const HK& hk) {
  constexpr int CW = SWAP ? 8 : 4;             // columns per lane and step
  constexpr int NS = SWAP ? (NJ + 1) / 2 : NJ; // steps per row tile (SWAP with an odd NJ: the last step pairs the lone column tile with
                                               // nothing -- the lanes that would own the missing tile's columns are masked out)
  constexpr uint32_t OOB = 0x80000000u;        // a byte offset past every buffer below: the hardware drops the access
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
  const int M = P.M, N = P.N;
  // Raw buffer descriptors over the wave's MI*16 rows of C / the residual (num_records = the rows that exist): rows >= M and
  // masked columns fall outside and are dropped / read as zero by the bounds check -- no exec-mask branch per step, so the whole
  // epilogue is ONE basic block the scheduler can interleave (a taken branch costs a lone wave ~30 cycles of refetch).
  const int rows_ok = min(max(M - row_base, 0), MI * 16);
  const int64_t ldc = P.ldc;
  const int up = CONV ? P.up_phase : 0;                           // (uniform) folded upsample: rows are scattered over the whole 2x grid
  __amdgpu_buffer_rsrc_t rc = (CONV && up != 0) ? uniform_rsrc(P.C, (int)((int64_t)(2 * P.conv_hp - 2) * (2 * P.conv_wp - 2) * ldc * 2))
                                                : uniform_rsrc(P.C + (int64_t)row_base * ldc, (int)(rows_ok * ldc * 2));
  const int ldc2 = (int)(ldc * 2);
  int ldr2 = 0, rpb = 1, ldg4 = 0;
  float inv_rpb = 1.f;
  bool has_gate = false;
  __amdgpu_buffer_rsrc_t rr_ = rc, rg_ = rc;
  if constexpr (EPI == EPI_GATE_RES) {
    const int64_t ldr = P.ldr;
    ldr2 = (int)(ldr * 2);
    rr_ = uniform_rsrc(const_cast<uint16_t*>(P.res) + (int64_t)row_base * ldr, (int)(rows_ok * ldr * 2));
    rpb = P.rows_per_batch;
    inv_rpb = 1.0f / (float)rpb;
    has_gate = P.gate != nullptr;
    ldg4 = (int)(P.ldg * 4);
    const int nb = (M + rpb - 1) / rpb;                           // gate rows
    rg_ = uniform_rsrc(const_cast<float*>(P.gate), has_gate ? (int)(((nb - 1) * P.ldg + N) * 4) : 0);   // (ldg may be 0: one gate row)
  }
  const int gelu_col0 = EPI == EPI_GELU ? P.gelu_col0 : 0;
  int gcol[NS];
  uint32_t coff[NS];             // byte offset of the lane's CW columns in a bf16 row, or OOB
  float bias[NS][CW];
  float wsc[FP8 ? NS : 1][CW];
  __amdgpu_buffer_rsrc_t rp_ = rc, ra_ = rc;
  int ldp2 = 0;
  if constexpr (PRE) {
    const int64_t ldp = P.ldp;
    ldp2 = (int)(ldp * 2);
    rp_ = uniform_rsrc(const_cast<uint16_t*>(P.pre) + (int64_t)row_base * ldp, (int)(rows_ok * ldp * 2));
  }
  if constexpr (FP8) ra_ = uniform_rsrc(const_cast<float*>(P.a_scale) + row_base, rows_ok * 4);
  const uint16_t* const biasp = P.bias;
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    gcol[st] = SWAP ? col_base + (2 * st + (fq & 1)) * 16 + (fq >> 1) * 8 : col_base + st * 16 + fq * 4;
    const bool col_ok = gcol[st] < N && (!SWAP || 2 * st + (fq & 1) < NJ);
    coff[st] = col_ok ? (uint32_t)gcol[st] * 2u : OOB;
#ifdef AFX_GEMM_TRACE
    if (EPI == EPI_NONE && P.gelu_col0 == -12345) coff[st] = OOB;     // tools/gemm_trace.hip TRACE_NOSTORE: epilogue without write traffic
#endif
#pragma unroll
    for (int e = 0; e < CW; ++e) bias[st][e] = 0.f;
    if constexpr (FP8) {
#pragma unroll
      for (int e = 0; e < CW; ++e) wsc[st][e] = col_ok ? P.w_scale[gcol[st] + e] : 0.f;
    }
    if (biasp != nullptr && col_ok && !ROWB) {
      if constexpr (SWAP) {
        unpack8(*reinterpret_cast<const u32x4_t*>(biasp + gcol[st]), bias[st]);
      } else {
        const u32x2_t w = *reinterpret_cast<const u32x2_t*>(biasp + gcol[st]);
        bias[st][0] = __uint_as_float(w[0] << 16); bias[st][1] = __uint_as_float(w[0] & 0xffff0000u);
        bias[st][2] = __uint_as_float(w[1] << 16); bias[st][3] = __uint_as_float(w[1] & 0xffff0000u);
      }
    }
  }
  // GATE_RES: the residual words of row tile ii + PF are requested before row tile ii is converted and stored (requesting ALL
  // rows up front serialises the read burst and the write burst: 25 k cycles per 256 x 256 tile against 15 k).  The gate vector is
  // loaded once when every row of the wave belongs to one batch sample (uniform test; always so for batch 1), per row tile otherwise.
  constexpr int PF = 2;
  uint32_t rw[PF + 1][NS][CW / 2];
  uint32_t pw[PRE ? PF + 1 : 1][NS][CW / 2];
  float asc[FP8 ? PF + 1 : 1];
  auto fetch_pre = [&](int ii) {        // the pre-add words and the activation scale of row tile ii
    if constexpr (PRE) {
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const int off = (int)((uint32_t)((ii * 16 + frow) * ldp2) + coff[st]);
        if constexpr (SWAP) {
          const u32x4_t w = __builtin_amdgcn_raw_buffer_load_b128(rp_, off, 0, 0);
          pw[ii % (PF + 1)][st][0] = w[0]; pw[ii % (PF + 1)][st][1] = w[1]; pw[ii % (PF + 1)][st][2] = w[2]; pw[ii % (PF + 1)][st][3] = w[3];
        } else {
          const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(rp_, off, 0, 0);
          pw[ii % (PF + 1)][st][0] = w[0]; pw[ii % (PF + 1)][st][1] = w[1];
        }
      }
    }
    if constexpr (FP8) asc[ii % (PF + 1)] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra_, (ii * 16 + frow) * 4, 0, 0));
  };
  auto fetch_res = [&](int ii, uint32_t (&r)[NS][CW / 2]) {
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const int off = (int)((uint32_t)((ii * 16 + frow) * ldr2) + coff[st]);
      if constexpr (SWAP) {
        const u32x4_t w = __builtin_amdgcn_raw_buffer_load_b128(rr_, off, 0, 0);
        r[st][0] = w[0]; r[st][1] = w[1]; r[st][2] = w[2]; r[st][3] = w[3];
      } else {
        const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(rr_, off, 0, 0);
        r[st][0] = w[0]; r[st][1] = w[1];
      }
    }
  };
  auto fetch_gate = [&](int b, float (&g)[NS][CW]) {
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const uint32_t go = (uint32_t)(b * ldg4) + 2u * coff[st];   // (OOB * 2 wraps to 0: harmless, the store is dropped)
#pragma unroll
      for (int h = 0; h < CW / 4; ++h) {
        const f32x4_t w = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rg_, (int)(go + 16u * h), 0, 0));
        g[st][4 * h] = w[0]; g[st][4 * h + 1] = w[1]; g[st][4 * h + 2] = w[2]; g[st][4 * h + 3] = w[3];
      }
    }
  };
  float g1[NS][CW];
  bool single = true;
  if constexpr (PRE || FP8) {
#pragma unroll
    for (int ii = 0; ii < PF && ii < MI; ++ii) fetch_pre(ii);
  }
  if constexpr (EPI == EPI_GATE_RES) {
#pragma unroll
    for (int ii = 0; ii < PF && ii < MI; ++ii) fetch_res(ii, rw[ii % (PF + 1)]);
    const int b_first = row_base / rpb, b_last = (row_base + MI * 16 - 1) / rpb;      // scalar divisions, once
    single = b_first == b_last;
    if constexpr (!CONV) fetch_gate(b_first, g1);      // (convolutions: a plain residual add, no gate vector -- 64 registers less)
  }
  // GN (256x128-tile convolutions of the one-wave-per-SIMD kernel, GemmProblem::gn_stats): this lane's share of the output's GroupNorm sums --
  // a lane's 8 columns are two 4-channel halves, 8 running floats for the two steps of a row tile.  Only the narrow tile has it: in the epilogue
  // of a 256x256 tile hipcc already parks VGPRs in accumulator registers as row tiles free them, and 8 more live floats make it spill 1.1 KB of
  // ACCUMULATORS to scratch instead (tried: one prefetch stage less, no gate arrays, a scheduling barrier per row tile -- no change); the
  // register-free alternative, private LDS cells updated by ds_add_f32, costs 70 % more time per tile (the LDS atomic unit is slow).  The wide
  // layers (>= 256 channels: 4x smaller grids per channel) keep the separate statistics pass.
  constexpr int GH = GN ? (NS > 2 ? 1 : 2) : 1;
  constexpr int GQ = 2 * NS * GH;                                  // quantities per lane: (step, half) x {sum, sum of squares}
  float gsum[GN ? NS : 1][GH], gsq[GN ? NS : 1][GH];
  if constexpr (GN) {
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
      for (int h = 0; h < GH; ++h) {
        float z = 0.f;
        asm volatile("" : "+v"(z));          // opaque zero, created HERE: a plain 0.f is materialised above the main loop and lives through it
        gsum[st][h] = z;
        asm volatile("" : "+v"(z));
        gsq[st][h] = z;
      }
  }
  hk.template at<-1>();
  static_for<MI>([&](auto ii_c) AFX_INL {
    constexpr int ii = decltype(ii_c)::value;
    uint32_t roff = (uint32_t)((ii * 16 + frow) * ldc2);
    bool border = false;
    if constexpr (CONV) {
      const int grow = row_base + ii * 16 + frow, wp = P.conv_wp;
      int yy = (int)((float)grow * (1.0f / (float)wp));           // grow / wp up to +-1: fix up exactly
      int xx = grow - yy * wp;
      if (xx < 0) { xx += wp; --yy; } else if (xx >= wp) { xx -= wp; ++yy; }
      border = yy <= 0 || yy >= P.conv_hp - 1 || xx == 0 || xx == wp - 1;        // (rows past the grid count as border: nothing of them is kept)
      if (up != 0) {                                              // source pixel (yy, xx) -> pixel (2 yy + py - 1, 2 xx + px - 1) of the 2x grid
        const int oy = 2 * yy + ((up - 1) >> 1) - 1, ox = 2 * xx + ((up - 1) & 1) - 1;
        const int oh = 2 * P.conv_hp - 2, ow = 2 * wp - 2;
        const bool inside = oy >= 0 && oy < oh && ox >= 0 && ox < ow;
        border = oy <= 0 || oy >= oh - 1 || ox <= 0 || ox >= ow - 1;   // source border pixels land on the 2x grid's border (or outside)
        roff = inside ? (uint32_t)(oy * ow + ox) * (uint32_t)ldc2 : OOB;
      }
    }
    float rb = 0.f;                                               // ROWB: bias[row] (the transposed V projection)
    if constexpr (ROWB) {
      if (biasp != nullptr) rb = __uint_as_float((uint32_t)biasp[min(row_base + ii * 16 + frow, M - 1)] << 16);
    }
    float gi[NS][CW];
    if constexpr (PRE || FP8) {
      if (ii + PF < MI) fetch_pre(ii + PF);
    }
    if constexpr (EPI == EPI_GATE_RES) {
      if (ii + PF < MI) fetch_res(ii + PF, rw[(ii + PF) % (PF + 1)]);
      if constexpr (!CONV) {
#pragma unroll
      for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int e = 0; e < CW; ++e) gi[st][e] = g1[st][e];
      }
      if (!CONV && !single) {                                            // (uniform) rows of several samples in this wave's tile
        const int grow = row_base + ii * 16 + frow;
        int b = (int)((float)grow * inv_rpb);                   // floor(grow / rpb) up to +-1: fix up exactly
        const int rem = grow - b * rpb;
        b += rem >= rpb ? 1 : (rem < 0 ? -1 : 0);
        fetch_gate(b, gi);
      }
    }
    static_for<NS>([&](auto st_c) AFX_INL {
      constexpr int st = decltype(st_c)::value;
      float v[CW];
      float c0[4], c1[4] = {0.f, 0.f, 0.f, 0.f};
      acc.template tile<ii, SWAP ? 2 * st : st>(c0);
      if constexpr (SWAP && 2 * st + 1 < NJ) acc.template tile<ii, (2 * st + 1) % NJ>(c1);
      if constexpr (SWAP) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {     // all 64 lanes take part in the exchange
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(c0[e]), __float_as_uint(c1[e]), false, false);
          v[e] = __uint_as_float(sw[0]);
          v[4 + e] = __uint_as_float(sw[1]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c0[e];
      }
      hk.template at<(ii * NS + st) * 4 + 0>();
      if constexpr (FP8) {
#pragma unroll
        for (int e = 0; e < CW; ++e) v[e] *= asc[ii % (PF + 1)] * wsc[st][e];
      }
#pragma unroll
      for (int e = 0; e < CW; ++e) v[e] += ROWB ? rb : bias[st][e];
      if constexpr (PRE) {
#pragma unroll
        for (int e = 0; e < CW; ++e) {
          const uint32_t w = pw[ii % (PF + 1)][st][e >> 1];
          v[e] += __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
        }
      }
      hk.template at<(ii * NS + st) * 4 + 1>();
      if constexpr (EPI == EPI_GELU) {
        if (gcol[st] >= gelu_col0) {
#pragma unroll
          for (int e = 0; e < CW; ++e) v[e] = gelu_tanh(v[e]);
        }
      } else if constexpr (EPI == EPI_GELU_ALL) {
#pragma unroll
        for (int e = 0; e < CW; ++e) v[e] = gelu_tanh(v[e]);
      } else if constexpr (EPI == EPI_GATE_RES) {
        if constexpr (DROP) {        // the LoRA branch's input gradient: the product is masked (keep / (1 - p) or 0), then added to the residual
          const uint32_t hr = drop_mix32(P.drop_seed ^ (uint32_t)((P.drop_row0 + row_base + ii * 16 + frow) * 0x9e3779b1u));
#pragma unroll
          for (int e = 0; e < CW; ++e) {
            const bool keep = drop_mix32(hr + (uint32_t)(gcol[st] + e) * 0x85ebca77u) >= P.drop_thresh;
            v[e] = keep ? v[e] * P.drop_inv_keep : 0.f;
          }
        }
#pragma unroll
        for (int e = 0; e < CW; ++e) {
          const uint32_t w = rw[ii % (PF + 1)][st][e >> 1];
          const float rr = __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
          v[e] = CONV ? rr + v[e] : rr + (has_gate ? gi[st][e] : 1.0f) * v[e];       // no gate: plain residual add
        }
      }
      hk.template at<(ii * NS + st) * 4 + 2>();
      if constexpr (CONV) {
#pragma unroll
        for (int e = 0; e < CW; ++e) v[e] = border ? 0.f : v[e];
        if constexpr (GN) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            gsum[st][GH == 2 ? e >> 2 : 0] += v[e];
            gsq[st][GH == 2 ? e >> 2 : 0] += v[e] * v[e];
          }
        }
      }
      if constexpr (SWAP) {
        float v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v8[e] = v[e % CW];
        __builtin_amdgcn_raw_buffer_store_b128(pack8(v8), rc, (int)(roff + coff[st]), 0, 0);
      } else {
        const u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        __builtin_amdgcn_raw_buffer_store_b64(o, rc, (int)(roff + coff[st]), 0, 0);
      }
      hk.template at<(ii * NS + st) * 4 + 3>();
    });
  });
  if constexpr (GN) {
    if (P.gn_stats != nullptr) {                                  // (uniform)
      static_assert(SWAP && GQ <= 16, "one quantity per lane of a 16-lane row");
      // sum over the 16 lanes (frow) that own the same columns -- a DPP row -- then lane frow = k adds quantity k = 2 (GH st + half) + {sum, sumsq}
      auto row_sum = [](float x) {
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));   // row_half_mirror
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));   // row_mirror
        return x;
      };
      float mine = 0.f;
      int gc = 0;
      bool ok = false;
#pragma unroll
      for (int q = 0; q < GQ; ++q) {
        const float x = row_sum((q & 1) ? gsq[(q >> 1) / GH][(q >> 1) % GH] : gsum[(q >> 1) / GH][(q >> 1) % GH]);
        mine = frow == q ? x : mine;
        if (frow == q) { gc = gcol[(q >> 1) / GH] + 4 * ((q >> 1) % GH); ok = coff[(q >> 1) / GH] != OOB; }
      }
      if (ok && frow < GQ) {
        const int g = gc / P.gn_gs;
        const int slot = ((row_base >> 7) + (col_base >> 6)) & (GN_SLOTS - 1);
        double* dst = P.gn_stats + ((int64_t)slot * P.gn_groups + g) * 2 + (frow & 1);
        __builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double*)dst, (double)mine);
      }
    }
  }
}

template <int EPI, int MI, int NJ, bool SWAP, bool FP8 = false, bool PRE = false, bool ROWB = false, bool CONV = false, bool GN = false, bool DROP = false, class HK = NoHook>
AFX_DEV void epi_store_fast(const GemmProblem& P, f32x4_t (&acc)[MI][NJ], int row_base, int col_base, int frow, int fq, const HK& hk = HK{}) {
  epi_store_fast_acc<EPI, MI, NJ, SWAP, FP8, PRE, ROWB, CONV, GN, DROP, HK>(P, AccArr<MI, NJ>{acc}, row_base, col_base, frow, fq, hk);
}

// fp32 output of the weight-gradient products (GemmProblem::out_f32 1: store, 2: C += result; no bias / activation there): 8
// consecutive columns per lane after the permlane exchange = two 16-byte accesses; rows >= M and masked columns dropped by the
// bounds check.  The accumulate mode's read of C is requested one row tile ahead.
template <int MI, int NJ, class ACC>
AFX_DEV void epi_store_f32_acc(const GemmProblem& P, const ACC& acc, int row_base, int col_base, int frow, int fq) {
  constexpr int NS = NJ / 2;
  constexpr uint32_t OOB = 0x80000000u;
  const int M = P.M, N = P.N;
  const bool accum = P.out_f32 == 2;
  const int rows_ok = min(max(M - row_base, 0), MI * 16);
  const int64_t ldc = P.ldc;
  float* const Cf = reinterpret_cast<float*>(P.C);
  __amdgpu_buffer_rsrc_t rc = uniform_rsrc(Cf + (int64_t)row_base * ldc, (int)(rows_ok * ldc * 4));
  const int ldc4 = (int)(ldc * 4);
  uint32_t coff[NS];
  float bias[NS][8];
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    const int gcol = col_base + (2 * st + (fq & 1)) * 16 + (fq >> 1) * 8;
    coff[st] = gcol < N ? (uint32_t)gcol * 4u : OOB;
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[st][e] = 0.f;
    if (P.bias != nullptr && gcol < N) unpack8(*reinterpret_cast<const u32x4_t*>(P.bias + gcol), bias[st]);
  }
  f32x4_t old[2][NS][2];
  auto fetch = [&](int ii) {
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const int off = (int)((uint32_t)((ii * 16 + frow) * ldc4) + coff[st]);
      old[ii & 1][st][0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rc, off, 0, 0));
      old[ii & 1][st][1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rc, off + 16, 0, 0));
    }
  };
  if (accum) fetch(0);
  static_for<MI>([&](auto ii_c) AFX_INL {
    constexpr int ii = decltype(ii_c)::value;
    if (accum && ii + 1 < MI) fetch(ii + 1);
    const uint32_t roff = (uint32_t)((ii * 16 + frow) * ldc4);
    static_for<NS>([&](auto st_c) AFX_INL {
      constexpr int st = decltype(st_c)::value;
      f32x4_t o0, o1;
      float c0[4], c1[4];
      acc.template tile<ii, 2 * st>(c0);
      acc.template tile<ii, 2 * st + 1>(c1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {     // all 64 lanes take part in the exchange
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(c0[e]), __float_as_uint(c1[e]), false, false);
        o0[e] = __uint_as_float(sw[0]) + bias[st][e];
        o1[e] = __uint_as_float(sw[1]) + bias[st][4 + e];
      }
      if (accum) {
        o0 += old[ii & 1][st][0];
        o1 += old[ii & 1][st][1];
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o0), rc, (int)(roff + coff[st]), 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o1), rc, (int)(roff + coff[st]) + 16, 0, 0);
    });
  });
}
template <int MI, int NJ>
AFX_DEV void epi_store_f32(const GemmProblem& P, f32x4_t (&acc)[MI][NJ], int row_base, int col_base, int frow, int fq) {
  epi_store_f32_acc<MI, NJ>(P, AccArr<MI, NJ>{acc}, row_base, col_base, frow, fq);
}

// (uniform) the forward's epilogue modes: bf16 out, no fp8 scales / convolution border / pre-add
AFX_DEV bool epi_is_fast(const GemmProblem& P) { return P.out_f32 == 0 && P.conv_wp == 0; }

// Keys / queries of a fused k|v|q projection (GemmProblem::qk_D): the wave's 128 columns are one head.  After the permlane
// exchange lane (frow, fq) owns, for row ii*16 + frow, the 8-column chunks 4 st + 2 (fq & 1) + (fq >> 1), st = 0..3, of that head:
// its 32 values + those of the three other fq lanes are the 128 of the RMSNorm (two cross-lane adds); a chunk holds 4 whole
// rotation pairs.  cos / sin of the NEXT row tile are requested before this one is finished.
// FP8: the accumulators are first scaled by a_scale[row] * w_scale[col] (the fp8 kernels' products).
template <int MI, bool FP8, class HK, class ACC>
AFX_DEV void epi_store_qk_acc(const GemmProblem& P, const ACC& acc, int row_base, int col_base, int frow, int fq, const float* wn, const HK& hk) {
  constexpr uint32_t OOB = 0x80000000u;
  const int M = P.M, N = P.N;
  const int rows_ok = min(max(M - row_base, 0), MI * 16);
  const int64_t ldc = P.ldc;
  __amdgpu_buffer_rsrc_t rc = uniform_rsrc(P.C + (int64_t)row_base * ldc, (int)(rows_ok * ldc * 2));
  __amdgpu_buffer_rsrc_t ra = rc;
  if constexpr (FP8) ra = uniform_rsrc(const_cast<float*>(P.a_scale) + row_base, rows_ok * 4);
  const float* const wscp = P.w_scale;       // (FP8: re-read per row tile out of L1, requested with the cos / sin rows -- 32 more live floats make hipcc spill)
  __amdgpu_buffer_rsrc_t rcs = uniform_rsrc(const_cast<float*>(P.rope_cos), P.rope_rows * 256);
  __amdgpu_buffer_rsrc_t rsn = uniform_rsrc(const_cast<float*>(P.rope_sin), P.rope_rows * 256);
  const int ldc2 = (int)(ldc * 2);
  const int period = P.rope_period, pos0 = P.rope_row0 + row_base;
  const float inv_period = 1.0f / (float)period;
  uint32_t coff[4], toff[4];
  float bias[4][8], w[4][8];
  const uint16_t* const biasp = P.bias;
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int c8 = 4 * st + 2 * (fq & 1) + (fq >> 1);          // chunk of 8 columns inside the head
    const int gcol = col_base + c8 * 8;
    const bool col_ok = gcol < N;
    coff[st] = col_ok ? (uint32_t)gcol * 2u : OOB;
    toff[st] = (uint32_t)c8 * 16u;                              // 4 pairs = 4 floats of a [., 64] table row
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[st][e] = 0.f;
    if (biasp != nullptr && col_ok) unpack8(*reinterpret_cast<const u32x4_t*>(biasp + gcol), bias[st]);

    const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(wn + c8 * 8), w1 = *reinterpret_cast<const f32x4_t*>(wn + c8 * 8 + 4);
    w[st][0] = w0[0]; w[st][1] = w0[1]; w[st][2] = w0[2]; w[st][3] = w0[3];
    w[st][4] = w1[0]; w[st][5] = w1[1]; w[st][6] = w1[2]; w[st][7] = w1[3];
  }
  f32x4_t cs[2][4], sn[2][4];
  auto fetch = [&](int ii, f32x4_t (&c)[4], f32x4_t (&s_)[4]) {
    int pos = pos0 + ii * 16 + frow;
    int q = (int)((float)pos * inv_period);                       // pos % period (period may be huge: q = 0)
    int rem = pos - q * period;
    rem += rem < 0 ? period : (rem >= period ? -period : 0);
    const uint32_t ro = (uint32_t)rem * 256u;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      c[st] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rcs, (int)(ro + toff[st]), 0, 0));
      s_[st] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsn, (int)(ro + toff[st]), 0, 0));
    }
  };
  fetch(0, cs[0], sn[0]);
  hk.template at<-1>();
  static_for<MI>([&](auto ii_c) AFX_INL {
    constexpr int ii = decltype(ii_c)::value;
    if (ii + 1 < MI) fetch(ii + 1, cs[(ii + 1) & 1], sn[(ii + 1) & 1]);
    const uint32_t roff = (uint32_t)((ii * 16 + frow) * ldc2);
    float v[4][8];
    float ss = 0.f;
    float asc = 1.f;
    f32x4_t ws[FP8 ? 4 : 1][2];
    if constexpr (FP8) {
      asc = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (ii * 16 + frow) * 4, 0, 0));
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int gc = min(col_base + (4 * st + 2 * (fq & 1) + (fq >> 1)) * 8, N - 8);
        ws[st][0] = *reinterpret_cast<const f32x4_t*>(wscp + gc);
        ws[st][1] = *reinterpret_cast<const f32x4_t*>(wscp + gc + 4);
      }
    }
    static_for<4>([&](auto st_c) AFX_INL {
      constexpr int st = decltype(st_c)::value;
      float c0[4], c1[4];
      acc.template tile<ii, 2 * st>(c0);
      acc.template tile<ii, 2 * st + 1>(c1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {     // all 64 lanes take part in the exchange
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(c0[e]), __float_as_uint(c1[e]), false, false);
        v[st][e] = __uint_as_float(sw[0]);
        v[st][4 + e] = __uint_as_float(sw[1]);
      }
      hk.template at<ii * 16 + 2 * st + 0>();             // (tail hook: 16 points per row tile IN PROGRAM ORDER -- the hook issues MFMA p of the next row tile at point p: 8 in the sum-of-squares pass, 8 in the rotation pass)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if constexpr (FP8) v[st][e] *= asc * ws[st][e >> 2][e & 3];
        v[st][e] += bias[st][e];
        ss += v[st][e] * v[st][e];
      }
      hk.template at<ii * 16 + 2 * st + 1>();
    });
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    const float rstd = rsqrtf(ss * (1.0f / 128.0f) + 1e-6f);
    static_for<4>([&](auto st_c) AFX_INL {
      constexpr int st = decltype(st_c)::value;
      float r[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a = v[st][2 * i] * rstd * w[st][2 * i];
        const float b = v[st][2 * i + 1] * rstd * w[st][2 * i + 1];
        const float c = cs[ii & 1][st][i], s_ = sn[ii & 1][st][i];
        r[2 * i] = a * c - b * s_;
        r[2 * i + 1] = a * s_ + b * c;
      }
      hk.template at<ii * 16 + 8 + 2 * st + 0>();
      __builtin_amdgcn_raw_buffer_store_b128(pack8(r), rc, (int)(roff + coff[st]), 0, 0);
      hk.template at<ii * 16 + 8 + 2 * st + 1>();
    });
  });
}
template <int MI, bool FP8 = false, class HK = NoHook>
AFX_DEV void epi_store_qk(const GemmProblem& P, f32x4_t (&acc)[MI][8], int row_base, int col_base, int frow, int fq, const float* wn, const HK& hk = HK{}) {
  epi_store_qk_acc<MI, FP8, HK>(P, AccArr<MI, 8>{acc}, row_base, col_base, frow, fq, wn, hk);
}

#ifndef V3_F32_EPI
#define V3_F32_EPI 1
#endif
template <int MI, int NJ, bool SWAP, bool FP8, bool PRE, class HK, class ACC>
AFX_DEV void epi_store_fast_any_acc(const GemmProblem& P, const ACC& acc, int row_base, int col_base, int frow, int fq, const HK& hk) {
  if constexpr (NJ == 8 && SWAP && !PRE) {
    if (P.qk_D > 0) {                                            // (uniform) which 128-column head of k | v | q (| mlp) is this wave's?
      const int region = col_base / P.qk_D;
      if constexpr (FP8) {        // (the fp8 form holds weight scales + cos / sin beside bias and RMSNorm weights: with the tail's fragment ring it needs > 256 arch VGPRs -- hipcc then
                                  // parks values in accumulator registers, which the build's audit refuses: it runs the tail's MFMAs up front)
        if (region == 0) { hk.all(); epi_store_qk_acc<MI, FP8, NoHook>(P, acc, row_base, col_base, frow, fq, P.qk_wk, NoHook{}); return; }
        if (region == 2) { hk.all(); epi_store_qk_acc<MI, FP8, NoHook>(P, acc, row_base, col_base, frow, fq, P.qk_wq, NoHook{}); return; }
      }
      if (region == 0) { epi_store_qk_acc<MI, FP8, HK>(P, acc, row_base, col_base, frow, fq, P.qk_wk, hk); return; }
      if (region == 2) { epi_store_qk_acc<MI, FP8, HK>(P, acc, row_base, col_base, frow, fq, P.qk_wq, hk); return; }
    }
  }
  if constexpr (SWAP && !FP8 && !PRE && V3_F32_EPI && !(MI == 8 && NJ == 4) && NJ % 2 == 0) {      // (8 x 4 is the 8-phase kernel's patch: it keeps its own)
    if (P.out_f32 == 1 || P.out_f32 == 2) { hk.all(); epi_store_f32_acc<MI, NJ>(P, acc, row_base, col_base, frow, fq); return; }
  }
  if (P.epi == EPI_GATE_RES) {
    if constexpr (SWAP && !FP8 && !PRE) {
      if (P.drop_on) { epi_store_fast_acc<EPI_GATE_RES, MI, NJ, SWAP, false, false, false, false, false, true, HK>(P, acc, row_base, col_base, frow, fq, hk); return; }
    }
    epi_store_fast_acc<EPI_GATE_RES, MI, NJ, SWAP, FP8, PRE, false, false, false, false, HK>(P, acc, row_base, col_base, frow, fq, hk);
  } else if (P.epi == EPI_GELU) {
    // The per-lane test `column >= gelu_col0` made every one of the 32 steps its own basic block behind an exec-mask branch (a lone wave pays ~30 cycles of
    // refetch per taken branch, and no step overlaps the next one's exchange / loads): 14.6 k cycles per 256 x 256 tile against 10.2 k without GELU.  A wave's
    // columns are (in every launch of the forward) all activated or none: decide once, wave-uniformly, and run straight-line code.
#ifndef AFX_GELU_PER_LANE         // (-DAFX_GELU_PER_LANE: the per-lane test everywhere, for A/B builds)
    if (col_base >= P.gelu_col0) epi_store_fast_acc<EPI_GELU_ALL, MI, NJ, SWAP, FP8, PRE, false, false, false, false, HK>(P, acc, row_base, col_base, frow, fq, hk);
    else if (col_base + 16 * NJ <= P.gelu_col0) epi_store_fast_acc<EPI_NONE, MI, NJ, SWAP, FP8, PRE, false, false, false, false, HK>(P, acc, row_base, col_base, frow, fq, hk);
    else
#endif
    epi_store_fast_acc<EPI_GELU, MI, NJ, SWAP, FP8, PRE, false, false, false, false, HK>(P, acc, row_base, col_base, frow, fq, hk);
  }
  else if (SWAP && !FP8 && !PRE && !(MI == 8 && NJ == 4) && P.bias_rows) epi_store_fast_acc<EPI_NONE, MI, NJ, SWAP, false, false, true, false, false, false, HK>(P, acc, row_base, col_base, frow, fq, hk);
  else epi_store_fast_acc<EPI_NONE, MI, NJ, SWAP, FP8, PRE, false, false, false, false, HK>(P, acc, row_base, col_base, frow, fq, hk);
}

template <int MI, int NJ, bool SWAP, bool FP8 = false, bool PRE = false, class HK = NoHook>
AFX_DEV void epi_store_fast_any(const GemmProblem& P, f32x4_t (&acc)[MI][NJ], int row_base, int col_base, int frow, int fq, const HK& hk = HK{}) {
  epi_store_fast_any_acc<MI, NJ, SWAP, FP8, PRE, HK>(P, AccArr<MI, NJ>{acc}, row_base, col_base, frow, fq, hk);
}

template <bool FP8K = false>
AFX_DEV void epi_store_direct(const GemmProblem& P, f32x4_t (&acc)[8][4], int row_base, int col_base, int frow, int fq, int chunk) {
  if constexpr (!FP8K) {
    if (P.conv_wp > 0 && P.out_f32 == 0 && P.pre == nullptr && P.fp8 == 0 && P.epi != EPI_GELU) {       // the VAE's convolutions
      if (P.epi == EPI_GATE_RES) epi_store_fast<EPI_GATE_RES, 8, 4, true, false, false, false, true>(P, acc, row_base, col_base, frow, fq);
      else epi_store_fast<EPI_NONE, 8, 4, true, false, false, false, true>(P, acc, row_base, col_base, frow, fq);
      return;
    }
  }
  if (epi_is_fast(P)) {                                          // bf16 output without a convolution border: straight-line code
    if constexpr (FP8K) {
      if (P.pre == nullptr) { epi_store_fast_any<8, 4, true, true, false>(P, acc, row_base, col_base, frow, fq); return; }
    } else {
      if (P.pre != nullptr) epi_store_fast_any<8, 4, true, false, true>(P, acc, row_base, col_base, frow, fq);
      else epi_store_fast_any<8, 4, true, false, false>(P, acc, row_base, col_base, frow, fq);
      return;
    }
  }
  const bool first_chunk = chunk == 0;
  float wsc[2][8];            // fp8: per-output-channel weight scales of this lane's 2 x 8 columns
  int gcol[2];
  bool col_ok[2];
  float bias[2][8];
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    gcol[jp] = col_base + (2 * jp + (fq & 1)) * 16 + (fq >> 1) * 8;
    col_ok[jp] = gcol[jp] < P.N;
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[jp][e] = 0.f;
    if (P.bias != nullptr && col_ok[jp] && first_chunk) {
      const u32x4_t bw = *reinterpret_cast<const u32x4_t*>(P.bias + gcol[jp]);
      unpack8(bw, bias[jp]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) wsc[jp][e] = 1.f;
    if (P.fp8 && col_ok[jp]) {
      const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(P.w_scale + gcol[jp]);
      const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(P.w_scale + gcol[jp] + 4);
      wsc[jp][0] = s0[0]; wsc[jp][1] = s0[1]; wsc[jp][2] = s0[2]; wsc[jp][3] = s0[3];
      wsc[jp][4] = s1[0]; wsc[jp][5] = s1[1]; wsc[jp][6] = s1[2]; wsc[jp][7] = s1[3];
    }
  }
#pragma unroll
  for (int ii = 0; ii < 8; ++ii) {
    const int grow = row_base + ii * 16 + frow;
    const bool row_ok = grow < P.M;
    bool border = false;
    if (P.conv_wp > 0) {        // convolution on the padded grid: keep the 1-pixel border zero for the next layer
      const int yy = grow / P.conv_wp, xx = grow - yy * P.conv_wp;
      border = yy == 0 || yy == P.conv_hp - 1 || xx == 0 || xx == P.conv_wp - 1;
    }
    const float* grow_gate = nullptr;
    if (P.epi == EPI_GATE_RES && P.gate != nullptr && row_ok) grow_gate = P.gate + (int64_t)(grow / P.rows_per_batch) * P.ldg;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {     // all 64 lanes take part in the exchange
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[ii][2 * jp][e]), __float_as_uint(acc[ii][2 * jp + 1][e]), false, false);
        v[e] = __uint_as_float(sw[0]);
        v[4 + e] = __uint_as_float(sw[1]);
      }
      if (!row_ok || !col_ok[jp]) continue;
      if (P.fp8) {
        const float as = P.a_scale[grow];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= as * wsc[jp][e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bias[jp][e];
      if (P.pre != nullptr) {
        float pr[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(P.pre + (int64_t)grow * P.ldp + gcol[jp]), pr);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += pr[e];
      }
      if (P.epi == EPI_GELU) {
        if (gcol[jp] >= P.gelu_col0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
        }
      } else if (P.epi == EPI_GATE_RES) {
        float g[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};      // gate == nullptr: plain residual add
        if (grow_gate != nullptr) {
          const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(grow_gate + gcol[jp]);
          const f32x4_t g1 = *reinterpret_cast<const f32x4_t*>(grow_gate + gcol[jp] + 4);
          g[0] = g0[0]; g[1] = g0[1]; g[2] = g0[2]; g[3] = g0[3]; g[4] = g1[0]; g[5] = g1[1]; g[6] = g1[2]; g[7] = g1[3];
        }
        const u32x4_t rw = *reinterpret_cast<const u32x4_t*>(P.res + (int64_t)grow * P.ldr + gcol[jp]);
        float rr[8];
        unpack8(rw, rr);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rr[e] + g[e] * v[e];
      }
      if (border) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
      if (P.out_f32 == 0) {
        *reinterpret_cast<u32x4_t*>(P.C + (int64_t)grow * P.ldc + gcol[jp]) = pack8(v);
      } else if (P.out_f32 == 3) {          // split-K partial tile into this chunk's slab
        float* cp = reinterpret_cast<float*>(P.C) + chunk * P.split_stride + (int64_t)grow * P.ldc + gcol[jp];
        *reinterpret_cast<f32x4_t*>(cp) = (f32x4_t){v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4_t*>(cp + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
      } else {
        float* cp = reinterpret_cast<float*>(P.C) + (int64_t)grow * P.ldc + gcol[jp];
        f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        if (P.out_f32 == 2) {
          o0 += *reinterpret_cast<const f32x4_t*>(cp);
          o1 += *reinterpret_cast<const f32x4_t*>(cp + 4);
        }
        *reinterpret_cast<f32x4_t*>(cp) = o0;
        *reinterpret_cast<f32x4_t*>(cp + 4) = o1;
      }
    }
  }
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
  const int GM_ = batch.group_m;
  const int per_group = GM_ * P.tiles_n;
  const int grp = wg / per_group;
  const int first_m = grp * GM_;
  const int gsz = min(P.tiles_m - first_m, GM_);
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
  AFX_SYNC_DMA();       // drains the DMA (explicit vmcnt(0)) and releases the work-group

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
    AFX_SYNC_DMA();     // next stage landed (explicit vmcnt(0)) and every wave is done with this one
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
    epi_store_rows(P, patch, m0 + wm * 128 + h * 64, gcol, col_ok, er, ec, bias);
  }
}


// =================================================================================================
// v2: 8-phase schedule -- two wave groups ping-pong on every SIMD, 4-deep counted DMA pipeline.
//
// Same 256x256x64 tile / 8 waves (wr = wave>>2 in M, wc = wave&3 in N) / 128 accumulators per lane,
// but the K-tile is consumed in FOUR phases of 16 MFMAs (one 64x32 quadrant of the wave's 128x64 patch
// each) and the operand tile is cut into four 16 KiB "half tiles", each consumed in exactly ONE phase:
//     X0 / X1 : A rows of quadrant row-half mh = 0 / 1 of both wave rows   (local row = wr*64 + r)
//     Y0 / Y1 : W rows of quadrant col-half nh = 0 / 1 of all four wave cols (local row = wc*32 + c)
//     phase q : 0 -> (mh0,nh0) reads X0+Y0   1 -> (mh0,nh1) reads Y1   2 -> (mh1,nh1) reads X1
//               3 -> (mh1,nh0) reads nothing (the nh0 fragments stay in registers)
// Every phase also issues the LDS-DMA of ONE half tile (2 x global_load_lds_dwordx4 per lane):
//     q=0: Y1(t+1)   q=1: X1(t+1)   q=2: Y0(t+2)   q=3: X0(t+2)
// A slot is re-staged >= 2 phases after its only read and >= 4 phases before its next read, so the
// wait in front of a phase's first barrier is always `s_waitcnt vmcnt(8)` (4 half tiles stay in
// flight across the barriers; never vmcnt(0) in the loop).  Wave group wr=1 runs one barrier behind
// group wr=0: on every SIMD one wave issues its 16 MFMAs while its partner does ds_reads + DMA issue.
// Phase anatomy:  ds_read quadrant | stage | vmcnt(8) | s_barrier | lgkmcnt(0) | 16 MFMA | s_barrier.
// RAW: a half tile is waited for (by its issuing waves) before the first barrier of phase r-1 and read in
// phase r; WAR: see above.  Loads for tiles past K are clamped to the last tile (keeps the count uniform).
constexpr int HALF_BYTES = 128 * BK * 2;     // 16 KiB

// Cycle stamps for tools/gemm_trace.hip (compiled only with -DAFX_GEMM_TRACE): s_memtime at fixed points of ONE
// K-iteration, kept in SGPRs and dumped after the loop.  Empty in the product build.
#ifdef AFX_GEMM_TRACE
__device__ unsigned int g_gemm_trace[2][8][32];
#if AFX_GEMM_TRACE == 1
#define AFX_TR(i) if (t == trace_t) { tr[i] = (unsigned)__builtin_readcyclecounter(); }
#else
#define AFX_TR(i)
#endif
#define AFX_TRC(i) tr[i] = (unsigned)__builtin_readcyclecounter();
#else
#define AFX_TR(i)
#define AFX_TRC(i)
#endif

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

AFX_DEV i32x8_t cat8(bf16x8_t lo, bf16x8_t hi) {      // two 16-byte fragments -> the 32-byte operand of the K = 128 fp8 MFMA
  const u32x4_t a = __builtin_bit_cast(u32x4_t, lo), b = __builtin_bit_cast(u32x4_t, hi);
  return (i32x8_t){(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
}

AFX_DEV void stage_half(const char* p0, const char* p1, int64_t kbyte, char* slot, int wave) {
  const char* s0 = p0 + kbyte;
  const char* s1 = p1 + kbyte;
  __builtin_amdgcn_global_load_lds((gbl_void_t*)s0, (lds_void_t*)(slot + (wave * 64) * 16), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gbl_void_t*)s1, (lds_void_t*)(slot + (GEMM_THREADS + wave * 64) * 16), 16, 0, 0);
}

#define AFX_WAIT_VM8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define AFX_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// FP8 = true: operands are OCP e4m3 bytes.  A 128-byte LDS row is then 128 k-values and one v_mfma_scale_f32_16x16x128_f8f6f4
// (unit block scales; 2x the bf16 rate) covers the whole K-tile of a 16x16 output tile: the DMA / LDS / swizzle / phase code is
// byte-for-byte the bf16 one, a phase is 8 MFMAs instead of 16.  The contraction order inside the instruction is free as
// long as both operands agree, so lane (row, fq) simply feeds the two 16-byte chunks fq and 4 + fq it already reads.
// ---- Stream-K tail (batch.sk_cus > 0) -----------------------------------------------------------------------------------
// 216 / 648 / 864 tiles on 256 CUs leave the last "round" of a launch 84 % / 53 % / 38 % full.  With sk_cus = C CUs per XCD and
// T = tiles / 8 tiles per XCD, every XCD runs its first full = floor(T / C) * C tiles as whole tiles (work-groups in dispatch
// order, exactly as before) and splits the remaining rem = T - full tiles EVENLY over C more work-groups: the rem * nk K-tile
// units are one linear range cut into C contiguous pieces, so a work-group computes (at most) the tail of one tile and the
// head of the next.  The work-group that reaches a tile's last K-tile owns it (epilogue); a piece that stops short dumps its
// raw fp32 accumulators (256 KB, lane-major: owner and helper share the lane <-> element map, no transposition) with 16-byte
// write-through (sc1) stores into its slab and raises its flag; the owner polls the flags of the work-groups before it (one
// lane, relaxed, s_sleep), ONE agent-scope acquire, adds the slabs and runs the normal epilogue.  Every work-group computes its
// publishing piece FIRST, and an owner only ever waits for work-groups with a LOWER id: no cycle, no wait inside a publisher.
// Hand-off protocol = cdna_hip_programming.md Guideline 16 R1 (sc1 payload, every wave drains vmcnt, one lane stores the flag,
// consumer: relaxed poll -> one acquire -> __syncthreads -> loads); the consumer re-arms the flag (a slab has one reader).
constexpr int SK_SLAB_FLOATS = BM * BN;
constexpr unsigned SK_SPIN_LIMIT = 1u << 22;

struct SkPiece { int nseg, tile, t0, n, role, g, s, j0, tot, nk; };

// Piece ``sg`` (0 or 1) of this work-group: logical tile id, first K-tile, number of K-tiles (-1: the whole tile / split-K chunk),
// role 0: whole tile, 1: owner of a tile whose head other work-groups computed, 2: publish a partial.  Pure scalar arithmetic
// on blockIdx and kernel arguments.
AFX_DEV SkPiece sk_piece(const GemmBatch& batch, int sg, int ES) {
  SkPiece r{1, 0, 0, -1, 0, 0, 0, 0, 0, 0};
  if (batch.sk_cus <= 0) {
    r.tile = xcd_remap(blockIdx.x, gridDim.x);
    return r;
  }
  const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
  const int T = batch.sk_tiles_per_xcd, full = batch.sk_full, C = batch.sk_cus;
  if (l < full) {
    r.tile = xcd * T + l;
    return r;
  }
  r.s = l - full;
  r.g = xcd * C + r.s;
  r.nk = batch.p[0].K * ES / (BK * 2);               // the launcher guarantees one K for every problem of a stream-K launch
  r.tot = (T - full) * r.nk;
  const int u0 = (int)((int64_t)r.s * r.tot / C), u1 = (int)((int64_t)(r.s + 1) * r.tot / C);
  r.j0 = u0 / r.nk;
  const int it0 = u0 - r.j0 * r.nk;
  const int tile_end = (r.j0 + 1) * r.nk;
  const int base = xcd * T + full;
  if (u1 <= tile_end) {                              // one piece
    r.tile = base + r.j0; r.t0 = it0; r.n = u1 - u0;
    r.role = u1 == tile_end ? (it0 > 0 ? 1 : 0) : 2;
  } else {                                           // head of the next tile FIRST (published), then the owned tail
    r.nseg = 2;
    if (sg == 0) {
      r.tile = base + r.j0 + 1; r.t0 = 0; r.n = u1 - tile_end; r.role = 2;
    } else {
      r.tile = base + r.j0; r.t0 = it0; r.n = tile_end - u0; r.role = it0 > 0 ? 1 : 0;
    }
  }
  return r;
}

template <bool FP8>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_kernel_v2(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = FP8 ? 1 : 2;       // bytes per operand element

  // ---- this work-group's pieces (decoded from blockIdx on demand: nothing but the piece index stays live across the main loop)
  const int nseg = sk_piece(batch, 0, ES).nseg;

#pragma unroll 1
  for (int sg = 0; sg < nseg; ++sg) {
  // Every lane constant is re-derived from an OPAQUE copy of threadIdx inside the piece loop: as loop invariants the compiler
  // hoists ~60 of them (swizzle offsets, epilogue address pieces) in front of the loop and spills them around the main loop.
  int tid_o = threadIdx.x;
  asm volatile("" : "+v"(tid_o));
  const int tid = tid_o;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int frow = lane & 15, fq = lane >> 4;
  const int arow = wr * 64 + frow;      // + i*16
  const int brow = wc * 32 + frow;      // + j*16
  int role, wg;
  int t0 = 0, nk = -1;
  {
    const SkPiece pc = sk_piece(batch, sg, ES);
    wg = pc.tile; role = pc.role;
    if (pc.n >= 0) { t0 = pc.t0; nk = pc.n; }
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
    if (i < batch.nprob && wg >= batch.p[i].tile_start) pi = i;
  const GemmProblem& P = batch.p[pi];
  wg -= P.tile_start;
  int chunk = 0;
  if (P.split_k > 1) {                  // chunk fastest: the work-groups of one output tile run side by side
    chunk = wg % P.split_k;
    wg /= P.split_k;
  }
  const int GM_ = batch.group_m;
  const int per_group = GM_ * P.tiles_n;
  const int grp = wg / per_group;
  const int first_m = grp * GM_;
  const int gsz = min(P.tiles_m - first_m, GM_);
  const int in_grp = wg - grp * per_group;
  const int tm = first_m + in_grp % gsz;
  const int tn = in_grp / gsz;
  const int m0 = tm * BM, n0 = tn * BN;
  if (nk < 0) {                         // whole tile or split-K chunk (stream-K pieces come with their own range)
    nk = P.K * ES / (BK * 2);
    if (P.split_k > 1) {
      const int per = (nk + P.split_k - 1) / P.split_k;
      t0 = chunk * per;
      nk = min(per, nk - t0);
      if (nk <= 0) return;              // (the launcher's chunking leaves no empty chunk; uniform, before any barrier)
    }
  }
#ifdef AFX_GEMM_TRACE
  unsigned tr[24];
  const int trace_t = nk / 2;
  AFX_TRC(17)
  tr[22] = (unsigned)__builtin_amdgcn_s_memrealtime();      // 100 MHz reference clock: shader clock = d(memtime) / d(realtime)
#endif

  // ---- per-lane DMA source pointers (k = 0) of the two 16-byte chunks this lane moves per half tile
  const char* src[4][2];       // [X0, X1, Y0, Y1][chunk i]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = i * GEMM_THREADS + tid;
    const int lr = p >> 3;                              // local row of the half tile
    const int c = (p & 7) ^ ((lr >> 1) & 7);            // logical 16-byte chunk stored at physical p & 7
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int ar = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
      ar = ar < P.M ? ar : P.M - 1;
      src[h][i] = reinterpret_cast<const char*>(P.A) + (int64_t)ar * P.lda * ES + c * 16;
      int br = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);
      br = br < P.N ? br : P.N - 1;
      src[2 + h][i] = reinterpret_cast<const char*>(P.W) + (int64_t)br * P.ldw * ES + c * 16;
    }
  }
  // slot(buffer b, half h) = smem + (b*4 + h) * HALF_BYTES with h: 0 X0, 1 X1, 2 Y0, 3 Y1
  auto slot = [&](int t, int h) -> char* { return smem + (((t & 1) << 2) + h) * HALF_BYTES; };
  auto kb = [&](int t) -> int64_t { return (int64_t)(t0 + (t < nk ? t : nk - 1)) * (BK * 2); };
  // implicit 3x3 convolution on a zero-bordered NHWC grid (conv_cin_tiles > 0): K-tile t = (tap, 64-channel chunk);
  // its A rows are the SAME flat pixel rows shifted by (dy * row_pitch + dx) -- no im2col, the tap is a row offset.
  const int ct = P.conv_cin_tiles;
  auto ka = [&](int t) -> int64_t {
    t = t < nk ? t : nk - 1;
    if (ct == 0) return (int64_t)(t0 + t) * (BK * 2);
    const int tap = t / ct, cc = t - tap * ct;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    return ((int64_t)(dy * P.conv_wp + dx) * P.lda + cc * BK) * 2;
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- prologue = the staging of virtual phases -6..-1 -------------------------------------
  stage_half(src[2][0], src[2][1], kb(0), slot(0, 2), wave);   // Y0(0)
  stage_half(src[0][0], src[0][1], ka(0), slot(0, 0), wave);   // X0(0)
  stage_half(src[3][0], src[3][1], kb(0), slot(0, 3), wave);   // Y1(0)
  stage_half(src[1][0], src[1][1], ka(0), slot(0, 1), wave);   // X1(0)
  stage_half(src[2][0], src[2][1], kb(1), slot(1, 2), wave);   // Y0(1)
  stage_half(src[0][0], src[0][1], ka(1), slot(1, 0), wave);   // X0(1)
  AFX_WAIT_VM8();                       // Y0(0), X0(0) landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();         // ... and everybody else's
  if (wr == 1) __builtin_amdgcn_s_barrier();   // stagger: group 1 runs one barrier behind group 0

  bf16x8_t af[2][4], b0[2][2], b1[2][2];    // [kk][tile]

#define AFX_MFMA_QUAD(MH, NH, BF)                                                                   \
  if constexpr (FP8) {                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
      acc[(MH) * 4 + i][(NH) * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(           \
          cat8(BF[0][j], BF[1][j]), cat8(af[0][i], af[1][i]), acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
  } else {                                                                                          \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
      acc[(MH) * 4 + i][(NH) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                    \
          BF[kk][j], af[kk][i], acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0);                          \
  }

#define AFX_PHASE_TAIL(MH, NH, BF, TI)                                                              \
  __builtin_amdgcn_sched_barrier(0);                                                                \
  __builtin_amdgcn_s_barrier();                                                                     \
  AFX_TR(TI)                                                                                        \
  AFX_WAIT_LGKM0();                                                                                 \
  AFX_TR(TI + 1)                                                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                                \
  __builtin_amdgcn_s_setprio(1);                                                                    \
  AFX_MFMA_QUAD(MH, NH, BF)                                                                         \
  __builtin_amdgcn_s_setprio(0);                                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                                \
  AFX_TR(TI + 2)                                                                                    \
  __builtin_amdgcn_s_barrier();                                                                     \
  AFX_TR(TI + 3)                                                                                    \
  __builtin_amdgcn_sched_barrier(0);

  AFX_TRC(18)
  for (int t = 0; t < nk; ++t) {
    AFX_TR(16)
    const char* x0 = slot(t, 0);
    const char* x1 = slot(t, 1);
    const char* y0 = slot(t, 2);
    const char* y1 = slot(t, 3);
    // ---- phase 0: quadrant (mh0, nh0) -----------------------------------------------------
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int j = 0; j < 2; ++j) b0[kk][j] = lds_frag(y0, brow + j * 16, kk * 4 + fq);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[kk][i] = lds_frag(x0, arow + i * 16, kk * 4 + fq);
    }
    stage_half(src[3][0], src[3][1], kb(t + 1), slot(t + 1, 3), wave);   // Y1(t+1)
    AFX_WAIT_VM8();
    AFX_PHASE_TAIL(0, 0, b0, 0)
    // ---- phase 1: quadrant (mh0, nh1) -----------------------------------------------------
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j) b1[kk][j] = lds_frag(y1, brow + j * 16, kk * 4 + fq);
    stage_half(src[1][0], src[1][1], ka(t + 1), slot(t + 1, 1), wave);   // X1(t+1)
    AFX_WAIT_VM8();
    AFX_PHASE_TAIL(0, 1, b1, 4)
    // ---- phase 2: quadrant (mh1, nh1) -----------------------------------------------------
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) af[kk][i] = lds_frag(x1, arow + i * 16, kk * 4 + fq);
    stage_half(src[2][0], src[2][1], kb(t + 2), slot(t, 2), wave);       // Y0(t+2)
    AFX_PHASE_TAIL(1, 1, b1, 8)
    // ---- phase 3: quadrant (mh1, nh0), operands already in registers --------------------------
    stage_half(src[0][0], src[0][1], ka(t + 2), slot(t, 0), wave);       // X0(t+2)
    AFX_WAIT_VM8();
    AFX_PHASE_TAIL(1, 0, b0, 12)
  }
  AFX_TRC(19)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // clamped tail DMAs must land before LDS is reused
  if (wr == 0) __builtin_amdgcn_s_barrier();         // undo the stagger
  __syncthreads();
  AFX_TRC(20)

  if (role == 2) {
    // ---- publish the raw accumulators: slab[v][tid] 16-byte words, write-through, then this work-group's flag ----------
    const int sk_g = sk_piece(batch, sg, ES).g;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(batch.sk_slab + (int64_t)sk_g * SK_SLAB_FLOATS, 0,
                                                                 SK_SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][j]), rs, ((i * 4 + j) * GEMM_THREADS + tid) * 16, 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its own stores
    __syncthreads();
    if (tid == 0) __hip_atomic_store(batch.sk_flags + sk_g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    continue;
  }
  if (role == 1) {
    // ---- add the partials of the work-groups that computed this tile's head: s-1, s-2, ... while their range reaches into it
    const SkPiece pc = sk_piece(batch, sg, ES);
    const int tile_u0 = pc.j0 * pc.nk, sk_tot = pc.tot, sk_s = pc.s, sk_g = pc.g;
    for (int sp = sk_s - 1; sp >= 0 && (int)((int64_t)(sp + 1) * sk_tot / batch.sk_cus) > tile_u0; --sp) {
      const int g = sk_g - sk_s + sp;
      if (tid == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(batch.sk_flags + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < SK_SPIN_LIMIT)
          __builtin_amdgcn_s_sleep(8);
        if (spins >= SK_SPIN_LIMIT) __hip_atomic_store(batch.sk_flags + 8 * 64, 0xdeadu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(batch.sk_slab + (int64_t)g * SK_SLAB_FLOATS, 0, SK_SLAB_FLOATS * 4,
                                                                   0x00020000);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] += __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, ((i * 4 + j) * GEMM_THREADS + tid) * 16, 0, 0));
      __syncthreads();                                    // every wave has its slab words (the adds above waited for them)
      if (tid == 0) __hip_atomic_store(batch.sk_flags + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
    }
  }

  // ---- epilogue: straight from the (transposed) accumulators ------------------------------------
  {   // lane constants of the epilogue from an OPAQUE copy of threadIdx: otherwise its column / row offsets are hoisted above the
      // main loop, where every register is spoken for (the fp8 instance spilled inside the loop: 158 -> 187 us at N = 9216)
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lane2 = tid2 & 63, wave2 = __builtin_amdgcn_readfirstlane(tid2 >> 6);
    epi_store_direct<FP8>(P, acc, m0 + (wave2 >> 2) * 128, n0 + (wave2 & 3) * 64, lane2 & 15, lane2 >> 4, chunk);
  }
#ifdef AFX_GEMM_TRACE
  AFX_TRC(21)
  tr[23] = (unsigned)__builtin_amdgcn_s_memrealtime();
  if ((blockIdx.x == 0 || blockIdx.x == 300) && lane == 0)
    for (int i = 0; i < 24; ++i) g_gemm_trace[blockIdx.x ? 1 : 0][wave][i] = tr[i];
#endif
  }   // pieces
}


// =================================================================================================
// v3: ONE wave per SIMD, tile shape a template parameter.  4 waves (2 x 2), each (16 MI) x (16 NJ) of a (32 MI) x (32 NJ) tile with
// 4 MI NJ fp32 accumulators per lane in the accumulator file (<= 256: the 512-register budget of a 256-thread work-group):
//   <8, 8>  256 x 256, 128 MFMAs / 32 ds_read_b128 / 16 DMA instructions per wave and K-tile (the 8-phase kernel: 2.67 MFMAs per
//           read, 196 KB of LDS reads per CU and K-tile against 128 KB here -- this kernel runs at a higher clock under the power cap)
//   <9, 6>  288 x 192: M = 4608 = 16 x 288 and N = 3072 j = 16 j x 192, so EVERY joint-stream FLUX GEMM is a whole number of
//           256-tile rounds (N = 3072: 256 tiles instead of 216 256 x 256 tiles in one 84 %-filled round)
//   <10, 6> 320 x 192: the two-problem launches of the double blocks (4096 + 512 rows): 15 x N/192 tiles
// Schedule per K-tile (64), per wave: k-half 0 multiplies from registers while the k-half-1 fragments are read (one ds_read_b128
// every second MFMA, all issued in the first half of the phase) and W(t+2) is issued (one DMA instruction every few MFMAs in the
// second half); ONE barrier in the middle of the tile (every LDS read of tile t is done, A(t+1) / W(t+1) have landed); k-half 1
// multiplies while tile t+1's k-half-0 fragments are read and A(t+2) is issued.  Never two memory instructions between two
// MFMAs: a ds_read_b128 / DMA issue costs the lone wave of a SIMD ~6-20 cycles of MFMA issue when they queue up (measured with
// tools/gemm_trace.hip: 2664 -> 2376 cycles per K-tile from the interleave alone; the 8-phase kernel: 2440).
//   * LDS = A x 2 slots + W x 3 slots: W(t+2) goes to the slot W(t-1) left at mid-tile t-1 (1.5 tiles ahead of its first read),
//     A(t+2) to the slot A(t) leaves at mid-tile t (one tile ahead).
//   * epilogue: epi_store_fast_any_acc on the asm-owned accumulator file (AccLit); round 6: the last K-tile's MFMAs are issued from inside it (TAIL, below).
#ifndef V3_EXP
#define V3_EXP 0
#endif
#ifndef V3F8_TAIL            // the same for the fp8 kernel (gemm_kernel_v3f8)
#define V3F8_TAIL 1
#endif
#ifndef V3_TAIL              // 0: the last K-tile in front of the epilogue (round 5's schedule; A/B builds)
#define V3_TAIL 1
#endif
#ifndef V3_EPI_SWAP          // 0: the 4-columns-per-lane epilogue without the permlane exchange (A/B builds; it has no masked residual add: gemm_dropres_available() is false there)
#define V3_EPI_SWAP 1
#endif
#define V3_EPI_SWAP_DEFAULT V3_EPI_SWAP
AFX_DEV uint64_t v3_uniform_u64(uint64_t v) {      // a wave-uniform 64-bit value, in an SGPR pair
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
// ASM-OWNED accumulators (round 6): gemm_kernel_v3's 4 MI NJ fp32 accumulators per lane are a[0 : 4 MI NJ) BY LITERAL NAME -- tile (i, j) = a[4 (i NJ + j) : + 3] --
// in every MFMA and every epilogue read (AccLit), and `amdgpu_num_vgpr` confines hipcc's own allocation to the arch VGPRs: hipcc does not know the accumulator
// file is in use.  (With "+a" operands hipcc owned the placement: fine for ONE MFMA chain in front of the epilogue, but once the last K-tile's MFMAs were issued
// from inside the epilogue branches -- V3_TAIL -- it moved tiles between registers around every MFMA, parked them in arch VGPRs and spilled accumulators, also in
// the main loop.  Physical-register constraints "+{a[n:m]}" made it spill the tiles it could no longer move.)  What hipcc no longer does for us: the MFMA ->
// accumulator-read wait states (explicit s_nop / program order below) and "parking" arch VGPRs in retired accumulator rows during the epilogue.
// Which registers hipcc may still take: the arch VGPRs up to `amdgpu_num_vgpr`, and -- it allocates values that only move (copies, loads, stores) to either
// file -- the accumulator registers from a0 up, as many as the launch bounds leave beside its arch VGPRs.  The big tiles (one work-group per CU) leave it no
// reason to (arcflow_amd/build.py audits every build: no compiler-generated instruction may name an accumulator register at or above the kernel's BASE); the
// 128 x 128 tile (two work-groups per CU: 256 registers per lane) gives it v[0:159] + a[0:31] and keeps its 64 accumulators at a[32:95].
template <int TILE, int BASE>
AFX_DEV void v3_mfma_lit(const bf16x8_t& a_, const bf16x8_t& b_) {
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c0:%c1], %2, %3, a[%c0:%c1]" ::"n"(BASE + 4 * TILE), "n"(BASE + 4 * TILE + 3), "v"(a_), "v"(b_));
}
template <int NACC, int BASE>
AFX_DEV void v3_acc_zero() {      // + the clobber that makes the kernel descriptor allocate the accumulator file (hipcc counts only what it knows of)
  static_assert((BASE == 0 && (NACC == 256 || NACC == 240 || NACC == 224 || NACC == 216 || NACC == 128)) || (BASE == 32 && NACC == 64), "add the clobber of this accumulator range");
  if constexpr (NACC == 256) asm volatile("" ::: "a255");
  else if constexpr (NACC == 240) asm volatile("" ::: "a239");
  else if constexpr (NACC == 224) asm volatile("" ::: "a223");
  else if constexpr (NACC == 216) asm volatile("" ::: "a215");
  else if constexpr (NACC == 128) asm volatile("" ::: "a127");
  else asm volatile("" ::: "a95");
  static_for<NACC / 8>([&](auto c) AFX_INL {
    constexpr int R = BASE + 8 * decltype(c)::value;
    asm volatile("v_accvgpr_write_b32 a%c0, 0\n\tv_accvgpr_write_b32 a%c1, 0\n\tv_accvgpr_write_b32 a%c2, 0\n\tv_accvgpr_write_b32 a%c3, 0\n\t"
                 "v_accvgpr_write_b32 a%c4, 0\n\tv_accvgpr_write_b32 a%c5, 0\n\tv_accvgpr_write_b32 a%c6, 0\n\tv_accvgpr_write_b32 a%c7, 0"
                 ::"n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3), "n"(R + 4), "n"(R + 5), "n"(R + 6), "n"(R + 7));
  });
}
constexpr int V3_THREADS = 256;
constexpr int v3_lds_bytes(int MI, int NJ) { return 2 * (32 * MI) * 128 + 3 * (32 * NJ) * 128; }

// PERSIST (round 4, DESIGN 4.1 "persistent tile loop"): a work-group walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... of the launch (grid =
// the CU count, a multiple of 8: the walk stays on the work-group's XCD chunk) and issues the NEXT tile's first four DMA batches
// (A(0) W(0) W(1) A(1)) between its last K-tile and its epilogue, so their latency -- and the launch of a fresh work-group -- hide
// behind the epilogue's stores.  VMEM operations retire in order: the epilogue's own residual loads then queue behind those batches.
template <int MI, int NJ, bool CONV, int PERSIST>      // PERSIST 1: next tile's DMA in front of the epilogue, 2: behind it; 3: no walk, ONE copy of the K-tile body (see the loop)
AFX_DEV void gemm_v3_body(const GemmBatch& batch) {
  constexpr int TM = 32 * MI, TN = 32 * NJ;
  constexpr int A_SLOT = TM * 128, W_SLOT = TN * 128;          // bytes: rows of 64 bf16, chunk-swizzled like every tile here
  constexpr int NM = MI * NJ;                                  // MFMAs per k-half
  constexpr int W_SP = (NM / 2) / NJ, A_SP = (NM / 2) / MI;    // MFMAs between two DMA issues in the second half of a phase
  static_assert(4 * MI * NJ <= 256 && 2 * (MI + NJ) <= NM && W_SP >= 2 && A_SP >= 2, "v3 tile shape");
  // TAIL (round 6): the last K-tile is not multiplied in front of the epilogue but FROM INSIDE it, row tile by row tile -- row tile 0 behind the
  // epilogue's preamble, then one MFMA of row tile i + 1 at each of the 4 NS points of row tile i's steps (TailHook above) -- so 2 MI NJ - 2 NJ of the tile's last
  // 2 MI NJ MFMAs run under the epilogue's VALU / store issue instead of in front of it.  Everything the last tile needs has landed at the mid-tile barrier of
  // tile nk - 2 (or at the very first barrier for nk = 1), so the tail has no barrier and no wait of its own; accumulation order per accumulator is unchanged
  // (k-half 0, then k-half 1 of the last tile): the results are bit-identical to the plain schedule (-DV3_TAIL=0).
  constexpr bool TAIL = V3_TAIL != 0 && !CONV && PERSIST == 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const smem_w = smem + 2 * A_SLOT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // ---- per-tile state (set by setup(); PERSIST re-runs it for every tile of the walk) --------------------------------------------
  int pi_cur = 0;                         // (an index, not a pointer: a loop-carried pointer into the by-value kernel argument makes hipcc copy the whole GemmBatch to scratch)
  int m0 = 0, n0 = 0, nk = 1;
  uint32_t aoff[MI], woff[NJ];
  const char* abase = nullptr;
  const char* wbase = nullptr;
  float inv_ct = 0.f;
  int up = 0;
  auto setup = [&](int vt, int tid_) {
    int wg = xcd_remap(vt, PERSIST != 0 ? batch.total_tiles : (int)gridDim.x);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
      if (i < batch.nprob && wg >= batch.p[i].tile_start) pi = i;
    const GemmProblem& Q = batch.p[pi];
    pi_cur = pi;
    wg -= Q.tile_start;
    const int GM_ = batch.group_m;
    const int per_group = GM_ * Q.tiles_n;
    const int grp = wg / per_group;
    const int first_m = grp * GM_;
    const int gsz = min(Q.tiles_m - first_m, GM_);
    const int in_grp = wg - grp * per_group;
    const int tm = first_m + in_grp % gsz;
    const int tn = in_grp / gsz;
    m0 = tm * TM; n0 = tn * TN;
    nk = Q.K / BK;
    inv_ct = CONV ? 1.0f / (float)Q.conv_cin_tiles : 0.f;
    up = CONV ? Q.up_phase : 0;                           // 0: 3x3 taps from (-1, -1); 1 + 2 py + px: 2x2 taps from (py - 1, px - 1)
    // per-lane byte offsets of this lane's chunks of an A tile (MI pieces of 32 rows) / a W tile (NJ pieces), k = 0
    const int prow = tid_ >> 3;                                       // + 32 i
    const int pc = ((tid_ & 7) ^ ((prow >> 1) & 7)) * 16;             // logical 16-byte chunk stored at physical chunk tid & 7
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      int ar = m0 + prow + 32 * i;
      ar = ar < Q.M ? ar : Q.M - 1;
      aoff[i] = (uint32_t)((int64_t)ar * Q.lda * 2 + pc);
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      int br = n0 + prow + 32 * i;
      if (Q.w_perm16) br = (br & ~15) + 4 * ((br & 15) >> 3) + (br & 3) + 8 * ((br >> 2) & 1);   // key_of_pos: column p <- key of position p
      br = br < Q.N ? br : Q.N - 1;
      woff[i] = (uint32_t)((int64_t)br * Q.ldw * 2 + pc);
    }
    abase = reinterpret_cast<const char*>(Q.A);
    // (Measured and dropped, r02u: W stored TILE-MAJOR -- the TN x 64 block of a (column tile, K-tile) as one contiguous run, so a
    // tile's weight stream is sequential in HBM instead of 128-byte pieces 6 KB apart.  With weights streaming from HBM the loop loses
    // its DMA waits, 2561 -> 2388 cycles per K-tile at N = 12288, and the launch takes the same 287 us: the clock drops by what the
    // stall cycles had saved in power.)
    wbase = reinterpret_cast<const char*>(Q.W);
  };
  int vt = blockIdx.x;
  setup(vt, tid);
  // CONV (the VAE decoders' 3x3 convolutions, afx_vae.hip): implicit GEMM on a zero-bordered NHWC grid -- K-tile t = (tap, 64-channel chunk)
  // reads the SAME pixel rows shifted by dy * row pitch + dx, so the A stream's K offset becomes a (uniform) row shift + channel offset.
  auto ka = [&](int t) -> int64_t {
    if constexpr (!CONV) return (int64_t)t * (BK * 2);
    const GemmProblem& P = batch.p[pi_cur];
    const int ct = P.conv_cin_tiles;
    const int tap = (int)(((float)t + 0.5f) * inv_ct);            // t / ct for t < 9 ct <= 72
    const int cc = t - tap * ct;
    int dy, dx;
    if (up == 0) {
      const int ty = (tap * 11) >> 5;                             // tap / 3
      dy = ty - 1; dx = tap - 3 * ty - 1;
    } else {
      dy = (tap >> 1) + ((up - 1) >> 1) - 1; dx = (tap & 1) + ((up - 1) & 1) - 1;
    }
    return ((int64_t)(dy * P.conv_wp + dx) * P.lda + cc * BK) * 2;
  };
  auto stage_a = [&](int t) {
    t = t < nk ? t : nk - 1;
    const char* src = abase + ka(t);
    char* dst = smem + (t & 1) * A_SLOT;
#pragma unroll
    for (int i = 0; i < MI; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + aoff[i]), (lds_void_t*)(dst + (i * V3_THREADS + wave * 64) * 16), 16, 0, 0);
  };
  auto stage_w = [&](int t) {
    t = t < nk ? t : nk - 1;
    const char* src = wbase + (int64_t)t * (BK * 2);
    char* dst = smem_w + (t % 3) * W_SLOT;
#pragma unroll
    for (int i = 0; i < NJ; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + woff[i]), (lds_void_t*)(dst + (i * V3_THREADS + wave * 64) * 16), 16, 0, 0);
  };

  constexpr int ABASE = 4 * MI * NJ <= 64 ? 32 : 0;      // first accumulator register (v3_mfma_lit above)
  const AccLit<NJ, ABASE> acc{};            // the accumulators: a[ABASE : ABASE + 4 MI NJ), asm-owned
  const int frow = lane & 15, fq = lane >> 4;
  const int arow = wr * (16 * MI) + frow, brow = wc * (16 * NJ) + frow;

#ifdef AFX_GEMM_TRACE
  unsigned tr[24];
  const int trace_t = nk / 2;
  AFX_TRC(17)
  tr[22] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
  // prologue: A(0) W(0) | W(1) A(1) stay in flight
  stage_a(0); stage_w(0); stage_w(1); stage_a(1);
  for (;;) {        // one pass per tile (PERSIST: the walk; otherwise left by the break behind the epilogue)
  v3_acc_zero<4 * MI * NJ, ABASE>();
  // (PERSIST, second tile on: in-order retirement -- at most MI + NJ operations outstanding means the four batches issued in front of
  // the previous epilogue have landed, whatever that epilogue issued behind them)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI + NJ) : "memory");        // A(0), W(0)
  __builtin_amdgcn_s_barrier();
  bf16x8_t a0[MI], b0[NJ], a1[MI], b1[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) b0[i] = lds_frag(smem_w, brow + i * 16, fq);
#pragma unroll
  for (int i = 0; i < MI; ++i) a0[i] = lds_frag(smem, arow + i * 16, fq);

  // The MFMAs are inline asm on asm-owned accumulator registers (v3_mfma_lit above; left to itself hipcc keeps part of the accumulators in arch
  // VGPRs and shuttles them through v_accvgpr moves around every MFMA: 452 moves per 128 MFMAs).  An asm statement with a "memory" clobber is also
  // the ordering tool: the fragment reads and DMA issues written between two MFMAs stay there, which is the interleave sched_group_barrier would
  // give for builtin MFMAs.  The m loops are compile-time loops (static_for): the tile index is part of the instruction text.
#define V3_FENCE() asm volatile("" ::: "memory")
  // LDS-DMA in the SGPR-base + 32-bit-lane-offset form (hipcc widens the builtin's address to a 64-bit VGPR pair with a
  // v_lshl_add_u64 in front of every issue); M0 = LDS byte address of the wave's 1 KiB destination
#define V3_DMA(BASE_U64, VOFF, LDS_PTR)                                                                                            \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"((uint32_t)(uintptr_t)(LDS_PTR)), "v"(VOFF), \
               "s"(BASE_U64)                                                                                                       \
               : "memory", "m0")   /* M0 is compiler-reserved (the clobber entry draws a warning and is otherwise recorded as an   \
                                     implicit def): the prologue's builtin LDS-DMAs must never see a stale M0 */
  // MFMA m of a k-half: A row tile m / NJ, W column tile m % NJ (operands swapped: the accumulator holds C^T, see epi_store_fast)
#define V3_MFMA_AT(m, AF, BF) v3_mfma_lit<(m), ABASE>(BF[(m) % NJ], AF[(m) / NJ])          // (tile index of (row tile m / NJ, column tile m % NJ) = m)

  AFX_TRC(18)
  // Two copies of the K-tile body: tiles whose successor t+2 exists issue its DMA, the last two tiles issue nothing -- as a
  // compile-time flag, because a scalar branch around each DMA issue costs the lone wave ~30 cycles of instruction refetch
  // (16 per K-tile: +20 %).
  // PERSIST == 3: ONE copy instead -- the last two tiles re-fetch tile nk - 1 into slots nobody reads any more (as the attention kernel does).
  // The two copies meet in a block where hipcc moves all 4 MI NJ accumulators to the registers the second copy was allocated
  // (255 v_accvgpr_mov_b32 per tile for 8 x 8: ~1-2 k cycles of a 125 k-cycle K = 3072 tile); one copy has no such seam.
  int t = 0;
#pragma unroll
  for (int part = 0; part < (PERSIST == 3 ? 1 : 2); ++part) {
  const bool more = part == 0;                 // a constant once the two parts are unrolled
  const int t_end = (more && PERSIST != 3) ? nk - 2 : (TAIL ? nk - 1 : nk);      // TAIL: the last K-tile's MFMAs are issued from inside the epilogue (below)
#pragma unroll 1
  for (; t < t_end; ++t) {
    const char* sa = smem + (t & 1) * A_SLOT;
    const char* sw = smem_w + (t % 3) * W_SLOT;
    AFX_TR(0)
    // ---- k-half 0 multiplies; the k-half-1 fragments stream in; W(t+2) -> slot (t+2) % 3 ------------------------------------
    {
      const uint64_t wsrc_u = v3_uniform_u64((uintptr_t)(wbase + (int64_t)(PERSIST == 3 ? min(t + 2, nk - 1) : t + 2) * (BK * 2)));
      char* wdst = smem_w + ((t + 2) % 3) * W_SLOT;
      static_for<NM>([&](auto m_c) AFX_INL {          // one memory instruction at most between two MFMAs (12 free issue cycles)
        constexpr int m = decltype(m_c)::value;
        V3_MFMA_AT(m, a0, b0);
        V3_FENCE();
        if ((m & 1) && (m >> 1) < MI + NJ) {
          const int r = m >> 1;               // b1[0..NJ), a1[0..MI)
          if (V3_EXP == 2) {
          } else if (r < NJ) b1[r] = lds_frag(sw, brow + r * 16, 4 + fq);
          else a1[r - NJ] = lds_frag(sa, arow + (r - NJ) * 16, 4 + fq);
        }
        if (m >= NM / 2 && (m - NM / 2) % W_SP == 1 && (m - NM / 2) / W_SP < NJ && V3_EXP != 1) {
          const int q = (m - NM / 2) / W_SP;
          const uint64_t src_ = wsrc_u;             // (locals: operands of an asm statement inside a generic lambda do not capture by themselves)
          const uint32_t off_ = woff[q];
          char* const dst_ = wdst + (q * V3_THREADS + wave * 64) * 16;
          if (more) V3_DMA(src_, off_, dst_);
        }
        V3_FENCE();
      });
    }
    // ---- mid-tile: every LDS read of tile t is done; W(t+1), A(t+1) have landed (W(t+2), just issued, may still fly)
    AFX_TR(1)
    if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NJ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    AFX_TR(2)
    __builtin_amdgcn_s_barrier();
    AFX_TR(3)
    // ---- k-half 1 multiplies; DMA of A(t+2) -> slot t & 1 and the k-half-0 fragments of tile t+1 ---------------------------
    {
      const uint64_t asrc_u = v3_uniform_u64((uintptr_t)(abase + ka(PERSIST == 3 ? min(t + 2, nk - 1) : t + 2)));
      char* adst = smem + (t & 1) * A_SLOT;
      const char* na = smem + ((t + 1) & 1) * A_SLOT;
      const char* nw = smem_w + ((t + 1) % 3) * W_SLOT;
      static_for<NM>([&](auto m_c) AFX_INL {
        constexpr int m = decltype(m_c)::value;
        V3_MFMA_AT(m, a1, b1);
        V3_FENCE();
        if ((m & 1) && (m >> 1) < MI + NJ) {
          const int r = m >> 1;               // (past the last tile: re-reads a landed slot, unused)
          if (V3_EXP == 2) {
          } else if (TAIL && !more) {          // (the tail reads the last tile's fragments itself, a few MFMAs ahead of their use: nothing of them is live across the epilogue's preamble)
          } else if (r < NJ) b0[r] = lds_frag(nw, brow + r * 16, fq);
          else a0[r - NJ] = lds_frag(na, arow + (r - NJ) * 16, fq);
        }
        if (m >= NM / 2 && (m - NM / 2) % A_SP == 1 && (m - NM / 2) / A_SP < MI && V3_EXP != 1) {
          const int q = (m - NM / 2) / A_SP;
          const uint64_t src_ = asrc_u;
          const uint32_t off_ = aoff[q];
          char* const dst_ = adst + (q * V3_THREADS + wave * 64) * 16;
          if (more) V3_DMA(src_, off_, dst_);
        }
        V3_FENCE();
      });
    }
    AFX_TR(4)
  }
  }
  AFX_TRC(19)
  // MFMA -> accumulator-read wait states hipcc cannot see inside the asm (nothing is in flight any more: the last two tiles issue no DMA)
  if constexpr (!TAIL) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  {   // lane constants of the epilogue from an OPAQUE copy of threadIdx: otherwise they are hoisted above the main loop (all 512
      // registers are spoken for there) and the accumulators get shuffled through v_accvgpr moves to make room
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lane2 = tid2 & 63, wave2 = __builtin_amdgcn_readfirstlane(tid2 >> 6);
    const int wr2 = wave2 >> 1, wc2 = wave2 & 1, frow2 = lane2 & 15, fq2 = lane2 >> 4;
    const GemmProblem& P = batch.p[pi_cur];  // this tile's problem and origin (setup() below moves pi_cur / m0 / n0 on to the next tile)
    const int m0e = m0, n0e = n0;
    bool has_next = false;
    auto next_tile = [&]() {
      int vn = vt + (int)gridDim.x;
      asm volatile("" : "+v"(vn));           // opaque: the next tile's address arithmetic must not be hoisted above the main loop
      vn = __builtin_amdgcn_readfirstlane(vn);
      has_next = vn < batch.total_tiles;
      if (has_next) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();        // every wave has finished its fragment reads: the LDS slots are free
        vt = vn;
        setup(vn, tid2);
        stage_a(0); stage_w(0); stage_w(1); stage_a(1);
      }
    };
    if constexpr (PERSIST == 1) next_tile();
    AFX_TRC(20)
    if constexpr (CONV) {      // bias (+ residual) + re-zeroing of the border pixels: the output grid is the next layer's padded input
      if (P.epi == EPI_GATE_RES) epi_store_fast_acc<EPI_GATE_RES, MI, NJ, true, false, false, false, true, (NJ <= 4), false, NoHook>(P, acc, m0e + wr2 * (16 * MI), n0e + wc2 * (16 * NJ), frow2, fq2, NoHook{});
      else epi_store_fast_acc<EPI_NONE, MI, NJ, true, false, false, false, true, (NJ <= 4), false, NoHook>(P, acc, m0e + wr2 * (16 * MI), n0e + wc2 * (16 * NJ), frow2, fq2, NoHook{});
    } else
#ifdef V3_NO_EPI      // timing bound only (WRONG results: nothing is stored): what hiding the WHOLE epilogue behind matrix work could buy at most (profiles/r05b_gemm_no_epilogue_bound.txt)
    asm volatile("" ::"v"(frow2), "v"(fq2), "s"(m0e), "s"(n0e));
#else
    if constexpr (TAIL) {
      // tile nk - 1 is multiplied from here on: ALL of its fragments are read by the tail itself (the loop's second copy no longer pre-reads them), a few MFMAs
      // ahead of their use, out of the slots tile nk - 1 landed in
      const char* const sa_l = smem + ((nk - 1) & 1) * A_SLOT;
      const char* const sw_l = smem_w + ((nk - 1) % 3) * W_SLOT;
      constexpr int PER_ROW = 2 * NJ;                                  // MFMAs per row tile: k-half 0 then k-half 1, column tiles in order
      constexpr int PTS = 4 * (V3_EPI_SWAP != 0 ? (NJ + 1) / 2 : NJ);  // hook points per row tile (>= PER_ROW)
      static_assert(PTS >= PER_ROW, "one MFMA per hook point");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (nothing in flight for nk >= 2; nk = 1: the prologue's clamped re-fetches must not outlive the work-group's LDS)
      // Register budget: the epilogues' own state (bias, RMSNorm weights, cos / sin, residual and gate words: up to ~160 VGPRs) sits beside ALL 256 accumulators
      // until row tiles retire, so the tail holds no fragment array: W fragments stream through a ring of RING registers (read RING MFMAs ahead of their use --
      // every row tile re-reads its 2 NJ W fragments: 2 NJ + 2 LDS reads per row tile against 2 NJ MFMAs), A fragments one row tile ahead.
      const int arow2 = wr2 * (16 * MI) + frow2, brow2 = wc2 * (16 * NJ) + frow2;      // (from the opaque thread id: nothing of the tail's addressing may be live across the main loop -- it has no register to spare)
      constexpr int RING = 4;
      bf16x8_t tb[RING], ta0[MI], ta1[MI];
      auto b_of = [&](int g) {                                         // W fragment of tail MFMA g = row * PER_ROW + m
        const int m = g % PER_ROW;
        return lds_frag(sw_l, brow2 + (m % NJ) * 16, (m < NJ ? 0 : 4) + fq2);
      };
      auto prime = [&]() {                                             // (inside the chosen epilogue, behind its preamble: nothing of the ring is live across the mode dispatch)
#pragma unroll
        for (int g = 0; g < RING; ++g) tb[g] = b_of(g);
        ta0[0] = lds_frag(sa_l, arow2, fq2);
        ta1[0] = lds_frag(sa_l, arow2, 4 + fq2);
      };
      auto tail_mfma = [&](auto row_c, auto m_c) AFX_INL {
        constexpr int row = decltype(row_c)::value, m = decltype(m_c)::value, g = row * PER_ROW + m;
        if constexpr (m < NJ) v3_mfma_lit<row * NJ + m % NJ, ABASE>(tb[g % RING], ta0[row]);
        else v3_mfma_lit<row * NJ + m % NJ, ABASE>(tb[g % RING], ta1[row]);
        V3_FENCE();
        if constexpr (g + RING < MI * PER_ROW) tb[g % RING] = b_of(g + RING);
        if constexpr (m == NJ - 2 && row + 1 < MI) ta0[row + 1] = lds_frag(sa_l, arow2 + (row + 1) * 16, fq2);
        if constexpr (m == NJ + 2 && row + 1 < MI) ta1[row + 1] = lds_frag(sa_l, arow2 + (row + 1) * 16, 4 + fq2);
        V3_FENCE();
      };
      // MFMA -> accumulator-read wait states (hipcc cannot see the dependency): a step's reads of row tile ii's tiles (2 st, 2 st + 1) come >= 6 MFMAs behind the
      // MFMA that finished them (k-half 1 of column tile j is MFMA NJ + j of the row tile, issued at point NJ + j of row tile ii - 1; the reads of step st sit behind
      // point 4 (NS - 1) + 3 of that row tile and in front of point 4 st of this one) -- each independent MFMA holds the issue port >= 4 cycles, so >= 24 cycles
      // against the 18 the 8-pass MFMA needs; the un-hooked path and the last points of a row tile carry explicit s_nops.
      auto hook = [&](auto code_c) AFX_INL {
        constexpr int code = decltype(code_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (code == -1) {                                    // row tile 0, behind the epilogue's preamble
          prime();
          static_for<PER_ROW>([&](auto m_c) AFX_INL { tail_mfma(std::integral_constant<int, 0>{}, m_c); });
          asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");           // (row tile 0's last MFMAs -> its first step's reads)
        } else if constexpr (code == -2) {                             // everything (an epilogue without hook points), then the MFMA -> accumulator-read wait states
          prime();
          static_for<MI>([&](auto r_c) AFX_INL { static_for<PER_ROW>([&](auto m_c) AFX_INL { tail_mfma(r_c, m_c); }); });
          asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        } else {
          constexpr int row = code / PTS + 1, m = code % PTS;
          if constexpr (row < MI && m < PER_ROW) tail_mfma(std::integral_constant<int, row>{}, std::integral_constant<int, m>{});
          if constexpr (row < MI && m == PTS - 1) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // (the row tile's last MFMAs -> the next step's reads: see above)
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      const TailHook<decltype(hook)> hk{hook};
      epi_store_fast_any_acc<MI, NJ, V3_EPI_SWAP != 0, false, false, TailHook<decltype(hook)>>(P, acc, m0e + wr2 * (16 * MI), n0e + wc2 * (16 * NJ), frow2, fq2, hk);
    } else
    epi_store_fast_any_acc<MI, NJ, V3_EPI_SWAP != 0, false, false, NoHook>(P, acc, m0e + wr2 * (16 * MI), n0e + wc2 * (16 * NJ), frow2, fq2, NoHook{});
#endif
#ifdef AFX_GEMM_TRACE
    AFX_TRC(21)
    tr[23] = (unsigned)__builtin_amdgcn_s_memrealtime();
    if ((blockIdx.x == 0 || blockIdx.x == 300) && lane2 == 0)
      for (int i = 0; i < 24; ++i) g_gemm_trace[blockIdx.x ? 1 : 0][wave2][i] = tr[i];
#endif
    if constexpr (PERSIST == 2) next_tile();
    if constexpr (PERSIST == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-fetched tiles must have landed before the LDS is handed on
    if (!has_next) break;
  }
  }
}
// `amdgpu_num_vgpr` takes a literal: one kernel per register budget.  gemm_kernel_v3: one work-group per CU, hipcc in v[0:255], up to 256 accumulators;
// gemm_kernel_v3s: the 128 x 128 tile, TWO work-groups per CU (256 registers per lane: hipcc's v[0:159] + a[0:31], 64 accumulators a[32:95]).
template <int MI, int NJ, bool CONV = false, int PERSIST = 0>
__global__ __launch_bounds__(V3_THREADS, 1) __attribute__((amdgpu_num_vgpr(256))) void gemm_kernel_v3(const GemmBatch batch) {
  static_assert(4 * MI * NJ > 64, "the small tile runs as gemm_kernel_v3s");
  gemm_v3_body<MI, NJ, CONV, PERSIST>(batch);
}
template <int MI, int NJ, bool CONV = false, int PERSIST = 0>
__global__ __launch_bounds__(V3_THREADS, 2) __attribute__((amdgpu_num_vgpr(160))) void gemm_kernel_v3s(const GemmBatch batch) {
  static_assert(4 * MI * NJ <= 64, "two work-groups per CU: 64 accumulators at most");
  gemm_v3_body<MI, NJ, CONV, PERSIST>(batch);
}

// fp8 GEMM -> block-scaled fp8 output (GemmProblem::c8): a wave's 128 columns are ONE block of the next GEMM's A operand.  Per row tile: the
// 32 values a lane owns (4 steps x 8 columns after the permlane exchange) are finished in fp32 (scales, bias, GELU), their absolute maximum
// is combined over the 4 lanes of the row (two cross-lane exchanges), and the row's 128 values leave as e4m3 bytes (8 per lane and step)
// with one E8M0 byte.  Half the store traffic of the bf16 epilogue and no quantisation pass behind it.
template <bool GELU, int MI, int NJ, class HK, class ACC>
AFX_DEV void epi_store_mx8(const GemmProblem& P, const ACC& acc, int row_base, int col_base, int frow, int fq, const HK& hk) {
  static_assert(NJ == 8, "one wave = 128 columns = one scale block");
  constexpr int NS = NJ / 2;
  constexpr uint32_t OOB = 0x80000000u;
  const int M = P.M, N = P.N;
  if (col_base >= N) { hk.all(); return; }           // (uniform) a wave past the last column owns no block: its scale byte would land in the next row (the tail's MFMAs still run: uniform control flow for the hook's LDS reads)
  const int rows_ok = min(max(M - row_base, 0), MI * 16);
  const int64_t ldc8 = P.ldc8;
  __amdgpu_buffer_rsrc_t rc = uniform_rsrc(P.c8 + (int64_t)row_base * ldc8, (int)(rows_ok * ldc8));
  __amdgpu_buffer_rsrc_t rm = uniform_rsrc(P.c_mx + (int64_t)row_base * P.ld_cmx, (int)(rows_ok * P.ld_cmx));
  __amdgpu_buffer_rsrc_t ra = uniform_rsrc(const_cast<float*>(P.a_scale) + row_base, rows_ok * 4);
  const int blk = (col_base - P.c8_col0) >> 7;
  uint32_t coff[NS];
  float bias[NS][8], wsc[NS][8];
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    const int gcol = col_base + (2 * st + (fq & 1)) * 16 + (fq >> 1) * 8;
    const bool col_ok = gcol < N;
    coff[st] = col_ok ? (uint32_t)(gcol - P.c8_col0) : OOB;
#pragma unroll
    for (int e = 0; e < 8; ++e) { bias[st][e] = 0.f; wsc[st][e] = col_ok ? P.w_scale[gcol + e] : 0.f; }
    if (P.bias != nullptr && col_ok) unpack8(*reinterpret_cast<const u32x4_t*>(P.bias + gcol), bias[st]);
  }
  float asc_n = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, frow * 4, 0, 0));
  hk.template at<-1>();
  static_for<MI>([&](auto ii_c) AFX_INL {
    constexpr int ii = decltype(ii_c)::value;
    const float asc = asc_n;
    if (ii + 1 < MI) asc_n = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, ((ii + 1) * 16 + frow) * 4, 0, 0));
    float v[NS][8];
    float amax = 0.f;
    static_for<NS>([&](auto st_c) AFX_INL {
      constexpr int st = decltype(st_c)::value;
      float c0[4], c1[4];
      acc.template tile<ii, 2 * st>(c0);
      acc.template tile<ii, 2 * st + 1>(c1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(c0[e]), __float_as_uint(c1[e]), false, false);
        v[st][e] = __uint_as_float(sw[0]);
        v[st][4 + e] = __uint_as_float(sw[1]);
      }
      hk.template at<ii * 16 + 2 * st + 0>();          // (tail hook: 16 points per row tile in program order, 8 in this pass, 8 in the packing pass)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = v[st][e] * (asc * wsc[st][e]) + bias[st][e];
        if constexpr (GELU) x = gelu_tanh(x);
        x = coff[st] != OOB ? x : 0.f;
        v[st][e] = x;
        amax = fmaxf(amax, fabsf(x));
      }
      hk.template at<ii * 16 + 2 * st + 1>();
    });
    amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
    const int eb = mx_exp(amax);
    const float inv = mx_inv(eb);
    const uint32_t roff = (uint32_t)((ii * 16 + frow) * (int)ldc8);
    static_for<NS>([&](auto st_c) AFX_INL {
      constexpr int st = decltype(st_c)::value;
      uint32_t w0, w1;
      mx_pack8(v[st], inv, w0, w1);
      hk.template at<ii * 16 + 8 + 2 * st + 0>();
      __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){w0, w1}, rc, (int)(roff + coff[st]), 0, 0);
      hk.template at<ii * 16 + 8 + 2 * st + 1>();
    });
    if (fq == 0) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)eb, rm, (ii * 16 + frow) * (int)P.ld_cmx + blk, 0, 0);
  });
}

// =================================================================================================
// v3 / fp8: the one-wave-per-SIMD kernel for OCP e4m3 operands (v_mfma_f32_16x16x128_f8f6f4: 2x the bf16 rate).  LDS rings, DMA stream,
// swizzle and the ONE mid-tile barrier are gemm_kernel_v3's -- a 128-byte LDS row is now 128 k-values, so a K-tile is a single MFMA per
// 16x16 output tile, and that MFMA needs BOTH chunk halves (fq and 4 + fq: the contraction order inside the instruction is free as long
// as the two operands agree) of its operands.  The two phases of a tile therefore split the COLUMN tiles instead of K:
//   phase 0   MFMAs (i, j <  NJ/2) from A(t) and W_lo(t)   | LDS: W_hi(t)               | DMA: W(t+2) -> the slot W(t-1) left
//   mid-tile  every LDS read of tile t is done; A(t+1), W(t+1) have landed (W(t+2), just issued, may still fly)
//   phase 1   MFMAs (i, j >= NJ/2) from A(t) and W_hi(t)   | LDS: A(t+1), W_lo(t+1)     | DMA: A(t+2) -> the slot A(t) left
// A(t) is needed by both phases; phase 1 runs row tile by row tile, so row tile i's fragment of tile t+1 is read into the SAME registers right behind
// the last MFMA of tile t that uses row tile i (the last row tile at the start of the next phase 0): MI x 8 (A) + NJ x 8 (W) = 128 arch VGPRs at
// 8 x 8 beside the 256 accumulators.  Per wave and K-tile: 64 MFMAs of 32 cycles, 32 fragment reads (16 bytes), 16 DMA issues -- at most one
// memory instruction behind an MFMA (the slot tables in the kernel).  A K-tile costs the matrix pipe what the bf16 kernel's costs and covers twice
// the k-values: the prologue / epilogue share of a tile's time doubles (K = 3072: 24 K-tiles; measured 17.7 us per launch of two rounds against
// 1.3 us per K-tile and round, tools/microbench.py).
//
// MX: the activation operand carries BLOCK scales instead of one fp32 scale per row (GemmProblem::a_mx: one E8M0 byte per row and K-tile,
// i.e. per 128 k-values -- the MX layout with the block = this kernel's K-tile).  The matrix pipe applies them for free
// (v_mfma_scale_f32_16x16x128_f8f6f4: the four k-groups of a lane row take the same byte), so a PRODUCER can quantise the 128 columns it
// holds without ever seeing the whole row (GELU / attention epilogues, LayerNorm-modulate), and no separate quantisation pass is left.
// A lane fetches the bytes of 4 K-tiles of its MI row tiles as one dword each (8 buffer loads per 4 tiles, issued behind the A DMA of
// the group's second tile: in-order retirement makes the next mid-tile wait cover them); the byte is picked by op_sel, so the loop is
// unrolled over the 4 tiles of a group.  K % 512 == 0.
template <int MI, int NJ, bool MX = false>
__global__ __launch_bounds__(V3_THREADS, 1) __attribute__((amdgpu_num_vgpr(256))) void gemm_kernel_v3f8(const GemmBatch batch) {
  constexpr int TM = 32 * MI, TN = 32 * NJ, KB = 128;          // K-tile: 128 fp8 values = 128 bytes per row
  constexpr int A_SLOT = TM * 128, W_SLOT = TN * 128;
  constexpr int NH = NJ / 2, NM = MI * NH;                     // column tiles / MFMAs per phase
  constexpr int R0 = NJ + 2, R1 = 2 * (MI - 1) + NJ;           // 16-byte fragment reads of phase 0 (W_hi + the last A row tile) / phase 1 (A row tiles 0 .. MI - 2, W_lo)
  static_assert(NJ % 2 == 0 && 4 * MI * NJ <= 256 && NJ + R0 <= NM && MI + R1 <= NM, "v3f8 tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const smem_w = smem + 2 * A_SLOT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int wg = xcd_remap(blockIdx.x, (int)gridDim.x);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAX_PROBLEMS; ++i)
    if (i < batch.nprob && wg >= batch.p[i].tile_start) pi = i;
  const GemmProblem& Q = batch.p[pi];
  wg -= Q.tile_start;
  const int GM_ = batch.group_m;
  const int per_group = GM_ * Q.tiles_n;
  const int grp = wg / per_group;
  const int first_m = grp * GM_;
  const int gsz = min(Q.tiles_m - first_m, GM_);
  const int in_grp = wg - grp * per_group;
  const int m0 = (first_m + in_grp % gsz) * TM, n0 = (in_grp / gsz) * TN;
  const int nk = Q.K / KB;
  uint32_t aoff[MI], woff[NJ];
  {
    const int prow = tid >> 3;                                        // + 32 i
    const int pc = ((tid & 7) ^ ((prow >> 1) & 7)) * 16;              // logical 16-byte chunk stored at physical chunk tid & 7
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      int ar = m0 + prow + 32 * i;
      ar = ar < Q.M ? ar : Q.M - 1;
      aoff[i] = (uint32_t)((int64_t)ar * Q.lda + pc);
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      int br = n0 + prow + 32 * i;
      br = br < Q.N ? br : Q.N - 1;
      woff[i] = (uint32_t)((int64_t)br * Q.ldw + pc);
    }
  }
  const char* const abase = reinterpret_cast<const char*>(Q.A);
  const char* const wbase = reinterpret_cast<const char*>(Q.W);
  auto stage_a = [&](int t) {
    t = t < nk ? t : nk - 1;
    const char* src = abase + (int64_t)t * KB;
    char* dst = smem + (t & 1) * A_SLOT;
#pragma unroll
    for (int i = 0; i < MI; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + aoff[i]), (lds_void_t*)(dst + (i * V3_THREADS + wave * 64) * 16), 16, 0, 0);
  };
  auto stage_w = [&](int t) {
    t = t < nk ? t : nk - 1;
    const char* src = wbase + (int64_t)t * KB;
    char* dst = smem_w + (t % 3) * W_SLOT;
#pragma unroll
    for (int i = 0; i < NJ; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + woff[i]), (lds_void_t*)(dst + (i * V3_THREADS + wave * 64) * 16), 16, 0, 0);
  };

  const AccLit<NJ, 0> acc{};              // the accumulators: a[0 : 4 MI NJ), asm-owned as in gemm_kernel_v3 (round 6): tile (i, j) = a[4 (i NJ + j) : + 3]
  const int frow = lane & 15, fq = lane >> 4;
  const int arow = wr * (16 * MI) + frow, brow = wc * (16 * NJ) + frow;
  // ---- MX: scale bytes of this lane's MI row tiles, 4 K-tiles per dword --------------------------------------------------------
  // scale of the weight operand: 2^0 in every byte.  OPAQUE (asm-defined): a plain constant may be re-materialised by a v_mov right in front of an
  // MFMA, and hipcc's hazard recogniser does not pad a VALU write -> scale-operand read when the reader is inline asm.
  int mx_one;
  asm volatile("v_mov_b32 %0, 0x7f7f7f7f\n\ts_nop 7\n\ts_nop 7" : "=v"(mx_one));
  uint32_t sc_cur[MX ? MI : 1], sc_nxt[MX ? MI : 1];
  u32x4_t mx_rsrc = (u32x4_t){0u, 0u, 0u, 0u};
  uint32_t mx_voff = 0;
  int mx_step = 0;                                   // bytes between two row tiles of a lane
  if constexpr (MX) {
    const uint64_t base = (uint64_t)(uintptr_t)Q.a_mx;
    mx_rsrc[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    mx_rsrc[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
    mx_rsrc[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)((int64_t)Q.M * Q.ld_mx));      // rows past M read as zero (their products are never stored)
    mx_rsrc[3] = 0x00020000u;
    mx_voff = (uint32_t)((m0 + wr * (16 * MI) + frow) * (int)Q.ld_mx);
    mx_step = __builtin_amdgcn_readfirstlane(16 * (int)Q.ld_mx);
  }
  // group g = K-tiles 4 g .. 4 g + 3; hand-waited (the compiler does not count the LDS-DMA issues of the asm statements around them)
  auto mx_fetch = [&](uint32_t (&dst)[MX ? MI : 1], int g) {
    if constexpr (MX) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst[i]) : "v"(mx_voff), "s"(mx_rsrc), "s"(i * mx_step + 4 * g) : "memory");
    }
  };
  if constexpr (MX) mx_fetch(sc_nxt, 0);          // in front of the prologue's DMA: the wait for A(0) / W(0) below covers them
  stage_a(0); stage_w(0); stage_w(1); stage_a(1);
  v3_acc_zero<4 * MI * NJ, 0>();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI + NJ) : "memory");        // A(0), W(0)
  __builtin_amdgcn_s_barrier();
  // 32-byte operands: words 0..3 = chunk fq, words 4..7 = chunk 4 + fq of the lane's row.  ONE register set for A: row tile i's fragment of tile
  // t+1 is read right behind the last MFMA of tile t that uses row tile i (phase 1 runs row tile by row tile), the last one at the start of the next
  // phase 0 -- 64 + 64 fragment registers instead of 128 + 64 (with two A sets the kernel sat at 250+ of 256 arch VGPRs and hipcc spilled: an LDS
  // address at best, and once an ACCUMULATOR, stored straight behind the inline-asm MFMA that was still writing it).
  static_assert((MI == 8 || MI == 7) && NJ == 8, "the slot tables below are written for 8 column tiles and 7 or 8 row tiles");
  i32x8_t af[MI], bl[NH], bh[NH];
  // LDS address of half h of row tile i of a slot = slot + lane part(h) + 2048 i: the swizzle term ((row >> 1) & 7) does not depend on i, so a tile
  // needs FOUR address registers (A / W x two halves: slot + lane part, made opaque per tile) and every read is `base offset:2048 i`.  Left to
  // itself hipcc hoists one register per (i, h) and slot out of the loop (`lane | 0x800 i`: 32+ of them) until the register file is full.
  typedef __attribute__((address_space(3))) const bf16x8_t* lds_frag_ptr;
  const uint32_t la0 = (uint32_t)(arow * 128 + (((0 + fq) ^ ((arow >> 1) & 7)) << 4)), la1 = (uint32_t)(arow * 128 + (((4 + fq) ^ ((arow >> 1) & 7)) << 4));
  const uint32_t lb0 = (uint32_t)(brow * 128 + (((0 + fq) ^ ((brow >> 1) & 7)) << 4)), lb1 = (uint32_t)(brow * 128 + (((4 + fq) ^ ((brow >> 1) & 7)) << 4));
  auto ld_at = [&](i32x8_t& f, int h, uint32_t base, int i) {      // base = slot + lane part of half h (a VGPR), i = row tile
    const u32x4_t w = __builtin_bit_cast(u32x4_t, *reinterpret_cast<lds_frag_ptr>(base + (uint32_t)(i * 2048)));
    f[4 * h + 0] = (int)w[0]; f[4 * h + 1] = (int)w[1]; f[4 * h + 2] = (int)w[2]; f[4 * h + 3] = (int)w[3];
  };
  auto slot_bases = [&](const char* slot, uint32_t l0, uint32_t l1, uint32_t (&out)[2]) {
    const uint32_t sb = (uint32_t)(uintptr_t)slot;
    out[0] = sb + l0; out[1] = sb + l1;
    asm volatile("" : "+v"(out[0]), "+v"(out[1]));              // opaque: computed here, once per tile and slot, not hoisted in 32 variants
  };
  // prologue: A row tiles 0..6 and W_lo of tile 0 (row tile 7 is read by tile 0's phase 0 like every tile's)
  {
    uint32_t pa[2], pw[2];
    slot_bases(smem, la0, la1, pa);
    slot_bases(smem_w, lb0, lb1, pw);
#pragma unroll
    for (int r = 0; r < NJ; ++r) ld_at(bl[r >> 1], r & 1, pw[r & 1], r >> 1);
#pragma unroll
    for (int r = 0; r < 2 * (MI - 1); ++r) ld_at(af[r >> 1], r & 1, pa[r & 1], r >> 1);
  }
  // (the host pass parses this body too, and x86's "v" constraint does not take a 256-bit operand without AVX: the function would be dropped
  // from the host object -- silently, as a deferred diagnostic -- and its launch stub with it)
#if defined(__HIP_DEVICE_COMPILE__)
#define V3F8_ONE(TILE, A_, B_) \
  asm volatile("v_mfma_f32_16x16x128_f8f6f4 a[%c0:%c1], %2, %3, a[%c0:%c1]" ::"n"(4 * (TILE)), "n"(4 * (TILE) + 3), "v"(A_), "v"(B_))
  // scaled form: src A = the weight fragment (scale 2^0: byte 0 of `one`), src B = the activation fragment, its scale = byte BT of SC
#define V3F8_SC(TILE, A_, B_, SC, OPS)                                                                                                                   \
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 a[%c0:%c1], %2, %3, a[%c0:%c1], %4, %5 " OPS ::"n"(4 * (TILE)), "n"(4 * (TILE) + 3), "v"(A_), "v"(B_), \
               "v"(mx_one), "v"(SC))
#else
#define V3F8_ONE(TILE, A_, B_) (void)(A_)
#define V3F8_SC(TILE, A_, B_, SC, OPS) (void)(A_)
#endif
#define V3F8_MFMA(BT, TILE, A_, B_, SC)                                                        \
  if constexpr (!MX) V3F8_ONE(TILE, A_, B_);                                                   \
  else if constexpr ((BT) == 0) V3F8_SC(TILE, A_, B_, SC, "op_sel:[0,0,0] op_sel_hi:[0,0,0]"); \
  else if constexpr ((BT) == 1) V3F8_SC(TILE, A_, B_, SC, "op_sel:[0,1,0] op_sel_hi:[0,0,0]"); \
  else if constexpr ((BT) == 2) V3F8_SC(TILE, A_, B_, SC, "op_sel:[0,0,0] op_sel_hi:[0,1,0]"); \
  else V3F8_SC(TILE, A_, B_, SC, "op_sel:[0,1,0] op_sel_hi:[0,1,0]")
  // One K-tile.  MORE: tile t+2 exists -> issue its DMA (compile-time: a scalar branch around each DMA issue costs the lone wave ~30 cycles of
  // instruction refetch).  Memory instruction behind MFMA m (at most one per gap):
  //   phase 0   even m < 16: W(t+2) piece m / 2 | odd m < 16: W_hi(t) half (m - 1) / 2 | m = 16, 17: A(t) row tile 7 (first used by MFMA 28)
  //   phase 1   m = 0..3, 6, 7, 10, 11: W_lo(t+1) halves | m = 4 i + 4, 4 i + 5 (i < 7): A(t+1) row tile i (free since MFMA 4 i + 3)
  //             m = 14, 15, 18, 19, 22, 23, 26, 27: A(t+2) pieces 0..7
  auto tile = [&](auto MORE_, int t, auto BT_, auto FETCH_, auto NOREAD_) {
    constexpr bool more = decltype(MORE_)::value;
    constexpr bool noread = decltype(NOREAD_)::value;     // the next tile is the TAIL (below): it reads its fragments itself -- no pre-reads of tile t + 1 here
    constexpr int bt = decltype(BT_)::value;            // MX: byte of the scale dwords = tile index inside its group of 4
    constexpr bool fetch = decltype(FETCH_)::value;     // MX: request the next group's scale dwords behind this tile's A DMA
    uint32_t pa[2], pw[2];
    slot_bases(smem + (t & 1) * A_SLOT, la0, la1, pa);
    slot_bases(smem_w + (t % 3) * W_SLOT, lb0, lb1, pw);
    {
      const uint64_t wsrc_u = v3_uniform_u64((uintptr_t)(wbase + (int64_t)(t + 2) * KB));
      char* wdst = smem_w + ((t + 2) % 3) * W_SLOT;
      static_for<NM>([&](auto m_c) AFX_INL {
        constexpr int m = decltype(m_c)::value;
        {
          const i32x8_t& wa_ = bl[m % NH];              // (locals: operands of an asm statement inside a generic lambda do not capture by themselves)
          const i32x8_t& xa_ = af[m / NH];
          const uint32_t sc_ = sc_cur[MX ? m / NH : 0];
          (void)sc_;
          V3F8_MFMA(bt, (m / NH) * NJ + m % NH, wa_, xa_, sc_);
        }
        V3_FENCE();
        if constexpr (m < 16 && (m & 1)) { constexpr int r = m >> 1; ld_at(bh[r >> 1], r & 1, pw[r & 1], NH + (r >> 1)); }
        if constexpr (m < 16 && !(m & 1) && more) {
          constexpr int q = m >> 1;
          const uint64_t src_ = wsrc_u;
          const uint32_t off_ = woff[q];
          char* const dst_ = wdst + (q * V3_THREADS + wave * 64) * 16;
          V3_DMA(src_, off_, dst_);
        }
        if constexpr (m == 16 || m == 17) ld_at(af[MI - 1], m - 16, pa[m - 16], MI - 1);
        V3_FENCE();
      });
    }
    if constexpr (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NJ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      const uint64_t asrc_u = v3_uniform_u64((uintptr_t)(abase + (int64_t)(t + 2) * KB));
      char* adst = smem + (t & 1) * A_SLOT;
      uint32_t na[2], nw[2];
      slot_bases(smem + ((t + 1) & 1) * A_SLOT, la0, la1, na);
      slot_bases(smem_w + ((t + 1) % 3) * W_SLOT, lb0, lb1, nw);
      static_for<NM>([&](auto m_c) AFX_INL {
        constexpr int m = decltype(m_c)::value;
        {
          const i32x8_t& wa_ = bh[m % NH];
          const i32x8_t& xa_ = af[m / NH];
          const uint32_t sc_ = sc_cur[MX ? m / NH : 0];
          (void)sc_;
          V3F8_MFMA(bt, (m / NH) * NJ + NH + m % NH, wa_, xa_, sc_);
        }
        V3_FENCE();
        constexpr int kind[32] = {1, 1, 1, 1, 2, 2, 1, 1, 2, 2, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3, 2, 2, 3, 3, 2, 2, 3, 3, 2, 2, 0, 0};   // 1 W_lo, 2 A, 3 DMA
        constexpr int arg[32] = {0, 1, 2, 3, 0, 1, 4, 5, 2, 3, 6, 7, 4, 5, 0, 1, 6, 7, 2, 3, 8, 9, 4, 5, 10, 11, 6, 7, 12, 13, 0, 0};
        // (MI = 7, the 224x256 shape: the same table cut at 28 MFMAs -- row tiles 0..5 are read where the 8-row table reads them, the seventh DMA piece
        // sits at m = 26 and the table's eighth is dropped)
        if constexpr (kind[m] == 1 && !noread) ld_at(bl[arg[m] >> 1], arg[m] & 1, nw[arg[m] & 1], arg[m] >> 1);
        else if constexpr (kind[m] == 2 && (arg[m] >> 1) < MI - 1 && !noread) ld_at(af[arg[m] >> 1], arg[m] & 1, na[arg[m] & 1], arg[m] >> 1);
        else if constexpr (kind[m] == 3 && arg[m] < MI && more) {
          const uint64_t src_ = asrc_u;
          const uint32_t off_ = aoff[arg[m]];
          char* const dst_ = adst + (arg[m] * V3_THREADS + wave * 64) * 16;
          V3_DMA(src_, off_, dst_);
        }
        V3_FENCE();
      });
      if constexpr (fetch) mx_fetch(sc_nxt, (t >> 2) + 1);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using T_ = std::true_type;
  using F_ = std::false_type;
  // TAIL (round 6, as gemm_kernel_v3's): the LAST K-tile is not multiplied in front of the epilogue but from inside it, row tile by row tile (NJ MFMAs of 32
  // cycles each per row tile, one at every second hook point of the previous row tile's steps).  Everything tile nk - 1 needs has landed at the mid-tile
  // barrier of tile nk - 2 (its wait is vmcnt(0)); per accumulator the order of the K-tiles is unchanged: bit-identical results (-DV3F8_TAIL=0).
  constexpr bool TAIL = V3F8_TAIL != 0;
  using TL = std::integral_constant<bool, TAIL>;
  int t = 0;
  if constexpr (MX) {
    // nk = 4 G tiles; the first group's scales were requested in front of the prologue's DMA
    auto take = [&]() {      // sc_cur <- sc_nxt (asm: the copies must not float above the wait that makes the loads' data valid)
#pragma unroll
      for (int i = 0; i < MI; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(sc_cur[i]) : "v"(sc_nxt[i]));
      asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // VALU write -> scale operand of the next v_mfma_scale: hipcc does not pad inline-asm pairs, and the
                                                         // wait states this pair needs are not documented here -- 16, once per 4 K-tiles
    };
    take();
#pragma unroll 1
    for (; t < nk - 4; t += 4) {
      tile(T_{}, t, I0{}, F_{}, F_{});
      tile(T_{}, t + 1, I1{}, T_{}, F_{});
      tile(T_{}, t + 2, I2{}, F_{}, F_{});            // its mid-tile wait leaves only W(t+4) in flight: the scale dwords have landed
      tile(T_{}, t + 3, I3{}, F_{}, F_{});
      take();
    }
    tile(T_{}, t, I0{}, F_{}, F_{});
    tile(T_{}, t + 1, I1{}, F_{}, F_{});
    tile(F_{}, t + 2, I2{}, F_{}, TL{});
    if constexpr (!TAIL) tile(F_{}, t + 3, I3{}, F_{}, F_{});
  } else {
#pragma unroll 1
    for (; t < nk - 2; ++t) tile(T_{}, t, I0{}, F_{}, F_{});
    tile(F_{}, t, I0{}, F_{}, TL{});                  // (nk >= 2: the launcher sends K >= 256 here)
    if constexpr (!TAIL) tile(F_{}, t + 1, I0{}, F_{}, F_{});
  }
  if constexpr (!TAIL) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");       // MFMA -> accumulator-read wait states (hipcc does not know the dependency)
  {
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lane2 = tid2 & 63, wave2 = __builtin_amdgcn_readfirstlane(tid2 >> 6);
    const int wr2 = wave2 >> 1, wc2 = wave2 & 1, frow2 = lane2 & 15, fq2 = lane2 >> 4;
    const int rb = m0 + wr2 * (16 * MI), cb = n0 + wc2 * (16 * NJ);
    auto epilogue = [&](const auto& hk) AFX_INL {
      using HK = std::decay_t<decltype(hk)>;
      if (Q.c8 != nullptr && cb >= Q.c8_col0) {       // (uniform) this wave's 128 columns go out as the next GEMM's block-scaled operand
        if (Q.epi == EPI_GELU) epi_store_mx8<true, MI, NJ, HK>(Q, acc, rb, cb, frow2, fq2, hk);
        else epi_store_mx8<false, MI, NJ, HK>(Q, acc, rb, cb, frow2, fq2, hk);
      } else
        epi_store_fast_any_acc<MI, NJ, true, true, false, HK>(Q, acc, rb, cb, frow2, fq2, hk);
    };
    if constexpr (TAIL) {
      // tile nk - 1: every fragment is read here, out of the slots it landed in, a few MFMAs ahead of its use: W fragments (32 bytes per lane = two reads) through
      // a ring of RING register sets -- every row tile re-reads its NJ W fragments: the epilogues' own state sits beside all accumulators --, A one row tile ahead
      const int tl = nk - 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int arow2 = wr2 * (16 * MI) + frow2, brow2 = wc2 * (16 * NJ) + frow2;      // (from the opaque thread id: nothing of the tail's addressing is live across the main loop)
      const uint32_t ta0_ = (uint32_t)(arow2 * 128 + (((0 + fq2) ^ ((arow2 >> 1) & 7)) << 4)), ta1_ = (uint32_t)(arow2 * 128 + (((4 + fq2) ^ ((arow2 >> 1) & 7)) << 4));
      const uint32_t tb0_ = (uint32_t)(brow2 * 128 + (((0 + fq2) ^ ((brow2 >> 1) & 7)) << 4)), tb1_ = (uint32_t)(brow2 * 128 + (((4 + fq2) ^ ((brow2 >> 1) & 7)) << 4));
      uint32_t pa[2], pw[2];
      slot_bases(smem + (tl & 1) * A_SLOT, ta0_, ta1_, pa);
      slot_bases(smem_w + (tl % 3) * W_SLOT, tb0_, tb1_, pw);
      constexpr int RING = 2, PTS = 16, NA = 2;
      constexpr int BT_LAST = 3;                           // MX: nk is a multiple of 4, the last tile is byte 3 of its scale dwords
      i32x8_t tw[RING], ta[NA];
      auto w_of = [&](i32x8_t& f, int g) { ld_at(f, 0, pw[0], g % NJ); ld_at(f, 1, pw[1], g % NJ); };      // W fragment of tail MFMA g = row * NJ + j
      auto a_of = [&](i32x8_t& f, int row) { ld_at(f, 0, pa[0], row); ld_at(f, 1, pa[1], row); };
      auto prime = [&]() {
#pragma unroll
        for (int g = 0; g < RING; ++g) w_of(tw[g], g);
        a_of(ta[0], 0);
      };
      auto tail_mfma = [&](auto row_c, auto j_c) AFX_INL {
        constexpr int row = decltype(row_c)::value, j = decltype(j_c)::value, g = row * NJ + j;
        {
          const i32x8_t& wa_ = tw[g % RING];
          const i32x8_t& xa_ = ta[row % NA];
          const uint32_t sc_ = sc_cur[MX ? row : 0];
          (void)sc_;
          V3F8_MFMA(BT_LAST, row * NJ + j, wa_, xa_, sc_);
        }
        V3_FENCE();
        if constexpr (g + RING < MI * NJ) w_of(tw[g % RING], g + RING);
        if constexpr (j == (NA == 2 ? 2 : NJ - 1) && row + 1 < MI) a_of(ta[(row + 1) % NA], row + 1);
        V3_FENCE();
      };
      auto hook = [&](auto code_c) AFX_INL {
        constexpr int code = decltype(code_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (code == -1) {                                    // row tile 0, behind the epilogue's preamble
          prime();
          static_for<NJ>([&](auto j_c) AFX_INL { tail_mfma(std::integral_constant<int, 0>{}, j_c); });
          asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        } else if constexpr (code == -2) {                             // everything at once (an epilogue without hook points)
          prime();
          static_for<MI>([&](auto r_c) AFX_INL { static_for<NJ>([&](auto j_c) AFX_INL { tail_mfma(r_c, j_c); }); });
          asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        } else {
          constexpr int row = code / PTS + 1, pt = code % PTS;         // one MFMA at every second point (NJ MFMAs, 16 points per row tile)
          if constexpr (row < MI && pt % 2 == 0 && pt / 2 < NJ) tail_mfma(std::integral_constant<int, row>{}, std::integral_constant<int, pt / 2>{});
          if constexpr (row < MI && pt == PTS - 1) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      const TailHook<decltype(hook)> hk{hook};
      epilogue(hk);
    } else {
      epilogue(NoHook{});
    }
  }
}

int& last_sk_cus() {           // CUs per XCD the last 8-phase launch split its tail over (0: plain launch) -- read by the stream-K tests
  static thread_local int v = 0;
  return v;
}
LaunchTimer& launch_timer() {
  static thread_local LaunchTimer t;
  return t;
}

#ifdef V3_KERNELS_ONLY      // experiment builds (register-allocation turnarounds in seconds, never the product): `-DV3_KERNELS_ONLY="8,8,false,0"` compiles ONE instance of
template __global__ void gemm_kernel_v3<V3_KERNELS_ONLY>(const GemmBatch);      // gemm_kernel_v3 and none of the launchers
template __global__ void gemm_kernel_v3s<4, 4, false, 0>(const GemmBatch);
#ifdef V3F8_ONLY
template __global__ void gemm_kernel_v3f8<V3F8_ONLY>(const GemmBatch);
#endif
}  // namespace afx
#else
// ---- tile shape / kernel choice -------------------------------------------------------------------------------------------
struct GemmMode { int impl = -1, tile = 0; };
bool gemm_qk_fusion_available();
static GemmMode& gemm_mode() {
  static GemmMode m;
  return m;
}
bool gemm_conv_stats_available() {                      // convolution launches go to the kernel whose epilogue accumulates GroupNorm sums
  if (gemm_mode().impl < 0) (void)gemm_qk_fusion_available();
  return gemm_mode().impl == 3 && gemm_mode().tile == 0;
}
bool gemm_qk_fusion_available() {
  if (gemm_mode().impl < 0) {                           // same defaults as launch_gemm's first call
    const char* e = getenv("AFX_GEMM_IMPL");
    gemm_mode().impl = (e && e[0] == '1') ? 1 : (e && e[0] == '2') ? 2 : 3;
    if (const char* t = getenv("AFX_GEMM_TILE")) gemm_mode().tile = atoi(t);
  }
  const char* k = getenv("AFX_GEMM_SK");
  const char* f = getenv("AFX_QK_FUSE");                // AFX_QK_FUSE=0: keep the separate kv_prep launch (A/B)
  return gemm_mode().impl == 3 && !(k && atoi(k) != 0) && !(f && f[0] == '0');
}
bool gemm_dropres_available() {                         // launch_gemm would take a problem with drop_on: the masked residual add exists in the one-wave-per-SIMD
  if (gemm_mode().impl < 0) (void)gemm_qk_fusion_available();      // kernel's permlane-paired epilogue only (kernel mode 3, no stream-K request)
  const char* k = getenv("AFX_GEMM_SK");
  return V3_EPI_SWAP_DEFAULT != 0 && gemm_mode().impl == 3 && !(k && atoi(k) != 0);
}
void gemm_set_mode(int impl, int tile) {
  gemm_mode().impl = (impl >= 1 && impl <= 3) ? impl : 3;
  gemm_mode().tile = (tile >= 0 && tile <= 6) ? tile : 0;
}
struct TileCfg { int tm, tn, group_m; };
// {4,4} = 128x128: 64 accumulators and 80 KiB of LDS, TWO work-groups per CU -- for launches that would leave most CUs without a
// 256x256 tile (the rank-256 LoRA products of the distillation step: N = 256 or M = 256, 12-84 tiles at 256x256)
// {8,7} = 256x224: the shape that makes the forward's N = 3072 launches (18 row tiles of the 4096 + 512 row problems x 14 column
// tiles = 252) and the N = 12288 launch (990 tiles = 3.87 rounds of 7/8-size tiles) fill their last round -- what hipBLASLt's
// MT256x224 kernels do for these shapes (1295 vs 1161 TF at 4608 x 3072 x 3072 in profiles/r02s_microbench.log)
// {7,8} = 224x256: the k|q|v^T launch of the Qwen-Image shape (4096 + 128 rows: 612 tiles of 256x256 = 2.39 rounds -> 718 tiles of 7/8 the size =
// 2.8 rounds); a wave keeps its 128 columns = one head, so the fused q / k epilogue works unchanged (FLUX's 4096 + 512 rows stay 256x256: 648 tiles)
static const TileCfg kTileCfg[6] = {{256, 256, GROUP_M}, {288, 192, 5}, {320, 192, 4}, {128, 128, 8}, {256, 224, GROUP_M}, {224, 256, GROUP_M}};

static int count_tiles(GemmBatch& batch, int tm, int tn, bool fill) {
  int total = 0;
  for (int i = 0; i < batch.nprob; ++i) {
    GemmProblem& p = batch.p[i];
    const int tiles_m = (p.M + tm - 1) / tm, tiles_n = (p.N + tn - 1) / tn;
    const int sk = (p.split_k < 1 || p.out_f32 != 3) ? 1 : p.split_k;
    if (fill) {
      p.tiles_m = tiles_m;
      p.tiles_n = tiles_n;
      p.tile_start = total;
      p.split_k = sk;
    }
    total += tiles_m * tiles_n * sk;
  }
  return total;
}

template <int MI, int NJ, bool CONV = false, int PERSIST = 0>
static hipError_t launch_v3_impl(GemmBatch& batch, int total, hipStream_t stream) {
  static bool attr = false;
  void (*kern)(const GemmBatch);
  if constexpr (4 * MI * NJ <= 64) kern = gemm_kernel_v3s<MI, NJ, CONV, PERSIST>;
  else kern = gemm_kernel_v3<MI, NJ, CONV, PERSIST>;
  if (!attr) {
    hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, v3_lds_bytes(MI, NJ));
    if (r != hipSuccess) return r;
    attr = true;
  }
  if (launch_timer().start != nullptr && launch_timer().stop != nullptr)
    hipExtLaunchKernelGGL(kern, dim3(total), dim3(V3_THREADS), v3_lds_bytes(MI, NJ), stream, launch_timer().start, launch_timer().stop, 0, batch);
  else
    hipLaunchKernelGGL(kern, dim3(total), dim3(V3_THREADS), v3_lds_bytes(MI, NJ), stream, batch);
  return hipGetLastError();
}

// AFX_GEMM_PERSIST=1 (or afx_gemm_set_persist(1)): launches with more tiles than resident work-groups run the persistent tile walk
// (grid = CUs x resident work-groups per CU); off by default -- see DESIGN 4.1 for the same-box A/B.
int& gemm_persist() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AFX_GEMM_PERSIST");
    v = e ? atoi(e) : 0;
  }
  return v;
}
template <int MI, int NJ, bool CONV = false>
static hipError_t launch_v3(GemmBatch& batch, int total, hipStream_t stream) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const int slots = cus * (4 * MI * NJ <= 64 ? 2 : 1);
  if constexpr (!CONV) {
    if (gemm_persist() == 1 && total > slots && slots % 8 == 0) return launch_v3_impl<MI, NJ, CONV, 1>(batch, slots, stream);
    if (gemm_persist() == 2 && total > slots && slots % 8 == 0) return launch_v3_impl<MI, NJ, CONV, 2>(batch, slots, stream);
  }
  // (The VAE's convolutions ran the persistent walk as an A/B in round 5 -- AFX_CONV_PERSIST, measured level, profiles/r05*: with the accumulator file asm-owned
  // (round 6) those instances no longer fit hipcc's arch VGPRs and it parked values in accumulator registers: dropped rather than shipped unsafe.)
  return launch_v3_impl<MI, NJ, CONV, 0>(batch, total, stream);
}

bool gemm_fp8_mx_ok(int64_t rows_total, int N, int K) {
  (void)rows_total; (void)N;                       // a block-scaled launch always takes the one-wave-per-SIMD kernel, whatever its tile count
  const char* e = getenv("AFX_FP8_V3");
  const char* k = getenv("AFX_GEMM_SK");
  if (gemm_mode().impl < 0) (void)gemm_qk_fusion_available();      // (reads AFX_GEMM_IMPL once, like launch_gemm's first call)
  // the same predicate launch_gemm applies: the kernel choice may have been overridden by afx_gemm_set_mode() (parity tests, A/B runs), not only by the environment
  return !(e && e[0] == '0') && !(k && atoi(k) != 0) && gemm_mode().impl == 3 && K % 512 == 0 && K >= 512;
}

hipError_t launch_gemm(GemmBatch& batch, hipStream_t stream) {
  static int cus = 256, sk_env = 0;
  static int group_m_env = 0;
  static bool init = false;
  int& impl = gemm_mode().impl;
  int& tile_env = gemm_mode().tile;
  if (!init) {
    init = true;
    if (const char* g = getenv("AFX_GEMM_GROUP_M")) group_m_env = atoi(g) > 0 ? atoi(g) : 0;
    // AFX_GEMM_IMPL: 1 = simple 2-stage kernel (reference / A-B), 2 = 8-phase kernel only, 3 (default) = one-wave-per-SIMD kernel
    // for the forward's bf16 epilogue modes (8-phase for everything else).  AFX_GEMM_TILE (impl 3): 0 = pick per launch,
    // 1 ... 6 = force 256x256 / 288x192 / 320x192 / 128x128 / 256x224 / 224x256.  afx_gemm_set_mode() overrides both (parity tests, A/B runs).
    if (impl < 0) {
      const char* e = getenv("AFX_GEMM_IMPL");
      impl = (e && e[0] == '1') ? 1 : (e && e[0] == '2') ? 2 : 3;
      if (const char* t = getenv("AFX_GEMM_TILE")) tile_env = atoi(t);
    }
    if (const char* k = getenv("AFX_GEMM_SK")) sk_env = atoi(k);             // the stream-K tail lives in the 8-phase kernel
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
    if (r != hipSuccess) return r;
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel_v2<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
    if (r != hipSuccess) return r;
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel_v2<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
    if (r != hipSuccess) return r;
  }
  // ---- one-wave-per-SIMD kernel: bf16 launches whose every problem is in a fast epilogue mode.  The tile shape is the one with
  // the least (rounds of `cus` tiles) x (tile area): the launch is as long as its fullest CU.
  bool v3_ok = impl == 3 && sk_env == 0 && batch.sk_force == 0;
  for (int i = 0; i < batch.nprob; ++i) {
    const GemmProblem& p = batch.p[i];
    const bool bf16_out = p.out_f32 == 0, f32_out = (p.out_f32 == 1 || p.out_f32 == 2) && p.epi == EPI_NONE;     // (3 = split-K slabs: 8-phase)
    v3_ok = v3_ok && (bf16_out || f32_out) && p.fp8 == 0 && p.conv_cin_tiles == 0 && p.conv_wp == 0 && p.pre == nullptr && p.K >= BK;
    if (p.drop_on && (p.epi != EPI_GATE_RES || p.gate != nullptr || p.out_f32 != 0)) return hipErrorInvalidValue;
  }
  for (int i = 0; i < batch.nprob; ++i)
    if (batch.p[i].drop_on && !v3_ok) return hipErrorInvalidValue;        // the masked residual add lives in the one-wave-per-SIMD kernel's epilogue only
  // ---- the VAE decoders' 3x3 convolutions: the same kernel with the implicit-GEMM address stream and the border-zeroing epilogue;
  // 256x128 tiles for the <= 128-channel layers (the full-resolution stage and conv_out, half of a 256-wide tile otherwise)
  bool conv_all = impl == 3 && tile_env == 0 && batch.nprob >= 1;
  for (int i = 0; i < batch.nprob; ++i) {
    const GemmProblem& p = batch.p[i];
    conv_all = conv_all && p.conv_cin_tiles > 0 && p.conv_wp > 0 && p.out_f32 == 0 && p.fp8 == 0 && p.pre == nullptr && p.epi != EPI_GELU &&
               p.split_k <= 1 && p.N == batch.p[0].N && (p.up_phase == 0 || (p.epi == EPI_NONE && p.gn_stats == nullptr));
  }
  if (conv_all) {
    // 256x128 tiles for the <= 128-channel layers only: for the 128^2 stage (134 tiles of 256x256 for 256 CUs, K = 4608) 268 narrow tiles measured
    // 116 us per launch against 93
    const bool narrow = batch.p[0].N <= 128;
    const int total = count_tiles(batch, 256, narrow ? 128 : 256, true);
    batch.total_tiles = total;
    if (total == 0) return hipSuccess;
    batch.group_m = group_m_env ? group_m_env : GROUP_M;
    batch.sk_cus = 0;
    return narrow ? launch_v3<8, 4, true>(batch, total, stream) : launch_v3<8, 8, true>(batch, total, stream);
  }
  bool qk = false;
  for (int i = 0; i < batch.nprob; ++i) {
    const GemmProblem& p = batch.p[i];
    if (p.qk_D > 0) {
      qk = true;
      if (p.qk_D % 128 || p.N < p.qk_D || !p.qk_wk || !p.qk_wq || !p.rope_cos || !p.rope_sin || p.rope_period < 1 || p.rope_rows < 1 ||
          p.epi == EPI_GATE_RES)
        return hipErrorInvalidValue;
    }
  }
  for (int i = 0; i < batch.nprob; ++i)
    if (batch.p[i].w_perm16 || batch.p[i].bias_rows) {
      qk = true;                                         // same kernel requirement (and the 256x256 shape: tested there)
      if (batch.p[i].out_f32 != 0 || batch.p[i].epi != EPI_NONE || (batch.p[i].w_perm16 && batch.p[i].N % 16)) return hipErrorInvalidValue;
    }
  bool all_fp8 = batch.nprob >= 1;
  for (int i = 0; i < batch.nprob; ++i) all_fp8 = all_fp8 && batch.p[i].fp8 != 0;
  if (qk && !v3_ok && !all_fp8) return hipErrorInvalidValue;        // callers ask gemm_qk_fusion_available() first (fp8: checked below)
  if (v3_ok) {
    int best = 0;
    bool f32_any = false;                               // fp32-output launches: the shapes with an even number of column tiles per wave only
    for (int i = 0; i < batch.nprob; ++i) f32_any = f32_any || batch.p[i].out_f32 != 0;
    if (qk) {                                           // one head = one wave's 128 columns: the two 256-wide shapes only
      best = 0;
      if (tile_env == 6) best = 5;
      else if (tile_env == 0) {
        const int t0 = count_tiles(batch, 256, 256, false), t5 = count_tiles(batch, 224, 256, false);
        if (t0 == 0) return hipSuccess;
        static double pen_qk224 = -1;
        if (pen_qk224 < 0) {
          const char* e = getenv("AFX_GEMM_PEN_QK224");   // A/B knob (1e9 = never)
          pen_qk224 = e ? atof(e) : 1.03;
        }
        const double c0 = (double)((t0 + cus - 1) / cus) * 256, c5 = (double)((t5 + cus - 1) / cus) * 224 * pen_qk224;
        if (c5 < c0) best = 5;
      }
    } else if (tile_env >= 1 && tile_env <= 6) best = (tile_env == 5 && f32_any) ? 0 : tile_env - 1;
    else {
      double best_cost = 0;
      int tiles256 = 0;
      static double pen_224 = -1;
      if (pen_224 < 0) {
        const char* e = getenv("AFX_GEMM_PEN224");       // A/B knob: cost factor of the 256x224 shape (1e9 = never pick it)
        pen_224 = e ? atof(e) : 1.03;
      }
      for (int c = 0; c < 5; ++c) {
        const int tiles = count_tiles(batch, kTileCfg[c].tm, kTileCfg[c].tn, false);
        if (tiles == 0) return hipSuccess;
        if (c == 0) tiles256 = tiles;
        if (c == 3 && tiles256 * 2 > cus) continue;   // 128x128 only where 256x256 leaves half the CUs idle (measured: the 864-tile
                                                      // mlp GEMM as 3456 small tiles takes 362 us against 287)
        const int slots = c == 3 ? 2 * cus : cus;
        const int rounds = (tiles + slots - 1) / slots;
        // 256x256 has the best MFMA : LDS-read ratio (4 : 1 against 3.6 : 1 / 3.75 : 1) and the chip is power-capped: a tile
        // shape that fills the last round only makes every CU clock lower.  Measured with weights streaming from HBM
        // (tools/gemm_trace.hip TRACE_COLD=1, r02s): 288x192 wins 4-6 % at K = 3072 where it saves a round or fills a 216-tile
        // launch, is level at K = 12288 and loses 3 % at K = 15360 (its W slots leave the DMA the shorter lead); 320x192 never won.
        double pen = 1.0;
        if (c == 1) pen = batch.p[0].K <= 8192 ? 1.05 : 1.5;
        if (c == 2) pen = 1.10;
        if (c == 3) pen = 2.0;          // 16 MFMAs per 8 fragment reads and 4 DMA issues per k-half: the loop runs at about half rate
        if (c == 4) pen = f32_any ? 1e9 : pen_224;   // 56 MFMAs per 15 fragment reads (256x256: 64 per 16); bf16 epilogues only
        const double cost = (double)rounds * kTileCfg[c].tm * kTileCfg[c].tn * pen;
        if (c == 0 || cost < best_cost) { best = c; best_cost = cost; }
      }
    }
    const int total = count_tiles(batch, kTileCfg[best].tm, kTileCfg[best].tn, true);
    batch.total_tiles = total;
    if (total == 0) return hipSuccess;
    batch.group_m = group_m_env ? group_m_env : kTileCfg[best].group_m;
    batch.sk_cus = 0;
    return best == 0 ? launch_v3<8, 8>(batch, total, stream) : best == 1 ? launch_v3<9, 6>(batch, total, stream)
         : best == 2 ? launch_v3<10, 6>(batch, total, stream) : best == 3 ? launch_v3<4, 4>(batch, total, stream)
         : best == 4 ? launch_v3<8, 7>(batch, total, stream) : launch_v3<7, 8>(batch, total, stream);
  }
  // ---- fp8 launches with at least one full round of 256x256 tiles: the one-wave-per-SIMD fp8 kernel (AFX_FP8_V3=0: 8-phase kernel, A/B)
  {
    static int f8v3 = -1;
    if (f8v3 < 0) {
      const char* e = getenv("AFX_FP8_V3");
      f8v3 = (e && e[0] == '0') ? 0 : 1;
    }
    bool ok = f8v3 != 0 && impl == 3 && sk_env == 0 && batch.sk_force == 0 && batch.nprob >= 1;
    bool mx_any = false, mx_all = true, c8_any = false, qk_any = false;
    for (int i = 0; i < batch.nprob; ++i) {
      const GemmProblem& p = batch.p[i];
      ok = ok && p.fp8 != 0 && p.out_f32 == 0 && p.conv_cin_tiles == 0 && p.conv_wp == 0 && p.pre == nullptr && p.K % 128 == 0 && p.K >= 256 &&
           !p.w_perm16 && !p.bias_rows && p.split_k <= 1;
      qk_any = qk_any || p.qk_D > 0;
      mx_any = mx_any || p.a_mx != nullptr;
      if (p.c8 != nullptr && (p.epi == EPI_GATE_RES || p.c8_col0 % 128 || !p.c_mx || p.ldc8 % 8 || (p.gelu_col0 != 0 && p.gelu_col0 != p.c8_col0))) return hipErrorInvalidValue;
      c8_any = c8_any || p.c8 != nullptr;
      mx_all = mx_all && p.a_mx != nullptr && p.K % 512 == 0 && p.ld_mx % 4 == 0;
    }
    static int min_tiles = -1;
    if (min_tiles < 0) {
      const char* e = getenv("AFX_FP8_V3_MIN");      // fewest 256x256 tiles of a launch that takes this kernel
      min_tiles = e ? atoi(e) : cus / 2;         // 216-tile launches (N = 3072): 2.1-2.2 -> 2.7 PF; below half a round the 8-phase kernel's 2 waves per SIMD win
    }
    if (mx_any && !(ok && mx_all)) return hipErrorInvalidValue;      // block scales are this kernel's format only (callers ask gemm_fp8_mx_ok() first)
    if ((c8_any || qk_any) && !ok) return hipErrorInvalidValue;     // (the fused q / k epilogue: this kernel only; the engine asks for it with block scales only)
    if (ok && (mx_any || c8_any || qk_any || count_tiles(batch, 256, 256, false) >= min_tiles)) {
      // 224x256 (round 5): a launch costs ceil(rounds) x tile area and the fp8 kernel had ONE shape -- Qwen-Image's 4096 + 128 rows are 17 + 1
      // row tiles of 256 (N = 3072: 216 tiles = 0.84 round, N = 12288: 864 = 3.4 -> 4 rounds, N = 9216: 648 = 2.5 -> 3) but 19 + 1 of 224 (240 tiles = 0.94,
      // 960 = 3.75 -> 4, 720 = 2.8 -> 3 rounds of tiles 7/8 the size); FLUX's joint 4608 rows of the single blocks likewise (out-projection: 252 tiles).
      // AFX_FP8_TILE=1 / 2 force 256x256 / 224x256 (A/B).  The fused q / k epilogue keeps 256x256 (one head = one wave's 128 columns either way, but its
      // row-tile loop is written for 8).
      static int tile_env8 = -1;
      if (tile_env8 < 0) {
        const char* e = getenv("AFX_FP8_TILE");
        tile_env8 = e ? atoi(e) : 0;
      }
      const int t8 = count_tiles(batch, 256, 256, false), t7 = count_tiles(batch, 224, 256, false);
      const int r8 = (t8 + cus - 1) / cus, r7 = (t7 + cus - 1) / cus;
      bool use7 = !qk_any && (tile_env8 == 2 || (tile_env8 == 0 && (double)r7 * 224 * 1.02 < (double)r8 * 256));
      const int total = use7 ? count_tiles(batch, 224, 256, true) : count_tiles(batch, 256, 256, true);
      batch.total_tiles = total;
      if (total == 0) return hipSuccess;
      batch.group_m = group_m_env ? group_m_env : GROUP_M;
      batch.sk_cus = 0;
      static bool attr = false;
      if (!attr) {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel_v3f8<8, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, v3_lds_bytes(8, 8));
        if (r != hipSuccess) return r;
        r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel_v3f8<8, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, v3_lds_bytes(8, 8));
        if (r != hipSuccess) return r;
        r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel_v3f8<7, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, v3_lds_bytes(7, 8));
        if (r != hipSuccess) return r;
        r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel_v3f8<7, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, v3_lds_bytes(7, 8));
        if (r != hipSuccess) return r;
        attr = true;
      }
      const bool timed = launch_timer().start != nullptr && launch_timer().stop != nullptr;
      if (use7) {
        if (mx_any) {
          if (timed) hipExtLaunchKernelGGL((gemm_kernel_v3f8<7, 8, true>), dim3(total), dim3(V3_THREADS), v3_lds_bytes(7, 8), stream, launch_timer().start, launch_timer().stop, 0, batch);
          else hipLaunchKernelGGL((gemm_kernel_v3f8<7, 8, true>), dim3(total), dim3(V3_THREADS), v3_lds_bytes(7, 8), stream, batch);
        } else {
          if (timed) hipExtLaunchKernelGGL((gemm_kernel_v3f8<7, 8, false>), dim3(total), dim3(V3_THREADS), v3_lds_bytes(7, 8), stream, launch_timer().start, launch_timer().stop, 0, batch);
          else hipLaunchKernelGGL((gemm_kernel_v3f8<7, 8, false>), dim3(total), dim3(V3_THREADS), v3_lds_bytes(7, 8), stream, batch);
        }
        return hipGetLastError();
      }
      if (mx_any) {
        if (timed) hipExtLaunchKernelGGL((gemm_kernel_v3f8<8, 8, true>), dim3(total), dim3(V3_THREADS), v3_lds_bytes(8, 8), stream, launch_timer().start, launch_timer().stop, 0, batch);
        else hipLaunchKernelGGL((gemm_kernel_v3f8<8, 8, true>), dim3(total), dim3(V3_THREADS), v3_lds_bytes(8, 8), stream, batch);
      } else {
        if (timed) hipExtLaunchKernelGGL((gemm_kernel_v3f8<8, 8, false>), dim3(total), dim3(V3_THREADS), v3_lds_bytes(8, 8), stream, launch_timer().start, launch_timer().stop, 0, batch);
        else hipLaunchKernelGGL((gemm_kernel_v3f8<8, 8, false>), dim3(total), dim3(V3_THREADS), v3_lds_bytes(8, 8), stream, batch);
      }
      return hipGetLastError();
    }
  }
  int total = count_tiles(batch, BM, BN, true);
  batch.total_tiles = total;
  if (total == 0) return hipSuccess;
  const int group_m = group_m_env ? group_m_env : GROUP_M;
  batch.group_m = group_m;
  batch.sk_cus = 0;
  int use = impl == 1 ? 1 : 2;
  bool conv = false;
  for (int i = 0; i < batch.nprob; ++i) conv = conv || batch.p[i].conv_cin_tiles > 0;
  if (conv) use = 2;                                 // the implicit-conv addressing lives in the 8-phase kernel
  for (int i = 0; i < batch.nprob; ++i)
    if (batch.p[i].out_f32 == 3) use = 2;            // ... and so do split-K and the atomic epilogue
  bool fp8 = false;
  for (int i = 0; i < batch.nprob; ++i) fp8 = fp8 || batch.p[i].fp8 != 0;     // a launch is all-bf16 or all-fp8
  // ---- stream-K tail: only with a caller-provided slab / flag workspace (the engine's), one K, plain bf16 / fp8 output tiles
  static int sk_mode = -1, cus_per_xcd = 32;
  if (sk_mode < 0) {
    // 0 (default): off, 1: launches with >= 1 full round in front of the tail, 2: every eligible launch.  OFF by default: in the
    // FLUX forward the tail costs more than it wins (r02p, same box, 3 interleaved runs: 7.06 / 7.02 / 6.93 images/s for 0 / 1 / 2)
    // although the isolated qkv GEMM gains 6-9 % -- the 64 MB of write-through partial slabs per launch compete with the next
    // launches' operands for the L2 / Infinity Cache.  Kept (and tested through afx_linear_bf16_sk) as the measured negative result.
    const char* e = getenv("AFX_GEMM_SK");
    sk_mode = e ? atoi(e) : 0;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8)
      cus_per_xcd = prop.multiProcessorCount / 8;
  }
  // (the gate used to read sk_mode only, so afx_linear_bf16_sk -- sk_force -- compared the plain kernel with itself)
  if ((sk_mode || batch.sk_force) && use == 2 && !conv && batch.sk_slab != nullptr && batch.sk_flags != nullptr && total % 8 == 0 && cus_per_xcd <= 32) {
    bool ok = true;
    for (int i = 0; i < batch.nprob; ++i)
      ok = ok && batch.p[i].out_f32 == 0 && batch.p[i].split_k == 1 && batch.p[i].K == batch.p[0].K && batch.p[i].fp8 == batch.p[0].fp8;
    const int T = total / 8, C = cus_per_xcd;
    const int full = T / C * C, rem = T - full;
    const int nk = batch.p[0].K * (fp8 ? 1 : 2) / (BK * 2);
    // Worth it when the tail round is visibly under-filled and a piece keeps a useful number of K-tiles.  Launches that are ONE
    // under-filled round (216 tiles: full == 0) split EVERY tile: 256 partial slabs = 64 MB written through + read back per
    // launch, which costs what the 16 % tail would win (r02d: 4608x3072xK, K = 3072 / 12288 / 15360: -10 % / -3 % / -1.5 %);
    // launches with full rounds in front split only the remainder tiles (N = 9216: +6 %, N = 12288: +3 %).
    if (ok && rem > 0 && rem * 8 <= C * 7 && (int64_t)rem * nk / C >= 6 && (full > 0 || sk_mode >= 2 || batch.sk_force)) {
      batch.sk_cus = C;
      batch.sk_tiles_per_xcd = T;
      batch.sk_full = full;
      total = 8 * (full + C);
    }
  }
  last_sk_cus() = batch.sk_cus;
  if (fp8 && launch_timer().start != nullptr && launch_timer().stop != nullptr)
    hipExtLaunchKernelGGL(gemm_kernel_v2<true>, dim3(total), dim3(GEMM_THREADS), GEMM_LDS_BYTES, stream, launch_timer().start,
                          launch_timer().stop, 0, batch);
  else if (fp8)
    hipLaunchKernelGGL(gemm_kernel_v2<true>, dim3(total), dim3(GEMM_THREADS), GEMM_LDS_BYTES, stream, batch);
  else if (use == 1)
    hipLaunchKernelGGL(gemm_bf16_kernel, dim3(total), dim3(GEMM_THREADS), GEMM_LDS_BYTES, stream, batch);
  else if (launch_timer().start != nullptr && launch_timer().stop != nullptr)
    hipExtLaunchKernelGGL(gemm_kernel_v2<false>, dim3(total), dim3(GEMM_THREADS), GEMM_LDS_BYTES, stream, launch_timer().start,
                          launch_timer().stop, 0, batch);
  else
    hipLaunchKernelGGL(gemm_kernel_v2<false>, dim3(total), dim3(GEMM_THREADS), GEMM_LDS_BYTES, stream, batch);
  return hipGetLastError();
}

}  // namespace afx
#endif      // V3_KERNELS_ONLY
