// Joint flash attention forward for gfx950 -- ONE wave per SIMD, 64 queries per wave, hand-placed instruction stream.
// head_dim 128, bf16 in / out, no mask, any S > 64.  The MMDiT product path (afx_attn.hip's 4-wave kernel stays for shorter
// sequences and the text encoders' EXT variants; launch_attention dispatches).
//
// Same mathematics and operand layouts as attention_kernel (afx_attn.hip): both products transposed on v_mfma_f32_32x32x16_bf16,
//     S^T = K . Q^T   D[key][query]         O^T = V^T . P^T   D[d][query]
// so lane (q, hi) owns query q of a 32-query slab in both, the softmax row is lane-local, and the exponentiated scores are
// already the B operand of the second product given the key permutation baked into V^T (key_of_pos).
//
// What is different -- the structure DESIGN.md 4.2 / 12.1 asked for (round 2 measured two compiler-scheduled waves per SIMD
// over-subscribing the issue port: ~1550 issue cycles per wave and tile beside 1024 matrix-pipe cycles):
//   * a work-group = 4 waves = one 256-query block, one wave per SIMD with the whole 512-register file:
//       accumulator file: O^T 2 slabs x 4 d-tiles x 16 = 128 | Q^T fragments 2 x 8 x 4 = 64 | K fragments 2 x 8 x 4 = 64
//       arch VGPRs:       S^T 2 slabs x 2 key blocks x 16 = 64 | P^T 2 x 16 | V^T fragments 64 | addresses, m, l
//     K / V^T fragments are read ONCE per 64 queries (32 + 32 ds_read_b128 per 64 MFMAs; the 4-wave kernel: per 32).
//   * the two 32-query slabs A, B of a wave run half a tile apart: while the matrix pipe does slab A's 32 MFMAs of a tile
//     (S_A^T(t+1) = K(t+1) Q_A^T, then O_A^T += V^T(t) P_A^T(t)), the VALU does slab B's softmax of tile t, and vice versa --
//     the overlap two independent work-groups per CU gave by accident, now by construction inside one in-order wave.
//   * every instruction of the loop is an `asm volatile` statement, so the source order IS the issue order: after each MFMA
//     at most one memory instruction (ds_read_b128 or LDS-DMA) and 4-5 VALU (MI355X_MICROARCH: <= 5 single-issue fillers hide
//     in a 32-cycle MFMA gap of a lone wave).  The stream is GENERATED (tools/gen_attn3.py -> gen/a3_*.inc, committed) and all
//     wide operands are ASM-OWNED: literal register names in the instruction text (map in the generator's header), while
//     `amdgpu_num_vgpr(96)` confines hipcc's own allocation to v[0:95].  Left to allocate 461 of 512 registers itself hipcc
//     split live ranges and spilled (464 registers, 2600 accumulator moves in the first build; the generator's header lists
//     what else was tried).  hipcc inserts no waits and no hazard padding inside this stream, so every dependency is kept
//     >= 16 MFMAs apart by the phase structure or carries an explicit s_nop / s_waitcnt; arcflow_amd/build.py audits the ISA
//     after every build (no accumulator move, no scratch access outside the asm statements).
//   * K and V^T tiles stream by LDS-DMA into two 4-slot rings (128 KiB): K(t+4) and V^T(t+2) are issued in iteration t, ONE
//     `s_waitcnt vmcnt(8)` + s_barrier per tile retires everything issued two iterations ago (K(t+2), V^T(t)): two tile times
//     of lead for the K read at the end of iteration t, and never vmcnt(0) in the loop.  The loop is unrolled x4 so that ring
//     slots are immediates of the ds_read / M0 offsets (no address arithmetic in the loop).
//   * the O rescale is deferred (row max grown by more than 2^5, as in the 4-wave kernel) and lives on a cold path
//     (accumulator-file reads / writes with explicit wait states).
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include <hip/hip_ext.h>

#include "afx_common.h"
#include "afx_kernels.h"

namespace afx {
namespace a3 {

constexpr int KVB = 64;                 // keys per tile
constexpr int QBLK = 256;               // queries per work-group (4 waves x 2 slabs x 32)
constexpr int THREADS = 256;
constexpr int TILE_BYTES = KVB * 128 * 2;            // 16 KiB: K tile [64 keys][128 d] or V^T tile [128 d][64 keys]
constexpr int SLOTS = 4;
constexpr int V_BASE = SLOTS * TILE_BYTES;           // V^T ring behind the K ring
constexpr int LDS_BYTES = 2 * SLOTS * TILE_BYTES;    // 131072
constexpr float RESCALE_LOG2 = 5.0f;

#ifdef AFX_ATTN_TRACE
__device__ unsigned g_attn3_trace[2 * 4 * 16];
#endif

AFX_DEV uint64_t uniform_u64(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// the generated pieces: gen/ by default; -DA3_GEN=<dir> selects another output of tools/gen_attn3.py (timing ablations)
#ifndef A3_GEN
#define A3_GEN gen
#endif
#define A3_XSTR(x) #x
#define A3_STR(x) A3_XSTR(x)
#define A3_INC(name) A3_STR(A3_GEN/name)
#include A3_INC(a3_rescale.inc)
#include A3_INC(a3_readout.inc)
#include A3_INC(a3_mask.inc)

// One tile = iteration t with ring slot J = t & 3 (gen/a3_body{J}.inc):
//   phase A   MFMA: S_A^T(t+1) (16), O_A^T += V^T(t) P_A^T(t) (16)   VALU: softmax of S_B(t)     LDS: V^T(t) fragments | DMA K(t+4)
//   phase B   MFMA: S_B^T(t+1) (16), O_B^T += V^T(t) P_B^T(t) (16)   VALU: softmax of S_A(t+1)   DMA V^T(t+2) | LDS: K(t+2) fragments
// Balanced last round (round 5; launch_attention_v3 builds the table): 432 work-groups on 256 CUs are 1.69 rounds and cost two
// (tools/round_probe.py: a launch's time follows ceil(rounds), not its work).  With a work table a work-group runs a list of SEGMENTS =
// (head, 256-query block, key tiles [t0, t0 + n)).  The blocks of the under-filled last round are cut once, at key tile L: the SHORT ends
// [L, ntiles) run first, beside the previous round's whole blocks, and leave a PARTIAL (O normalised by its own row sum as bf16 [256][128] + the
// row's log2-sum-exp, published with an agent-scope release + a generation flag); the LONG parts [0, L) run last, take the partials as their
// initial (m, l, O) state -- m = lse / c, l = 1, O = the partial rows (several partials: merged with exp2(lse_p - max) weights) -- and write the
// final rows themselves: no merge pass, no atomics, and every CU ends up with (blocks x tiles) / CUs key tiles.  A long part that does not see its
// partials published within a bounded number of polls (HIP promises no dispatch order) computes the whole key range itself.
// Softmax of one slab and tile = 134 instructions (gen_attn3.softmax_ops): two interleaved v_max3 chains, the xor-32 exchange,
// the rescale decision (cold branch), then per score fma / exp2 / row-sum add and per pair cvt_pk, software-pipelined so that a
// v_exp_f32 result is never consumed by the next instruction (trans -> VALU use needs a wait state hipcc cannot add here).
constexpr int MAX_SEG = 8;
struct Seg { int h, b, qb, t0, n, out, nin, in0; };   // out >= 0: write partial slot `out` (else the final rows); nin > 0: start from partial slots [in0, in0 + nin) (then t0 = 0)
struct Item { int nseg, pad_[3]; Seg seg[MAX_SEG]; }; // 272 bytes: one work-group's list

__global__ __launch_bounds__(THREADS, 1) __attribute__((amdgpu_num_vgpr(96))) void attention_v3_kernel(const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k,
                                                                  int64_t ldk, const bf16_t* __restrict__ vt, bf16_t* __restrict__ o,
                                                                  int64_t ldo, int H, int S, int S_pad, int nqb, int B, float* __restrict__ lse, int dbg,
                                                                  uint8_t* __restrict__ o8, int64_t ldo8, uint8_t* __restrict__ omx, int64_t ld_omx,
                                                                  const Item* __restrict__ work, bf16_t* __restrict__ po, float* __restrict__ plse,
                                                                  int* __restrict__ pflag, int gen, int wt, unsigned long long* __restrict__ tl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nseg = work != nullptr ? __builtin_amdgcn_readfirstlane(work[blockIdx.x].nseg) : 1;
  if (tl != nullptr && threadIdx.x == 0) tl[4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();       // (afx_debug_attn_timeline: when each work-group ran, 100 MHz ticks; [1], [2]: its LAST segment's prologue issued / main loop entered)
#pragma unroll 1
  for (int si = 0; si < nseg; ++si) {
  // every lane constant is derived INSIDE the segment loop from an opaque copy of threadIdx: nothing a segment computes lives through the
  // previous segment's epilogue (hipcc's 96 registers are full in the main loop; with the constants hoisted it parked values in the asm-owned accumulator file)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hi = lane >> 5;
  int h, b, qblk, t0, ntiles, pidx, nin = 0, in0 = 0;
  if (work != nullptr) {                                         // (uniform)
    const Seg& sg = work[blockIdx.x].seg[si];
    h = __builtin_amdgcn_readfirstlane(sg.h); b = __builtin_amdgcn_readfirstlane(sg.b); qblk = __builtin_amdgcn_readfirstlane(sg.qb);
    t0 = __builtin_amdgcn_readfirstlane(sg.t0); ntiles = __builtin_amdgcn_readfirstlane(sg.n); pidx = __builtin_amdgcn_readfirstlane(sg.out);
    nin = __builtin_amdgcn_readfirstlane(sg.nin); in0 = __builtin_amdgcn_readfirstlane(sg.in0);
    if (nin > 0) {
      // consumer side of the hand-over: ONE lane polls the generation flags (relaxed, agent scope, bounded), the verdict goes through LDS (free between
      // segments), then every wave takes an agent-scope acquire before the plain loads of the partial rows
      if (tid == 0) {
        int ok = (wt & 2) ? 0 : 1;               // (bit 1 of the mode word, AFX_ATTN_HANDOVER=lost: pretend no partial is ever published -- the tests' way into the fallback)
        for (int p = 0; p < nin && ok; ++p) {
          int spins = 0;
          while (__hip_atomic_load(pflag + in0 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) {
            if (++spins > (1 << 15)) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(32);
          }
        }
        *reinterpret_cast<volatile int*>(smem) = ok;
      }
      __syncthreads();
      const int ok = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(smem));
      __syncthreads();                                           // (the word is read before the prologue's DMA may overwrite it)
      if (ok) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      } else {                                                   // never published (dispatch order?): the whole key range, from zero
        nin = 0;
        ntiles = S_pad / KVB;
      }
    }
  } else {
    // head -> XCD affinity as in the 4-wave kernel: XCD x owns heads x, x + 8, ...
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int per_head = nqb * B;
    h = xcd + 8 * (slot_id / per_head);
    if (h >= H) return;
    const int rem = slot_id % per_head;
    b = rem / nqb;
    qblk = rem % nqb;
    t0 = 0;
    ntiles = S_pad / KVB;                                        // >= 2 (launcher); S_pad = roundup(S, 64), V^T is zero-padded to it
    pidx = -1;
  }
  const int q0 = qblk * QBLK + wave * 64;
  const int KS = min(S - t0 * KVB, ntiles * KVB), KS_pad = ntiles * KVB;      // keys of this segment / its tiles x 64 (ragged only in the block's last segment)
  const float c = 0.08838834764831845f * 1.4426950408889634f;   // 1/sqrt(128) * log2(e)
  const float neg_c = -c;
  const float thr = RESCALE_LOG2 / c;

  // bring-up aid (AFX_ATTN3_DBG=n, 0 in production): the whole work-group leaves after stage n with everything drained
#define A3_DBG(n)                                                      \
  if (dbg == (n)) {                                                    \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        \
    return;                                                            \
  }
  A3_DBG(1)
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t wave_lds = lds0 + wave * 1024;

  // ---- lane constants ---------------------------------------------------------------------------------------------------------
  // fragment read addresses inside a slot (swizzles as the DMA sources below): K row ql (+32 kb), chunk (2s + hi) ^ (row & 15);
  // V^T row ql (+32 d), chunk (2g + hi) ^ ((row >> 1) & 7)
#define A3_KADDR(s) (lds0 + ql * 256 + (((2 * (s) + hi) ^ (ql & 15)) << 4))
#define A3_VADDR(g) (lds0 + V_BASE + ql * 128 + (((2 * (g) + hi) ^ ((ql >> 1) & 7)) << 4))
  const uint32_t kaddr0 = A3_KADDR(0), kaddr1 = A3_KADDR(1), kaddr2 = A3_KADDR(2), kaddr3 = A3_KADDR(3), kaddr4 = A3_KADDR(4),
                 kaddr5 = A3_KADDR(5), kaddr6 = A3_KADDR(6), kaddr7 = A3_KADDR(7);
  const uint32_t vaddr0 = A3_VADDR(0), vaddr1 = A3_VADDR(1), vaddr2 = A3_VADDR(2), vaddr3 = A3_VADDR(3);
  // DMA sources: K piece i = rows 16 i + tid / 16, physical chunk tid % 16; V^T piece i = rows 32 i + tid / 8, physical chunk tid % 8
  const int dk_r = tid >> 4, dk_c = ((tid & 15) ^ (dk_r & 15)) << 3;
  const int dv_r = tid >> 3, dv_c = ((tid & 7) ^ ((dv_r >> 1) & 7)) << 3;
#define A3_KOFF(i) ((uint32_t)(((int64_t)(dk_r + 16 * (i)) * ldk + dk_c) * 2))
#define A3_VOFF(i) ((uint32_t)(((int64_t)(dv_r + 32 * (i)) * S_pad + dv_c) * 2))
  const uint32_t koff0 = A3_KOFF(0), koff1 = A3_KOFF(1), koff2 = A3_KOFF(2), koff3 = A3_KOFF(3);
  const uint32_t voff0 = A3_VOFF(0), voff1 = A3_VOFF(1), voff2 = A3_VOFF(2), voff3 = A3_VOFF(3);
  // Ragged S: the K rows of the LAST tile past the end are clamped to row S - 1 (finite data; their scores get -inf added in front of the last tile's softmax streams: gen/a3_mask.inc, cold),
  // so the last tile's DMA pieces use their own per-lane offsets; tiles past the end re-fetch the last one.
  const int last0 = (ntiles - 1) * KVB;
#define A3_KOFFL(i) ((uint32_t)(((int64_t)(min(last0 + dk_r + 16 * (i), KS - 1) - last0) * ldk + dk_c) * 2))
  const uint32_t koffl0 = A3_KOFFL(0), koffl1 = A3_KOFFL(1), koffl2 = A3_KOFFL(2), koffl3 = A3_KOFFL(3);
#define A3_KOFS(TILE)                                                                                                     \
  const bool klast_ = (TILE) >= ntiles - 1;                                                                               \
  const uint32_t kofs0 = klast_ ? koffl0 : koff0, kofs1 = klast_ ? koffl1 : koff1, kofs2 = klast_ ? koffl2 : koff2,       \
                 kofs3 = klast_ ? koffl3 : koff3;
  const char* kbase = reinterpret_cast<const char*>(k + ((int64_t)b * S + (int64_t)t0 * KVB) * ldk + h * 128);
  const char* vbase = reinterpret_cast<const char*>(vt + ((int64_t)(b * H + h) * 128) * S_pad + (int64_t)t0 * KVB);
  const int64_t ktile_bytes = (int64_t)KVB * ldk * 2;
  // tiles past the end re-fetch the last one into a free slot: the DMA count per iteration stays 8
  auto k_src = [&](int tt) { tt = tt < ntiles ? tt : ntiles - 1; return uniform_u64((uint64_t)(uintptr_t)(kbase + tt * ktile_bytes)); };
  auto v_src = [&](int tt) { tt = tt < ntiles ? tt : ntiles - 1; return uniform_u64((uint64_t)(uintptr_t)(vbase + (int64_t)tt * (KVB * 2))); };

#include A3_INC(a3_init.inc)
  float mA = -INFINITY, lA0 = 0.f, lA1 = 0.f, mB = -INFINITY, lB0 = 0.f, lB1 = 0.f;
  float rs_tmp;
  (void)rs_tmp;
  // cold path: advance the running max of a slab and rescale its l and O^T (O^T: gen/a3_rescale.inc)
#define A3_RESCALE_A(M_NEW)                                        \
  const float alpha = __builtin_amdgcn_exp2f((mA - (M_NEW)) * c); \
  mA = (M_NEW);                                                    \
  lA0 *= alpha;                                                    \
  lA1 *= alpha;                                                    \
  A3_OSCALE_A
#define A3_RESCALE_B(M_NEW)                                        \
  const float alpha = __builtin_amdgcn_exp2f((mB - (M_NEW)) * c); \
  mB = (M_NEW);                                                    \
  lB0 *= alpha;                                                    \
  lB1 *= alpha;                                                    \
  A3_OSCALE_B

  // ---- prologue: Q fragments (accumulator file), K(0..3), V^T(0..1) ------------------------------------------------------------
  {
    const int r0 = min(q0 + ql, S - 1), r1 = min(q0 + 32 + ql, S - 1);
    const bf16_t* qptr0 = q + ((int64_t)b * S + r0) * ldq + h * 128 + hi * 8;
    const bf16_t* qptr1 = q + ((int64_t)b * S + r1) * ldq + h * 128 + hi * 8;
#include A3_INC(a3_qload.inc)
  }
#include A3_INC(a3_prologue_dma.inc)
  A3_DBG(2)
  if (tl != nullptr && threadIdx.x == 0) tl[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  if (nin > 0) {      // (uniform) continue from the handed-over partial results of this block's other key tiles: m = lse / c, l = sum of weights, O = weighted rows
    // (the prologue's loads stay in flight; hipcc's own waits for these loads are conservative -- VMEM retires in order -- and the counted waits of tile 0 only
    // ever wait for MORE than they need when other loads are outstanding)
    const float inv_c = 1.0f / c;
    float ox[16];
    // Plain loads: the acquire above emptied this CU's L1, and no L2 can hold an older copy of a slot -- a launch starts with the L2s' stale lines
    // invalidated and nobody reads a slot before its producer has written it through.  The weights of the (at most 4) partials are formed once per
    // slab and every load of a tile is independent of the others, so hipcc batches them (the first version paid two serial fabric latencies per tile: 13-16 us).
#define A3_LD_LSE(P) plse[(int64_t)(in0 + (P)) * QBLK + rloc_]
#define A3_INIT_TILE(SL, D)                                                                                        \
    {                                                                                                              \
      _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) ox[r_] = 0.f;                                              \
      _Pragma("unroll") for (int p_ = 0; p_ < 4; ++p_) {                                                           \
        if (p_ < nin) {                                                                                            \
          const bf16_t* src_ = po + ((int64_t)(in0 + p_) * QBLK + rloc_) * 128 + (D) * 32 + hi * 4;                \
          _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                       \
            const u32x2_t v_ = *reinterpret_cast<const u32x2_t*>(src_ + 8 * g_);                                   \
            ox[4 * g_ + 0] += w_[p_] * __uint_as_float(v_[0] << 16);                                               \
            ox[4 * g_ + 1] += w_[p_] * __uint_as_float(v_[0] & 0xffff0000u);                                       \
            ox[4 * g_ + 2] += w_[p_] * __uint_as_float(v_[1] << 16);                                               \
            ox[4 * g_ + 3] += w_[p_] * __uint_as_float(v_[1] & 0xffff0000u);                                       \
          }                                                                                                        \
        }                                                                                                          \
      }                                                                                                            \
      A3_WRITE_##SL##_##D                                                                                          \
    }
#define A3_INIT_SLAB(SL, MV, L0V, L1V)                                                                             \
    {                                                                                                              \
      const int rloc_ = wave * 64 + (SL) * 32 + ql;                                                                \
      float w_[4], lmax_ = -INFINITY, wsum_ = 0.f;                                                                 \
      _Pragma("unroll") for (int p_ = 0; p_ < 4; ++p_) { w_[p_] = p_ < nin ? A3_LD_LSE(p_) : -INFINITY; lmax_ = fmaxf(lmax_, w_[p_]); } \
      _Pragma("unroll") for (int p_ = 0; p_ < 4; ++p_) { w_[p_] = __builtin_amdgcn_exp2f(w_[p_] - lmax_); wsum_ += w_[p_]; }           \
      A3_INIT_TILE(SL, 0) A3_INIT_TILE(SL, 1) A3_INIT_TILE(SL, 2) A3_INIT_TILE(SL, 3)                              \
      MV = lmax_ * inv_c;                                                                                          \
      L0V = hi == 0 ? wsum_ : 0.f;      /* a row's sum lives as two partial sums, one per half-wave */                \
      L1V = 0.f;                                                                                                   \
    }
    A3_INIT_SLAB(0, mA, lA0, lA1)
    A3_INIT_SLAB(1, mB, lB0, lB1)
    asm volatile("s_nop 7" ::: "memory");                          // accumulator write -> MFMA SrcC
  }

#ifdef AFX_ATTN_TRACE
  unsigned tr[16];
  const unsigned tr_c0 = (unsigned)__builtin_readcyclecounter(), tr_r0 = (unsigned)__builtin_amdgcn_s_memrealtime();
#define A3_TR(i)                                                                        \
  if (t == 36) {                                                                        \
    uint64_t st_;                                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_)::"memory");        \
    tr[i] = (unsigned)st_;                                                              \
  }
#else
#define A3_TR(i)
#endif

  // tile 0, not pipelined: K(0) fragments, S^T(0) of both slabs, slab A's softmax, K(1) fragments
#include A3_INC(a3_tile0.inc)
  A3_DBG(3)
  if (tl != nullptr && threadIdx.x == 0) tl[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();

  int t = 0;
  if (ntiles > 1) {
#pragma unroll 1
    for (;;) {
      {
#include A3_INC(a3_body0.inc)
      }
      A3_DBG(4)
      if (++t == ntiles - 1) break;
      {
#include A3_INC(a3_body1.inc)
      }
      if (++t == ntiles - 1) break;
      {
#include A3_INC(a3_body2.inc)
      }
      if (++t == ntiles - 1) break;
      {
#include A3_INC(a3_body3.inc)
      }
      if (++t == ntiles - 1) break;
    }
  }
  A3_DBG(5)
  // ---- last tile t = ntiles - 1: V^T(t) fragments | softmax of S_B(t) | O_A^T += ..., O_B^T += ... ---------------------------
  {
    const uint32_t vs = (uint32_t)(t & 3) * TILE_BYTES;
#include A3_INC(a3_final.inc)
  }
  A3_DBG(6)
  // every DMA piece must have landed before this work-group's LDS can be handed to another one; MFMA -> accumulator-read wait states
  A3_DRAIN
  A3_DBG(7)
  // lane constants of the epilogue from an OPAQUE copy of threadIdx: computed before the loop they would have to live through it
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const int ql2 = tid2 & 31, hi2 = (tid2 >> 5) & 1;
  const int q02 = qblk * QBLK + __builtin_amdgcn_readfirstlane(tid2 >> 6) * 64;
  // where this segment's rows go: the caller's O / lse, or partial slot pidx ([256][128] bf16 + [256] f32, rows relative to the block)
  const bool part = pidx >= 0;                                   // (uniform)
  bf16_t* const obase = part ? po + ((int64_t)pidx * QBLK - (int64_t)qblk * QBLK) * 128 : o + (int64_t)b * S * ldo + h * 128;
  const int64_t ld_o = part ? 128 : ldo;
  float* const lbase = part ? plse + ((int64_t)pidx * QBLK - (int64_t)qblk * QBLK) : (lse != nullptr ? lse + ((int64_t)b * H + h) * S_pad : nullptr);

  // ---- normalise and store, one 32-row tile of O^T at a time: lane (q, hi) holds O[q][32 d + 8 g + 4 hi + 0..3] in ox[4 g + 0..3].
  // One permlane32 exchange per register pairs the two half-waves' 8-byte pieces into 16 contiguous bytes per lane (8 instead of
  // 16 stores per slab).
  float ox[16];
  auto store_tile = [&](int sl, int d, float inv, int row, bf16_t* op) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int g0 = 2 * kk, g1 = 2 * kk + 1;
      const uint32_t ax = pack_bf16x2(ox[4 * g0 + 0] * inv, ox[4 * g0 + 1] * inv);
      const uint32_t ay = pack_bf16x2(ox[4 * g0 + 2] * inv, ox[4 * g0 + 3] * inv);
      const uint32_t bx = pack_bf16x2(ox[4 * g1 + 0] * inv, ox[4 * g1 + 1] * inv);
      const uint32_t by = pack_bf16x2(ox[4 * g1 + 2] * inv, ox[4 * g1 + 3] * inv);
      const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
      const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
      const u32x4_t val = (u32x4_t){rx[0], ry[0], rx[1], ry[1]};
      // write-through: the payload of a hand-over (see the publish below).  The s_nop is the wait a VALU write of a > 8-byte store's data registers needs
      // behind the store (gfx940+: 2 states) -- hipcc's hazard recogniser cannot see into the asm, and without it one build stored the NEXT step's values in some lanes
      if (part && (wt & 1)) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(op + d * 32 + kk * 16), "v"(val) : "memory");
      else if (part || row < S) *reinterpret_cast<u32x4_t*>(op + d * 32 + kk * 16) = val;
    }
  };
#define A3_STORE_SLAB(SL, LSUM, MRUN)                                                                              \
  {                                                                                                                \
    const float lsum = (LSUM);                                                                                     \
    const float l_tot = lsum + __shfl_xor(lsum, 32, 64);                                                           \
    const float inv = 1.0f / l_tot;                                                                                \
    const int row = q02 + (SL) * 32 + ql2;                                                                         \
    if (part) {                                                                                                    \
      const float lv_ = (MRUN) * c + __log2f(l_tot);                                                               \
      if (hi2 == 0 && (wt & 1)) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(lbase + row), "v"(lv_) : "memory"); \
      else if (hi2 == 0) lbase[row] = lv_;                                                                         \
    } else if (lbase != nullptr && hi2 == 0 && row < S) lbase[row] = (MRUN) * c + __log2f(l_tot);                  \
    bf16_t* op = obase + (int64_t)row * ld_o + hi2 * 8;                                                            \
    A3_READ_##SL##_0 store_tile(SL, 0, inv, row, op);                                                              \
    A3_READ_##SL##_1 store_tile(SL, 1, inv, row, op);                                                              \
    A3_READ_##SL##_2 store_tile(SL, 2, inv, row, op);                                                              \
    A3_READ_##SL##_3 store_tile(SL, 3, inv, row, op);                                                              \
  }
  // o8 != nullptr: O leaves as the next fp8 GEMM's block-scaled operand instead of bf16 (afx_common.h mx_exp / mx_inv: one E8M0 byte per row and
  // head -- a head's 128 columns are one block; a row's values sit in the two half-waves' lanes).  Two passes over the accumulator file (the
  // block maximum first), 8 instead of 16 bytes per lane and store, no quantisation pass behind the attention.
  auto store_tile8 = [&](int d, float sc, int row, uint8_t* op) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int g0 = 2 * kk, g1 = 2 * kk + 1;
      int a = 0, b_ = 0;
      a = __builtin_amdgcn_cvt_pk_fp8_f32(ox[4 * g0 + 0] * sc, ox[4 * g0 + 1] * sc, a, false);
      a = __builtin_amdgcn_cvt_pk_fp8_f32(ox[4 * g0 + 2] * sc, ox[4 * g0 + 3] * sc, a, true);
      b_ = __builtin_amdgcn_cvt_pk_fp8_f32(ox[4 * g1 + 0] * sc, ox[4 * g1 + 1] * sc, b_, false);
      b_ = __builtin_amdgcn_cvt_pk_fp8_f32(ox[4 * g1 + 2] * sc, ox[4 * g1 + 3] * sc, b_, true);
      const auto r = __builtin_amdgcn_permlane32_swap((uint32_t)a, (uint32_t)b_, false, false);
      if (row < S) *reinterpret_cast<u32x2_t*>(op + d * 32 + kk * 16) = (u32x2_t){r[0], r[1]};
    }
  };
  auto amax16 = [&](float am) {
#pragma unroll
    for (int j = 0; j < 16; ++j) am = fmaxf(am, fabsf(ox[j]));
    return am;
  };
#define A3_STORE_SLAB_MX8(SL, LSUM, MRUN)                                                                          \
  {                                                                                                                \
    const float lsum = (LSUM);                                                                                     \
    const float l_tot = lsum + __shfl_xor(lsum, 32, 64);                                                           \
    const float inv = 1.0f / l_tot;                                                                                \
    const int row = q02 + (SL) * 32 + ql2;                                                                         \
    if (lse != nullptr && hi2 == 0 && row < S) lse[((int64_t)b * H + h) * S_pad + row] = (MRUN) * c + __log2f(l_tot); \
    float am = 0.f;                                                                                                \
    A3_READ_##SL##_0 am = amax16(am);                                                                              \
    A3_READ_##SL##_1 am = amax16(am);                                                                              \
    A3_READ_##SL##_2 am = amax16(am);                                                                              \
    A3_READ_##SL##_3 am = amax16(am);                                                                              \
    am = fmaxf(am, __shfl_xor(am, 32, 64)) * inv;                                                                  \
    const int eb = mx_exp(am);                                                                                     \
    const float sc = inv * mx_inv(eb);                                                                             \
    uint8_t* op = o8 + ((int64_t)b * S + row) * ldo8 + h * 128 + hi2 * 8;                                          \
    A3_READ_##SL##_0 store_tile8(0, sc, row, op);                                                                  \
    A3_READ_##SL##_1 store_tile8(1, sc, row, op);                                                                  \
    A3_READ_##SL##_2 store_tile8(2, sc, row, op);                                                                  \
    A3_READ_##SL##_3 store_tile8(3, sc, row, op);                                                                  \
    if (hi2 == 0 && row < S) omx[((int64_t)b * S + row) * ld_omx + h] = (uint8_t)eb;                               \
  }
  if (o8 != nullptr) {        // (uniform)
    A3_STORE_SLAB_MX8(0, lA0 + lA1, mA)
    A3_STORE_SLAB_MX8(1, lB0 + lB1, mB)
  } else {
    A3_STORE_SLAB(0, lA0 + lA1, mA)
    A3_STORE_SLAB(1, lB0 + lB1, mB)
  }
#ifdef AFX_ATTN_TRACE
  if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 300)) {
    unsigned* t4 = g_attn3_trace + ((blockIdx.x ? 1 : 0) * 4 + wave) * 16;
    t4[0] = (unsigned)__builtin_readcyclecounter() - tr_c0;
    t4[1] = (unsigned)__builtin_amdgcn_s_memrealtime() - tr_r0;
    t4[2] = (unsigned)ntiles;
    t4[3] = tr[4] - tr[3];      // wait for the DMA of two iterations ago + own LDS reads
    t4[4] = tr[0] - tr[4];      // barrier
    t4[5] = tr[1] - tr[0];      // phase A
    t4[6] = tr[2] - tr[1];      // phase B
  }
#endif
  if (part) {                   // producer side of the hand-over: the rows went out as write-through (sc0 sc1) stores -- no L2 write-back fence, which stalls the
                                // whole XCD's L2 for microseconds per publish -- every wave drains its own, then ONE lane raises the slot's generation flag (sc1 store)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid2 == 0) {
      if (!(wt & 1)) {          // (AFX_ATTN_HANDOVER=fence, A/B: plain stores + an agent-scope release)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_store(pflag + pidx, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (si + 1 < nseg) {          // the next segment's DMA overwrites ring slots other waves may still be reading
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  }   // segments
  if (tl != nullptr && threadIdx.x == 0) tl[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace a3

extern "C" int afx_debug_attn3_trace(unsigned* host_out) {      // [2 blocks][4 waves][16]
#ifdef AFX_ATTN_TRACE
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(a3::g_attn3_trace), 2 * 4 * 16 * sizeof(unsigned)) == hipSuccess ? 0 : -1;
#else
  (void)host_out;
  return -1;
#endif
}

bool attention_v3_eligible(int S) { return S > a3::KVB; }        // >= 2 KV tiles (the pipeline's tile 0 is never the last one)

namespace {
// The balanced schedule of one (B, H, S) shape on one stream (see the kernel's header): per XCD the 256-query blocks of its heads in dispatch
// order.  How work-groups reach CUs (tools/dispatch_probe.hip, a one-work-group-per-CU kernel with table-driven run times; for speed only, nothing
// below depends on it for correctness): work-group b runs on XCD b % 8; inside an XCD the j-th work-group goes to shader engine perm[j % 4] (8 CUs
// each) and waits -- IN ORDER, blocking every work-group behind it -- until that engine has a free CU.  So a schedule is four interleaved
// sub-lists, one per engine, and the start times it plans must not decrease along the merged list.  With R whole rounds and `rem` blocks left
// over, the last whole round and the remainder are dealt to the engines (8 whole blocks + rem_k = rem / 4 (+ 1) long parts each), and engine k runs
//     [rem_k whole blocks] [8 - rem_k runs of SHORT ends] [8 - rem_k whole blocks] [rem_k LONG parts]
// so that a CU gets either whole + long or short run + whole: the same number of key tiles (+ the segment overheads folded into the cut point L).
// Engines with more runs come first in the rotation: where the sub-lists differ, a whole block that starts when a run ends precedes a long part
// that starts when a whole block ends.
struct SplitPlan {
  bool split = false;
  int grid = 0, nparts = 0, gen = 0;
  a3::Item* items = nullptr;
  uint16_t* po = nullptr;
  float* plse = nullptr;
  int* pflag = nullptr;
};
constexpr int MIN_SEG = 4;          // key tiles: no segment shorter than this
constexpr int SIG_SHORT = 7;        // cost of a short segment beyond its key tiles, in key tiles (prologue, two unpipelined tiles, epilogue, publish): 11.5 us measured (tools/attn_timeline.py)
constexpr int SIG_LONG = 2;         // cost of a long part beyond a whole block's own overhead (flag wait, acquire, init from the partials)

bool build_plan(int B, int H, int S, int ncu, std::vector<a3::Item>& items, int& nparts, int& grid) {
  constexpr int NSE = 4;
  const int nqb = (S + a3::QBLK - 1) / a3::QBLK, ntiles = (int)(attn_spad(S) / a3::KVB), hx = (H + 7) / 8, cu = ncu / 8, cs = cu / NSE;
  if (cu < 2 * NSE || cu % NSE != 0 || ntiles < 4 * MIN_SEG) return false;
  std::vector<std::vector<a3::Item>> per(8);
  bool any = false;
  nparts = 0;
  for (int x = 0; x < 8; ++x) {
    std::vector<a3::Seg> blocks;
    for (int hh = 0; hh < hx; ++hh)
      for (int b = 0; b < B; ++b)
        for (int qb = 0; qb < nqb; ++qb)
          if (x + 8 * hh < H) blocks.push_back(a3::Seg{x + 8 * hh, b, qb, 0, ntiles, -1, 0, 0});
    const int nx = (int)blocks.size(), R = nx / cu, rem = nx % cu;
    auto item1 = [](const a3::Seg& sg) { a3::Item it{}; it.nseg = 1; it.seg[0] = sg; return it; };
    // engine k (in rotation order) gets rem_k long parts and cs - rem_k runs, fewest long parts first (= most runs first).  ONE cut point L for the
    // XCD: the short ends of all `rem` blocks, end to end, are cut into cu - rem equal runs and dealt to the engines -- a run may hold the short end
    // of a block whose long part runs on another engine (the engines hold 13 or 14 blocks' worth of work otherwise, and the fullest one sets the time)
    int remk[NSE];
    const int n2t = cu - rem;
    const int L = rem > 0 ? (int)(((int64_t)rem * (ntiles + SIG_SHORT) - (int64_t)SIG_LONG * n2t) / cu) : 0, s = ntiles - L;
    bool ok = rem >= NSE && R >= 1 && rem * 4 <= cu * 3 && L >= 2 * MIN_SEG && s >= 2 * MIN_SEG;      // every engine gets a long part; a round in front to hide the short ends in; a last round at most 3/4 full
    for (int k = 0; k < NSE && ok; ++k) {
      remk[k] = rem / NSE + (k >= NSE - rem % NSE ? 1 : 0);
      ok = cs - remk[k] >= 1;
    }
    if (!ok) {
      for (int i = 0; i < nx; ++i) per[x].push_back(item1(blocks[i]));
      continue;
    }
    any = true;
    const int first_full = (R - 1) * cu, first_split = R * cu;
    for (int i = 0; i < first_full; ++i) per[x].push_back(item1(blocks[i]));
    const a3::Seg* X = &blocks[first_split];                   // the blocks to cut
    const int64_t Ts = (int64_t)rem * s;
    std::vector<int64_t> bd(n2t + 1);
    for (int j = 0; j <= n2t; ++j) {
      int64_t v = j * Ts / n2t;
      const int r = (int)(v % s);
      if (r > 0 && r < MIN_SEG) v -= r;
      else if (r > s - MIN_SEG) v += s - r;
      bd[j] = v;
    }
    std::vector<int> in0(rem, 0), nin(rem, 0);
    std::vector<a3::Item> runs;
    for (int j = 0; j < n2t; ++j) {
      a3::Item it{};
      int64_t pos = bd[j];
      while (pos < bd[j + 1]) {
        const int bi = (int)(pos / s), off = (int)(pos % s);
        const int n = (int)std::min<int64_t>(bd[j + 1] - pos, s - off);
        if (it.nseg >= a3::MAX_SEG) return false;
        a3::Seg sg = X[bi];
        sg.t0 = L + off; sg.n = n; sg.out = nparts++;
        if (nin[bi]++ == 0) in0[bi] = sg.out;
        if (nin[bi] > 4) return false;                         // (the hand-over init merges at most 4 partials)
        it.seg[it.nseg++] = sg;
        pos += n;
      }
      if (it.nseg == 0) return false;                          // (an empty work-group would still wait for a CU of its engine)
      runs.push_back(it);
    }
    std::vector<a3::Item> sub[NSE];
    int run0 = 0, long0 = 0;
    for (int k = 0; k < NSE; ++k) {
      const int rk = remk[k], n2 = cs - rk;
      const a3::Seg* F = &blocks[first_full + k * cs];         // this engine's whole blocks
      for (int i = 0; i < rk; ++i) sub[k].push_back(item1(F[i]));
      for (int j = 0; j < n2; ++j) sub[k].push_back(runs[run0 + j]);
      run0 += n2;
      for (int i = rk; i < cs; ++i) sub[k].push_back(item1(F[i]));
      for (int i = 0; i < rk; ++i) {
        a3::Seg sg = X[long0 + i];
        sg.n = L; sg.nin = nin[long0 + i]; sg.in0 = in0[long0 + i];
        if (sg.nin < 1) return false;
        sub[k].push_back(item1(sg));
      }
      long0 += rk;
      if ((int)sub[k].size() != 2 * cs) return false;
    }
    for (int i = 0; i < 2 * cs; ++i)
      for (int k = 0; k < NSE; ++k) per[x].push_back(sub[k][i]);
  }
  if (!any) return false;
  size_t slots = 0;
  for (int x = 0; x < 8; ++x) slots = std::max(slots, per[x].size());
  grid = (int)slots * 8;
  items.assign(grid, a3::Item{});
  for (int x = 0; x < 8; ++x)
    for (size_t i = 0; i < per[x].size(); ++i) items[i * 8 + x] = per[x][i];
  return true;
}

// One plan per (device, B, H, S, stream), at most PLAN_CACHE_MAX of them (prompt lengths vary batch to batch in training, S = T + N, and every plan
// owns its partial slots -- up to ~25 MB): the least recently used plan is freed when a new shape arrives (after its stream has drained: a launch of it may still
// be running).  A new plan is uploaded and cleared on the LAUNCH stream (no device-wide synchronisation); its host table lives as long as the plan, so the
// asynchronous copy always has its source.  The generation number is advanced under the cache mutex and handed out by value.
constexpr size_t PLAN_CACHE_MAX = 16;
struct PlanEntry {
  SplitPlan pl;
  std::vector<a3::Item> host_items;
  char* base = nullptr;
  uint64_t last_use = 0;
};
bool plan_for(int B, int H, int S, hipStream_t stream, SplitPlan& out) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, int, hipStream_t>, PlanEntry> cache;
  static uint64_t tick = 0;
  static int enabled = -1;
  std::lock_guard<std::mutex> lock(mu);
  if (enabled < 0) {
    const char* e = getenv("AFX_ATTN_SPLIT");            // 0: the plain grid (A/B runs)
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  if (!enabled) return false;
  // the generation number is a kernel argument: a captured launch would replay a stale one (and nothing may be allocated inside a capture)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const auto key = std::make_tuple(dev, B, H, S, stream);
  auto it = cache.find(key);
  if (it == cache.end()) {
    if (cache.size() >= PLAN_CACHE_MAX) {
      auto victim = cache.begin();
      for (auto j = cache.begin(); j != cache.end(); ++j)
        if (j->second.last_use < victim->second.last_use) victim = j;
      if (victim->second.base != nullptr) {
        int cur = dev;
        const int vdev = std::get<0>(victim->first);
        if (vdev != cur) (void)hipSetDevice(vdev);
        (void)hipStreamSynchronize(std::get<4>(victim->first));      // (a destroyed stream: the error is dropped, hipFree below synchronises by itself)
        (void)hipFree(victim->second.base);
        (void)hipGetLastError();
        if (vdev != cur) (void)hipSetDevice(cur);
      }
      cache.erase(victim);
    }
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    PlanEntry& en = cache[key];
    SplitPlan& pl = en.pl;
    if (build_plan(B, H, S, ncu, en.host_items, pl.nparts, pl.grid)) {
      const size_t ib = en.host_items.size() * sizeof(a3::Item);
      const size_t pb = (size_t)pl.nparts * a3::QBLK * 128 * 2, lb = (size_t)pl.nparts * a3::QBLK * 4, fb = (size_t)pl.nparts * 4;
      char* base = nullptr;
      const size_t o1 = (ib + 255) / 256 * 256, o2 = o1 + (pb + 255) / 256 * 256, o3 = o2 + (lb + 255) / 256 * 256;
      if (hipMalloc((void**)&base, o3 + fb) == hipSuccess && hipMemcpyAsync(base, en.host_items.data(), ib, hipMemcpyHostToDevice, stream) == hipSuccess &&
          hipMemsetAsync(base + o1, getenv("AFX_ATTN_FILL") ? 0xff : 0, o3 - o1, stream) == hipSuccess && hipMemsetAsync(base + o3, 0, fb, stream) == hipSuccess) {
        en.base = base;
        pl.items = reinterpret_cast<a3::Item*>(base);
        pl.po = reinterpret_cast<uint16_t*>(base + o1);
        pl.plse = reinterpret_cast<float*>(base + o2);
        pl.pflag = reinterpret_cast<int*>(base + o3);
        pl.split = true;
      } else {
        (void)hipGetLastError();
        if (base != nullptr) (void)hipFree(base);
      }
    }
    if (!pl.split) { en.host_items.clear(); en.host_items.shrink_to_fit(); }
    it = cache.find(key);
  }
  it->second.last_use = ++tick;
  if (!it->second.pl.split) return false;
  ++it->second.pl.gen;
  out = it->second.pl;
  return true;
}
}  // namespace

// debug hook (tools/attn_timeline.py): record start / prologue issued / main loop entered (last segment) / end of every work-group of the NEXT launches (enable = 1), read them back: 4 x grid
namespace {
unsigned long long* g_tl = nullptr;
int g_tl_cap = 0, g_tl_grid = 0;
}
extern "C" int afx_debug_attn_timeline(int enable) {
  if (!enable) { g_tl_cap = -1; return 0; }
  if (g_tl == nullptr && hipMalloc((void**)&g_tl, 4 * 8192 * sizeof(unsigned long long)) != hipSuccess) return -1;
  g_tl_cap = 8192;
  return 0;
}
extern "C" int afx_debug_attn_timeline_read(unsigned long long* host, int max_wg) {
  if (g_tl == nullptr || g_tl_grid > max_wg) return -1;
  if (hipMemcpy(host, g_tl, 4 * (size_t)g_tl_grid * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return g_tl_grid;
}

// test hook: the schedule of a shape as plain integers (per item: nseg, then 8 per segment x MAX_SEG)
extern "C" int afx_debug_attn_plan(int B, int H, int S, int ncu, int* items_out, int max_items, int* nparts, int* grid) {
  std::vector<a3::Item> items;
  int np = 0, g = 0;
  if (!build_plan(B, H, S, ncu, items, np, g)) return 0;
  *nparts = np; *grid = g;
  if ((int)items.size() > max_items) return -1;
  constexpr int W = 1 + 8 * a3::MAX_SEG;
  for (size_t i = 0; i < items.size(); ++i) {
    int* d = items_out + i * W;
    d[0] = items[i].nseg;
    for (int s_ = 0; s_ < a3::MAX_SEG; ++s_) {
      const a3::Seg& sg = items[i].seg[s_];
      int* e = d + 1 + 8 * s_;
      e[0] = sg.h; e[1] = sg.b; e[2] = sg.qb; e[3] = sg.t0; e[4] = sg.n; e[5] = sg.out; e[6] = sg.nin; e[7] = sg.in0;
    }
  }
  return (int)items.size();
}

hipError_t launch_attention_v3(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* vt, uint16_t* o,
                               int64_t ldo, int B, int H, int S, hipStream_t stream, float* lse, const AttnMx8* mx8, bool split) {
  uint8_t* o8 = mx8 ? mx8->o8 : nullptr;
  uint8_t* omx = mx8 ? mx8->mx : nullptr;
  const int64_t ldo8 = mx8 ? mx8->ldo8 : 0, ld_omx = mx8 ? mx8->ld_mx : 0;
  static bool attr = false;
  if (!attr) {
    hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(a3::attention_v3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       a3::LDS_BYTES);
    if (r != hipSuccess) return r;
    attr = true;
  }
  const int nqb = (S + a3::QBLK - 1) / a3::QBLK;
  const int S_pad = (int)attn_spad(S);
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = getenv("AFX_ATTN3_DBG");
    dbg = e ? atoi(e) : 0;
  }
  SplitPlan plan;
  const SplitPlan* pl = (split && o8 == nullptr && dbg == 0 && plan_for(B, H, S, stream, plan)) ? &plan : nullptr;      // (the block-scaled fp8 output keeps the plain grid)
  const dim3 grid(pl ? pl->grid : 8 * ((H + 7) / 8) * nqb * B);
  const a3::Item* items = pl ? pl->items : nullptr;
  uint16_t* po = pl ? pl->po : nullptr;
  float* plse = pl ? pl->plse : nullptr;
  int* pflag = pl ? pl->pflag : nullptr;
  const int gen = pl ? pl->gen : 0;
  unsigned long long* tl = nullptr;
  if (g_tl_cap > 0 && (int)grid.x <= g_tl_cap) { tl = g_tl; g_tl_grid = (int)grid.x; }
  // AFX_ATTN_HANDOVER: "fence" = plain stores + release fence instead of write-through stores (A/B); "lost" = the long parts never see a partial published and
  // compute their whole key range themselves (the fallback for a dispatch order the schedule does not expect; read per launch: the tests flip it)
  const char* hv = getenv("AFX_ATTN_HANDOVER");
  const int wt = (hv && hv[0] == 'f') ? 0 : (hv && hv[0] == 'l') ? 3 : 1;
  if (launch_timer().start != nullptr && launch_timer().stop != nullptr)
    hipExtLaunchKernelGGL(a3::attention_v3_kernel, grid, dim3(a3::THREADS), a3::LDS_BYTES, stream, launch_timer().start, launch_timer().stop, 0,
                          q, ldq, k, ldk, vt, o, ldo, H, S, S_pad, nqb, B, lse, dbg, o8, ldo8, omx, ld_omx, items, po, plse, pflag, gen, wt, tl);
  else
    hipLaunchKernelGGL(a3::attention_v3_kernel, grid, dim3(a3::THREADS), a3::LDS_BYTES, stream, q, ldq, k, ldk, vt, o, ldo, H, S, S_pad, nqb, B, lse, dbg, o8, ldo8, omx, ld_omx,
                       items, po, plse, pflag, gen, wt, tl);
  return hipGetLastError();
}

}  // namespace afx
