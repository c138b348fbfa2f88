// C-ABI layer + MMDiT forward engine (see include/arcflow_hip.h for the contract).
//
// The forward is a fixed launch plan over the kernels in afx_gemm / afx_attn / afx_elementwise:
//   temb MLPs (gemv) -> ONE gemv over all stacked AdaLN modulation linears -> embedders (grouped GEMM)
//   -> per block { LN+modulate, QKV GEMM (rows k|v|q), RMSNorm+RoPE in place, V transpose,
//                  flash attention (O overwrites Q), out-proj GEMM with fused gate*x+residual,
//                  LN+modulate, MLP-up GEMM + GELU, MLP-down GEMM with fused gate*x+residual }
//   -> norm_out + head GEMM -> log_softmax split.
// Activations live in the caller-provided workspace in the joint [B][text;image] token layout, so the
// FLUX single-stream blocks run on the same buffers without a concat, and attention / proj_out read
// the [O | mlp] operand in place (lda-strided) from the fused QKV+MLP buffer.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/arcflow_hip.h"
#include "afx_kernels.h"

using namespace afx;

#include "afx_api_util.h"

thread_local char afx_g_err[512] = "";

int afx_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(afx_g_err, sizeof(afx_g_err), fmt, ap);
  va_end(ap);
  return code;
}

namespace {

struct Weight {
  const void* ptr = nullptr;
  int dtype = 0;
  std::vector<int64_t> shape;
};

inline int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

}  // namespace

// Weight pointers of one linear / one block, resolved once in afx_finalize (the launch plan does no string work)
struct LinW {
  const uint16_t* w = nullptr;    // bf16 [out, in]
  const uint16_t* b = nullptr;    // bf16 [out]
  const void* wq = nullptr;       // fp8 mode: e4m3 [out, in]
  const float* wscale = nullptr;  //           per-output-channel scale [out]
};
struct DoubleW { LinW qkv[2], out[2], mlp1[2], mlp2[2]; const float* qkn = nullptr; };   // [0] image stream, [1] text stream
struct SingleW { LinW fused, out; const float* qkn = nullptr; };

struct afx_ctx {
  afx_model_desc d;
  std::unordered_map<std::string, Weight> w;
  bool finalized = false;
  std::vector<DoubleW> dbl;
  std::vector<SingleW> sgl;
  // side stream for the weight-streaming modulation GEMV (HBM-bound, 6.5 GB for FLUX): it runs under the first blocks' MFMA work
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int mod_overlap = -1;          // -1: read AFX_MOD_OVERLAP on first use (default off)
  char* ws = nullptr;
  int64_t ws_bytes = 0;
  int D = 0;
  int64_t n_mod = 0;       // rows of the stacked modulation linear
  int head_n = 0;          // padded head width
  uint16_t* ckpt = nullptr;   // optional [num_blocks][B*S, D] block-input checkpoints (gradient checkpointing)
  bool fp8 = false;            // block linears on the fp8 MFMA (afx_set_fp8_linear)
  int fp8_mx = -1;             // fp8: block-scaled activations, quantisation in the producers' epilogues (AFX_FP8_MX=0: one scale per row + a pass per GEMM)
  bool sk_flags_dirty = false; // a new workspace was bound: its stream-K flag words are cleared on the next forward's stream
  const float* temb_override = nullptr;   // optional [B, D] f32 replacing timestep_embedder(t) (training student with its LoRA pair)
  // conditioning of several denoising steps prepared in one pass over the stacked modulation matrix (afx_mmdit_prepare_steps)
  int prep_steps = 0, prep_B = 0, prep_use = -1;
  // optional per-launch-class timing (HIP events on the forward's stream)
  bool prof_on = false;
  int prof_stride = 1;          // time one launch in prof_stride (afx_profile_enable(ctx, N)): an event pair on a dispatch costs ~4 us
  uint64_t prof_count = 0;
  struct ProfRec { hipEvent_t a, b; int klass; double flops; };
  std::vector<ProfRec> prof_pool;
  size_t prof_used = 0;
};

namespace {

const uint16_t* W16(const afx_ctx* c, const std::string& n) {
  auto it = c->w.find(n);
  return it == c->w.end() ? nullptr : (const uint16_t*)it->second.ptr;
}
const float* W32(const afx_ctx* c, const std::string& n) {
  auto it = c->w.find(n);
  return it == c->w.end() ? nullptr : (const float*)it->second.ptr;
}

const void* WQ(const afx_ctx* c, const std::string& n) {
  auto it = c->w.find(n);
  return it == c->w.end() ? nullptr : it->second.ptr;
}

int need(const afx_ctx* c, const std::string& n, int dtype, std::vector<int64_t> shape) {
  auto it = c->w.find(n);
  if (it == c->w.end()) return fail(AFX_E_MISSING, "weight '%s' is not bound", n.c_str());
  if (it->second.dtype != dtype) return fail(AFX_E_INVALID, "weight '%s': wrong dtype", n.c_str());
  if (it->second.shape != shape) {
    std::string got, want;
    for (auto v : it->second.shape) got += std::to_string(v) + ",";
    for (auto v : shape) want += std::to_string(v) + ",";
    return fail(AFX_E_INVALID, "weight '%s': shape [%s] expected [%s]", n.c_str(), got.c_str(), want.c_str());
  }
  return AFX_OK;
}

int need_linear(const afx_ctx* c, const std::string& n, int64_t out_f, int64_t in_f) {
  int r = need(c, n + ".weight", AFX_DT_BF16, {out_f, in_f});
  if (r) return r;
  return need(c, n + ".bias", AFX_DT_BF16, {out_f});
}

// modulation vector offsets inside one row of the stacked modulation output
struct ModLayout {
  int64_t D;
  int nd, ns;
  int64_t dbl(int i, int stream /*0 img, 1 txt*/, int chunk /*0..5*/) const {
    return ((int64_t)i * 12 + stream * 6 + chunk) * D;
  }
  int64_t sgl(int i, int chunk /*0..2*/) const { return (int64_t)nd * 12 * D + ((int64_t)i * 3 + chunk) * D; }
  int64_t fin(int chunk /*0 scale, 1 shift*/) const { return (int64_t)nd * 12 * D + (int64_t)ns * 3 * D + chunk * D; }
  int64_t total() const { return (int64_t)nd * 12 * D + (int64_t)ns * 3 * D + 2 * D; }
};

constexpr int AFX_PREP_ROWS = 8;      // (steps x samples) one modulation pass can serve: the GEMV's batch limit

struct Workspace {
  uint32_t* sk_flags;   // stream-K GEMM tail (afx_gemm.hip): 1024 flag words at offset 0 (zeroed by afx_set_workspace, re-armed by the kernel)
  float* sk_slab;       //   + 256 fp32 accumulator slabs of one 256x256 tile each
  uint16_t *X, *Xn, *F, *Vt, *head;
  float *sincos, *tmp, *temb, *semb, *mod, *pooled;
  float *prep_temb, *prep_semb, *prep_mod;   // [AFX_PREP_ROWS][D], [..][D], [..][n_mod]: prepared steps (step-major, then sample)
  uint8_t* q8;     // fp8 mode: the quantised A operand of the GEMM about to run [R, <= 5D]
  uint8_t* q8n;    // fp8, block-scaled: the D-wide operands (LayerNorm outputs, the double blocks' attention output) [R, D] -- q8 then holds the
                   // wide ones, written by the producing GEMM's epilogue while that GEMM still reads q8n
  uint8_t* mxn;    // [R, ld_mxn] / [R, ld_mxw] E8M0 scale bytes of q8n / q8
  uint8_t* mxw;
  float* ones;     // [R] 1.0f: a_scale of the block-scaled launches
  int64_t ld_mxn, ld_mxw;
  float* qs;       //           its per-row scales [R]
  int64_t total;
};

Workspace carve(const afx_ctx* c, char* base, int B, int N, int T) {
  const int64_t D = c->D, S = (int64_t)N + T, R = (int64_t)B * S;
  Workspace w;
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align256(bytes);
    return p;
  };
  w.sk_flags = (uint32_t*)take(GEMM_SK_FLAG_BYTES);          // FIRST: its offset must not depend on the shape
  w.sk_slab = (float*)take(GEMM_SK_SLAB_BYTES);
  w.prep_temb = (float*)take((int64_t)AFX_PREP_ROWS * D * 4);       // (shape-independent offsets: they outlive a forward)
  w.prep_semb = (float*)take((int64_t)AFX_PREP_ROWS * D * 4);
  w.prep_mod = (float*)take((int64_t)AFX_PREP_ROWS * c->n_mod * 4);
  w.X = (uint16_t*)take(R * D * 2);
  w.Xn = (uint16_t*)take(R * D * 2);
  w.F = (uint16_t*)take(R * 7 * D * 2);   // single: fused [k|v|q|mlp]; double: [QKV 3D] then [H 4D]
  w.Vt = (uint16_t*)take((int64_t)B * c->d.heads * 128 * attn_spad((int)S) * 2);
  w.head = (uint16_t*)take((int64_t)B * N * c->head_n * 2);
  w.sincos = (float*)take((int64_t)B * 256 * 4);
  w.tmp = (float*)take((int64_t)B * D * 4);
  w.temb = (float*)take((int64_t)B * D * 4);
  w.semb = (float*)take((int64_t)B * D * 4);
  w.pooled = (float*)take((int64_t)B * (c->d.pooled_dim > 0 ? c->d.pooled_dim : 8) * 4);
  w.mod = (float*)take((int64_t)B * c->n_mod * 4);
  w.q8 = nullptr;
  w.qs = nullptr;
  w.q8n = w.mxn = w.mxw = nullptr;
  w.ones = nullptr;
  w.ld_mxn = (D / 128 + 3) / 4 * 4;
  w.ld_mxw = (5 * D / 128 + 3) / 4 * 4;
  if (c->fp8) {
    w.q8 = (uint8_t*)take(R * 5 * D);
    w.qs = (float*)take(R * 4);
    w.q8n = (uint8_t*)take(R * D);
    w.mxn = (uint8_t*)take(R * w.ld_mxn);
    w.mxw = (uint8_t*)take(R * w.ld_mxw);
    w.ones = (float*)take(R * 4);
  }
  w.total = off;
  return w;
}

// Scoped timer of ONE launch when profiling is enabled: hands an event pair to the launcher, which attaches it to the kernel
// dispatch itself (hipExtLaunchKernelGGL: kernel begin / end timestamps, no extra queue packets).
struct ProfScope {
  afx_ctx* c; hipStream_t st; afx_ctx::ProfRec* r = nullptr;
  ProfScope(afx_ctx* c_, hipStream_t st_, int klass, double flops) : c(c_), st(st_) {
    if (!c->prof_on) return;
    if ((c->prof_count++ % (uint64_t)c->prof_stride) != 0) return;      // sampled: the forward has an odd number of launches, so the sample walks every launch position
    if (c->prof_used == c->prof_pool.size()) {
      afx_ctx::ProfRec n{};
      if (hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) return;
      c->prof_pool.push_back(n);
    }
    r = &c->prof_pool[c->prof_used++];
    r->klass = klass; r->flops = flops;
    launch_timer().start = r->a;
    launch_timer().stop = r->b;
  }
  ~ProfScope() { launch_timer() = LaunchTimer{}; }
};

double gemm_flops(const GemmBatch& gb) {
  double f = 0;
  for (int i = 0; i < gb.nprob; ++i) f += 2.0 * gb.p[i].M * (double)gb.p[i].N * gb.p[i].K;
  return f;
}

}  // namespace

extern "C" {

const char* afx_last_error(void) { return afx_g_err; }
const char* afx_version(void) { return "arcflow_hip 0.1 (gfx950)"; }

int afx_create(const afx_model_desc* desc, afx_ctx** out) {
  if (!desc || !out) return fail(AFX_E_INVALID, "null argument");
  if (desc->head_dim != 128) return fail(AFX_E_UNSUPPORTED, "head_dim must be 128");
  if (desc->heads <= 0 || desc->num_double < 0 || desc->num_single < 0 || desc->num_gaussians < 1 ||
      desc->num_gaussians > 32)
    return fail(AFX_E_INVALID, "bad model description");
  if (desc->in_channels % 64 != 0 || desc->joint_dim % 64 != 0)
    return fail(AFX_E_UNSUPPORTED, "in_channels and joint_dim must be multiples of 64");
  afx_ctx* c = new afx_ctx();
  c->d = *desc;
  c->D = desc->heads * desc->head_dim;
  ModLayout ml{c->D, desc->num_double, desc->num_single};
  c->n_mod = ml.total();
  const int K = desc->num_gaussians, ch = desc->in_channels, lw = desc->logweights_channels;
  const int raw = desc->head_mode == 0 ? K * ch + K * lw + (K - 1) * lw : ch;
  c->head_n = (raw + 7) / 8 * 8;
  *out = c;
  return AFX_OK;
}

int afx_destroy(afx_ctx* ctx) {
  if (ctx) {
    for (auto& r : ctx->prof_pool) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
  }
  delete ctx;
  return AFX_OK;
}

int afx_bind_weight(afx_ctx* ctx, const char* name, const void* dptr, int32_t dtype, int32_t ndim,
                    const int64_t* shape) {
  if (!ctx || !name || !dptr || ndim < 1 || ndim > 4 || !shape) return fail(AFX_E_INVALID, "bad bind_weight argument");
  if (dtype != AFX_DT_BF16 && dtype != AFX_DT_F32 && dtype != AFX_DT_FP8) return fail(AFX_E_INVALID, "bad dtype for '%s'", name);
  Weight w;
  w.ptr = dptr;
  w.dtype = dtype;
  w.shape.assign(shape, shape + ndim);
  ctx->w[name] = w;
  ctx->finalized = false;
  return AFX_OK;
}

int afx_finalize(afx_ctx* c) {
  if (!c) return fail(AFX_E_INVALID, "null ctx");
  const int64_t D = c->D;
  const afx_model_desc& d = c->d;
  int r;
#define NEED(x) \
  if ((r = (x)) != AFX_OK) return r
  NEED(need_linear(c, "x_in", D, d.in_channels));
  NEED(need_linear(c, "ctx_in", D, d.joint_dim));
  if (d.family == 1) NEED(need(c, "txt_norm.weight", AFX_DT_F32, {d.joint_dim}));
  NEED(need_linear(c, "temb.t.l1", D, 256));
  NEED(need_linear(c, "temb.t.l2", D, D));
  if (d.guidance_embeds) {
    NEED(need_linear(c, "temb.g.l1", D, 256));
    NEED(need_linear(c, "temb.g.l2", D, D));
  }
  if (d.pooled_dim > 0) {
    NEED(need_linear(c, "temb.p.l1", D, d.pooled_dim));
    NEED(need_linear(c, "temb.p.l2", D, D));
  }
  NEED(need_linear(c, "mod", c->n_mod, D));
  for (int i = 0; i < d.num_double; ++i) {
    const std::string p = "d" + std::to_string(i) + ".";
    for (const char* s : {"img", "txt"}) {
      NEED(need_linear(c, p + s + "_qkv", 3 * D, D));
      NEED(need_linear(c, p + s + "_out", D, D));
      NEED(need_linear(c, p + s + "_mlp1", 4 * D, D));
      NEED(need_linear(c, p + s + "_mlp2", D, 4 * D));
    }
    NEED(need(c, p + "qknorm", AFX_DT_F32, {4, 128}));
  }
  for (int i = 0; i < d.num_single; ++i) {
    const std::string p = "s" + std::to_string(i) + ".";
    NEED(need_linear(c, p + "fused", 7 * D, D));
    NEED(need_linear(c, p + "out", D, 5 * D));
    NEED(need(c, p + "qknorm", AFX_DT_F32, {2, 128}));
  }
  NEED(need_linear(c, "head", c->head_n, D));
  if (c->w.count("mod_final.weight")) NEED(need_linear(c, "mod_final", 2 * D, D));
#undef NEED
  auto lin = [&](const std::string& n) {
    LinW l;
    l.w = W16(c, n + ".weight"); l.b = W16(c, n + ".bias");
    l.wq = WQ(c, n + ".weight_q"); l.wscale = W32(c, n + ".wscale");
    return l;
  };
  c->dbl.assign(d.num_double, DoubleW{});
  for (int i = 0; i < d.num_double; ++i) {
    const std::string p = "d" + std::to_string(i) + ".";
    for (int s = 0; s < 2; ++s) {
      const std::string q = p + (s == 0 ? "img_" : "txt_");
      c->dbl[i].qkv[s] = lin(q + "qkv"); c->dbl[i].out[s] = lin(q + "out");
      c->dbl[i].mlp1[s] = lin(q + "mlp1"); c->dbl[i].mlp2[s] = lin(q + "mlp2");
    }
    c->dbl[i].qkn = W32(c, p + "qknorm");
  }
  c->sgl.assign(d.num_single, SingleW{});
  for (int i = 0; i < d.num_single; ++i) {
    const std::string p = "s" + std::to_string(i) + ".";
    c->sgl[i].fused = lin(p + "fused"); c->sgl[i].out = lin(p + "out");
    c->sgl[i].qkn = W32(c, p + "qknorm");
  }
  c->finalized = true;
  return AFX_OK;
}

int64_t afx_workspace_bytes(const afx_ctx* ctx, int32_t batch, int32_t n_img, int32_t n_txt) {
  if (!ctx || batch < 1 || n_img < 1 || n_txt < 0) return fail(AFX_E_INVALID, "bad workspace query");
  return carve(ctx, nullptr, batch, n_img, n_txt).total;
}

int afx_set_workspace(afx_ctx* ctx, void* dptr, int64_t bytes) {
  if (!ctx || !dptr || bytes <= 0) return fail(AFX_E_INVALID, "bad workspace");
  if (((uintptr_t)dptr & 255) != 0) return fail(AFX_E_INVALID, "workspace must be 256-byte aligned");
  if (bytes < GEMM_SK_FLAG_BYTES) return fail(AFX_E_WORKSPACE, "workspace too small");
  // The stream-K hand-off flags at the head of the workspace must start cleared (the kernel re-arms them).  The clear is issued on
  // the stream of the NEXT forward (sk_flags_dirty), not here on the NULL stream: torch side streams are non-blocking, so a memset
  // here would be unordered against a kernel still using a recycled allocator block and against the first forward.
  ctx->sk_flags_dirty = true;
  ctx->ws = (char*)dptr;
  ctx->ws_bytes = bytes;
  ctx->prep_steps = 0;                                       // prepared steps lived in the old workspace
  ctx->prep_use = -1;
  return AFX_OK;
}

int afx_mmdit_forward(afx_ctx* c, const void* x, const void* ctx_emb, const void* pooled, const float* t,
                      const float* g, const float* rope_cos, const float* rope_sin, int32_t B, int32_t N,
                      int32_t T, void* means, void* logw, void* logg, void* stream_) {
  return afx_mmdit_forward_stage(c, x, ctx_emb, pooled, t, g, rope_cos, rope_sin, B, N, T, means, logw, logg, 0, stream_);
}

int afx_mmdit_forward_stage(afx_ctx* c, const void* x, const void* ctx_emb, const void* pooled, const float* t,
                            const float* g, const float* rope_cos, const float* rope_sin, int32_t B, int32_t N,
                            int32_t T, void* means, void* logw, void* logg, int32_t stage, void* stream_) {
  if (stage < 0 || stage > 2) return fail(AFX_E_INVALID, "stage must be 0 (all), 1 (conditioning + embedders) or 2 (norm_out + head)");
  if (!c || !x || !ctx_emb || !t || !rope_cos || !rope_sin || !means)
    return fail(AFX_E_INVALID, "null argument to afx_mmdit_forward");
  if (!c->finalized) return fail(AFX_E_MISSING, "afx_finalize() has not succeeded on this context");
  const afx_model_desc& d = c->d;
  if (B < 1 || N < 1 || T < 1) return fail(AFX_E_INVALID, "bad shape: batch, N, T >= 1");
  if (B > AFX_MAX_MICRO_BATCH) {
    // A grouped launch holds 2 problems per sample (GEMM_MAX_PROBLEMS = 8) and the workspace is carved for 4 samples: larger
    // batches run as micro-batches of 4 on the same stream (samples are independent: the reference's batch dimension,
    // arcflux.py:134-257, has no cross-sample op).  The staged (training) calls keep one micro-batch per call.
    if (stage != 0 || c->ckpt) return fail(AFX_E_INVALID, "staged / checkpointed forward: batch must be 1..4 per call");
    const int64_t K = d.num_gaussians, Cc = d.in_channels, lw = d.logweights_channels;
    for (int b0 = 0; b0 < B; b0 += AFX_MAX_MICRO_BATCH) {
      const int nb = B - b0 < AFX_MAX_MICRO_BATCH ? B - b0 : AFX_MAX_MICRO_BATCH;
      const uint16_t* xb = (const uint16_t*)x + (int64_t)b0 * N * Cc;
      const uint16_t* cb = (const uint16_t*)ctx_emb + (int64_t)b0 * T * d.joint_dim;
      const uint16_t* pb = pooled ? (const uint16_t*)pooled + (int64_t)b0 * d.pooled_dim : nullptr;
      uint16_t* mb = (uint16_t*)means + (int64_t)b0 * N * (d.head_mode == 0 ? K * Cc : Cc);
      uint16_t* wb = logw ? (uint16_t*)logw + (int64_t)b0 * N * K * lw : nullptr;
      uint16_t* gb_ = logg ? (uint16_t*)logg + (int64_t)b0 * N * (K - 1) * lw : nullptr;
      const float* temb_all = c->temb_override;
      if (temb_all) c->temb_override = temb_all + (int64_t)b0 * c->D;
      const int rc = afx_mmdit_forward_stage(c, xb, cb, pb, t + b0, g ? g + b0 : nullptr, rope_cos, rope_sin, nb, N, T, mb, wb, gb_, 0, stream_);
      c->temb_override = temb_all;
      if (rc != AFX_OK) return rc;
    }
    return AFX_OK;
  }
  if (d.guidance_embeds && !g) return fail(AFX_E_INVALID, "guidance vector required");
  if (d.pooled_dim > 0 && !pooled) return fail(AFX_E_INVALID, "pooled projections required");
  if (d.head_mode == 0 && (!logw || !logg)) return fail(AFX_E_INVALID, "logw/logg outputs required");
  Workspace ws = carve(c, c->ws, B, N, T);
  if (!c->ws || ws.total > c->ws_bytes)
    return fail(AFX_E_WORKSPACE, "workspace too small: need %lld bytes, have %lld", (long long)ws.total,
                (long long)c->ws_bytes);
  hipStream_t st = (hipStream_t)stream_;
  if (c->sk_flags_dirty) {
    HIP_TRY(hipMemsetAsync(c->ws, 0, GEMM_SK_FLAG_BYTES, st));
    c->sk_flags_dirty = false;
  }
  const int64_t D = c->D;
  const int H = d.heads, S = N + T;
  const int64_t R = (int64_t)B * S;
  ModLayout ml{D, d.num_double, d.num_single};
  const int64_t ldm = c->n_mod;
  int overlap_join_block = -1;          // >= 0: the side-stream modulation GEMV must be joined in front of this block

  const bool use_prep = c->prep_use >= 0 && c->prep_use < c->prep_steps && c->prep_B == B && stage != 2 && c->temb_override == nullptr;
  const int prep_k = c->prep_use;
  if (stage != 2) c->prep_use = -1;                              // one-shot
  if (use_prep) {
    // the modulation vectors (and temb, for afx_mmdit_export) of this step were computed by afx_mmdit_prepare_steps
    const int64_t r0 = (int64_t)prep_k * B;
    HIP_TRY(hipMemcpyAsync(ws.mod, ws.prep_mod + r0 * ldm, (size_t)B * ldm * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(ws.temb, ws.prep_temb + r0 * D, (size_t)B * D * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(ws.semb, ws.prep_semb + r0 * D, (size_t)B * D * 4, hipMemcpyDeviceToDevice, st));
  }
  if (stage != 2 && !use_prep) {
  // ---- conditioning: temb = t_mlp(sincos(1000 t)) [+ g_mlp(sincos(1000 g))] [+ p_mlp(pooled)] ------
  if (c->temb_override != nullptr) {
    HIP_TRY(hipMemcpyAsync(ws.temb, c->temb_override, (size_t)B * D * 4, hipMemcpyDeviceToDevice, st));
  } else {
    HIP_TRY(launch_sincos(t, 1000.0f, ws.sincos, B, d.family == 0 ? 1 : 2, st));
    HIP_TRY(launch_gemv(ws.sincos, W16(c, "temb.t.l1.weight"), W16(c, "temb.t.l1.bias"), ws.tmp, B, (int)D, 256, 1, 0, st));
    HIP_TRY(launch_gemv(ws.tmp, W16(c, "temb.t.l2.weight"), W16(c, "temb.t.l2.bias"), ws.temb, B, (int)D, (int)D, 0, 0, st));
  }
  if (d.guidance_embeds) {
    HIP_TRY(launch_sincos(g, 1000.0f, ws.sincos, B, 1, st));
    HIP_TRY(launch_gemv(ws.sincos, W16(c, "temb.g.l1.weight"), W16(c, "temb.g.l1.bias"), ws.tmp, B, (int)D, 256, 1, 0, st));
    HIP_TRY(launch_gemv(ws.tmp, W16(c, "temb.g.l2.weight"), W16(c, "temb.g.l2.bias"), ws.temb, B, (int)D, (int)D, 0, 1, st));
  }
  if (d.pooled_dim > 0) {
    HIP_TRY(launch_bf16_to_f32((const uint16_t*)pooled, ws.pooled, (int64_t)B * d.pooled_dim, st));
    HIP_TRY(launch_gemv(ws.pooled, W16(c, "temb.p.l1.weight"), W16(c, "temb.p.l1.bias"), ws.tmp, B, (int)D, d.pooled_dim, 1, 0, st));
    HIP_TRY(launch_gemv(ws.tmp, W16(c, "temb.p.l2.weight"), W16(c, "temb.p.l2.bias"), ws.temb, B, (int)D, (int)D, 0, 1, st));
  }
  HIP_TRY(launch_silu(ws.temb, ws.semb, (int64_t)B * D, st));
  // Every AdaLN modulation vector of the whole network in one weight-streaming pass over the stacked [n_mod, D] matrix
  // (6.5 GB for FLUX: 1.3 ms of pure HBM streaming).  Only the first blocks' rows are needed right away: those run on the
  // forward's stream; with AFX_MOD_OVERLAP=1 the rest streams on a side stream under the embedders' and first blocks' MFMA work
  // and is joined in front of the first block that reads it (fork / join by events).  OFF by default: measured 140.2 vs 139.5 ms
  // per image (r02c) -- the GEMMs lose more to the shared HBM / issue slots than the 1.3 ms the stream hides.
  if (c->mod_overlap < 0) {
    const char* e = getenv("AFX_MOD_OVERLAP");
    c->mod_overlap = (e && e[0] == '1') ? 1 : 0;     // opt-in: measured -0.5 % on FLUX (r02c: the stream steals HBM + issue slots from the GEMMs)
  }
  const int nblocks = d.num_double + d.num_single;
  const int head_blocks = 2;                                  // blocks whose modulation rows stay on the main stream
  int64_t rows_main = c->n_mod;
  overlap_join_block = -1;
  if (c->mod_overlap && stage == 0 && nblocks > head_blocks + 2) {
    rows_main = head_blocks <= d.num_double ? ml.dbl(head_blocks, 0, 0) : ml.sgl(head_blocks - d.num_double, 0);
    if (!c->side) {
      HIP_TRY(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(c->ev_fork, st));                  // semb is ready
    HIP_TRY(hipStreamWaitEvent(c->side, c->ev_fork, 0));
    const uint16_t* mw = W16(c, "mod.weight");
    const uint16_t* mb = W16(c, "mod.bias");
    HIP_TRY(launch_gemv(ws.semb, mw + rows_main * D, mb + rows_main, ws.mod + rows_main, B, (int)(c->n_mod - rows_main), (int)D, 0, 0,
                        c->side, ldm));
    if (W16(c, "mod_final.weight") != nullptr)
      HIP_TRY(launch_gemv(ws.semb, W16(c, "mod_final.weight"), W16(c, "mod_final.bias"), ws.mod + ml.fin(0), B, (int)(2 * D),
                          (int)D, 0, 0, c->side, ldm));
    HIP_TRY(hipEventRecord(c->ev_join, c->side));
    overlap_join_block = head_blocks;
  }
  HIP_TRY(launch_gemv(ws.semb, W16(c, "mod.weight"), W16(c, "mod.bias"), ws.mod, B, (int)rows_main, (int)D, 0, 0, st, ldm));
  // a separately bound norm_out.linear (the distillation student trains its own copy while the teacher keeps the
  // frozen one inside the stacked matrix: lakonlab/configs/flux/arcflux_2nfe_k16.py:20-25 freeze_exclude 'norm_out')
  if (overlap_join_block < 0 && W16(c, "mod_final.weight") != nullptr)
    HIP_TRY(launch_gemv(ws.semb, W16(c, "mod_final.weight"), W16(c, "mod_final.bias"), ws.mod + ml.fin(0), B, (int)(2 * D),
                        (int)D, 0, 0, st, ldm));
  }   // conditioning
  if (stage != 2) {
  // ---- embedders into the joint layout X[b][text T | image N] ---------------------------------------
  const uint16_t* ctx_src = (const uint16_t*)ctx_emb;
  if (d.family == 1) {   // Qwen: RMSNorm(joint_dim) on the text states before txt_in (arcqwen.py:129)
    HIP_TRY(launch_norm_modulate(ctx_src, d.joint_dim, ws.F, d.joint_dim, B * T, d.joint_dim,
                                 W32(c, "txt_norm.weight"), nullptr, 0, B * T, 1, st));
    ctx_src = ws.F;
  }
  {
    GemmBatch gb{};
    for (int b = 0; b < B; ++b) {
      GemmProblem& pi = gb.p[gb.nprob++];
      pi = GemmProblem{};
      pi.A = (const uint16_t*)x + (int64_t)b * N * d.in_channels; pi.lda = d.in_channels;
      pi.W = W16(c, "x_in.weight"); pi.ldw = d.in_channels; pi.bias = W16(c, "x_in.bias");
      pi.C = ws.X + ((int64_t)b * S + T) * D; pi.ldc = D; pi.M = N; pi.N = (int)D; pi.K = d.in_channels;
      GemmProblem& pt = gb.p[gb.nprob++];
      pt = GemmProblem{};
      pt.A = ctx_src + (int64_t)b * T * d.joint_dim; pt.lda = d.joint_dim;
      pt.W = W16(c, "ctx_in.weight"); pt.ldw = d.joint_dim; pt.bias = W16(c, "ctx_in.bias");
      pt.C = ws.X + (int64_t)b * S * D; pt.ldc = D; pt.M = T; pt.N = (int)D; pt.K = d.joint_dim;
    }
    { ProfScope ps_(c, st, 0, gemm_flops(gb)); HIP_TRY(launch_gemm(gb, st)); }
  }
  }   // stage != 2
  if (stage == 1) return AFX_OK;       // the caller runs the blocks itself on the exported token matrix

  auto join_side = [&](int blk) -> int {
    if (overlap_join_block >= 0 && blk >= overlap_join_block) {
      HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0));
      overlap_join_block = -1;
    }
    return AFX_OK;
  };
  // fp8 with block-scaled activations: every block GEMM reads e4m3 rows + one E8M0 byte per row and 128 columns.  The D-wide
  // operands are quantised into q8n / mxn (by the pass below until their producers write them), the wide ones (mlp hidden, [O | mlp]) leave
  // the producing GEMM's epilogue in q8 / mxw.
  if (c->fp8 && c->fp8_mx < 0) {
    const char* e = getenv("AFX_FP8_MX");
    c->fp8_mx = (e && e[0] == '0') ? 0 : 1;
  }
  const bool mx = c->fp8 && c->fp8_mx == 1 && D % 512 == 0 && gemm_fp8_mx_ok(R, (int)D, (int)D);
  if (mx) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)ws.ones, 0x3f800000, (size_t)R, st));
  // The fused q / k epilogue also exists on the one-wave-per-SIMD fp8 kernel (epi_store_qk<MI, true>; V is then transposed by its own launch), opt-in:
  // measured level with the separate preparation launch (11.30 vs 11.30 images/s) -- with 256 accumulators in the file the fp8 variant of that epilogue
  // has to re-read the weight scales per row tile, which costs what the saved launch gave (AFX_FP8_QK_FUSE=1).
  // LayerNorm-produced operands (a row sits in one wave there) carry ONE scale per row and run on the plain fp8 MFMA; only the operands written by
  // GEMM / attention epilogues need block scales (AFX_FP8_NORM_MX=1: block scales everywhere, A/B)
  const bool norm_rows = mx && getenv("AFX_FP8_NORM_MX") == nullptr;
  const bool attn_mx = mx && H * 128 == D && getenv("AFX_FP8_ATTN_MX_OFF") == nullptr;      // the attention epilogue as the last producer of the format
  const char* qkf8 = getenv("AFX_FP8_QK_FUSE");
  const bool qk_fuse_fp8 = mx && qkf8 != nullptr && qkf8[0] == '1';
  // helper: one grouped GEMM over the image and text row ranges of every sample
  // q / k RMSNorm + RoPE inside the epilogue of the k|v|q projections (GemmProblem::qk_D) when the GEMM kernel in use offers it:
  // the separate launch then only transposes V.  Not in fp8 mode (that kernel has no such epilogue).
  const bool qk_fuse = gemm_qk_fusion_available() && (!c->fp8 || qk_fuse_fp8);
  // ... and V^T straight out of the projection (GemmProblem::w_perm16 / bias_rows: the V third computed transposed, A = the V rows
  // of the weight, W = the tokens) when the 16-key groups of the joint sequence do not straddle the text / image boundary and no
  // key padding is needed: then no preparation launch is left between the projection and the attention.
  const int S_pad_ = (int)attn_spad(S);
  const bool vt_fuse = qk_fuse && !c->fp8 && T % 16 == 0 && S % 64 == 0 && getenv("AFX_VT_FUSE_OFF") == nullptr;
  const int64_t vt_sample = (int64_t)H * 128 * S_pad_;           // elements of one sample's V^T
  // the k, q and transposed-v problems of one k|v|q(|mlp) projection over `rows` joint rows starting at joint row `row0` of sample b
  auto kqv_problems = [&](GemmBatch& gb, const uint16_t* A, int64_t lda, const LinW& lw, uint16_t* C, int64_t ldc, int64_t row0g,
                          int rows, int b, int pos0, int period, const float* wq, const float* wk, bool with_v) {
    for (int part = 0; part < (with_v ? 3 : 2); ++part) {        // 0: k (weight rows 0..D), 1: q (2D..3D), 2: v^T (D..2D)
      GemmProblem& p = gb.p[gb.nprob++];
      p = GemmProblem{};
      p.K = (int)D; p.epi = EPI_NONE;
      if (part < 2) {
        const int64_t wrow = part == 0 ? 0 : 2 * D;
        p.A = A + row0g * lda; p.lda = lda;
        p.W = lw.w + wrow * D; p.ldw = D; p.bias = lw.b ? lw.b + wrow : nullptr;
        p.C = C + row0g * ldc + wrow; p.ldc = ldc; p.M = rows; p.N = (int)D;
        p.qk_D = (int)D; p.qk_wk = part == 0 ? wk : wq; p.qk_wq = p.qk_wk;       // a problem of its own: its columns are "region 0"
        p.rope_cos = rope_cos; p.rope_sin = rope_sin; p.rope_row0 = pos0; p.rope_period = period; p.rope_rows = (int)S;
      } else {
        p.A = lw.w + D * D; p.lda = D; p.bias = lw.b ? lw.b + D : nullptr; p.bias_rows = 1;
        p.W = A + row0g * lda; p.ldw = lda; p.w_perm16 = 1;
        p.C = ws.Vt + (int64_t)b * vt_sample + pos0; p.ldc = S_pad_; p.M = (int)D; p.N = rows;
      }
    }
  };
  // mx_in: 0 = quantise A here; 1 = A is the wide operand a previous epilogue left in q8 / mxw; 2 = the LayerNorm kernel left it in q8n (+ qs: one scale
  // per row, or + mxn with AFX_FP8_NORM_MX); 3 = the attention kernel left it in q8n / mxn.
  // mx_out: this GEMM's epilogue writes the wide operand.
  const int split_txt_default = 0;
  auto stream_gemm = [&](const uint16_t* A, int64_t lda, int K, const LinW (&lw)[2], uint16_t* C, int64_t ldc, int Nout, int epi,
                         int blk, int gate_chunk, const float* qkn = nullptr, int mx_in = 0, bool mx_out = false, int kind = 0) -> int {
    GemmBatch gb{};
    if (mx && mx_in == 0) HIP_TRY(launch_quant_rows_mx8(A, lda, ws.q8n, K, ws.mxn, ws.ld_mxn, (int)R, K, st));      // (mx_in 2: the LayerNorm kernel wrote q8n / mxn itself)
    else if (c->fp8 && !mx) HIP_TRY(launch_quant_rows_fp8(A, lda, ws.q8, K, ws.qs, (int)R, K, st));     // per-token scales, all rows at once
    for (int b = 0; b < B; ++b)
      for (int s = 0; s < 2; ++s) {   // 0 image rows, 1 text rows
        GemmProblem& p = gb.p[gb.nprob++];
        p = GemmProblem{};
        const int64_t row0 = (int64_t)b * S + (s == 0 ? T : 0);
        p.A = A + row0 * lda; p.lda = lda;
        p.W = lw[s].w; p.ldw = K; p.bias = lw[s].b;
        if (mx) {
          p.W = (const uint16_t*)lw[s].wq; p.fp8 = 1; p.a_scale = ws.ones + row0; p.w_scale = lw[s].wscale; p.lda = K;
          if (mx_in == 2 && norm_rows) { p.A = (const uint16_t*)(ws.q8n + row0 * K); p.a_scale = ws.qs + row0; }      // LayerNorm rows: one scale per row, the plain fp8 MFMA
          else if (mx_in != 1) { p.A = (const uint16_t*)(ws.q8n + row0 * K); p.a_mx = ws.mxn + row0 * ws.ld_mxn; p.ld_mx = ws.ld_mxn; }
          else { p.A = (const uint16_t*)(ws.q8 + row0 * K); p.a_mx = ws.mxw + row0 * ws.ld_mxw; p.ld_mx = ws.ld_mxw; }
          if (mx_out) { p.c8 = ws.q8 + row0 * Nout; p.ldc8 = Nout; p.c_mx = ws.mxw + row0 * ws.ld_mxw; p.ld_cmx = ws.ld_mxw; p.c8_col0 = 0; }
        } else if (c->fp8) {
          p.A = (const uint16_t*)(ws.q8 + row0 * K); p.lda = K;
          p.W = (const uint16_t*)lw[s].wq; p.fp8 = 1; p.a_scale = ws.qs + row0; p.w_scale = lw[s].wscale;
        }
        p.C = C + row0 * ldc; p.ldc = ldc;
        p.M = (s == 0 ? N : T); p.N = Nout; p.K = K;
        p.epi = epi; p.gelu_col0 = 0;
        if (qkn != nullptr) {           // [img_q, img_k, txt_q, txt_k][128]
          p.qk_D = (int)D; p.qk_wq = qkn + (s == 0 ? 0 : 2) * 128; p.qk_wk = qkn + (s == 0 ? 1 : 3) * 128;
          p.rope_cos = rope_cos; p.rope_sin = rope_sin; p.rope_row0 = (s == 0 ? T : 0); p.rope_period = 1 << 30; p.rope_rows = (int)S;
        }
        if (epi == EPI_GATE_RES) {
          p.gate = ws.mod + (int64_t)b * ldm + ml.dbl(blk, s, gate_chunk); p.ldg = 0; p.rows_per_batch = 1 << 30;
          p.res = C + row0 * ldc; p.ldr = ldc;
        }
      }
    gb.sk_slab = ws.sk_slab; gb.sk_flags = ws.sk_flags;
    // A launch costs ceil(rounds) x tile area: a SHORT text stream (Qwen-Image: 128 rows) costs the grouped launch a whole row of half-empty tiles --
    // at N = 12288, 4096 + 128 rows are 816 tiles of 256x256 (3.2 -> 4 rounds; the launcher settles for 1024 tiles of 288x192 = 4 rounds) where the image rows
    // alone are 768 = exactly 3.  split_txt (bit per launch kind: 1 qkv, 2 out, 4 mlp1, 8 mlp2; AFX_SPLIT_TXT overrides): the text problems go out as a launch
    // of their own (small tiles, two work-groups per CU) behind the image problems'.
    static int split_env = -2;
    if (split_env == -2) {
      const char* e = getenv("AFX_SPLIT_TXT");
      split_env = e ? atoi(e) : -1;
    }
    const int split_mask = split_env >= 0 ? split_env : split_txt_default;
    if ((split_mask & kind) != 0 && T > 0 && N > 0 && !mx && !c->fp8) {
      GemmBatch gi{}, gt{};
      for (int j = 0; j < gb.nprob; ++j) ((j & 1) ? gt : gi).p[((j & 1) ? gt : gi).nprob++] = gb.p[j];
      gi.sk_slab = gt.sk_slab = ws.sk_slab; gi.sk_flags = gt.sk_flags = ws.sk_flags;
      { ProfScope ps_(c, st, 0, gemm_flops(gi)); HIP_TRY(launch_gemm(gi, st)); }
      { ProfScope ps_(c, st, 0, gemm_flops(gt)); HIP_TRY(launch_gemm(gt, st)); }
      return AFX_OK;
    }
    { ProfScope ps_(c, st, 0, gemm_flops(gb)); HIP_TRY(launch_gemm(gb, st)); }
    return AFX_OK;
  };
  // LN + modulate of both streams of every sample in one launch (text rows take the text stream's vectors)
  bool norm_fused = false;      // mx: the last stream_norm wrote the GEMM operand itself
  auto stream_norm = [&](int blk, int shift_chunk, int scale_chunk) -> int {
    norm_fused = false;
    if (mx) {
      HIP_TRY(launch_norm_modulate_mx8(ws.X, D, ws.q8n, D, ws.mxn, ws.ld_mxn, (int)R, (int)D, ws.mod + ml.dbl(blk, 0, scale_chunk),
                                       ws.mod + ml.dbl(blk, 0, shift_chunk), ws.mod + ml.dbl(blk, 1, scale_chunk),
                                       ws.mod + ml.dbl(blk, 1, shift_chunk), ldm, (int)S, T, st, &norm_fused, norm_rows ? ws.qs : nullptr));
      if (norm_fused) return AFX_OK;
    }
    HIP_TRY(launch_norm_modulate_joint(ws.X, D, ws.Xn, D, (int)R, (int)D, ws.mod + ml.dbl(blk, 0, scale_chunk),
                                       ws.mod + ml.dbl(blk, 0, shift_chunk), ws.mod + ml.dbl(blk, 1, scale_chunk),
                                       ws.mod + ml.dbl(blk, 1, shift_chunk), ldm, S, T, st));
    return AFX_OK;
  };

  int rc;
  // ---- dual-stream blocks ---------------------------------------------------------------------------
  uint16_t* QKV = ws.F;                 // [R, 3D]  rows k|v|q
  uint16_t* Hb = ws.F + R * 3 * D;      // [R, 4D]  MLP hidden
  for (int i = 0; stage == 0 && i < d.num_double; ++i) {
    const DoubleW& bw = c->dbl[i];
    const float* qkn = bw.qkn;          // [img_q, img_k, txt_q, txt_k][128]
    if ((rc = join_side(i))) return rc;
    if (c->ckpt) HIP_TRY(hipMemcpyAsync(c->ckpt + (int64_t)i * R * D, ws.X, (size_t)R * D * 2, hipMemcpyDeviceToDevice, st));
    if ((rc = stream_norm(i, 0, 1))) return rc;
    if (vt_fuse) {                        // per sample: img k, q, v^T + txt k, q, v^T = 6 problems in one launch
      for (int b = 0; b < B; ++b) {
        GemmBatch gb{};
        for (int s = 0; s < 2; ++s)
          kqv_problems(gb, ws.Xn, D, bw.qkv[s], QKV, 3 * D, (int64_t)b * S + (s == 0 ? T : 0), s == 0 ? N : T, b, s == 0 ? T : 0, 1 << 30,
                       qkn + (s == 0 ? 0 : 2) * 128, qkn + (s == 0 ? 1 : 3) * 128, true);
        { ProfScope ps_(c, st, 0, gemm_flops(gb)); HIP_TRY(launch_gemm(gb, st)); }
      }
    } else
    if ((rc = stream_gemm(ws.Xn, D, (int)D, bw.qkv, QKV, 3 * D, (int)(3 * D), EPI_NONE, i, 0, qk_fuse ? qkn : nullptr, norm_fused ? 2 : 0, false, 1))) return rc;
    // k, q: RMSNorm + RoPE in place, v -> V^T: one launch
    if (vt_fuse) {
    } else if (qk_fuse)
      HIP_TRY(launch_v_transpose(QKV + D, 3 * D, ws.Vt, B, H, S, st));
    else
    HIP_TRY(launch_kv_prep(QKV, QKV + 2 * D, 3 * D, qkn + 3 * 128, qkn + 1 * 128, qkn + 2 * 128, qkn, rope_cos, rope_sin, T, QKV + D,
                           3 * D, ws.Vt, B, H, S, st));
    bool o_fused = false;            // mx: the attention kernel wrote the out-projection's operand itself (q8n / mxn)
    const AttnMx8 omx_d{ws.q8n, D, ws.mxn, ws.ld_mxn};
    { ProfScope ps_(c, st, 1, 4.0 * B * H * (double)S * S * 128);
      HIP_TRY(launch_attention(QKV + 2 * D, 3 * D, QKV, 3 * D, ws.Vt, QKV + 2 * D, 3 * D, B, H, S, st, nullptr, attn_mx ? &omx_d : nullptr, &o_fused)); }
    if ((rc = stream_gemm(QKV + 2 * D, 3 * D, (int)D, bw.out, ws.X, D, (int)D, EPI_GATE_RES, i, 2, nullptr, o_fused ? 3 : 0, false, 2))) return rc;
    if ((rc = stream_norm(i, 3, 4))) return rc;
    if ((rc = stream_gemm(ws.Xn, D, (int)D, bw.mlp1, Hb, 4 * D, (int)(4 * D), EPI_GELU, i, 0, nullptr, norm_fused ? 2 : 0, mx, 4))) return rc;      // (mx: the hidden leaves as the next GEMM's operand, Hb stays unwritten)
    if ((rc = stream_gemm(Hb, 4 * D, (int)(4 * D), bw.mlp2, ws.X, D, (int)D, EPI_GATE_RES, i, 5, nullptr, mx ? 1 : 0, false, 8))) return rc;
  }

  // ---- single-stream blocks on the joint sequence -------------------------------------------------
  for (int i = 0; stage == 0 && i < d.num_single; ++i) {
    const SingleW& bw = c->sgl[i];
    const float* qkn = bw.qkn;          // [q, k][128]
    if ((rc = join_side(d.num_double + i))) return rc;
    if (c->ckpt)
      HIP_TRY(hipMemcpyAsync(c->ckpt + (int64_t)(d.num_double + i) * R * D, ws.X, (size_t)R * D * 2, hipMemcpyDeviceToDevice, st));
    bool sgl_fused = false;
    if (mx) HIP_TRY(launch_norm_modulate_mx8(ws.X, D, ws.q8n, D, ws.mxn, ws.ld_mxn, (int)R, (int)D, ws.mod + ml.sgl(i, 1), ws.mod + ml.sgl(i, 0),
                                             nullptr, nullptr, ldm, (int)S, 0, st, &sgl_fused, norm_rows ? ws.qs : nullptr));
    if (!sgl_fused) HIP_TRY(launch_norm_modulate(ws.X, D, ws.Xn, D, (int)R, (int)D, ws.mod + ml.sgl(i, 1), ws.mod + ml.sgl(i, 0), ldm, S, 0, st));
    GemmBatch gb{};
    gb.nprob = 1;
    GemmProblem& f = gb.p[0];
    f = GemmProblem{};
    f.A = ws.Xn; f.lda = D; f.W = bw.fused.w; f.ldw = D; f.bias = bw.fused.b;
    f.C = ws.F; f.ldc = 7 * D; f.M = (int)R; f.N = (int)(7 * D); f.K = (int)D; f.epi = EPI_GELU; f.gelu_col0 = (int)(3 * D);
    if (mx) {             // A: the LayerNorm rows, block-scaled; the mlp columns leave as columns [D, 5D) of the proj_out operand
      if (!sgl_fused) HIP_TRY(launch_quant_rows_mx8(ws.Xn, D, ws.q8n, D, ws.mxn, ws.ld_mxn, (int)R, (int)D, st));
      f.A = (const uint16_t*)ws.q8n; f.W = (const uint16_t*)bw.fused.wq; f.fp8 = 1; f.a_scale = ws.ones; f.w_scale = bw.fused.wscale;
      if (sgl_fused && norm_rows) f.a_scale = ws.qs;      // the LayerNorm kernel wrote one scale per row: the plain fp8 MFMA
      else { f.a_mx = ws.mxn; f.ld_mx = ws.ld_mxn; }
      f.c8 = ws.q8 + D; f.ldc8 = 5 * D; f.c_mx = ws.mxw + D / 128; f.ld_cmx = ws.ld_mxw; f.c8_col0 = (int)(3 * D);
    } else if (c->fp8) {
      HIP_TRY(launch_quant_rows_fp8(ws.Xn, D, ws.q8, D, ws.qs, (int)R, (int)D, st));
      f.A = (const uint16_t*)ws.q8; f.W = (const uint16_t*)bw.fused.wq; f.fp8 = 1; f.a_scale = ws.qs; f.w_scale = bw.fused.wscale;
    }
    if (vt_fuse && B + 3 <= GEMM_MAX_PROBLEMS) {
      // k and q over the joint rows of every sample (position = row % S), the mlp columns, one transposed v per sample
      gb.nprob = 0;
      kqv_problems(gb, ws.Xn, D, bw.fused, ws.F, 7 * D, 0, (int)R, 0, 0, (int)S, qkn, qkn + 128, false);
      GemmProblem& m = gb.p[gb.nprob++];
      m = GemmProblem{};
      m.A = ws.Xn; m.lda = D; m.W = bw.fused.w + 3 * D * D; m.ldw = D; m.bias = bw.fused.b ? bw.fused.b + 3 * D : nullptr;
      m.C = ws.F + 3 * D; m.ldc = 7 * D; m.M = (int)R; m.N = (int)(4 * D); m.K = (int)D; m.epi = EPI_GELU; m.gelu_col0 = 0;
      for (int b = 0; b < B; ++b) {
        GemmProblem& v = gb.p[gb.nprob++];
        v = GemmProblem{};
        v.A = bw.fused.w + D * D; v.lda = D; v.bias = bw.fused.b ? bw.fused.b + D : nullptr; v.bias_rows = 1;
        v.W = ws.Xn + (int64_t)b * S * D; v.ldw = D; v.w_perm16 = 1;
        v.C = ws.Vt + (int64_t)b * vt_sample; v.ldc = S_pad_; v.M = (int)D; v.N = (int)S; v.K = (int)D; v.epi = EPI_NONE;
      }
    } else if (qk_fuse) {               // [q, k][128]; the joint rows of every sample: position = row % S
      f.qk_D = (int)D; f.qk_wq = qkn; f.qk_wk = qkn + 128;
      f.rope_cos = rope_cos; f.rope_sin = rope_sin; f.rope_row0 = 0; f.rope_period = (int)S; f.rope_rows = (int)S;
    }
    gb.sk_slab = ws.sk_slab; gb.sk_flags = ws.sk_flags;
    { ProfScope ps_(c, st, 0, gemm_flops(gb)); HIP_TRY(launch_gemm(gb, st)); }
    if (vt_fuse && B + 3 <= GEMM_MAX_PROBLEMS) {
    } else if (qk_fuse)
      HIP_TRY(launch_v_transpose(ws.F + D, 7 * D, ws.Vt, B, H, S, st));
    else
    HIP_TRY(launch_kv_prep(ws.F, ws.F + 2 * D, 7 * D, qkn + 128, qkn + 128, qkn, qkn, rope_cos, rope_sin, T, ws.F + D, 7 * D, ws.Vt, B,
                           H, S, st));
    bool o_fused = false;            // mx: ... columns [0, D) of the [O | mlp] operand (q8 / mxw)
    const AttnMx8 omx_s{ws.q8, 5 * D, ws.mxw, ws.ld_mxw};
    { ProfScope ps_(c, st, 1, 4.0 * B * H * (double)S * S * 128);
      HIP_TRY(launch_attention(ws.F + 2 * D, 7 * D, ws.F, 7 * D, ws.Vt, ws.F + 2 * D, 7 * D, B, H, S, st, nullptr, attn_mx ? &omx_s : nullptr, &o_fused)); }
    GemmBatch go{};
    go.nprob = 1;
    GemmProblem& o = go.p[0];
    o = GemmProblem{};
    o.A = ws.F + 2 * D; o.lda = 7 * D; o.W = bw.out.w; o.ldw = 5 * D; o.bias = bw.out.b;
    o.C = ws.X; o.ldc = D; o.M = (int)R; o.N = (int)D; o.K = (int)(5 * D); o.epi = EPI_GATE_RES;
    o.gate = ws.mod + ml.sgl(i, 2); o.ldg = ldm; o.rows_per_batch = S; o.res = ws.X; o.ldr = D;
    if (mx) {             // the attention output joins the mlp columns the projection's epilogue left in q8
      if (!o_fused) HIP_TRY(launch_quant_rows_mx8(ws.F + 2 * D, 7 * D, ws.q8, 5 * D, ws.mxw, ws.ld_mxw, (int)R, (int)D, st));
      o.A = (const uint16_t*)ws.q8; o.lda = 5 * D; o.W = (const uint16_t*)bw.out.wq; o.fp8 = 1; o.a_scale = ws.ones; o.w_scale = bw.out.wscale;
      o.a_mx = ws.mxw; o.ld_mx = ws.ld_mxw;
    } else if (c->fp8) {
      HIP_TRY(launch_quant_rows_fp8(ws.F + 2 * D, 7 * D, ws.q8, 5 * D, ws.qs, (int)R, (int)(5 * D), st));
      o.A = (const uint16_t*)ws.q8; o.lda = 5 * D; o.W = (const uint16_t*)bw.out.wq; o.fp8 = 1; o.a_scale = ws.qs;
      o.w_scale = bw.out.wscale;
    }
    go.sk_slab = ws.sk_slab; go.sk_flags = ws.sk_flags;
    { ProfScope ps_(c, st, 0, gemm_flops(go)); HIP_TRY(launch_gemm(go, st)); }
  }
  if (stage == 0 && (rc = join_side(1 << 30))) return rc;      // short trunks: the head reads the norm_out rows

  // ---- norm_out (scale first) + velocity head on the image tokens --------------------------------------
  for (int b = 0; b < B; ++b)
    HIP_TRY(launch_norm_modulate(ws.X + ((int64_t)b * S + T) * D, D, ws.Xn + (int64_t)b * N * D, D, N, (int)D,
                                 ws.mod + (int64_t)b * ldm + ml.fin(0), ws.mod + (int64_t)b * ldm + ml.fin(1), 0,
                                 1 << 30, 0, st));
  {
    GemmBatch gb{};
    gb.nprob = 1;
    GemmProblem& hp = gb.p[0];
    hp = GemmProblem{};
    hp.A = ws.Xn; hp.lda = D; hp.W = W16(c, "head.weight"); hp.ldw = D; hp.bias = W16(c, "head.bias");
    hp.M = B * N; hp.N = c->head_n; hp.K = (int)D; hp.epi = EPI_NONE;
    if (d.head_mode == 0) {
      hp.C = ws.head; hp.ldc = c->head_n;
    } else {
      hp.C = (uint16_t*)means; hp.ldc = c->head_n;    // teacher: velocity [B*N, in_channels] written directly
    }
    { ProfScope ps_(c, st, 0, gemm_flops(gb)); HIP_TRY(launch_gemm(gb, st)); }
  }
  if (d.head_mode == 0)
    HIP_TRY(launch_head_split(ws.head, c->head_n, (uint16_t*)means, (uint16_t*)logw, (uint16_t*)logg, (int64_t)B * N,
                              d.num_gaussians, d.in_channels, d.logweights_channels, st));
  return AFX_OK;
}

int afx_mmdit_prepare_steps(afx_ctx* c, const void* pooled, const float* t_steps, const float* g, int32_t B, int32_t nsteps,
                            void* stream_) {
  if (!c || !t_steps) return fail(AFX_E_INVALID, "null argument to afx_mmdit_prepare_steps");
  if (!c->finalized) return fail(AFX_E_MISSING, "afx_finalize() has not succeeded on this context");
  const afx_model_desc& d = c->d;
  if (B < 1 || nsteps < 1 || B > AFX_MAX_MICRO_BATCH || (int64_t)B * nsteps > AFX_PREP_ROWS)
    return fail(AFX_E_INVALID, "afx_mmdit_prepare_steps: batch <= 4 and batch * steps <= %d", AFX_PREP_ROWS);
  if (d.guidance_embeds && !g) return fail(AFX_E_INVALID, "guidance vector required");
  if (d.pooled_dim > 0 && !pooled) return fail(AFX_E_INVALID, "pooled projections required");
  if (c->temb_override) return fail(AFX_E_INVALID, "afx_mmdit_prepare_steps: not with a timestep-embedding override");
  Workspace ws = carve(c, c->ws, 1, 1, 1);                       // (only the shape-independent head of the workspace is used)
  if (!c->ws || ws.total > c->ws_bytes) return fail(AFX_E_WORKSPACE, "workspace not set / too small");
  hipStream_t st = (hipStream_t)stream_;
  const int64_t D = c->D, ldm = c->n_mod;
  const int rows = B * nsteps;
  c->prep_steps = 0;
  c->prep_use = -1;
  // temb of every (step, sample): the same three tiny MLPs as the forward, B rows at a time into row block k
  float* sc = ws.prep_semb;                                      // scratch: sincos / hidden rows live in the semb block until the SiLU
  for (int k = 0; k < nsteps; ++k) {
    float* temb = ws.prep_temb + (int64_t)k * B * D;
    float* hid = sc + (int64_t)k * B * D;                        // [B, D] hidden of the current MLP (overwritten by the SiLU below)
    float* sin_ = ws.prep_mod;                                   // [B, 256] / [B, pooled_dim]: free until the big pass
    HIP_TRY(launch_sincos(t_steps + (int64_t)k * B, 1000.0f, sin_, B, d.family == 0 ? 1 : 2, st));
    HIP_TRY(launch_gemv(sin_, W16(c, "temb.t.l1.weight"), W16(c, "temb.t.l1.bias"), hid, B, (int)D, 256, 1, 0, st));
    HIP_TRY(launch_gemv(hid, W16(c, "temb.t.l2.weight"), W16(c, "temb.t.l2.bias"), temb, B, (int)D, (int)D, 0, 0, st));
    if (d.guidance_embeds) {
      HIP_TRY(launch_sincos(g, 1000.0f, sin_, B, 1, st));
      HIP_TRY(launch_gemv(sin_, W16(c, "temb.g.l1.weight"), W16(c, "temb.g.l1.bias"), hid, B, (int)D, 256, 1, 0, st));
      HIP_TRY(launch_gemv(hid, W16(c, "temb.g.l2.weight"), W16(c, "temb.g.l2.bias"), temb, B, (int)D, (int)D, 0, 1, st));
    }
    if (d.pooled_dim > 0) {
      HIP_TRY(launch_bf16_to_f32((const uint16_t*)pooled, sin_, (int64_t)B * d.pooled_dim, st));
      HIP_TRY(launch_gemv(sin_, W16(c, "temb.p.l1.weight"), W16(c, "temb.p.l1.bias"), hid, B, (int)D, d.pooled_dim, 1, 0, st));
      HIP_TRY(launch_gemv(hid, W16(c, "temb.p.l2.weight"), W16(c, "temb.p.l2.bias"), temb, B, (int)D, (int)D, 0, 1, st));
    }
  }
  HIP_TRY(launch_silu(ws.prep_temb, ws.prep_semb, (int64_t)rows * D, st));
  // ONE pass over the stacked [n_mod, D] matrix for all steps: the matrix (6.5 GB for FLUX) is what the time goes into, the
  // number of right-hand sides is free up to the GEMV's 8
  HIP_TRY(launch_gemv(ws.prep_semb, W16(c, "mod.weight"), W16(c, "mod.bias"), ws.prep_mod, rows, (int)c->n_mod, (int)D, 0, 0, st, ldm));
  if (W16(c, "mod_final.weight") != nullptr) {
    ModLayout ml{D, d.num_double, d.num_single};
    HIP_TRY(launch_gemv(ws.prep_semb, W16(c, "mod_final.weight"), W16(c, "mod_final.bias"), ws.prep_mod + ml.fin(0), rows, (int)(2 * D),
                        (int)D, 0, 0, st, ldm));
  }
  c->prep_steps = nsteps;
  c->prep_B = B;
  return AFX_OK;
}

int afx_mmdit_use_prepared_step(afx_ctx* c, int32_t k) {
  if (!c) return fail(AFX_E_INVALID, "null ctx");
  if (k >= c->prep_steps) return fail(AFX_E_INVALID, "afx_mmdit_use_prepared_step: step %d of %d prepared", k, c->prep_steps);
  c->prep_use = k < 0 ? -1 : k;
  return AFX_OK;
}

int afx_set_fp8_linear(afx_ctx* c, int32_t on) {
  if (!c) return fail(AFX_E_INVALID, "null ctx");
  if (on) {
    const int64_t D = c->D;
    int r;
    auto chk = [&](const std::string& n, int64_t out_f, int64_t in_f) -> int {
      if ((r = need(c, n + ".weight_q", AFX_DT_FP8, {out_f, in_f})) != AFX_OK) return r;
      return need(c, n + ".wscale", AFX_DT_F32, {out_f});
    };
    for (int i = 0; i < c->d.num_double; ++i)
      for (const char* s : {"img_", "txt_"}) {
        const std::string p = "d" + std::to_string(i) + "." + s;
        if ((r = chk(p + "qkv", 3 * D, D)) || (r = chk(p + "out", D, D)) || (r = chk(p + "mlp1", 4 * D, D)) || (r = chk(p + "mlp2", D, 4 * D))) return r;
      }
    for (int i = 0; i < c->d.num_single; ++i) {
      const std::string p = "s" + std::to_string(i) + ".";
      if ((r = chk(p + "fused", 7 * D, D)) || (r = chk(p + "out", D, 5 * D))) return r;
    }
  }
  c->fp8 = on != 0;
  return AFX_OK;
}

int afx_set_temb_override(afx_ctx* ctx, const float* temb_t) {
  if (!ctx) return fail(AFX_E_INVALID, "null ctx");
  ctx->temb_override = temb_t;
  return AFX_OK;
}

int afx_set_checkpoint_buffer(afx_ctx* ctx, void* dptr) {
  if (!ctx) return fail(AFX_E_INVALID, "null ctx");
  ctx->ckpt = (uint16_t*)dptr;     // nullptr switches checkpointing off
  return AFX_OK;
}

int afx_profile_enable(afx_ctx* ctx, int32_t on) {
  if (!ctx) return fail(AFX_E_INVALID, "null ctx");
  ctx->prof_on = on != 0;
  ctx->prof_stride = on > 1 ? on : 1;
  ctx->prof_count = 0;
  ctx->prof_used = 0;
  return AFX_OK;
}

int afx_profile_read(afx_ctx* ctx, int32_t klass, double* total_ms, int64_t* launches, double* flops) {
  if (!ctx || !total_ms || !launches || !flops) return fail(AFX_E_INVALID, "null argument to afx_profile_read");
  double ms = 0, fl = 0;
  int64_t n = 0;
  for (size_t i = 0; i < ctx->prof_used; ++i) {
    auto& r = ctx->prof_pool[i];
    if (r.klass != klass) continue;
    HIP_TRY(hipEventSynchronize(r.b));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
    ms += t; fl += r.flops; ++n;
  }
  *total_ms = ms; *launches = n; *flops = fl;
  return AFX_OK;
}

int afx_arcflow_step(const float* x_in, const void* means, const void* logw, const void* logg, int32_t mix_dtype,
                     float sigma_src, float sigma_start, float sigma_end, const float* sigma_vec, float eps,
                     float* x_out, int32_t batch, int32_t n_tok, int32_t K, int32_t ch, int32_t pp, void* stream) {
  if (!x_in || !means || !logw || !logg || !x_out) return fail(AFX_E_INVALID, "null argument to afx_arcflow_step");
  if (mix_dtype != AFX_DT_BF16 && mix_dtype != AFX_DT_F32) return fail(AFX_E_INVALID, "bad mix_dtype");
  if (batch < 0 || n_tok < 0 || K < 1 || K > 32 || ch < 1 || pp < 1 || ch % pp != 0)
    return fail(AFX_E_INVALID, "bad shape for afx_arcflow_step");
  HIP_TRY(launch_arcflow_step(x_in, means, logw, logg, mix_dtype == AFX_DT_BF16, sigma_src, sigma_start, sigma_end,
                              sigma_vec, eps, x_out, batch, n_tok, K, ch, pp, 0, nullptr, (hipStream_t)stream));
  return AFX_OK;
}

int afx_arcflow_step_dropout(const float* x_in, const void* means, const void* logw, const void* logg, int32_t mix_dtype,
                             const float* sigma_vec, const uint8_t* drop_mask, float eps, float* x_out, int32_t batch,
                             int32_t n_tok, int32_t K, int32_t ch, int32_t pp, void* stream) {
  if (!x_in || !means || !logw || !logg || !x_out || !sigma_vec)
    return fail(AFX_E_INVALID, "null argument to afx_arcflow_step_dropout");
  if (mix_dtype != AFX_DT_BF16 && mix_dtype != AFX_DT_F32) return fail(AFX_E_INVALID, "bad mix_dtype");
  if (batch < 0 || n_tok < 0 || K < 1 || K > 32 || ch < 1 || pp < 1 || ch % pp != 0)
    return fail(AFX_E_INVALID, "bad shape for afx_arcflow_step_dropout");
  HIP_TRY(launch_arcflow_step(x_in, means, logw, logg, mix_dtype == AFX_DT_BF16, 0.f, 0.f, 0.f, sigma_vec, eps, x_out,
                              batch, n_tok, K, ch, pp, 0, drop_mask, (hipStream_t)stream));
  return AFX_OK;
}

int afx_linear_tn_f32out(const void* X, int64_t ldx, const void* Y, int64_t ldy, float* C, int64_t ldc, int32_t M, int32_t N1, int32_t N2,
                         int32_t accumulate, void* stream) {
  if (!X || !Y || !C) return fail(AFX_E_INVALID, "null argument to afx_linear_tn_f32out");
  if (M < 0 || N1 < 0 || N2 < 0 || N1 % 8 || N2 % 8 || ldx % 8 || ldy % 8 || ldc % 4 || ldx < N1 || ldy < N2 || ldc < N2)
    return fail(AFX_E_INVALID, "afx_linear_tn_f32out: need N1%%8==0, N2%%8==0, ldx/ldy%%8==0, ldc%%4==0, leading dimensions >= the widths");
  if (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)C) & 15) return fail(AFX_E_INVALID, "afx_linear_tn_f32out: operands must be 16-byte aligned");
  HIP_TRY(launch_gemm_tn_f32((const uint16_t*)X, ldx, (const uint16_t*)Y, ldy, C, ldc, M, N1, N2, accumulate, (hipStream_t)stream));
  return AFX_OK;
}

int64_t afx_linear_tn_ws_bytes(int32_t M, int32_t N1, int32_t N2) {
  if (M < 0 || N1 < 0 || N2 < 0) return fail(AFX_E_INVALID, "bad shape to afx_linear_tn_ws_bytes");
  return gemm_tn_ws_bytes(M, N1, N2);
}

int afx_linear_tn_f32out_ws(const void* X, int64_t ldx, const void* Y, int64_t ldy, float* C, int64_t ldc, int32_t M, int32_t N1, int32_t N2,
                            int32_t accumulate, void* ws, void* stream) {
  if (!X || !Y || !C) return fail(AFX_E_INVALID, "null argument to afx_linear_tn_f32out_ws");
  if (M < 0 || N1 < 0 || N2 < 0 || N1 % 8 || N2 % 8 || ldx % 8 || ldy % 8 || ldc % 4 || ldx < N1 || ldy < N2 || ldc < N2)
    return fail(AFX_E_INVALID, "afx_linear_tn_f32out_ws: need N1%%8==0, N2%%8==0, ldx/ldy%%8==0, ldc%%4==0, leading dimensions >= the widths");
  if (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)C | (uintptr_t)ws) & 15) return fail(AFX_E_INVALID, "afx_linear_tn_f32out_ws: operands must be 16-byte aligned");
  if (!ws && gemm_tn_ws_bytes(M, N1, N2) > 0) return fail(AFX_E_INVALID, "afx_linear_tn_f32out_ws: this shape needs afx_linear_tn_ws_bytes() of workspace");
  HIP_TRY(launch_gemm_tn_f32((const uint16_t*)X, ldx, (const uint16_t*)Y, ldy, C, ldc, M, N1, N2, accumulate, (hipStream_t)stream, (float*)ws));
  return AFX_OK;
}

int afx_linear_bf16_f32out(const void* A, int64_t lda, const void* W, int64_t ldw, float* C, int64_t ldc, int32_t M,
                           int32_t N, int32_t K, int32_t accumulate, void* stream) {
  if (!A || !W || !C) return fail(AFX_E_INVALID, "null argument to afx_linear_bf16_f32out");
  if (M < 0 || N < 0 || K <= 0 || K % 64 || N % 8 || lda % 8 || ldw % 8 || ldc % 4)
    return fail(AFX_E_INVALID, "afx_linear_bf16_f32out: need K%%64==0, N%%8==0, lda/ldw%%8==0, ldc%%4==0");
  GemmBatch gb{};
  gb.nprob = 1;
  GemmProblem& p = gb.p[0];
  p = GemmProblem{};
  p.A = (const uint16_t*)A; p.lda = lda; p.W = (const uint16_t*)W; p.ldw = ldw; p.C = (uint16_t*)C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.epi = EPI_NONE; p.rows_per_batch = 1; p.out_f32 = accumulate ? 2 : 1;
  HIP_TRY(launch_gemm(gb, (hipStream_t)stream));
  return AFX_OK;
}

int afx_mmdit_import_tokens(afx_ctx* c, const void* src, int32_t batch, int32_t n_img, int32_t n_txt, void* stream) {
  if (!c || !src || !c->ws || batch < 1 || n_img < 1 || n_txt < 0) return fail(AFX_E_INVALID, "bad argument to afx_mmdit_import_tokens");
  Workspace ws = carve(c, c->ws, batch, n_img, n_txt);
  if (ws.total > c->ws_bytes)
    return fail(AFX_E_WORKSPACE, "afx_mmdit_import_tokens: shape (%d, %d, %d) needs %lld workspace bytes, have %lld", batch, n_img, n_txt,
                (long long)ws.total, (long long)c->ws_bytes);
  HIP_TRY(hipMemcpyAsync(ws.X, src, (size_t)batch * (n_img + n_txt) * c->D * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return AFX_OK;
}

int afx_linear_splitk_chunks(int32_t M, int32_t N, int32_t K, int32_t split_k) {
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256), nk = K / 64;
  if (split_k <= 0) {                    // ~2 work-groups per CU (512 in flight), at least 4 K-tiles per chunk
    split_k = (512 + tiles / 2) / tiles;
    if (split_k > nk / 4) split_k = nk / 4;
    if (split_k < 1) split_k = 1;
  }
  split_k = split_k < nk ? split_k : nk;
  const int per = (nk + split_k - 1) / split_k;
  return (nk + per - 1) / per;           // no empty chunk
}

int afx_linear_bf16_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, float* partials, int32_t M,
                           int32_t N, int32_t K, int32_t split_k, void* stream) {
  if (!A || !W || !partials) return fail(AFX_E_INVALID, "null argument to afx_linear_bf16_splitk");
  if (M < 0 || N < 0 || K <= 0 || K % 64 || N % 8 || lda % 8 || ldw % 8 || split_k < 0)
    return fail(AFX_E_INVALID, "afx_linear_bf16_splitk: need K%%64==0, N%%8==0, lda/ldw%%8==0, split_k >= 0");
  GemmBatch gb{};
  gb.nprob = 1;
  GemmProblem& p = gb.p[0];
  p = GemmProblem{};
  p.A = (const uint16_t*)A; p.lda = lda; p.W = (const uint16_t*)W; p.ldw = ldw; p.bias = (const uint16_t*)bias; p.C = (uint16_t*)partials;
  p.ldc = N; p.M = M; p.N = N; p.K = K; p.epi = EPI_NONE; p.rows_per_batch = 1; p.out_f32 = 3;
  p.split_k = afx_linear_splitk_chunks(M, N, K, split_k);
  p.split_stride = (int64_t)M * N;
  HIP_TRY(launch_gemm(gb, (hipStream_t)stream));
  return AFX_OK;
}

int afx_mmdit_export(afx_ctx* c, const char* what, void* dst, int32_t batch, int32_t n_img, int32_t n_txt, void* stream) {
  if (!c || !what || !dst || !c->ws || batch < 1 || n_img < 1 || n_txt < 0) return fail(AFX_E_INVALID, "bad argument to afx_mmdit_export");
  Workspace ws = carve(c, c->ws, batch, n_img, n_txt);
  // the layout is a function of the shape: a shape the bound workspace was not sized for would read past it / from another layout
  if (ws.total > c->ws_bytes)
    return fail(AFX_E_WORKSPACE, "afx_mmdit_export: shape (%d, %d, %d) needs %lld workspace bytes, have %lld", batch, n_img, n_txt,
                (long long)ws.total, (long long)c->ws_bytes);
  const int64_t D = c->D;
  ModLayout ml{D, c->d.num_double, c->d.num_single};
  hipStream_t st = (hipStream_t)stream;
  const std::string w(what);
  if (w == "head_in") {            // [B*N, D] bf16: norm_out output = input of the velocity head
    HIP_TRY(hipMemcpyAsync(dst, ws.Xn, (size_t)batch * n_img * D * 2, hipMemcpyDeviceToDevice, st));
  } else if (w == "x_final") {     // [B*N, D] bf16: image tokens entering norm_out
    for (int b = 0; b < batch; ++b)
      HIP_TRY(hipMemcpyAsync((char*)dst + (size_t)b * n_img * D * 2, ws.X + ((int64_t)b * (n_img + n_txt) + n_txt) * D,
                             (size_t)n_img * D * 2, hipMemcpyDeviceToDevice, st));
  } else if (w == "x_tokens") {    // [B*(T+N), D] bf16: the joint token matrix (after the embedders / after the last block)
    HIP_TRY(hipMemcpyAsync(dst, ws.X, (size_t)batch * (n_img + n_txt) * D * 2, hipMemcpyDeviceToDevice, st));
  } else if (w == "temb") {        // [B, D] f32: the summed conditioning embedding before SiLU
    HIP_TRY(hipMemcpyAsync(dst, ws.temb, (size_t)batch * D * 4, hipMemcpyDeviceToDevice, st));
  } else if (w == "silu_temb") {   // [B, D] f32
    HIP_TRY(hipMemcpyAsync(dst, ws.semb, (size_t)batch * D * 4, hipMemcpyDeviceToDevice, st));
  } else if (w == "mod_all") {     // [B, n_mod] f32: every AdaLN modulation vector of the network
    HIP_TRY(hipMemcpyAsync(dst, ws.mod, (size_t)batch * c->n_mod * 4, hipMemcpyDeviceToDevice, st));
  } else if (w == "mod_final") {   // [B, 2D] f32: (scale | shift) of norm_out
    for (int b = 0; b < batch; ++b)
      HIP_TRY(hipMemcpyAsync((char*)dst + (size_t)b * 2 * D * 4, ws.mod + (int64_t)b * c->n_mod + ml.fin(0), (size_t)2 * D * 4,
                             hipMemcpyDeviceToDevice, st));
  } else {
    return fail(AFX_E_INVALID, "afx_mmdit_export: unknown buffer '%s'", what);
  }
  return AFX_OK;
}

int afx_arcflow_velocity(const void* means, const void* logw, const void* logg, int32_t mix_dtype, float sigma_src,
                         float sigma_t, const float* sigma_vec, float* u_out, int32_t batch, int32_t n_tok,
                         int32_t K, int32_t ch, int32_t pp, void* stream) {
  if (!means || !logw || !logg || !u_out) return fail(AFX_E_INVALID, "null argument to afx_arcflow_velocity");
  if (mix_dtype != AFX_DT_BF16 && mix_dtype != AFX_DT_F32) return fail(AFX_E_INVALID, "bad mix_dtype");
  if (batch < 0 || n_tok < 0 || K < 1 || K > 32 || ch < 1 || pp < 1 || ch % pp != 0)
    return fail(AFX_E_INVALID, "bad shape for afx_arcflow_velocity");
  HIP_TRY(launch_arcflow_step(u_out, means, logw, logg, mix_dtype == AFX_DT_BF16, sigma_src, sigma_t, sigma_t,
                              sigma_vec, 1e-4f, u_out, batch, n_tok, K, ch, pp, 1, nullptr, (hipStream_t)stream));
  return AFX_OK;
}

int afx_linear_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, void* C, int64_t ldc,
                    int32_t M, int32_t N, int32_t K, int32_t epi, int32_t gelu_col0, const float* gate,
                    int64_t ldg, int32_t rows_per_batch, const void* res, int64_t ldr, void* stream) {
  return afx_linear_bf16_pre(A, lda, W, ldw, bias, C, ldc, M, N, K, epi, gelu_col0, gate, ldg, rows_per_batch, res, ldr, nullptr, 0,
                             stream);
}

int64_t afx_linear_sk_ws_bytes(void) { return GEMM_SK_FLAG_BYTES + GEMM_SK_SLAB_BYTES; }
int afx_linear_sk_last_split(void) { return last_sk_cus(); }
int afx_gemm_dropres_available(void) { return gemm_dropres_available() ? 1 : 0; }
int afx_gemm_set_mode(int32_t impl, int32_t tile) {
  gemm_set_mode(impl, tile);
  return 0;
}
int afx_attn_set_impl(int32_t impl) {
  attn_set_impl(impl);
  return 0;
}
int afx_attn_bwd_set_impl(int32_t impl) {
  attn_bwd_set_impl(impl);
  return 0;
}

static int linear_impl(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, void* C, int64_t ldc,
                       int32_t M, int32_t N, int32_t K, int32_t epi, int32_t gelu_col0, const float* gate,
                       int64_t ldg, int32_t rows_per_batch, const void* res, int64_t ldr, const void* pre, int64_t ldp,
                       void* sk_ws, void* stream);

int afx_linear_bf16_sk(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, void* C, int64_t ldc,
                       int32_t M, int32_t N, int32_t K, int32_t epi, int32_t gelu_col0, const float* gate,
                       int64_t ldg, int32_t rows_per_batch, const void* res, int64_t ldr, void* sk_ws, void* stream) {
  if (!sk_ws || ((uintptr_t)sk_ws & 255) != 0) return fail(AFX_E_INVALID, "afx_linear_bf16_sk: sk_ws must be a 256-byte aligned device buffer");
  return linear_impl(A, lda, W, ldw, bias, C, ldc, M, N, K, epi, gelu_col0, gate, ldg, rows_per_batch, res, ldr, nullptr, 0, sk_ws, stream);
}

int afx_linear_bf16_pre(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, void* C, int64_t ldc,
                        int32_t M, int32_t N, int32_t K, int32_t epi, int32_t gelu_col0, const float* gate,
                        int64_t ldg, int32_t rows_per_batch, const void* res, int64_t ldr, const void* pre, int64_t ldp,
                        void* stream) {
  return linear_impl(A, lda, W, ldw, bias, C, ldc, M, N, K, epi, gelu_col0, gate, ldg, rows_per_batch, res, ldr, pre, ldp, nullptr, stream);
}

static int linear_impl(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, void* C, int64_t ldc,
                       int32_t M, int32_t N, int32_t K, int32_t epi, int32_t gelu_col0, const float* gate,
                       int64_t ldg, int32_t rows_per_batch, const void* res, int64_t ldr, const void* pre, int64_t ldp,
                       void* sk_ws, void* stream) {
  if (!A || !W || !C) return fail(AFX_E_INVALID, "null argument to afx_linear_bf16");
  if (pre && ldp % 8) return fail(AFX_E_INVALID, "afx_linear_bf16_pre: ldp %% 8 == 0");
  if (M < 0 || N < 0 || K <= 0 || K % 64 || N % 8 || lda % 8 || ldw % 8 || ldc % 8)
    return fail(AFX_E_INVALID, "afx_linear_bf16: need K%%64==0, N%%8==0, strides%%8==0");
  if (epi < 0 || epi > 2) return fail(AFX_E_INVALID, "bad epilogue");
  if (epi == EPI_GATE_RES && (!res || ldr % 8 || (gate && rows_per_batch < 1)))
    return fail(AFX_E_INVALID, "gated residual epilogue needs res (and rows_per_batch with a gate)");
  GemmBatch gb{};
  gb.nprob = 1;
  GemmProblem& p = gb.p[0];
  p = GemmProblem{};
  p.A = (const uint16_t*)A; p.lda = lda; p.W = (const uint16_t*)W; p.ldw = ldw; p.bias = (const uint16_t*)bias;
  p.C = (uint16_t*)C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.epi = epi; p.gelu_col0 = gelu_col0;
  p.gate = gate; p.ldg = ldg; p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; p.res = (const uint16_t*)res; p.ldr = ldr;
  p.pre = (const uint16_t*)pre; p.ldp = ldp;
  if (sk_ws) {
    gb.sk_flags = (uint32_t*)sk_ws;
    gb.sk_slab = (float*)((char*)sk_ws + GEMM_SK_FLAG_BYTES);
    gb.sk_force = 1;
  }
  HIP_TRY(launch_gemm(gb, (hipStream_t)stream));
  return AFX_OK;
}

int afx_linear_bf16_dropres(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                            const void* res, int64_t ldr, float p, uint32_t seed, int64_t row0, void* stream) {
  if (!A || !W || !C || !res) return fail(AFX_E_INVALID, "null argument to afx_linear_bf16_dropres");
  if (M < 0 || N < 0 || K <= 0 || K % 64 || N % 8 || lda % 8 || ldw % 8 || ldc % 8 || ldr % 8 || !(p >= 0.f && p < 1.f))
    return fail(AFX_E_INVALID, "afx_linear_bf16_dropres: need K%%64==0, N%%8==0, strides%%8==0, 0 <= p < 1");
  GemmBatch gb{};
  gb.nprob = 1;
  GemmProblem& q = gb.p[0];
  q = GemmProblem{};
  q.A = (const uint16_t*)A; q.lda = lda; q.W = (const uint16_t*)W; q.ldw = ldw; q.C = (uint16_t*)C; q.ldc = ldc; q.M = M; q.N = N; q.K = K;
  q.epi = EPI_GATE_RES; q.rows_per_batch = M > 0 ? M : 1; q.res = (const uint16_t*)res; q.ldr = ldr;
  q.drop_on = 1; q.drop_thresh = (uint32_t)((double)p * 4294967296.0); q.drop_seed = seed; q.drop_inv_keep = 1.0f / (1.0f - p); q.drop_row0 = row0;
  HIP_TRY(launch_gemm(gb, (hipStream_t)stream));
  return AFX_OK;
}

int64_t afx_attention_ws_bytes(int32_t batch, int32_t heads, int32_t S) {
  if (batch < 1 || heads < 1 || S < 1) return fail(AFX_E_INVALID, "bad attention shape");
  return (int64_t)batch * heads * 128 * attn_spad(S) * 2;
}

int afx_attention_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                       int64_t ldo, void* vt_ws, int32_t batch, int32_t heads, int32_t S, void* stream) {
  if (!q || !k || !v || !o || !vt_ws) return fail(AFX_E_INVALID, "null argument to afx_attention_bf16");
  if (batch < 1 || heads < 1 || S < 1 || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4)
    return fail(AFX_E_INVALID, "bad attention shape / stride");
  HIP_TRY(launch_v_transpose((const uint16_t*)v, ldv, (uint16_t*)vt_ws, batch, heads, S, (hipStream_t)stream));
  HIP_TRY(launch_attention((const uint16_t*)q, ldq, (const uint16_t*)k, ldk, (const uint16_t*)vt_ws, (uint16_t*)o, ldo,
                           batch, heads, S, (hipStream_t)stream));
  return AFX_OK;
}

int64_t afx_attention_bwd_ws_bytes(int32_t batch, int32_t heads, int32_t S) {
  if (batch < 1 || heads < 1 || S < 1) return fail(AFX_E_INVALID, "bad attention shape");
  return attn_bwd_ws_bytes(batch, heads, S);
}

int afx_attention_fwd_lse_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                               int64_t ldo, float* lse, void* vt_ws, int32_t batch, int32_t heads, int32_t S, void* stream) {
  if (!q || !k || !v || !o || !vt_ws || !lse) return fail(AFX_E_INVALID, "null argument to afx_attention_fwd_lse_bf16");
  if (batch < 1 || heads < 1 || S < 1 || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4) return fail(AFX_E_INVALID, "bad attention shape / stride");
  HIP_TRY(launch_v_transpose((const uint16_t*)v, ldv, (uint16_t*)vt_ws, batch, heads, S, (hipStream_t)stream));
  HIP_TRY(launch_attention((const uint16_t*)q, ldq, (const uint16_t*)k, ldk, (const uint16_t*)vt_ws, (uint16_t*)o, ldo,
                           batch, heads, S, (hipStream_t)stream, lse));
  return AFX_OK;
}

int afx_attention_to_mx8(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o8, int64_t ldo8,
                         void* mx, int64_t ld_mx, void* vt_ws, int32_t batch, int32_t heads, int32_t S, void* stream) {
  if (!q || !k || !v || !o8 || !mx || !vt_ws) return fail(AFX_E_INVALID, "null argument to afx_attention_to_mx8");
  if (batch < 1 || heads < 1 || S < 1 || ldq % 8 || ldk % 8 || ldv % 8 || ldo8 % 8 || ld_mx < heads) return fail(AFX_E_INVALID, "bad attention shape / stride");
  if (!attention_v3_eligible(S)) return fail(AFX_E_INVALID, "afx_attention_to_mx8: S > 64 (the one-wave-per-SIMD kernel's epilogue)");
  HIP_TRY(launch_v_transpose((const uint16_t*)v, ldv, (uint16_t*)vt_ws, batch, heads, S, (hipStream_t)stream));
  const AttnMx8 m{(uint8_t*)o8, ldo8, (uint8_t*)mx, ld_mx};
  bool fused = false;
  HIP_TRY(launch_attention((const uint16_t*)q, ldq, (const uint16_t*)k, ldk, (const uint16_t*)vt_ws, nullptr, 0, batch, heads, S, (hipStream_t)stream,
                           nullptr, &m, &fused));
  if (!fused) return fail(AFX_E_INVALID, "afx_attention_to_mx8: the one-wave-per-SIMD attention kernel is switched off (AFX_ATTN_IMPL)");
  return AFX_OK;
}

int afx_attention_bwd_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                           int64_t ldo, const void* dout, int64_t lddo, const float* lse, void* dq, int64_t lddq, void* dk,
                           int64_t lddk, void* dv, int64_t lddv, void* ws, int32_t batch, int32_t heads, int32_t S,
                           void* stream) {
  if (!q || !k || !v || !o || !dout || !lse || !dq || !dk || !dv || !ws) return fail(AFX_E_INVALID, "null argument to afx_attention_bwd_bf16");
  if (batch < 1 || heads < 1 || S < 1 || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8 || lddo % 8 || lddq % 4 || lddk % 4 || lddv % 4)
    return fail(AFX_E_INVALID, "bad attention backward shape / stride");
  HIP_TRY(launch_attention_backward((const uint16_t*)q, ldq, (const uint16_t*)k, ldk, (const uint16_t*)v, ldv,
                                    (const uint16_t*)o, ldo, (const uint16_t*)dout, lddo, lse, (uint16_t*)dq, lddq,
                                    (uint16_t*)dk, lddk, (uint16_t*)dv, lddv, ws, batch, heads, S, (hipStream_t)stream));
  return AFX_OK;
}

int afx_norm_modulate_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int32_t rows, int32_t D,
                           const float* scale, const float* shift, int64_t ldmod, int32_t rows_per_batch, int32_t rms,
                           void* stream) {
  if (!x || !out || !scale || (!rms && !shift)) return fail(AFX_E_INVALID, "null argument to afx_norm_modulate_bf16");
  if (rows < 0 || D < 8 || D % 8 || D > 4096 || ldx % 8 || ldo % 8) return fail(AFX_E_INVALID, "bad norm shape");
  HIP_TRY(launch_norm_modulate((const uint16_t*)x, ldx, (uint16_t*)out, ldo, rows, D, scale, shift, ldmod,
                               rows_per_batch, rms, (hipStream_t)stream));
  return AFX_OK;
}

int afx_qk_norm_rope_bf16(void* x, int64_t ldx, const float* w_txt, const float* w_img, const float* rope_cos,
                          const float* rope_sin, int32_t batch, int32_t S, int32_t n_txt, int32_t heads, void* stream) {
  if (!x || !w_txt || !w_img || !rope_cos || !rope_sin) return fail(AFX_E_INVALID, "null argument to afx_qk_norm_rope_bf16");
  if (batch < 1 || S < 1 || heads < 1 || ldx % 8) return fail(AFX_E_INVALID, "bad qk_norm_rope shape");
  HIP_TRY(launch_qk_norm_rope((uint16_t*)x, ldx, w_txt, w_img, rope_cos, rope_sin, batch, S, n_txt, heads,
                              (hipStream_t)stream));
  return AFX_OK;
}

int afx_gemv_bf16(const float* x, const void* W, const void* bias, float* y, int32_t B, int32_t N, int32_t K,
                  int32_t act, int32_t accumulate, void* stream) {
  if (!x || !W || !y) return fail(AFX_E_INVALID, "null argument to afx_gemv_bf16");
  if (B < 1 || B > 8 || N < 1 || K < 8 || K % 8) return fail(AFX_E_INVALID, "bad gemv shape (B<=8, K%%8==0)");
  HIP_TRY(launch_gemv(x, (const uint16_t*)W, (const uint16_t*)bias, y, B, N, K, act, accumulate, (hipStream_t)stream));
  return AFX_OK;
}

}  // extern "C"
