// Cycle-stamp trace of one K-iteration of the 8-phase GEMM (block 0 and block 300, all 8 waves).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAFX_GEMM_TRACE=1|2 -I arcflow_amd/csrc tools/gemm_trace.hip -o gpurun_out/gemm_trace
// Stamps per phase q (4 each): after barrier 1 | after lgkmcnt(0) | after the 16 MFMAs issued | after barrier 2; stamp 16 = loop top.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../arcflow_amd/csrc/afx_gemm.hip"

// Pure MFMA, no memory: the clock and rate the power cap leaves to a kernel that does nothing but 16x16x32 bf16 MFMAs.
typedef __attribute__((ext_vector_type(8))) __bf16 burn_bf16x8;
typedef __attribute__((ext_vector_type(4))) float burn_f32x4;
__global__ void mfma_burn(float* out, unsigned* stamps, int iters) {
  burn_f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (burn_f32x4){0.f, 0.f, 0.f, 0.f};
  burn_bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
  const unsigned c0 = (unsigned)__builtin_readcyclecounter(), r0 = (unsigned)__builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  const unsigned c1 = (unsigned)__builtin_readcyclecounter(), r1 = (unsigned)__builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
}

// Same, but 8 x 8 different random operand registers (toggling as in a real GEMM) -- and optionally an LDS read stream beside it.
template <int LDS_READS>
__global__ void mfma_burn_rand(float* out, unsigned* stamps, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  burn_f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (burn_f32x4){0.f, 0.f, 0.f, 0.f};
  burn_bf16x8 a[8], b[8];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int r = 0; r < 8; ++r)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u;
      a[r][i] = __builtin_bit_cast(__bf16, (unsigned short)(0x3c00 + ((h >> 8) & 0x3ff) - (((h >> 20) & 1) << 15)));
      h = h * 1664525u + 1013904223u;
      b[r][i] = __builtin_bit_cast(__bf16, (unsigned short)(0x3c00 + ((h >> 8) & 0x3ff) - (((h >> 20) & 1) << 15)));
    }
  for (int i = threadIdx.x; i < 65536 / 16; i += blockDim.x) {
    h = h * 1664525u + 1013904223u;
    reinterpret_cast<uint4*>(lds)[i] = make_uint4(h, h * 3u, h * 5u, h * 7u);
  }
  __syncthreads();
  const unsigned c0 = (unsigned)__builtin_readcyclecounter(), r0 = (unsigned)__builtin_amdgcn_s_memrealtime();
  unsigned la = (threadIdx.x * 16) & 65535;
  uint4 sink = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[(i + r) & 7], acc[i], 0, 0, 0);
      if (r < LDS_READS) {
        const uint4 v = *reinterpret_cast<const uint4*>(lds + la);
        sink.x ^= v.x; sink.y ^= v.y; sink.z ^= v.z; sink.w ^= v.w;
        la = (la + 4096 + 16) & 65535 & ~15u;
      }
    }
  }
  const unsigned c1 = (unsigned)__builtin_readcyclecounter(), r1 = (unsigned)__builtin_amdgcn_s_memrealtime();
  float s = __uint_as_float(sink.x ^ sink.y ^ sink.z ^ sink.w) * 1e-30f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
}

template <int LDS_READS>
static void burn_rand(int threads) {
  float* out; unsigned* st;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&st, 8);
  const int iters = 60000;                        // 64 MFMAs per iteration and wave
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_burn_rand<LDS_READS><<<256, threads>>>(out, st, 100);
  hipEventRecord(e0, 0);
  mfma_burn_rand<LDS_READS><<<256, threads>>>(out, st, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned h[2]; hipMemcpy(h, st, 8, hipMemcpyDeviceToHost);
  const double flops = 256.0 * (threads / 64) * iters * 64.0 * 2 * 16 * 16 * 32;
  printf("mfma_burn_rand %d waves/CU, %d ds_read_b128 per 64 MFMAs: %.2f ms  %.0f TF   shader clock %.1f MHz; %.2f cycles per MFMA per wave\n", threads / 64, LDS_READS, ms,
         flops / (ms * 1e-3) / 1e12, 100.0 * h[0] / h[1], (double)h[0] / (iters * 64.0));
  hipFree(out); hipFree(st);
}

typedef __attribute__((ext_vector_type(16))) float burn_f32x16;
__global__ void mfma_burn_rand32(float* out, unsigned* stamps, int iters) {
  burn_f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  burn_bf16x8 a[8], b[8];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int r = 0; r < 8; ++r)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u;
      a[r][i] = __builtin_bit_cast(__bf16, (unsigned short)(0x3c00 + ((h >> 8) & 0x3ff) - (((h >> 20) & 1) << 15)));
      h = h * 1664525u + 1013904223u;
      b[r][i] = __builtin_bit_cast(__bf16, (unsigned short)(0x3c00 + ((h >> 8) & 0x3ff) - (((h >> 20) & 1) << 15)));
    }
  const unsigned c0 = (unsigned)__builtin_readcyclecounter(), r0 = (unsigned)__builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * i + (r & 1)], b[(i + r) & 7], acc[i], 0, 0, 0);
  }
  const unsigned c1 = (unsigned)__builtin_readcyclecounter(), r1 = (unsigned)__builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
}
static void burn_rand32(int threads) {
  float* out; unsigned* st;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&st, 8);
  const int iters = 60000;                        // 32 MFMAs (32x32x16) per iteration and wave
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_burn_rand32<<<256, threads>>>(out, st, 100);
  hipEventRecord(e0, 0);
  mfma_burn_rand32<<<256, threads>>>(out, st, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned h[2]; hipMemcpy(h, st, 8, hipMemcpyDeviceToHost);
  const double flops = 256.0 * (threads / 64) * iters * 32.0 * 2 * 32 * 32 * 16;
  printf("mfma_burn_rand32 (32x32x16) %d waves/CU: %.2f ms  %.0f TF   shader clock %.1f MHz; %.2f cycles per MFMA per wave\n", threads / 64, ms,
         flops / (ms * 1e-3) / 1e12, 100.0 * h[0] / h[1], (double)h[0] / (iters * 32.0));
  hipFree(out); hipFree(st);
}

static void burn(int threads) {
  float* out; unsigned* st;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&st, 8);
  const int iters = 200000;                       // 1.6 M MFMAs per wave: ~12 ms
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_burn<<<256, threads>>>(out, st, 1000);
  hipEventRecord(e0, 0);
  mfma_burn<<<256, threads>>>(out, st, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned h[2]; hipMemcpy(h, st, 8, hipMemcpyDeviceToHost);
  const double flops = 256.0 * (threads / 64) * iters * 8.0 * 2 * 16 * 16 * 32;
  printf("mfma_burn %d waves/CU: %.2f ms  %.0f TF   memtime %u  realtime %u  -> memtime runs at %.1f MHz; %.2f memtime ticks per MFMA per wave\n", threads / 64, ms,
         flops / (ms * 1e-3) / 1e12, h[0], h[1], 100.0 * h[0] / h[1], (double)h[0] / (iters * 8.0));
  hipFree(out); hipFree(st);
}

int main(int argc, char** argv) {
  if (argc <= 3) {
  burn(256); burn(512); burn(1024);
  burn_rand<0>(256); burn_rand<0>(512); burn_rand<8>(256); burn_rand<8>(512); burn_rand32(256); burn_rand32(512);
  if (argc > 2) return 0;
  }
  const int M = getenv("TRACE_M") ? atoi(getenv("TRACE_M")) : 4608, N = argc > 1 ? atoi(argv[1]) : 9216, K = getenv("TRACE_K") ? atoi(getenv("TRACE_K")) : 3072;
  std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
  srand(1);
  for (auto& v : ha) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);   // bf16 in +-[0.0078, 0.0156): random mantissas
  for (auto& v : hw) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);
  uint16_t *a, *w, *c;
  // TRACE_COLD=1: rotate through enough copies of W (and C) that no launch finds its weights in the 256 MB Infinity Cache -- the
  // situation of the denoiser, where every layer's weights are read once per forward
  const size_t wbytes = hw.size() * 2;
  const int ncopy = getenv("TRACE_COLD") ? (int)((768ull << 20) / wbytes) + 2 : 1;
  hipMalloc(&a, ha.size() * 2); hipMalloc(&w, wbytes * ncopy); hipMalloc(&c, (size_t)M * N * 2);
  for (int i = 1; i < ncopy; ++i) hipMemcpy((char*)w + wbytes * i, hw.data(), wbytes, hipMemcpyHostToDevice);
  hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  afx::GemmBatch b{};
  b.nprob = 1;
  afx::GemmProblem& p = b.p[0];
  p.A = a; p.W = w; p.C = c; p.lda = K; p.ldw = K; p.ldc = N; p.M = M; p.N = N; p.K = K; p.rows_per_batch = M;
  if (getenv("TRACE_NOSTORE")) p.gelu_col0 = -12345;
  if (getenv("TRACE_EPI")) { p.epi = atoi(getenv("TRACE_EPI")); p.gelu_col0 = 0; p.res = c; p.ldr = N; }   // 1: GELU, 2: residual add in place
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) afx::launch_gemm(b, 0);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 200; ++i) {                           // ~50 ms: the clock governor has settled
    b.p[0].W = (const uint16_t*)((const char*)w + wbytes * (i % ncopy));
    afx::launch_gemm(b, 0);
  }
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("M=%d N=%d K=%d  %.1f us  %.0f TF (with stamps)\n", M, N, K, ms * 5, 2.0 * M * N * K / (ms * 5e-6) / 1e12);
  unsigned tr[2][8][32];
  hipMemcpyFromSymbol(tr, HIP_SYMBOL(afx::g_gemm_trace), sizeof(tr));
  for (int blk = 0; blk < 2; ++blk) {
    printf("block %d: per phase [load+vmcnt+bar1 | lgkm | mfma issue | bar2]\n", blk ? 300 : 0);
    for (int wv = 0; wv < 8; ++wv) {
      unsigned* t = tr[blk][wv];
      unsigned prev = t[16];
      if (AFX_GEMM_TRACE != 1) goto coarse;
      if (!getenv("AFX_GEMM_IMPL") || getenv("AFX_GEMM_IMPL")[0] == '3') {      // v3: k-half 0 | DMA / LDS wait | barrier | k-half 1
        if (wv < 4) printf(" w%d: k-half0 %4u  wait %4u  barrier %4u  k-half1 %4u  iter %u\n", wv, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[4] - t[0]);
        goto coarse;
      }
      printf(" w%d:", wv);
      for (int q = 0; q < 4; ++q) {
        printf("  %4u %4u %4u %4u |", t[q * 4] - prev, t[q * 4 + 1] - t[q * 4], t[q * 4 + 2] - t[q * 4 + 1], t[q * 4 + 3] - t[q * 4 + 2]);
        prev = t[q * 4 + 3];
      }
      printf("  iter %u\n", t[15] - t[16]);
    coarse:
      printf("      w%d kernel: prologue %u  loop %u (%u / K-tile)  drain %u  epilogue %u  total %u\n", wv, t[18] - t[17], t[19] - t[18],
             (t[19] - t[18]) / (K / 64), t[20] - t[19], t[21] - t[20], t[21] - t[17]);
      if (wv == 0) printf("      memtime / realtime over the kernel body: %u / %u -> %.1f MHz\n", t[21] - t[17], t[23] - t[22], 100.0 * (t[21] - t[17]) / (t[23] - t[22]));
    }
  }
  return 0;
}
