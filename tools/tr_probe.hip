// What does ds_read_b64_tr_b16 return?  LDS holds a [64][128] u16 image with value = row * 256 + col (row stride 256 B).  Every lane passes the address
//   base + (row0 + 4 * (l >> 4) * 2 ... see below) and the host prints, per lane, the (row, col) of the four 16-bit values it got.
//   hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o tools/bin/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>


__global__ void probe2(unsigned* out, int mode) {
  __shared__ unsigned short img[64 * 128];
  for (int i = threadIdx.x; i < 64 * 128; i += 64) img[i] = (unsigned short)((i / 128) * 256 + (i % 128));
  __syncthreads();
  const int l = threadIdx.x;
  const int g = l >> 4, i = l & 15;
  unsigned addr;
  if (mode == 0) addr = (unsigned)(uintptr_t)img + ((8 * g + i / 4) * 128 + 4 * (i % 4)) * 2;
  else addr = (unsigned)(uintptr_t)img + ((8 * g + 4 + i % 4) * 128 + 32 + 4 * (i / 4)) * 2;
  typedef unsigned __attribute__((ext_vector_type(2))) u2;
  u2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  out[2 * l] = v[0];
  out[2 * l + 1] = v[1];
}

int main() {
  unsigned* d;
  hipMalloc(&d, 128 * sizeof(unsigned));
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe2, dim3(1), dim3(64), 0, 0, d, mode);
    std::vector<unsigned> h(128);
    hipMemcpy(h.data(), d, 128 * sizeof(unsigned), hipMemcpyDeviceToHost);
    printf("mode %d: lane: (row,col) x 4\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("%2d:", l);
      for (int e = 0; e < 4; ++e) {
        const unsigned v = (h[2 * l + e / 2] >> (16 * (e & 1))) & 0xffff;
        printf(" (%2u,%3u)", v >> 8, v & 255);
      }
      printf("%s", (l % 2) ? "\n" : "   ");
    }
  }
  return 0;
}
