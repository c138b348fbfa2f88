// Which CU does a work-group of a one-work-group-per-CU kernel land on, and when?  Every work-group declares 128 KiB of LDS (so one fits a CU), spins for
// a duration taken from a table, and records (XCC id, HW_ID, start, end in the 100 MHz realtime counter).  The host prints, per XCC, the items in
// dispatch order with the (SE, CU) they ran on and their start / end times -- the dispatch policy the balanced attention / stream-K schedules rely on.
//   hipcc --offload-arch=gfx950 -O2 tools/dispatch_probe.hip -o tools/bin/dispatch_probe;  tools/bin/dispatch_probe [pattern]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

struct Rec { unsigned xcc, hwid; unsigned long long t0, t1; };

__global__ __launch_bounds__(256) void probe(const int* dur_us, Rec* rec) {
  extern __shared__ char smem[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) smem[0] = 1;
  const unsigned long long ticks = (unsigned long long)dur_us[blockIdx.x] * 100ull;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    rec[blockIdx.x] = Rec{xcc & 0xf, hwid, t0, __builtin_amdgcn_s_memrealtime()};
  }
}

int main(int argc, char** argv) {
  const int pattern = argc > 1 ? atoi(argv[1]) : 0;
  // per XCC list (32 CUs): pattern 0: 22 x 110 us, 10 x 80 us, 10 x 110 us, 22 x 80 us (the balanced attention schedule); 1: 54 x 110 us (plain)
  std::vector<int> per;
  if (pattern == 0) {
    for (int i = 0; i < 22; ++i) per.push_back(110);
    for (int i = 0; i < 10; ++i) per.push_back(80);
    for (int i = 0; i < 10; ++i) per.push_back(110);
    for (int i = 0; i < 22; ++i) per.push_back(80);
  } else if (pattern == 1) {
    for (int i = 0; i < 54; ++i) per.push_back(110);
  } else {           // 2: 32 long, then 64 short: does a freed CU take the next item at once?
    for (int i = 0; i < 32; ++i) per.push_back(40 + 5 * (i % 8));
    for (int i = 0; i < 64; ++i) per.push_back(30);
  }
  const int n = (int)per.size() * 8;
  std::vector<int> dur(n);
  for (int i = 0; i < n; ++i) dur[i] = per[i / 8];
  int* d_dur; Rec* d_rec;
  hipMalloc(&d_dur, n * sizeof(int)); hipMalloc(&d_rec, n * sizeof(Rec));
  hipMemcpy(d_dur, dur.data(), n * sizeof(int), hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  std::vector<Rec> rec(n);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 131072, 0, d_dur, d_rec);
    hipDeviceSynchronize();
  }
  hipMemcpy(rec.data(), d_rec, n * sizeof(Rec), hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull, tmax = 0;
  for (auto& r : rec) { tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); }
  printf("pattern %d: %d work-groups, makespan %.1f us\n", pattern, n, (tmax - tmin) / 100.0);
  int mism = 0;
  for (int i = 0; i < n; ++i) mism += (int)(rec[i].xcc != (unsigned)(i % 8));
  printf("work-groups whose XCC id != blockIdx %% 8: %d\n", mism);
  for (int x = 0; x < 2; ++x) {
    printf("== XCC %d: slot dur | se sh cu | start end (us)\n", x);
    for (int i = x; i < n; i += 8) {
      const Rec& r = rec[i];
      printf("%3d %4d | %u %u %2u | %7.1f %7.1f\n", i / 8, dur[i], (r.hwid >> 13) & 7, (r.hwid >> 12) & 1, (r.hwid >> 8) & 15, (r.t0 - tmin) / 100.0, (r.t1 - tmin) / 100.0);
    }
  }
  return 0;
}
