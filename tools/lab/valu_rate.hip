// How long does one wave take per instruction when it has a SIMD to itself?  (tuning aid)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)
#define R8(x) x x x x x x x x
#define R64(x) R8(R8(x))
template <int K> __global__ void __launch_bounds__(64, 1) rate(int n, int lanes, unsigned* out)
{
  unsigned a = threadIdx.x, b = threadIdx.x * 3, c = 1, d = 2;
  unsigned long long q = threadIdx.x, r = 5;
  if ((int)threadIdx.x < lanes) {
    for (int i = 0; i < n; ++i) {
      if (K == 0) asm volatile(R64("v_mov_b32 %0, %1\n") : "+v"(a) : "v"(b));
      if (K == 1) asm volatile(R64("v_mov_b32 %0, %2\n v_mov_b32 %1, %2\n") : "+v"(a), "+v"(c) : "v"(b));        // 128 independent
      if (K == 2) asm volatile(R64("v_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));                                  // dependent chain
      if (K == 3) asm volatile(R64("v_lshl_add_u64 %0, %0, 0, %1\n") : "+v"(q) : "v"(r));
      if (K == 4) asm volatile(R64("s_nop 0\n"));
      if (K == 5) asm volatile(R64("s_mov_b32 s20, s21\n") ::: "s20");
      if (K == 6) asm volatile(R64("v_mov_b32 %0, %1\n s_mov_b32 s20, s21\n") : "+v"(a) : "v"(b) : "s20");        // 64 + 64 alternating
      if (K == 7) asm volatile(R64("v_xor_b32 %0, %0, %1\n") : "+v"(a) : "v"(b));
      if (K == 8) asm volatile(R64("v_mov_b32 %0, %1\n s_nop 0\n s_nop 0\n s_nop 0\n") : "+v"(a) : "v"(b));
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a + c + d + (unsigned)q;
}
template <int K> static void run(const char* name, int per_iter, int waves, int lanes, unsigned* out)
{
  const int n = 2000; hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(t0)); hipLaunchKernelGGL(rate<K>, dim3(waves), dim3(64), 0, 0, n, lanes, out); CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1)); if (ms < best) best = ms; }
  printf("%-28s waves %5d lanes %2d: %.2f ns per instruction\n", name, waves, lanes, best * 1e6 / ((double)n * per_iter));
}
int main()
{
  unsigned* out; CK(hipMalloc(&out, 8192 * 64 * 4));
  for (int lanes : {1, 4, 15, 16, 17, 32, 33, 48, 49, 63, 64}) { run<0>("v_mov same dst", 64, 1024, lanes, out); run<3>("v_lshl_add_u64 chain", 64, 1024, lanes, out); }
  for (int lanes : {4, 64}) { run<0>("v_mov same dst", 64, 2048, lanes, out); run<8>("v_mov + 3 s_nop", 256, 1024, lanes, out); }
  return 0;
}
