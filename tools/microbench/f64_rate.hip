// f64 / f32 / int VALU issue rates on gfx950, one wave per SIMD up to 8: clock64() cycles per instruction per wave.
// hipcc --offload-arch=gfx950 -O3 -o f64_rate f64_rate.hip && ./f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void k(double* out, long long* cyc, int iters)
{
  double a[8]; float f[8]; int n[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3 + i; f[i] = (float)a[i]; n[i] = threadIdx.x + i; }
  const double m = 1.0000001, c = 1e-9;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) a[i] = fma(a[i], m, c);
        if (MODE == 1) a[i] = a[i] * m;
        if (MODE == 2) a[i] = a[i] + c;
        if (MODE == 3) f[i] = fmaf(f[i], 1.0000001f, 1e-9f);
        if (MODE == 4) n[i] = n[i] * 3 + 1;
        if (MODE == 5) n[i] = (n[i] << 1) ^ 0x55;
      }
  }
  long long t1 = clock64();
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + f[i] + n[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int wg, int threads)
{
  double* out; long long* cyc; hipMalloc(&out, sizeof(double) * wg * threads); hipMalloc(&cyc, 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<wg, threads>>>(out, cyc, 10);
  hipEventRecord(e0); k<MODE><<<wg, threads>>>(out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double ninst = (double)iters * 32;
  const double lane_ops = ninst * wg * threads;
  printf("%-10s wg=%4d x %4d thr: %.2f clock64-ticks per wave-instr (WG 0), %.1f G lane-ops/s chip-wide (%.1f TFLOP/s if 2 flop)\n", name, wg, threads,
         h / ninst, lane_ops / (ms * 1e-3) / 1e9, 2 * lane_ops / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}
int main()
{
  for (int thr : {64, 256, 512, 1024}) {
    run<0>("fma_f64", 256, thr); run<1>("mul_f64", 256, thr); run<2>("add_f64", 256, thr);
    run<3>("fma_f32", 256, thr); run<4>("mad_i32", 256, thr); run<5>("shl_xor", 256, thr);
  }
  run<0>("fma_f64", 2048, 1024); run<3>("fma_f32", 2048, 1024);
  return 0;
}
