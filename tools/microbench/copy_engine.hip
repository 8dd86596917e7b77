// tools/microbench/copy_engine.hip -- which engine moves a pinned <-> device hipMemcpyAsync on this runtime: SDMA or the
// __amd_rocclr_copyBuffer blit kernel (which occupies compute units)?  Run with the runtime's copy log on:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/copy_engine tools/microbench/copy_engine.hip
//   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ce -- /tmp/copy_engine 2> /tmp/ce.cases
//   python tools/microbench/copy_engine_summary.py /tmp/ce /tmp/ce.cases
// (the release runtime has no copy log.)  Every case starts with a MARKER kernel k_marker<<<case index + 1, 64>>> and prints
// "CASE <index> <name>" to stderr; the summary cuts both traces at the markers and counts, per case, rows of the memory-copy trace
// (SDMA) and __amd_rocclr_copyBuffer launches (blit kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_spin(double* p, int n, int iters)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double v = i < n ? p[i] : 0.0;
  for (int k = 0; k < iters; ++k) v = fma(v, 1.0000001, 1e-9);
  if (i < n) p[i] = v;
}

__global__ void k_marker(int* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] = 1; }
static int g_case = 0;
static void mark(const char* name)
{
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_marker, dim3(++g_case), dim3(64), 0, 0, (int*)nullptr);
  CK(hipDeviceSynchronize());
  fprintf(stderr, "CASE %d %s\n", g_case, name);
}

int main()
{
  const size_t MAXB = size_t(64) << 20;
  char *h = nullptr, *d = nullptr; double* w = nullptr;
  CK(hipHostMalloc((void**)&h, MAXB, hipHostMallocDefault));
  CK(hipMalloc((void**)&d, MAXB));
  CK(hipMalloc((void**)&w, sizeof(double) << 20));
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const size_t sizes[] = {1024, 16 << 10, 64 << 10, 1 << 20, 6 << 20, size_t(60) << 20};
  for (int dir = 0; dir < 2; ++dir)
    for (size_t nb : sizes) {
      for (int busy = 0; busy < 3; ++busy) {
        // busy 0: idle stream; 1: a kernel in front of the copy on the SAME stream; 2: a long kernel running on ANOTHER stream
        char nm[128];
        snprintf(nm, sizeof(nm), "%s %zu %s", dir ? "D2H" : "H2D", nb, busy == 0 ? "idle" : busy == 1 ? "after-kernel" : "other-stream-busy");
        mark(nm);
        for (int rep = 0; rep < 4; ++rep) {
          if (busy == 1) hipLaunchKernelGGL(k_spin, dim3(4096), dim3(256), 0, s, w, 1 << 20, 2000);
          if (busy == 2) hipLaunchKernelGGL(k_spin, dim3(4096), dim3(256), 0, s2, w, 1 << 20, 20000);
          if (dir) CK(hipMemcpyAsync(h, d, nb, hipMemcpyDeviceToHost, s));
          else CK(hipMemcpyAsync(d, h, nb, hipMemcpyHostToDevice, s));
          CK(hipStreamSynchronize(s));
        }
      }
    }
  // two copies back to back in opposite directions (both engines), then several H2D copies in flight on different streams
  mark("both-directions 6291456 two-streams");
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemcpyAsync(d, h, 6 << 20, hipMemcpyHostToDevice, s));
    CK(hipMemcpyAsync(h + (32 << 20), d + (32 << 20), 6 << 20, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
  }
  mark("H2D 6291456 while-a-48MB-H2D-runs-on-another-stream");
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemcpyAsync(d + (8 << 20), h + (8 << 20), size_t(48) << 20, hipMemcpyHostToDevice, s2));
    CK(hipMemcpyAsync(d, h, 6 << 20, hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
  }
  mark("D2H 6291456 while-a-48MB-H2D-runs-on-another-stream");
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemcpyAsync(d + (8 << 20), h + (8 << 20), size_t(48) << 20, hipMemcpyHostToDevice, s2));
    CK(hipMemcpyAsync(h, d, 6 << 20, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
  }
  mark("end");
  return 0;
}
