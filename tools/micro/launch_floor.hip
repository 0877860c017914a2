// Micro-benchmark: cost of a dependent kernel boundary on one stream, launched directly vs replayed from a HIP graph.
// hipcc --offload-arch=gfx950 -O2 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void tiny(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

int main() {
  const int n = 1 << 16, chain = 40, reps = 200;
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemset(d, 0, n * sizeof(float)));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int grid : {1, 256, 1024}) {
    // direct launches
    for (int w = 0; w < 3; ++w) {
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r)
        for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, n);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("grid %4d: direct  %.2f us per kernel\n", grid, ms * 1e3 / (reps * chain));
    // graph of the same chain
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, n);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) {
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("grid %4d: graph   %.2f us per kernel\n", grid, ms * 1e3 / (reps * chain));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
