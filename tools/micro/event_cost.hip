// Micro-benchmark: what an event record (a fork to a side stream) between two dependent kernels costs the
// stream it is recorded on.  hipcc --offload-arch=gfx950 -O2 -o event_cost event_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void spin(float* p, long long cycles) {  // keeps the GPU busy while the host enqueues what follows
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (threadIdx.x == 0) p[0] += 1.0f;
}

__global__ void tiny(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

int main(int argc, char** argv) {
  const unsigned flags = argc > 1 ? strtoul(argv[1], nullptr, 0) : 0;
  const int n = 1 << 16, chain = 40, reps = 20;
  float *d, *d2;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMalloc(&d2, n * sizeof(float)));
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, ev[chain];
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int k = 0; k < chain; ++k) CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming | flags));
  float ms;
  hipEvent_t old;
  CK(hipEventCreateWithFlags(&old, hipEventDisableTiming | flags));
  hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s2, d2, n);
  CK(hipEventRecord(old, s2));
  CK(hipStreamSynchronize(s2));
  for (int mode = 0; mode < 5; ++mode) {
    for (int w = 0; w < 3; ++w) {
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d2 + 1024, 3000000LL);  // 100 MHz clock: 30 ms
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r)
        for (int k = 0; k < chain; ++k) {
          hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d, n);
          if (mode == 4) { CK(hipStreamWaitEvent(s, old, 0)); continue; }
          if (mode >= 1) CK(hipEventRecord(ev[k], s));
          if (mode >= 2) CK(hipStreamWaitEvent(s2, ev[k], 0));
          if (mode >= 3) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s2, d2, n);
        }
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      CK(hipStreamSynchronize(s2));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    const char* what[] = {"kernels only", "+ event record after each", "+ side stream waits on it", "+ side stream runs a kernel", "kernels + wait on a long-complete event"};
    printf("flags 0x%x  %-32s %.2f us per kernel on the recording stream\n", flags, what[mode], ms * 1e3 / (reps * chain));
  }
  return 0;
}
