// Philox4x32-10 counter-based random numbers (Salmon et al. 2011), shared by the count sampler (sampler.hip)
// and the step's own noise kernel (elementwise.hip).  A draw is a pure function of (key, counter): it does
// not depend on the launch geometry, and two ranks with the same key and counter draw the same numbers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cradle {

struct Philox {
  uint32_t k0, k1;
  __device__ static inline void round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  __device__ inline uint4 operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) const {
    uint32_t a = k0, b = k1;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      round(c0, c1, c2, c3, a, b);
      a += 0x9E3779B9u;
      b += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
  }
};

__device__ __forceinline__ float u01(uint32_t r) { return ((r >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

}  // namespace cradle
