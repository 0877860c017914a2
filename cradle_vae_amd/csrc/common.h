// Shared helpers for the CRADLE-VAE HIP kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// the public C ABI: status codes, descriptor structs and the prototypes every definition is checked against
#include "../../include/cradle_hip.h"

// thread-local last-error string, see api.hip
void cradle_set_error(const char* fmt, ...);

#define CRADLE_CHECK_ARG(cond, ...)          \
  do {                                       \
    if (!(cond)) {                           \
      cradle_set_error(__VA_ARGS__);         \
      return CRADLE_ERR_ARG;                 \
    }                                        \
  } while (0)

#define CRADLE_CHECK_LAUNCH(name)                                              \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      cradle_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return CRADLE_ERR_HIP;                                                   \
    }                                                                          \
  } while (0)

namespace cradle {

constexpr int kWave = 64;

// Wave-wide reductions on the DPP data path (six dependent VALU operations and one v_readlane) instead of six
// ds_bpermute round trips through the LDS pipe: a row kernel does four of these per particle row, and with two
// waves per SIMD nothing hides a shuffle's latency.  Every lane gets the result.
//   quad_perm [1,0,3,2] -> quad_perm [2,3,0,1] -> row_half_mirror -> row_mirror: every lane holds its row's (16 lanes) value;
//   row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3: lane 63 holds the wave's.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                               ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1, 0xF>(0.f, v);
  v += dpp_move<0x4E, 0xF>(0.f, v);
  v += dpp_move<0x141, 0xF>(0.f, v);
  v += dpp_move<0x140, 0xF>(0.f, v);
  v += dpp_move<0x142, 0xA>(0.f, v);
  v += dpp_move<0x143, 0xC>(0.f, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xB1, 0xF>(v, v));
  v = fmaxf(v, dpp_move<0x4E, 0xF>(v, v));
  v = fmaxf(v, dpp_move<0x141, 0xF>(v, v));
  v = fmaxf(v, dpp_move<0x140, 0xF>(v, v));
  v = fmaxf(v, dpp_move<0x142, 0xA>(v, v));
  v = fmaxf(v, dpp_move<0x143, 0xC>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Block-wide reductions over NT threads (NT multiple of 64, <= 1024).  `slot` points at
// NT/64 floats of LDS reserved for this call; the caller alternates slots so that one
// __syncthreads() per reduction is enough.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* slot) {
  constexpr int NW = NT / 64;
  v = wave_sum(v);
  if constexpr (NW == 1) return v;
  if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) t += slot[w];
  return t;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* slot) {
  constexpr int NW = NT / 64;
  v = wave_max(v);
  if constexpr (NW == 1) return v;
  if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = slot[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) t = fmaxf(t, slot[w]);
  return t;
}
template <int NT>
__device__ __forceinline__ float2 block_sum2(float a, float b, float2* slot) {
  constexpr int NW = NT / 64;
  a = wave_sum(a);
  b = wave_sum(b);
  if constexpr (NW == 1) return make_float2(a, b);
  if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = make_float2(a, b);
  __syncthreads();
  float2 t = make_float2(0.f, 0.f);
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    t.x += slot[w].x;
    t.y += slot[w].y;
  }
  return t;
}

// torch.nan_to_num(v, nan=1e-6, posinf=1e6, neginf=1e-6): the mapping the reference applies
// in mlp.py:138-139 and normalization.py:44 (note neginf -> +1e-6).
__device__ __forceinline__ float nan_to_num_ref(float v) {
  if (v != v) return 1e-6f;
  if (v == INFINITY) return 1e6f;
  if (v == -INFINITY) return 1e-6f;
  return v;
}

constexpr float kLeakySlope = 0.01f;  // torch.nn.LeakyReLU default (mlp.py:19)

}  // namespace cradle
