// K10: global-norm gradient clip + Adam in two passes (two launches) over the flat parameter arena.
//
// Reference: Lightning's `gradient_clip_val=100` (train_norman.py:96 ->
// torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (norm + 1e-6))) followed by
// torch.optim.Adam(lr, weight_decay=1e-6) with L2 (coupled) weight decay
// (cradle_vae/models/utils/perturbation_lightning_module.py:147-150).
// All parameters, gradients and both moments live in flat fp32 arenas, so the norm is one
// reduction and the update one streaming kernel (4 reads + 3 writes per parameter).
#include "common.h"
#include "kernels.h"

namespace cradle {

constexpr int kNormBlocks = 512;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = g4[i];
    acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += g[i] * g[i];
  acc = block_sum<256>(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float sqrt_bc2, float max_norm, float grad_scale,
                                                   float* __restrict__ ws, const int* __restrict__ skip_flags) {
  // the step that produced these gradients rejected its input (cradle_step_desc.flags != 0: a cell with more
  // than kmax dosages, qc outside {0,1}): its gradients are wrong, so parameters and moments stay as they are
  if (skip_flags && *skip_flags != 0) return;
  // every workgroup finishes the norm from the kNormBlocks partial sums itself (same order everywhere:
  // same value), which saves a single-workgroup launch between the two passes
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < kNormBlocks; i += 256) acc += ws[i];
  const float norm = sqrtf(block_sum<256>(acc, red)) * fabsf(grad_scale);  // norm of the scaled gradient
  if (blockIdx.x == 0 && threadIdx.x == 0) ws[kNormBlocks] = norm;
  float scale = grad_scale;
  if (max_norm > 0.f) scale *= fminf(1.0f, max_norm / (norm + 1e-6f));
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float pi = p[i];
    const float gi = fmaf(wd, pi, g[i] * scale);
    const float mi = fmaf(b1, m[i], (1.0f - b1) * gi);
    const float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step_size * mi / (sqrtf(vi) / sqrt_bc2 + eps);
  }
}

int adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
              float eps, float weight_decay, int step, float max_norm, float grad_scale, float* norm_ws,
              const int* skip_flags, hipStream_t stream) {
  CRADLE_CHECK_ARG(params && grads && m && v && norm_ws && n > 0 && step >= 1, "adam_step: bad arguments");
  CRADLE_CHECK_ARG((uintptr_t)grads % 16 == 0, "adam_step: grads must be 16-byte aligned");
  hipLaunchKernelGGL(sumsq_kernel, dim3(kNormBlocks), dim3(256), 0, stream, grads, n, norm_ws);
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float sqrt_bc2 = sqrtf(1.0f - powf(beta2, (float)step));
  const int blocks = (int)min((int64_t)2048, (n + 255) / 256);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, params, grads, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, sqrt_bc2, max_norm, grad_scale, norm_ws, skip_flags);
  CRADLE_CHECK_LAUNCH("adam_step");
  return CRADLE_OK;
}

}  // namespace cradle

extern "C" {

// norm_ws: cradle_adam_workspace_floats() floats; after the call its last element holds the
// (scaled, pre-clip) global gradient norm.  skip_flags (device int, may be NULL): non-zero = no update.
int cradle_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step, float max_norm,
                     float grad_scale, float* norm_ws, const int* skip_flags, hipStream_t stream) {
  return cradle::adam_step(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, max_norm,
                           grad_scale, norm_ws, skip_flags, stream);
}

int64_t cradle_adam_workspace_floats(void) { return cradle::kNormBlocks + 1; }

}  // extern "C"
