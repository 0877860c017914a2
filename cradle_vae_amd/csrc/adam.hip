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

// ---- region-wise form (data-parallel training with sharded optimiser work) ---------------------------
// The arena is cut into regions (gradient bucket x rank).  The squared norm of the gradient is DEFINED as the
// sum, in region order, of per-region sums S_i, each computed by the same two kernels below whoever computes it
// -- the rank that owns region i after a reduce-scatter, or every rank over all regions after an all-reduce --
// so that both ways of synchronising gradients give bit-identical clip coefficients.
constexpr int kMaxRegions = 64;
struct Regions {
  int64_t off[kMaxRegions];
  int64_t len[kMaxRegions];
  int n;
  int blocks;  // partial-sum workgroups per region
};

__global__ __launch_bounds__(256) void region_sumsq_kernel(const float* __restrict__ g, Regions r, float* __restrict__ part) {
  __shared__ float red[4];
  const int i = blockIdx.y;
  const float* src = g + r.off[i];
  const int64_t n = r.len[i], n4 = n >> 2;  // regions start 16-byte aligned (checked on the host)
  const float4* g4 = reinterpret_cast<const float4*>(src);
  float acc = 0.f;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += (int64_t)gridDim.x * 256) {
    const float4 v = g4[k];
    acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0)
    for (int64_t k = (n4 << 2) + threadIdx.x; k < n; k += 256) acc += src[k] * src[k];
  acc = block_sum<256>(acc, red);
  if (threadIdx.x == 0) part[(int64_t)i * gridDim.x + blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void region_finish_kernel(const float* __restrict__ part, int blocks, float* __restrict__ sums) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int k = threadIdx.x; k < blocks; k += 256) acc += part[(int64_t)blockIdx.x * blocks + k];
  acc = block_sum<256>(acc, red);
  if (threadIdx.x == 0) sums[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void adam_regions_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, Regions r, float lr, float b1, float b2,
                                                           float eps, float wd, float bc1, float sqrt_bc2, float max_norm,
                                                           float grad_scale, const float* __restrict__ all_sums, int n_sums,
                                                           float* __restrict__ norm_out, const int* __restrict__ skip_flags) {
  float tot = 0.f;
  for (int k = 0; k < n_sums; ++k) tot += all_sums[k];  // fixed order, the same on every rank
  const float norm = sqrtf(tot) * fabsf(grad_scale);
  const bool first = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
  // A rank whose step rejected its input puts a NaN into its gradients before they are summed
  // (cradle_grad_poison_if_flagged): the sum of squares every rank sees is then not finite and ALL ranks skip the
  // update together -- the decision is global although the flags word is local.  norm_out[1] counts skipped updates.
  if ((skip_flags && *skip_flags != 0) || !(fabsf(tot) <= 3.0e38f)) {
    if (first && norm_out) { norm_out[0] = norm; norm_out[1] += 1.0f; }
    return;
  }
  if (first && norm_out) norm_out[0] = norm;
  float scale = grad_scale;
  if (max_norm > 0.f) scale *= fminf(1.0f, max_norm / (norm + 1e-6f));
  const float step_size = lr / bc1;
  const int64_t o = r.off[blockIdx.y], n = r.len[blockIdx.y];
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
    const int64_t i = o + k;
    const float pi = p[i];
    const float gi = fmaf(wd, pi, g[i] * scale);
    const float mi = fmaf(b1, m[i], (1.0f - b1) * gi);
    const float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step_size * mi / (sqrtf(vi) / sqrt_bc2 + eps);
  }
}

__global__ void poison_if_flagged_kernel(float* __restrict__ element, const int* __restrict__ flags) {
  if (*flags != 0) *element = __builtin_nanf("");
}

static int fill_regions(Regions& r, const int64_t* off, const int64_t* len, int n, int n_total, const char* what) {
  CRADLE_CHECK_ARG(off && len && n >= 1 && n <= kMaxRegions && n_total >= n && n_total <= kMaxRegions,
                   "%s: 1..%d regions (got %d of %d)", what, kMaxRegions, n, n_total);
  r.n = n;
  // partial workgroups per region: a function of the TOTAL region count only, so that the owner of one region
  // and a rank that sums all of them add its elements in the same order
  int b = 512 / n_total;
  r.blocks = b < 8 ? 8 : b;
  for (int i = 0; i < n; ++i) {
    CRADLE_CHECK_ARG(off[i] >= 0 && len[i] > 0 && off[i] % 4 == 0, "%s: region %d must start at a multiple of 4 floats", what, i);
    r.off[i] = off[i];
    r.len[i] = len[i];
  }
  return CRADLE_OK;
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

// region_sums[i] = sum of squares of grads over region i (offsets / lengths in floats, host arrays), i < n_regions;
// n_total = number of regions the whole arena is cut into (fixes the summation order, see above).
// ws: cradle_adam_regions_workspace_floats() floats.
int cradle_grad_region_sumsq(const float* grads, const int64_t* region_off, const int64_t* region_len, int n_regions,
                             int n_total, float* region_sums, float* ws, hipStream_t stream) {
  using namespace cradle;
  CRADLE_CHECK_ARG(grads && region_sums && ws && (uintptr_t)grads % 16 == 0, "grad_region_sumsq: bad arguments");
  Regions r;
  const int rc = fill_regions(r, region_off, region_len, n_regions, n_total, "grad_region_sumsq");
  if (rc) return rc;
  hipLaunchKernelGGL(region_sumsq_kernel, dim3(r.blocks, r.n), dim3(256), 0, stream, grads, r, ws);
  hipLaunchKernelGGL(region_finish_kernel, dim3(r.n), dim3(256), 0, stream, ws, r.blocks, region_sums);
  CRADLE_CHECK_LAUNCH("grad_region_sumsq");
  return CRADLE_OK;
}

int64_t cradle_adam_regions_workspace_floats(void) { return (int64_t)cradle::kMaxRegions * 512; }

// Data-parallel ranks decide together whether an update is applied: a rank whose step rejected its input (flags != 0)
// overwrites one gradient element with NaN before the gradients are summed over ranks, so that the squared norm
// cradle_adam_step_regions works with is not finite on every rank and every rank skips.  element: any float of the
// gradient arena that is summed after this call (the first of the bucket reduced at the end of the step).
int cradle_grad_poison_if_flagged(float* element, const int* flags, hipStream_t stream) {
  CRADLE_CHECK_ARG(element && flags, "grad_poison_if_flagged: null pointer");
  hipLaunchKernelGGL(cradle::poison_if_flagged_kernel, dim3(1), dim3(1), 0, stream, element, flags);
  CRADLE_CHECK_LAUNCH("grad_poison_if_flagged");
  return CRADLE_OK;
}

// cradle_adam_step on the listed regions only, with the gradient's squared norm given as all_sums (device,
// n_sums region sums of the whole arena, added in order).  norm_out (device float[2], may be NULL): [0] receives
// |g'|, [1] is incremented whenever the update is skipped (skip_flags set, or the squared norm not finite).
int cradle_adam_step_regions(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                             const int64_t* region_off, const int64_t* region_len, int n_regions, const float* all_sums,
                             int n_sums, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                             float max_norm, float grad_scale, float* norm_out, const int* skip_flags, hipStream_t stream) {
  using namespace cradle;
  CRADLE_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && all_sums && step >= 1, "adam_step_regions: bad arguments");
  Regions r;
  const int rc = fill_regions(r, region_off, region_len, n_regions, n_sums, "adam_step_regions");
  if (rc) return rc;
  int64_t longest = 0;
  for (int i = 0; i < r.n; ++i) longest = longest > r.len[i] ? longest : r.len[i];
  const int blocks = (int)min((int64_t)1024, (longest + 255) / 256);
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float sqrt_bc2 = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_regions_kernel, dim3(blocks, r.n), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, r, lr,
                     beta1, beta2, eps, weight_decay, bc1, sqrt_bc2, max_norm, grad_scale, all_sums, n_sums, norm_out,
                     skip_flags);
  CRADLE_CHECK_LAUNCH("adam_step_regions");
  return CRADLE_OK;
}

}  // extern "C"
