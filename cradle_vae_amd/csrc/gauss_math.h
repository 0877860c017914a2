// Scalar device math shared by the sampling / log-ratio kernels (elementwise.hip) and the single-workgroup q(qc_E)
// chain (qc_chain.hip): the Gaussian head of BaseGaussianLikelihoodMLP.forward (reference
// cradle_vae/models/utils/mlp.py:131-141) and the helpers of the relaxed-Bernoulli sample.
#pragma once
#include "common.h"

namespace cradle {

constexpr float kEps32 = 1.1920928955078125e-07f;  // torch.finfo(float32).eps: clamp_probs
constexpr float kVarEps = 1e-4f;                   // mlp.py:227

__device__ __forceinline__ float clamp_probs(float p) { return fminf(fmaxf(p, kEps32), 1.0f - kEps32); }
__device__ __forceinline__ bool interior(float p) { return p > kEps32 && p < 1.0f - kEps32; }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float softplusf_(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ bool finitef_(float x) { return fabsf(x) <= 3.4028234663852886e38f; }

// heads row = [mean_raw (L) | log_var_raw (L)]:
//   var = nan_to_num(|exp(lv)+1e-4|), mean = nan_to_num(mean_raw), sd = sqrt(var), s = mean + sd*eps
struct GaussVals {
  float mean, sd, e;
  bool mean_ok, var_ok;
};
__device__ __forceinline__ GaussVals gauss_vals(float mean_raw, float lv_raw) {
  GaussVals g;
  g.e = expf(lv_raw);
  const float var_raw = fabsf(g.e + kVarEps);
  g.var_ok = finitef_(var_raw) && !(var_raw != var_raw);
  g.mean_ok = finitef_(mean_raw) && !(mean_raw != mean_raw);
  g.mean = nan_to_num_ref(mean_raw);
  g.sd = sqrtf(nan_to_num_ref(var_raw));
  return g;
}

// The draw itself, written once so that every kernel that needs E[p,t,l] gets the same bits.
__device__ __forceinline__ float gauss_draw(const GaussVals& g, float ep) { return fmaf(g.sd, ep, g.mean); }

__device__ __forceinline__ void gauss_head_grads(const GaussVals& g, float ds, float ep, float c_over_sd, float& dmean,
                                                 float& dlv) {
  dmean = g.mean_ok ? ds : 0.f;
  const float dsd = ds * ep + c_over_sd;
  dlv = g.var_ok ? dsd * g.e / (2.0f * g.sd) : 0.f;
}

}  // namespace cradle
