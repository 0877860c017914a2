// Device math for the Gamma-Poisson (negative binomial) likelihood.
//
// log p(x | c, rate) as pyro-ppl 1.9.1 GammaPoisson.log_prob writes it (reference call
// sites cradle_vae/models/utils/loss_modules.py:134-135, predictor.py:158-159):
//   lgamma(c+x+1) - lgamma(c) - lgamma(x+1) - log(c+x) + c*log(rate) - (c+x)*log(1+rate)
// split here into a particle-independent part  L(x,c) = lgamma(c+x) - lgamma(c) - lgamma(x+1)
// and the main part  F = c*log(rate) - (c+x)*log1p(rate),  rate = max(c/(mu+1e-4), 1e-6)
// (mlp.py:160-163).  fp32 throughout; the formulas are chosen so that the fp32 result is at
// least as close to the fp64 value as the reference's own fp32 evaluation.
#pragma once
#include <hip/hip_runtime.h>

namespace cradle {

constexpr float kMuEps = 1e-4f;    // mlp.py:161
constexpr float kRateMin = 1e-6f;  // mlp.py:163

// A float4 of genes 4k .. 4k+3 of a row whose last valid gene is n_valid - 1: the components behind it take `fill`.
// Gene counts that are not a multiple of 4 are run as rows of the next multiple (the 16-byte vectors these kernels
// move); whatever the memory holds there never enters the arithmetic.
__device__ __forceinline__ float4 mask_gene_tail(float4 v, int k, int n_valid, float fill) {
  const int g = 4 * k;
  if (g + 3 >= n_valid) {
    if (g >= n_valid) v.x = fill;
    if (g + 1 >= n_valid) v.y = fill;
    if (g + 2 >= n_valid) v.z = fill;
    v.w = fill;
  }
  return v;
}

__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
// natural log / exp on the bare v_log_f32 / v_exp_f32 (1 ulp, no denormal fix-up code): every
// argument here is a normal float (rates >= 1e-6, 1 + rate >= 1, counts >= 1), and an
// exponent argument far below zero may flush to 0, which is its correct limit in a softmax
__device__ __forceinline__ float fast_log(float v) { return __builtin_amdgcn_logf(v) * 0.6931471805599453f; }
constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ float fast_exp(float v) { return __builtin_amdgcn_exp2f(v * kLog2e); }
// max of three in one instruction (NaN operands are passed over, like fmaxf)
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// log(1+r) with relative accuracy for small |r| (r > -1).
__device__ __forceinline__ float log1p_acc(float r) {
  // log(u) of the rounded u = fl(1+r) plus the first-order correction for the rounding error d:
  // log(1+r) = log(u) + log1p(d/u) ~ log(u) + d/u  (|d/u| <= 2^-24)
  const float u = 1.0f + r;
  const float d = r - (u - 1.0f);
  return fast_log(u) + d * fast_rcp(u);
}

// L(x,c) and (BWD) psi(c+x) - psi(c).  x must be a non-negative integer-valued float (the
// support the reference validates).  lgc = lgamma(c), psc = digamma(c) per gene.
template <bool BWD>
__device__ __forceinline__ float nb_lgamma_term(float x, float c, float lgc, float psc, float& dpsi) {
  dpsi = 0.f;
  if (x < 0.5f) return 0.f;
  if (x < 4.5f) {
    // lgamma(c+x) - lgamma(c) - lgamma(x+1) = log prod_{k<x} (c+k)/(k+1)
    float prod = c;
    float dp = BWD ? fast_rcp(c) : 0.f;
    if (x > 1.5f) {
      prod *= (c + 1.0f) * 0.5f;
      if (BWD) dp += fast_rcp(c + 1.0f);
    }
    if (x > 2.5f) {
      prod *= (c + 2.0f) * (1.0f / 3.0f);
      if (BWD) dp += fast_rcp(c + 2.0f);
    }
    if (x > 3.5f) {
      prod *= (c + 3.0f) * 0.25f;
      if (BWD) dp += fast_rcp(c + 3.0f);
    }
    dpsi = dp;
    return fast_log(prod);
  }
  // Stirling difference lgamma(a) - lgamma(b), a = c+x, b = x+1 (both >= 5), arranged so that
  // nothing large cancels: (a-.5)log(a) - (b-.5)log(b) = (a-.5)log1p((c-1)/b) + (c-1)log(b)
  const float a = c + x, b = x + 1.0f;
  const float ia = fast_rcp(a), ib = fast_rcp(b);
  const float cm1 = c - 1.0f;
  const float lb = fast_log(b);
  const float ia2 = ia * ia, ib2 = ib * ib;
  float A = (a - 0.5f) * log1p_acc(cm1 * ib) + cm1 * (lb - 1.0f);
  A += (ia - ib) * (1.0f / 12.0f) - (ia * ia2 - ib * ib2) * (1.0f / 360.0f) +
       (ia * ia2 * ia2 - ib * ib2 * ib2) * (1.0f / 1260.0f);
  if (BWD) {
    // digamma(a) asymptotic series, a >= 5
    const float la = lb + log1p_acc(cm1 * ib);
    dpsi = la - 0.5f * ia - ia2 * (1.0f / 12.0f - ia2 * (1.0f / 120.0f - ia2 * (1.0f / 252.0f))) - psc;
  }
  return A - lgc;
}

// Main term F(mu; x, c).  BWD also returns u = mu * dF/dmu (the softmax-Jacobian input) and
// du = dF/dc at fixed mu.
// NOCLAMP: the caller knows that rate0 = c/m >= 1e-6 for this element (the rows kernel tests it once per row and
// wave: min c of the thread against the row's largest possible mean), so the clamp and its selects drop out, and
// the two quotients c/m and 1/(m+c) come from ONE reciprocal, 1/(m(m+c)): the kernel is bound by its VALU work, and
// a transcendental costs four ordinary operations (m >= 1e-4 and m + c >= 1e-4: the product is a normal float).
template <bool BWD, bool NOCLAMP = false>
__device__ __forceinline__ float nb_main_term(float mu, float x, float c, float& u, float& du) {
  const float m = mu + kMuEps;
  if constexpr (NOCLAMP) {
    const float s = m + c;
    const float r = fast_rcp(m * s);
    const float rate = c * (s * r);  // c / m
    const float rmc = m * r;         // 1 / (m + c)
    const float lr = fast_log(rate);
    const float u1 = 1.0f + rate;
    const float d = rate - (u1 - 1.0f);
    const float l1p = fast_log(u1) + d * (m * rmc);  // log1p(rate), first-order correction d/u1, 1/u1 = m/(m+c)
    if (BWD) {
      const float t = (x - m) * rmc;
      u = mu * rate * t;
      du = lr - l1p - t;
    }
    return c * lr - (c + x) * l1p;
  }
  const float rate0 = c * fast_rcp(m);
  const bool clamped = rate0 < kRateMin;  // torch.clamp(min=1e-6): gradient passes where rate0 >= min
  const float rate = clamped ? kRateMin : rate0;
  const float lr = fast_log(rate);
  float l1p;
  float rmc = 0.f;
  if (BWD) {
    // 1/(1+rate) = m/(m+c): the reciprocal the gradient needs anyway also serves the first-order
    // correction of log1p (|d/u| <= 2^-24, so the clamped case may use the same value: rate is 1e-6)
    rmc = fast_rcp(m + c);
    const float u1 = 1.0f + rate;
    const float d = rate - (u1 - 1.0f);
    l1p = fast_log(u1) + d * (clamped ? 1.0f : m * rmc);
  } else {
    l1p = log1p_acc(rate);
  }
  if (BWD) {
    const float xm = x - m;
    u = clamped ? 0.f : mu * rate * xm * rmc;
    du = lr - l1p - (clamped ? 0.f : xm * rmc);
  }
  return c * lr - (c + x) * l1p;
}

}  // namespace cradle
