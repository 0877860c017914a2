// The GEMM epilogue shared by the fp32 kernels (gemm.hip), the split-bf16 kernels (gemm_bf16x3.hip) and the
// split-K reducer.
#pragma once
#include "common.h"
#include "kernels.h"

namespace cradle {

// The fused epilogue of NE output elements of one thread (shared by the GEMM kernel and the split-K reducer).
// Every feature is handled for all NE elements at once -- its NE loads are issued back to back and waited
// for once -- instead of element by element: with the features behind run-time branches, the element-wise
// form serialises up to 3 dependent global-load latencies per element, 16 elements per lane, which was
// most of the fixed cost of the small products.  Loads of elements outside the matrix go to a clamped
// (valid) address; only the stores are guarded.  Per element the arithmetic is, in this order:
//   v + bias[col] + (rowscale ? rowscale * rowbias : rowbias) -> LeakyReLU -> (act_out) -> + res -> + beta*C
template <int NE>
__device__ __forceinline__ void epilogue_batch(const GemmParams& p, const int (&row_in)[NE], const int (&col_in)[NE],
                                               const bool (&ok)[NE], float (&v)[NE]) {
  int row[NE], col[NE];
  bool ones[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    ones[e] = col_in[e] == p.N;  // virtual ones column -> bias gradient
    row[e] = ok[e] ? row_in[e] : 0;
    col[e] = ok[e] && !ones[e] ? col_in[e] : 0;
  }
  if (p.extra_col) {
    float old[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) old[e] = (p.beta != 0.f && ones[e]) ? p.extra_col[row[e]] : 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e)
      if (ok[e] && ones[e]) p.extra_col[row[e]] = v[e] + (p.beta != 0.f ? p.beta * old[e] : 0.f);
  }
  if (p.bias) {
    float b[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) b[e] = p.bias[col[e]];
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] += b[e];
  }
  if (p.rowbias) {
    float rb[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int rb_row = p.rb_mode == 1 ? row[e] / p.rb_n : row[e] % p.rb_n;
      rb[e] = p.rowbias[(int64_t)rb_row * p.ld_rb + col[e]];
    }
    if (p.rowscale) {
      float rs[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) rs[e] = p.rowscale[row[e] % p.rs_n];
#pragma unroll
      for (int e = 0; e < NE; ++e) v[e] += rs[e] * rb[e];
    } else {
#pragma unroll
      for (int e = 0; e < NE; ++e) v[e] += rb[e];
    }
  }
  if (p.act) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = v[e] > 0.f ? v[e] : kLeakySlope * v[e];
  }
  if (p.act_out) {
#pragma unroll
    for (int e = 0; e < NE; ++e)
      if (ok[e] && !ones[e]) p.act_out[(int64_t)row[e] * p.ld_act + col[e]] = v[e];
  }
  if (p.res) {
    float r[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) r[e] = p.res[(int64_t)row[e] * p.ld_res + col[e]];
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] += r[e];
  }
  if (p.beta != 0.f) {
    float c[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) c[e] = p.C[(int64_t)row[e] * p.ldc + col[e]];
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] += p.beta * c[e];
  }
#pragma unroll
  for (int e = 0; e < NE; ++e)
    if (ok[e] && !ones[e]) p.C[(int64_t)row[e] * p.ldc + col[e]] = v[e];
  if (p.C2) {
    float d[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) d[e] = p.dact_src[(int64_t)row[e] * p.ld_ds + col[e]];
#pragma unroll
    for (int e = 0; e < NE; ++e)
      if (ok[e] && !ones[e]) p.C2[(int64_t)row[e] * p.ldc2 + col[e]] = v[e] * (d[e] > 0.f ? 1.f : kLeakySlope);
  }
}

}  // namespace cradle
