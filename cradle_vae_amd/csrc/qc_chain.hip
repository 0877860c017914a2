// q(qc_E) of the ELBO step as three launches instead of ~35.
//
// Reference: the qc encoder of cradle_vae_CorrelatedNormalGuide.forward (cradle_vae/models/cradle_vae/guides/
// correlated_normal_guide.py:162-166: GaussianLikelihoodResidualMLP on the constant input ones(Q, L), mlp.py:111-142),
// the products matmul(qc, qc_E) / matmul(1 - qc, qc_E) that enter the basal encoder's first layer (:189-197, here the
// vectors qc_E . Wqc^T) and the bad-variant shift ones(N, Q) @ qc_E of the decoder input (model.py:122).
//
// The encoder runs on Q rows (Q = 1 in every shipped configuration): every layer is a matrix-vector product that cannot
// occupy more than a few CUs, and as GEMM launches the chain was ten kernels forward and about twenty-five backward on
// a side stream -- launch floors and first-load latencies, nothing else.  Here:
//   qc_chain_fwd   one workgroup: all layers, the heads, the reparameterised draw + log-ratio sums, qc_E . Wqc^T and
//                  the decoder shift, activations kept in LDS between layers;
//   qc_chain_bwd   one workgroup: d qc_E from the encoder and decoder sides, the Gaussian head's backward, the
//                  data-gradient chain through the layers (pre-activation gradients left in HBM), bias gradients;
//   qc_chain_wgrad many workgroups: every weight gradient of the chain, each a sum of <= 2P*Q outer products.
// Latency-bound by design (a few hundred KB of weights through one CU per layer); off the step's critical path.
#include "common.h"
#include "gauss_math.h"
#include "kernels.h"

#include <algorithm>

namespace cradle {

namespace {

constexpr int kQcThreads = 1024;
constexpr int kQcRows = 8;  // rows a matrix-vector pass carries at once

// out(r, h) = sum_k W[h * ldw + k] * x[r * ldx + k] for r < R <= kQcRows, h < H: a wave per output feature h, lanes
// along k.  x in LDS.  emit(h, r, value) is called by lane 0 of the wave that owns h.
template <typename F>
__device__ __forceinline__ void wg_rows_dot(const float* __restrict__ W, int64_t ldw, int H, int K, const float* x, int ldx,
                                            int R, F&& emit) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const bool vec = (K & 3) == 0 && (ldw & 3) == 0 && ((uintptr_t)W & 15) == 0 && (ldx & 3) == 0;
  for (int h = wave; h < H; h += nw) {
    float acc[kQcRows];
#pragma unroll
    for (int r = 0; r < kQcRows; ++r) acc[r] = 0.f;
    const float* w = W + (int64_t)h * ldw;
    if (vec) {
      for (int k4 = lane; k4 < (K >> 2); k4 += 64) {
        const float4 wv = reinterpret_cast<const float4*>(w)[k4];
#pragma unroll
        for (int r = 0; r < kQcRows; ++r)
          if (r < R) {
            const float4 xv = reinterpret_cast<const float4*>(x + r * ldx)[k4];
            acc[r] += (wv.x * xv.x + wv.y * xv.y) + (wv.z * xv.z + wv.w * xv.w);
          }
      }
    } else {
      for (int k = lane; k < K; k += 64) {
        const float wv = w[k];
#pragma unroll
        for (int r = 0; r < kQcRows; ++r)
          if (r < R) acc[r] = fmaf(wv, x[r * ldx + k], acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < kQcRows; ++r)
      if (r < R) {
        const float v = wave_sum(acc[r]);
        if (lane == 0) emit(h, r, v);
      }
  }
}

// out(r, n) = sum_h a[r * lda + h] * W[h * ldw + n] for r < R <= kQcRows, n < Nout: threads along n (coalesced rows of
// W), the h range cut into slices that meet in LDS.  a in LDS; part: LDS scratch of slices * R * Nout floats.
// emit(r, n, value) is called once per (r, n) by one thread.
template <typename F>
__device__ __forceinline__ void wg_rows_matT(const float* __restrict__ W, int64_t ldw, int Hin, int Nout, const float* a,
                                             int lda, int R, float* part, F&& emit) {
  const int cols = Nout;
  int slices = blockDim.x / cols;
  if (slices < 1) slices = 1;
  if (slices > 16) slices = 16;
  const int per = (Hin + slices - 1) / slices;
  for (int base = 0; base < cols; base += blockDim.x / slices) {  // (one pass when Nout * slices <= threads)
    const int width = min((int)(blockDim.x / slices), cols - base);
    const int sl = threadIdx.x / width, n = base + threadIdx.x % width;
    if (sl < slices) {
      float acc[kQcRows];
#pragma unroll
      for (int r = 0; r < kQcRows; ++r) acc[r] = 0.f;
      const int h1 = min(Hin, (sl + 1) * per);
      for (int h = sl * per; h < h1; ++h) {
        const float wv = W[(int64_t)h * ldw + n];
#pragma unroll
        for (int r = 0; r < kQcRows; ++r)
          if (r < R) acc[r] = fmaf(a[r * lda + h], wv, acc[r]);
      }
#pragma unroll
      for (int r = 0; r < kQcRows; ++r)
        if (r < R) part[(sl * kQcRows + r) * cols + n] = acc[r];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R * width; i += blockDim.x) {
      const int r = i / width, nn = base + i % width;
      float v = 0.f;
      for (int s = 0; s < slices; ++s) v += part[(s * kQcRows + r) * cols + nn];  // fixed order
      emit(r, nn, v);
    }
    __syncthreads();
  }
}

struct QcFwdArgs {
  cradle_mlp_ptrs m;
  float* y[CRADLE_MAX_LAYERS];  // (Q, Hm) layer outputs
  float* r[CRADLE_MAX_LAYERS];  // (Q, Hm) activations before the residual (k >= 1)
  float* heads;                 // (Q, 2L)
  const float *eps, *ploc, *pscale;
  float *qcE, *sd, *S, *qsum;
  const float* Wqc;  // (He, ldwqc): qc block of the basal encoder's first layer
  int64_t ldwqc;
  float* vq;         // (PP*Q, He)
  const float* Wbad;  // (Hd, ldwbad): qc block of the decoder's first layer
  int64_t ldwbad;
  float* wq;          // (P, Hd)
  int Q, L, Hm, He, Hd, PP, P;
};

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : kLeakySlope * v; }

__global__ __launch_bounds__(kQcThreads) void qc_chain_fwd_kernel(const QcFwdArgs a) {
  extern __shared__ float sm[];
  const int Q = a.Q, L = a.L, Hm = a.Hm, n = a.m.n_layers;
  const int Kmax = max(max(L, Hm), 4);
  const int ld = (Kmax + 3) & ~3;
  float* xa = sm;                      // (Q, ld) current layer input
  float* xb = xa + kQcRows * ld;       // (Q, ld) current layer output
  float* hd = xb + kQcRows * ld;       // (Q, 2L) heads
  float* qe = hd + kQcRows * 2 * L;    // (PP*Q, ldq) draws, then (P, ldq) their sums over q
  const int ldq = (L + 3) & ~3;
  float* qs = qe + (size_t)a.PP * Q * ldq;  // (P, ldq)

  for (int i = threadIdx.x; i < Q * ld; i += blockDim.x) xa[i] = (i % ld) < L ? 1.0f : 0.f;  // the input: ones(Q, L)
  __syncthreads();
  // layer 0: y0 = LeakyReLU(x W0^T + b0)
  wg_rows_dot(a.m.W[0], L, Hm, L, xa, ld, Q, [&](int h, int r, float v) {
    const float y = leaky(v + a.m.b[0][h]);
    xb[r * ld + h] = y;
    a.y[0][r * Hm + h] = y;
  });
  __syncthreads();
  for (int k = 1; k < n; ++k) {  // y_k = LeakyReLU(y_{k-1} W_k^T + b_k) + y_{k-1}
    float* t = xa; xa = xb; xb = t;
    wg_rows_dot(a.m.W[k], Hm, Hm, Hm, xa, ld, Q, [&](int h, int r, float v) {
      const float act = leaky(v + a.m.b[k][h]);
      a.r[k][r * Hm + h] = act;
      const float y = act + xa[r * ld + h];
      xb[r * ld + h] = y;
      a.y[k][r * Hm + h] = y;
    });
    __syncthreads();
  }
  wg_rows_dot(a.m.W_head, Hm, 2 * L, Hm, xb, ld, Q, [&](int o, int r, float v) {
    const float h = v + a.m.b_head[o];
    hd[r * 2 * L + o] = h;
    a.heads[r * 2 * L + o] = h;
  });
  __syncthreads();
  // the draws qc_E[pp, q, :] and their log-ratio sums: a wave per (pp, q)
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int row = wave; row < a.PP * Q; row += nw) {
      const int pp = row / Q, q = row % Q;
      float acc = 0.f;
      for (int l = lane; l < L; l += 64) {
        const GaussVals g = gauss_vals(hd[q * 2 * L + l], hd[q * 2 * L + L + l]);
        const int64_t i = (int64_t)row * L + l;
        const float ep = a.eps[i];
        const float s = g.mean + g.sd * ep;
        a.qcE[i] = s;
        qe[row * ldq + l] = s;
        if (pp == 0) a.sd[q * L + l] = g.sd;
        const float pl = a.ploc[q * L + l], ps = a.pscale[q * L + l];
        const float d = s - pl;
        acc += -(d * d) / (2.0f * ps * ps) - logf(ps) + 0.5f * ep * ep + logf(g.sd);
      }
      acc = wave_sum(acc);
      if (lane == 0) a.S[row] = acc;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.P * L; i += blockDim.x) {  // qsum[p] = sum_q qc_E[p, q]: the bad variant's shift
    const int p = i / L, l = i % L;
    float s = 0.f;
    for (int q = 0; q < Q; ++q) s += qe[(p * Q + q) * ldq + l];
    qs[p * ldq + l] = s;
    a.qsum[i] = s;
  }
  __syncthreads();
  // vq = qc_E . Wqc^T (PP*Q rows), wq = qsum . Wbad^T (P rows), kQcRows rows at a time
  for (int r0 = 0; r0 < a.PP * Q; r0 += kQcRows)
    wg_rows_dot(a.Wqc, a.ldwqc, a.He, L, qe + r0 * ldq, ldq, min(kQcRows, a.PP * Q - r0),
                [&](int h, int r, float v) { a.vq[(int64_t)(r0 + r) * a.He + h] = v; });
  for (int r0 = 0; r0 < a.P; r0 += kQcRows)
    wg_rows_dot(a.Wbad, a.ldwbad, a.Hd, L, qs + r0 * ldq, ldq, min(kQcRows, a.P - r0),
                [&](int h, int r, float v) { a.wq[(int64_t)(r0 + r) * a.Hd + h] = v; });
}

struct QcBwdArgs {
  cradle_mlp_ptrs m, g;  // parameters, gradients (the biases' are written here)
  const float* y[CRADLE_MAX_LAYERS];
  const float* r[CRADLE_MAX_LAYERS];
  float* da[CRADLE_MAX_LAYERS];  // (Q, Hm) d loss / d pre-activation of layer k (read by qc_chain_wgrad)
  const float* heads;
  float* dheads;  // (Q, 2L)
  const float *eps, *ploc, *pscale, *qcE, *sd, *coefQ;
  const float* dvq;   // (PP*Q, He) gradient of qc_E . Wqc^T
  const float* dwq;   // (P, Hd) gradient of the decoder's bad-variant shift
  const float* Wqc;
  int64_t ldwqc;
  const float* Wbad;
  int64_t ldwbad;
  int Q, L, Hm, He, Hd, PP, P;
};

__global__ __launch_bounds__(kQcThreads) void qc_chain_bwd_kernel(const QcBwdArgs a) {
  extern __shared__ float sm[];
  const int Q = a.Q, L = a.L, Hm = a.Hm, n = a.m.n_layers;
  const int Kmax = max(max(max(2 * L, Hm), max(a.He, a.Hd)), 4);
  float* in = sm;                                   // (kQcRows, Kmax) rows fed to a product
  float* G = in + kQcRows * Kmax;                   // (Q, Hm) running data gradient
  float* dav = G + kQcRows * Hm;                    // (Q, Hm) pre-activation gradient of the current layer
  float* dq = dav + kQcRows * Hm;                   // (PP*Q, L) d qc_E
  float* part = dq + (size_t)a.PP * Q * L;          // 16 * kQcRows * max(Hm, L) slice partials
  float* dh = part + 16 * kQcRows * max(Hm, 2 * L);  // (Q, 2L)

  // d qc_E from the encoder: dq[row] = dvq[row] . Wqc  (rows of qc_E . Wqc^T), kQcRows rows at a time
  for (int r0 = 0; r0 < a.PP * Q; r0 += kQcRows) {
    const int R = min(kQcRows, a.PP * Q - r0);
    for (int i = threadIdx.x; i < R * a.He; i += blockDim.x) in[(i / a.He) * Kmax + i % a.He] = a.dvq[(int64_t)r0 * a.He + i];
    __syncthreads();
    wg_rows_matT(a.Wqc, a.ldwqc, a.He, L, in, Kmax, R, part, [&](int r, int l, float v) { dq[(r0 + r) * L + l] = v; });
  }
  // ... and from the decoder's shift (the same for every q of a particle of the main pass)
  for (int r0 = 0; r0 < a.P; r0 += kQcRows) {
    const int R = min(kQcRows, a.P - r0);
    for (int i = threadIdx.x; i < R * a.Hd; i += blockDim.x) in[(i / a.Hd) * Kmax + i % a.Hd] = a.dwq[(int64_t)r0 * a.Hd + i];
    __syncthreads();
    wg_rows_matT(a.Wbad, a.ldwbad, a.Hd, L, in, Kmax, R, part, [&](int r, int l, float v) {
      for (int q = 0; q < Q; ++q) dq[((r0 + r) * Q + q) * L + l] += v;
    });
  }
  // the Gaussian head: particles summed in order -> dheads (Q, 2L)
  for (int i = threadIdx.x; i < Q * L; i += blockDim.x) {
    const int q = i / L, l = i % L;
    const GaussVals g = gauss_vals(a.heads[q * 2 * L + l], a.heads[q * 2 * L + L + l]);
    const float pl = a.ploc[i], ps = a.pscale[i], sdv = a.sd[i], cq = a.coefQ[q];
    float gm = 0.f, gv = 0.f;
    for (int pp = 0; pp < a.PP; ++pp) {
      const int64_t j = ((int64_t)pp * Q + q) * L + l;
      const float s = a.qcE[j], ep = a.eps[j];
      float ds = dq[j];
      float coef = 0.f;
      if (pp < a.P) {
        coef = cq;
        ds += coef * (-(s - pl) / (ps * ps));
      }
      float dmean, dlv;
      gauss_head_grads(g, ds, ep, coef / sdv, dmean, dlv);
      gm += dmean;
      gv += dlv;
    }
    dh[q * 2 * L + l] = gm;
    dh[q * 2 * L + L + l] = gv;
    a.dheads[q * 2 * L + l] = gm;
    a.dheads[q * 2 * L + L + l] = gv;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * L; o += blockDim.x) {  // db_head
    float s = 0.f;
    for (int q = 0; q < Q; ++q) s += dh[q * 2 * L + o];
    a.g.b_head[o] = s;
  }
  // G_{n-1} = dheads . W_head ; da_{n-1} = G_{n-1} * slope
  auto slope_src = [&](int k) { return k >= 1 ? a.r[k] : a.y[0]; };
  wg_rows_matT(a.m.W_head, Hm, 2 * L, Hm, dh, 2 * L, Q, part, [&](int r, int h, float v) {
    G[r * Hm + h] = v;
    const float d = v * (slope_src(n - 1)[r * Hm + h] > 0.f ? 1.0f : kLeakySlope);
    dav[r * Hm + h] = d;
    a.da[n - 1][r * Hm + h] = d;
  });
  for (int k = n - 1; k >= 1; --k) {
    for (int h = threadIdx.x; h < Hm; h += blockDim.x) {  // db_k
      float s = 0.f;
      for (int q = 0; q < Q; ++q) s += dav[q * Hm + h];
      a.g.b[k][h] = s;
    }
    for (int i = threadIdx.x; i < Q * Hm; i += blockDim.x) in[(i / Hm) * Kmax + i % Hm] = dav[i];
    __syncthreads();
    // G_{k-1} = da_k . W_k + G_k ; da_{k-1} = G_{k-1} * slope
    wg_rows_matT(a.m.W[k], Hm, Hm, Hm, in, Kmax, Q, part, [&](int r, int h, float v) {
      const float gsum = v + G[r * Hm + h];
      G[r * Hm + h] = gsum;
      const float d = gsum * (slope_src(k - 1)[r * Hm + h] > 0.f ? 1.0f : kLeakySlope);
      dav[r * Hm + h] = d;
      a.da[k - 1][r * Hm + h] = d;
    });
  }
  for (int h = threadIdx.x; h < Hm; h += blockDim.x) {  // db_0
    float s = 0.f;
    for (int q = 0; q < Q; ++q) s += dav[q * Hm + h];
    a.g.b[0][h] = s;
  }
}

// ---- weight gradients: out[o * ldo + h] = sum_{r < R} a[r * lda + o] * b[r * ldb + h]  (b == nullptr: ones)
struct OuterJob {
  const float* a;
  const float* b;
  float* out;
  int64_t lda, ldb, ldo;
  int R, O, H;
};
constexpr int kQcMaxJobs = CRADLE_MAX_LAYERS + 4;
struct OuterJobs {
  OuterJob j[kQcMaxJobs];
  int n;
};

__global__ __launch_bounds__(256) void qc_chain_wgrad_kernel(const OuterJobs jobs) {
  const OuterJob& j = jobs.j[blockIdx.y];
  const int64_t total = (int64_t)j.O * j.H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int o = (int)(i / j.H), h = (int)(i % j.H);
    float acc = 0.f;
    for (int r = 0; r < j.R; ++r) acc = fmaf(j.a[r * j.lda + o], j.b ? j.b[r * j.ldb + h] : 1.0f, acc);
    j.out[o * j.ldo + h] = acc;
  }
}

}  // namespace

bool qc_chain_supported(int Q, int L, int Hm, int He, int Hd, int PP) {
  // one workgroup, everything it carries between layers in LDS
  const int Kmax = std::max(std::max(std::max(2 * L, Hm), std::max(He, Hd)), 4);
  const size_t bwd = ((size_t)kQcRows * Kmax + 2 * (size_t)kQcRows * Hm + (size_t)PP * Q * L +
                      16 * (size_t)kQcRows * std::max(Hm, 2 * L) + (size_t)kQcRows * 2 * L) * 4;
  return Q >= 1 && Q <= kQcRows && bwd <= 150 * 1024;
}

int qc_chain_fwd(const cradle_mlp_ptrs& m, float* const* y, float* const* r, float* heads, const float* eps, const float* ploc,
                 const float* pscale, float* qcE, float* sd, float* S, float* qsum, const float* Wqc, int64_t ldwqc, float* vq,
                 const float* Wbad, int64_t ldwbad, float* wq, int Q, int L, int Hm, int He, int Hd, int PP, int P,
                 hipStream_t stream) {
  QcFwdArgs a = {};
  a.m = m;
  for (int k = 0; k < m.n_layers; ++k) { a.y[k] = y[k]; a.r[k] = r[k]; }
  a.heads = heads; a.eps = eps; a.ploc = ploc; a.pscale = pscale; a.qcE = qcE; a.sd = sd; a.S = S; a.qsum = qsum;
  a.Wqc = Wqc; a.ldwqc = ldwqc; a.vq = vq; a.Wbad = Wbad; a.ldwbad = ldwbad; a.wq = wq;
  a.Q = Q; a.L = L; a.Hm = Hm; a.He = He; a.Hd = Hd; a.PP = PP; a.P = P;
  const int ld = (std::max(std::max(L, Hm), 4) + 3) & ~3, ldq = (L + 3) & ~3;
  const size_t lds = ((size_t)2 * kQcRows * ld + (size_t)kQcRows * 2 * L + (size_t)PP * Q * ldq + (size_t)P * ldq) * 4;
  if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)qc_chain_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds) != hipSuccess) {
    cradle_set_error("qc_chain_fwd: cannot reserve %zu bytes of LDS", lds);
    return CRADLE_ERR_HIP;
  }
  hipLaunchKernelGGL(qc_chain_fwd_kernel, dim3(1), dim3(kQcThreads), lds, stream, a);
  CRADLE_CHECK_LAUNCH("qc_chain_fwd_kernel");
  return CRADLE_OK;
}

int qc_chain_bwd(const cradle_mlp_ptrs& m, const cradle_mlp_ptrs& g, float* const* y, float* const* r, float* const* da,
                 const float* heads, float* dheads, const float* eps, const float* ploc, const float* pscale, const float* qcE,
                 const float* sd, const float* coefQ, const float* dvq, const float* dwq, const float* qsum, const float* Wqc,
                 int64_t ldwqc, float* dWqc, const float* Wbad, int64_t ldwbad, float* dWbad, int Q, int L, int Hm, int He,
                 int Hd, int PP, int P, hipStream_t stream) {
  QcBwdArgs a = {};
  a.m = m; a.g = g;
  const int n = m.n_layers;
  for (int k = 0; k < n; ++k) { a.y[k] = y[k]; a.r[k] = r[k]; a.da[k] = da[k]; }
  a.heads = heads; a.dheads = dheads; a.eps = eps; a.ploc = ploc; a.pscale = pscale; a.qcE = qcE; a.sd = sd; a.coefQ = coefQ;
  a.dvq = dvq; a.dwq = dwq; a.Wqc = Wqc; a.ldwqc = ldwqc; a.Wbad = Wbad; a.ldwbad = ldwbad;
  a.Q = Q; a.L = L; a.Hm = Hm; a.He = He; a.Hd = Hd; a.PP = PP; a.P = P;
  const int Kmax = std::max(std::max(std::max(2 * L, Hm), std::max(He, Hd)), 4);
  const size_t lds = ((size_t)kQcRows * Kmax + 2 * (size_t)kQcRows * Hm + (size_t)PP * Q * L +
                      16 * (size_t)kQcRows * std::max(Hm, 2 * L) + (size_t)kQcRows * 2 * L) * 4;
  if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)qc_chain_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds) != hipSuccess) {
    cradle_set_error("qc_chain_bwd: cannot reserve %zu bytes of LDS", lds);
    return CRADLE_ERR_HIP;
  }
  hipLaunchKernelGGL(qc_chain_bwd_kernel, dim3(1), dim3(kQcThreads), lds, stream, a);
  CRADLE_CHECK_LAUNCH("qc_chain_bwd_kernel");
  // every weight gradient of the chain in one launch
  OuterJobs jobs = {};
  auto add = [&](const float* aa, int64_t lda, const float* bb, int64_t ldb, float* out, int64_t ldo, int R, int O, int H) {
    jobs.j[jobs.n++] = OuterJob{aa, bb, out, lda, ldb, ldo, R, O, H};
  };
  add(dheads, 2 * L, y[n - 1], Hm, g.W_head, Hm, Q, 2 * L, Hm);                // dW_head = dheads^T y_last
  for (int k = n - 1; k >= 1; --k) add(da[k], Hm, y[k - 1], Hm, g.W[k], Hm, Q, Hm, Hm);  // dW_k = da_k^T y_{k-1}
  add(da[0], Hm, nullptr, 0, g.W[0], L, Q, Hm, L);                             // dW_0 = da_0^T ones(Q, L)
  if (dWqc) add(dvq, He, qcE, L, dWqc, ldwqc, PP * Q, He, L);                  // dWqc = dvq^T qc_E
  if (dWbad) add(dwq, Hd, qsum, L, dWbad, ldwbad, P, Hd, L);                   // dWbad = dwq^T qsum
  hipLaunchKernelGGL(qc_chain_wgrad_kernel, dim3(64, jobs.n), dim3(256), 0, stream, jobs);
  CRADLE_CHECK_LAUNCH("qc_chain_wgrad_kernel");
  return CRADLE_OK;
}

}  // namespace cradle
