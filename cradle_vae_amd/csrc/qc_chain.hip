// q(qc_E) of the ELBO step: the Q-row work of its encoder as three launches instead of seventeen.
//
// Reference: the qc encoder of cradle_vae_CorrelatedNormalGuide.forward (cradle_vae/models/cradle_vae/guides/
// correlated_normal_guide.py:162-166: GaussianLikelihoodResidualMLP on the constant input ones(Q, L), mlp.py:111-142)
// and the reparameterised draws qc_E with their log-ratio sums (loss_modules.py:153-177).
//
// The encoder runs on Q rows (Q = 1 in every shipped configuration): every layer is a matrix-vector product that cannot
// occupy more than a few CUs, and as GEMM launches the chain was six kernels forward and eleven backward on a side
// stream -- launch floors and first-load latencies, nothing else.  Here:
//   qc_chain_fwd   one workgroup: all layers, the heads, the 2P*Q draws + log-ratio sums and the per-particle sum over
//                  q, activations kept in LDS between layers;
//   qc_chain_bwd   one workgroup: the Gaussian head's backward from d qc_E, the data-gradient chain through the layers
//                  (pre-activation gradients left in HBM), bias gradients;
//   qc_chain_wgrad many workgroups: every weight gradient of the chain, each a sum of Q outer products.
// The products over the 2P*Q particle rows (qc_E . Wqc^T into the basal encoder's first layer, correlated_normal_guide.py:
// 189-197; the decoder's bad-variant shift, model.py:122; their data gradients) stay GEMM launches in step.hip: in here
// they were most of the time (first version: 370 / 140 us forward / backward; now 69 / 81).
// Latency-bound by design (one CU streams the MLP's weights at ~45 GB/s); off the step's critical path.  Q <= 2; wider
// qc or an MLP beyond the LDS budget takes the GEMM chain (qc_chain_supported).
#include "common.h"
#include "gauss_math.h"
#include "kernels.h"

#include <algorithm>

namespace cradle {

namespace {

constexpr int kQcThreads = 1024;
constexpr int kQcFwdThreads = 512;  // the forward chain keeps 13 weight rows per wave in flight: 256 VGPRs per thread
constexpr int kQcRows = 2;       // rows a matrix-vector pass carries at once (Q <= 2)
constexpr int kQcPart = 16384;   // floats of LDS the slice partials of wg_rows_matT may take

// A single workgroup reading a few hundred KB of weights is bound by how many loads it keeps in flight, not by
// arithmetic: both products below issue the loads of several weight rows before they touch any of them.

// out(r, h) = sum_k W[h * ldw + k] * x[r * ldx + k] for r < R <= RMAX, h < H: a wave per output feature h, lanes
// along k, HB features per wave in flight at once (HB * RMAX <= 16).  x in LDS.  The HB * RMAX partial sums of a
// wave meet through LDS (red: kQcRedFloats per wave; 78 cross-lane shuffles per batch otherwise -- measured: they, not
// the loads, set the kernel's time): four lanes per sum add sixteen lanes' partials each, two shuffles finish it.
// emit(h, r, value) is called by one lane per result.  Every thread of the workgroup must call (barriers inside).
constexpr int kQcRedFloats = 16 * 65;
template <int RMAX, int HB, typename F>
__device__ __forceinline__ void wg_rows_dot_t(const float* __restrict__ W, int64_t ldw, int H, int K, const float* x, int ldx,
                                              int R, float* red, F&& emit) {
  static_assert(HB * RMAX <= 16, "four lanes per partial sum");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const bool vec = (K & 3) == 0 && (ldw & 3) == 0 && ((uintptr_t)W & 15) == 0 && (ldx & 3) == 0 && K <= 512;
  float* my = red + wave * kQcRedFloats;
  const int K4 = K >> 2;
  for (int hb = 0; hb < H; hb += nw * HB) {  // (the same trip count for every wave)
    const int h0 = hb + wave;
    float acc[HB][RMAX];
    if (vec) {
      float4 w0[HB], w1[HB];  // k4 = lane and lane + 64 of HB rows: all loads issued before the first use
#pragma unroll
      for (int i = 0; i < HB; ++i) {
        const int h = h0 + i * nw;
        const float4* wr = reinterpret_cast<const float4*>(W + (int64_t)(h < H ? h : 0) * ldw);
        w0[i] = lane < K4 ? wr[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        w1[i] = lane + 64 < K4 ? wr[lane + 64] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        const float4* xr = reinterpret_cast<const float4*>(x + (r < R ? r : 0) * ldx);
        const float4 x0 = lane < K4 ? xr[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 x1 = lane + 64 < K4 ? xr[lane + 64] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < HB; ++i)
          acc[i][r] = ((w0[i].x * x0.x + w0[i].y * x0.y) + (w0[i].z * x0.z + w0[i].w * x0.w)) +
                      ((w1[i].x * x1.x + w1[i].y * x1.y) + (w1[i].z * x1.z + w1[i].w * x1.w));
      }
    } else {  // any K, any alignment
#pragma unroll
      for (int i = 0; i < HB; ++i) {
        const int h = h0 + i * nw;
        const float* w = W + (int64_t)(h < H ? h : 0) * ldw;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) acc[i][r] = 0.f;
        for (int k = lane; k < K; k += 64) {
          const float wv = w[k];
#pragma unroll
          for (int r = 0; r < RMAX; ++r) acc[i][r] = fmaf(wv, x[(r < R ? r : 0) * ldx + k], acc[i][r]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < HB; ++i)
#pragma unroll
      for (int r = 0; r < RMAX; ++r) my[(i * RMAX + r) * 65 + lane] = acc[i][r];
    __syncthreads();
    {
      const int item = lane >> 2, q = lane & 3;
      float sacc = 0.f;
      if (item < HB * RMAX) {
#pragma unroll
        for (int t = 0; t < 16; ++t) sacc += my[item * 65 + q * 16 + t];
      }
      sacc += __shfl_xor(sacc, 1, 64);
      sacc += __shfl_xor(sacc, 2, 64);
      const int i = item / RMAX, r = item % RMAX, h = h0 + i * nw;
      if (q == 0 && item < HB * RMAX && h < H && r < R) emit(h, r, sacc);
    }
    __syncthreads();
  }
}
template <typename F>
__device__ __forceinline__ void wg_rows_dot(const float* __restrict__ W, int64_t ldw, int H, int K, const float* x, int ldx,
                                            int R, float* red, F&& emit) {
  if (R == 1) wg_rows_dot_t<1, 13>(W, ldw, H, K, x, ldx, R, red, emit);
  else wg_rows_dot_t<2, 8>(W, ldw, H, K, x, ldx, R, red, emit);  // R == 2 (qc_chain_supported)
}

// out(r, n) = sum_h a[r * lda + h] * W[h * ldw + n] for r < R <= kQcRows, n < Nout: threads along n (coalesced rows of
// W, four columns per thread where the layout allows), the h range cut into slices that meet in LDS, eight rows of W
// in flight per thread.  a in LDS; part: LDS scratch of kQcPart floats (slices * R * Nout of them are used;
// Nout * R <= kQcPart).  emit(r, n, value) is called once per (r, n) by one thread.
template <int V, typename F>
__device__ __forceinline__ void wg_rows_matT_t(const float* __restrict__ W, int64_t ldw, int Hin, int Nout, const float* a,
                                               int lda, int R, float* part, F&& emit) {
  const int cols = Nout, colsv = Nout / V;  // V == 4: Nout % 4 == 0 (checked by the caller)
  int slices = min(min((int)blockDim.x / colsv, 32), kQcPart / (R * cols));
  if (slices < 1) slices = 1;
  const int per = (Hin + slices - 1) / slices;
  const int lanes = (int)blockDim.x / slices;  // column groups handled per pass
  for (int base = 0; base < colsv; base += lanes) {
    const int width = min(lanes, colsv - base);
    const int sl = threadIdx.x / width, nv = base + threadIdx.x % width;
    if (sl < slices) {
      float acc[kQcRows][V];
#pragma unroll
      for (int r = 0; r < kQcRows; ++r)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[r][e] = 0.f;
      const int h1 = min(Hin, (sl + 1) * per);
      for (int h = sl * per; h < h1; h += 8) {
        float wv[8][V];
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // eight independent loads before any use
          const float* wp = W + (int64_t)min(h + i, h1 - 1) * ldw + nv * V;
          if constexpr (V == 4) {
            const float4 t = *reinterpret_cast<const float4*>(wp);
            wv[i][0] = t.x; wv[i][1] = t.y; wv[i][2] = t.z; wv[i][3] = t.w;
          } else {
            wv[i][0] = wp[0];
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (h + i < h1) {
#pragma unroll
            for (int r = 0; r < kQcRows; ++r)
              if (r < R) {
                const float av = a[r * lda + h + i];
#pragma unroll
                for (int e = 0; e < V; ++e) acc[r][e] = fmaf(av, wv[i][e], acc[r][e]);
              }
          }
      }
#pragma unroll
      for (int r = 0; r < kQcRows; ++r)
        if (r < R) {
#pragma unroll
          for (int e = 0; e < V; ++e) part[(sl * R + r) * cols + nv * V + e] = acc[r][e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R * width * V; i += blockDim.x) {
      const int r = i / (width * V), nn = base * V + i % (width * V);
      float v = 0.f;
      for (int s = 0; s < slices; ++s) v += part[(s * R + r) * cols + nn];  // fixed order
      emit(r, nn, v);
    }
    __syncthreads();
  }
}
template <typename F>
__device__ __forceinline__ void wg_rows_matT(const float* __restrict__ W, int64_t ldw, int Hin, int Nout, const float* a,
                                             int lda, int R, float* part, F&& emit) {
  if ((Nout & 3) == 0 && (ldw & 3) == 0 && ((uintptr_t)W & 15) == 0) wg_rows_matT_t<4>(W, ldw, Hin, Nout, a, lda, R, part, emit);
  else wg_rows_matT_t<1>(W, ldw, Hin, Nout, a, lda, R, part, emit);
}

struct QcFwdArgs {
  cradle_mlp_ptrs m;
  float* y[CRADLE_MAX_LAYERS];  // (Q, Hm) layer outputs
  float* r[CRADLE_MAX_LAYERS];  // (Q, Hm) activations before the residual (k >= 1)
  float* heads;                 // (Q, 2L)
  const float *eps, *ploc, *pscale;
  float *qcE, *sd, *S, *qsum;
  int Q, L, Hm, PP, P;
};

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : kLeakySlope * v; }

__global__ __launch_bounds__(kQcFwdThreads) void qc_chain_fwd_kernel(const QcFwdArgs a) {
  extern __shared__ float sm[];
  const int Q = a.Q, L = a.L, Hm = a.Hm, n = a.m.n_layers;
  const int Kmax = max(max(L, Hm), 4);
  const int ld = (Kmax + 3) & ~3;
  float* xa = sm;                      // (Q, ld) current layer input
  float* xb = xa + Q * ld;             // (Q, ld) current layer output
  float* hd = xb + Q * ld;             // (Q, 2L) heads
  float* qe = hd + Q * 2 * L;          // (PP*Q, ldq) draws
  const int ldq = (L + 3) & ~3;
  float* bs = qe + (size_t)a.PP * Q * ldq;  // (n, Hm) layer biases, then (2L) the heads': no global load where a result
                                            // is delivered (one lane per output: such a load would be a latency each)
  float* red = bs + n * Hm + 2 * L;         // (waves, kQcRedFloats)
  for (int k = 0; k < n; ++k)
    for (int i = threadIdx.x; i < Hm; i += blockDim.x) bs[k * Hm + i] = a.m.b[k][i];
  for (int i = threadIdx.x; i < 2 * L; i += blockDim.x) bs[n * Hm + i] = a.m.b_head[i];
  for (int i = threadIdx.x; i < Q * ld; i += blockDim.x) xa[i] = (i % ld) < L ? 1.0f : 0.f;  // the input: ones(Q, L)
  __syncthreads();
  // layer 0: y0 = LeakyReLU(x W0^T + b0)
  wg_rows_dot(a.m.W[0], L, Hm, L, xa, ld, Q, red, [&](int h, int r, float v) {
    const float y = leaky(v + bs[h]);
    xb[r * ld + h] = y;
    a.y[0][r * Hm + h] = y;
  });
  __syncthreads();
  for (int k = 1; k < n; ++k) {  // y_k = LeakyReLU(y_{k-1} W_k^T + b_k) + y_{k-1}
    float* t = xa; xa = xb; xb = t;
    wg_rows_dot(a.m.W[k], Hm, Hm, Hm, xa, ld, Q, red, [&](int h, int r, float v) {
      const float act = leaky(v + bs[k * Hm + h]);
      a.r[k][r * Hm + h] = act;
      const float y = act + xa[r * ld + h];
      xb[r * ld + h] = y;
      a.y[k][r * Hm + h] = y;
    });
    __syncthreads();
  }
  wg_rows_dot(a.m.W_head, Hm, 2 * L, Hm, xb, ld, Q, red, [&](int o, int r, float v) {
    const float h = v + bs[n * Hm + o];
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
    a.qsum[i] = s;
  }
}

struct QcBwdArgs {
  cradle_mlp_ptrs m, g;  // parameters, gradients (the biases' are written here)
  const float* y[CRADLE_MAX_LAYERS];
  const float* r[CRADLE_MAX_LAYERS];
  float* da[CRADLE_MAX_LAYERS];  // (Q, Hm) d loss / d pre-activation of layer k (read by qc_chain_wgrad)
  const float* heads;
  float* dheads;  // (Q, 2L)
  const float *eps, *ploc, *pscale, *qcE, *sd, *coefQ;
  const float* dq_enc;  // (PP*Q, L) d qc_E through the basal encoder's first layer
  const float* dq_dec;  // (P, L) d qc_E through the decoder's bad-variant shift (the same for every q)
  int Q, L, Hm, PP, P;
};

__global__ __launch_bounds__(kQcThreads) void qc_chain_bwd_kernel(const QcBwdArgs a) {
  extern __shared__ float sm[];
  const int Q = a.Q, L = a.L, Hm = a.Hm, n = a.m.n_layers;
  const int Kmax = max(max(2 * L, Hm), 4);
  float* in = sm;                 // (Q, Kmax) rows fed to a product
  float* G = in + Q * Kmax;       // (Q, Hm) running data gradient
  float* dav = G + Q * Hm;        // (Q, Hm) pre-activation gradient of the current layer
  float* part = dav + Q * Hm;     // kQcPart floats of slice partials
  float* dh = part + kQcPart;     // (Q, 2L)

  // the Gaussian head: particles summed in order -> dheads (Q, 2L)
  for (int i = threadIdx.x; i < Q * L; i += blockDim.x) {
    const int q = i / L, l = i % L;
    const GaussVals g = gauss_vals(a.heads[q * 2 * L + l], a.heads[q * 2 * L + L + l]);
    const float pl = a.ploc[i], ps = a.pscale[i], sdv = a.sd[i], cq = a.coefQ[q];
    float gm = 0.f, gv = 0.f;
    for (int pp = 0; pp < a.PP; ++pp) {
      const int64_t j = ((int64_t)pp * Q + q) * L + l;
      const float s = a.qcE[j], ep = a.eps[j];
      float ds = a.dq_enc[j];
      float coef = 0.f;
      if (pp < a.P) {
        ds += a.dq_dec[pp * L + l];
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

static size_t qc_fwd_lds(int Q, int L, int Hm, int PP, int n_layers) {
  const int ld = (std::max(std::max(L, Hm), 4) + 3) & ~3, ldq = (L + 3) & ~3;
  return ((size_t)2 * Q * ld + (size_t)Q * 2 * L + (size_t)PP * Q * ldq + (size_t)n_layers * Hm +
          2 * (size_t)L + (size_t)(kQcFwdThreads / 64) * kQcRedFloats) * 4;
}
static size_t qc_bwd_lds(int Q, int L, int Hm) {
  const int Kmax = std::max(std::max(2 * L, Hm), 4);
  return ((size_t)Q * Kmax + 2 * (size_t)Q * Hm + kQcPart + (size_t)Q * 2 * L) * 4;
}

bool qc_chain_supported(int Q, int L, int Hm, int PP) {
  // one workgroup, everything it carries between layers in LDS; a product's rows x columns within the partials' budget
  return Q >= 1 && Q <= 2 && (int64_t)Q * std::max(Hm, 2 * L) <= kQcPart &&
         qc_fwd_lds(Q, L, Hm, PP, CRADLE_MAX_LAYERS) <= 150 * 1024 && qc_bwd_lds(Q, L, Hm) <= 150 * 1024;
}

int qc_chain_fwd(const cradle_mlp_ptrs& m, float* const* y, float* const* r, float* heads, const float* eps, const float* ploc,
                 const float* pscale, float* qcE, float* sd, float* S, float* qsum, int Q, int L, int Hm, int PP, int P,
                 hipStream_t stream) {
  QcFwdArgs a = {};
  a.m = m;
  for (int k = 0; k < m.n_layers; ++k) { a.y[k] = y[k]; a.r[k] = r[k]; }
  a.heads = heads; a.eps = eps; a.ploc = ploc; a.pscale = pscale; a.qcE = qcE; a.sd = sd; a.S = S; a.qsum = qsum;
  a.Q = Q; a.L = L; a.Hm = Hm; a.PP = PP; a.P = P;
  const size_t lds = qc_fwd_lds(Q, L, Hm, PP, m.n_layers);
  if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)qc_chain_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds) != hipSuccess) {
    cradle_set_error("qc_chain_fwd: cannot reserve %zu bytes of LDS", lds);
    return CRADLE_ERR_HIP;
  }
  hipLaunchKernelGGL(qc_chain_fwd_kernel, dim3(1), dim3(kQcFwdThreads), lds, stream, a);
  CRADLE_CHECK_LAUNCH("qc_chain_fwd_kernel");
  return CRADLE_OK;
}

int qc_chain_bwd(const cradle_mlp_ptrs& m, const cradle_mlp_ptrs& g, float* const* y, float* const* r, float* const* da,
                 const float* heads, float* dheads, const float* eps, const float* ploc, const float* pscale, const float* qcE,
                 const float* sd, const float* coefQ, const float* dq_enc, const float* dq_dec, int Q, int L, int Hm, int PP,
                 int P, hipStream_t stream) {
  QcBwdArgs a = {};
  a.m = m; a.g = g;
  const int n = m.n_layers;
  for (int k = 0; k < n; ++k) { a.y[k] = y[k]; a.r[k] = r[k]; a.da[k] = da[k]; }
  a.heads = heads; a.dheads = dheads; a.eps = eps; a.ploc = ploc; a.pscale = pscale; a.qcE = qcE; a.sd = sd; a.coefQ = coefQ;
  a.dq_enc = dq_enc; a.dq_dec = dq_dec;
  a.Q = Q; a.L = L; a.Hm = Hm; a.PP = PP; a.P = P;
  const size_t lds = qc_bwd_lds(Q, L, Hm);
  if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)qc_chain_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds) != hipSuccess) {
    cradle_set_error("qc_chain_bwd: cannot reserve %zu bytes of LDS", lds);
    return CRADLE_ERR_HIP;
  }
  hipLaunchKernelGGL(qc_chain_bwd_kernel, dim3(1), dim3(kQcThreads), lds, stream, a);
  CRADLE_CHECK_LAUNCH("qc_chain_bwd_kernel");
  // every weight gradient of the encoder in one launch
  OuterJobs jobs = {};
  auto add = [&](const float* aa, int64_t lda, const float* bb, int64_t ldb, float* out, int64_t ldo, int R, int O, int H) {
    jobs.j[jobs.n++] = OuterJob{aa, bb, out, lda, ldb, ldo, R, O, H};
  };
  add(dheads, 2 * L, y[n - 1], Hm, g.W_head, Hm, Q, 2 * L, Hm);                // dW_head = dheads^T y_last
  for (int k = n - 1; k >= 1; --k) add(da[k], Hm, y[k - 1], Hm, g.W[k], Hm, Q, Hm, Hm);  // dW_k = da_k^T y_{k-1}
  add(da[0], Hm, nullptr, 0, g.W[0], L, Q, Hm, L);                             // dW_0 = da_0^T ones(Q, L)
  hipLaunchKernelGGL(qc_chain_wgrad_kernel, dim3(64, jobs.n), dim3(256), 0, stream, jobs);
  CRADLE_CHECK_LAUNCH("qc_chain_wgrad_kernel");
  return CRADLE_OK;
}

}  // namespace cradle

extern "C" int cradle_qc_chain_supported(int Q, int L, int Hm, int P) {
  return cradle::qc_chain_supported(Q, L, Hm, 2 * P) ? 1 : 0;
}
