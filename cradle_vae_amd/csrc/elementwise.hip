// K2-K5, K7, K8: the sampling, log-ratio, gather/scatter and bookkeeping kernels of the ELBO
// step.  All HBM/latency-bound elementwise or row-reduction work on small tensors; each kernel
// fuses what the reference spreads over a chain of ATen launches and keeps forward values the
// backward needs.  Reference lines are cited per kernel (paths relative to /root/reference).
#include "common.h"
#include "kernels.h"
#include "philox.h"
#include "gauss_math.h"

namespace cradle {


// =====================================================================================
// The step's own noise: every random draw of one ELBO step in one launch.  The reference draws with torch's
// generator inside the guide (uniforms for the relaxed-Bernoulli mask, correlated_normal_guide.py:139-141;
// standard normals for E, qc_E, z_basal, :157,166,202-204); here a caller that passes no noise tensors gets
// Philox4x32-10 draws keyed by two seeds -- one shared by all ranks for the perturbation-plated variables
// (mask, E, qc_E must agree across data-parallel ranks), one per rank for the per-cell z_basal -- with the
// step number in the counter.  Bit-parity with torch's generator is impossible either way; parity tests
// inject the noise.  Tensor k of {u_mask, eps_E, eps_qc_E, eps_z} occupies float4 chunks [c0[k], c0[k+1]).
// =====================================================================================
struct NoisePlan {
  float* dst[4];
  int64_t n[4];        // elements
  int64_t chunk0[5];   // first float4 chunk of each tensor, chunk0[4] = total
};

// float4 chunk `local` of tensor k: the one place that turns a key into four draws
__device__ __forceinline__ float4 noise_chunk(int k, int64_t local, const Philox& ph, uint32_t step_lo, uint32_t step_hi) {
  const uint4 r = ph((uint32_t)local, (uint32_t)(local >> 32) | ((uint32_t)k << 28), step_lo, step_hi);
  if (k == 0) return make_float4(u01(r.x), u01(r.y), u01(r.z), u01(r.w));  // uniforms in (0,1)
  // Box-Muller, two pairs
  const float r0 = sqrtf(-2.0f * logf(u01(r.x))), r1 = sqrtf(-2.0f * logf(u01(r.z)));
  float s0, c0, s1, c1;
  sincospif(2.0f * u01(r.y), &s0, &c0);
  sincospif(2.0f * u01(r.w), &s1, &c1);
  return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}

// u_mask and eps_E, (PP, T, L) each, for the treatments of the compact axis only: slot j of particle pp gets the
// chunks of row (pp, tmap[j]) -- the same values at the same places as draw_noise_kernel writes, the rows of
// absent treatments (which nothing reads) left alone.  L % 4 == 0.  At T = 10 000 with 600 treatments in the batch
// this is 6 % of the draws.
__global__ __launch_bounds__(256) void draw_noise_slots_kernel(float* __restrict__ u_mask, float* __restrict__ eps_E,
                                                               const int* __restrict__ tmap, int PP, int T, int Tc, int L4,
                                                               uint32_t plated_lo, uint32_t plated_hi, uint32_t step_lo,
                                                               uint32_t step_hi) {
  const int64_t per = (int64_t)PP * Tc * L4;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * per) return;
  const int k = i >= per ? 1 : 0;
  i -= k * per;
  const int l4 = (int)(i % L4);
  const int j = (int)((i / L4) % Tc), pp = (int)(i / ((int64_t)L4 * Tc));
  const int t = tmap[j];
  if (t < 0) return;
  const int64_t local = ((int64_t)pp * T + t) * L4 + l4;
  const float4 v = noise_chunk(k, local, Philox{plated_lo, plated_hi}, step_lo, step_hi);
  reinterpret_cast<float4*>(k ? eps_E : u_mask)[local] = v;
}

int draw_noise_slots(float* u_mask, float* eps_E, const int* tmap, int PP, int T, int Tc, int L, uint64_t seed_plated,
                     uint64_t step, hipStream_t stream) {
  CRADLE_CHECK_ARG(L % 4 == 0 && tmap && ((uintptr_t)u_mask % 16 == 0) && ((uintptr_t)eps_E % 16 == 0),
                   "draw_noise_slots: needs a slot map, n_latent %% 4 == 0 and 16-byte aligned buffers");
  const int64_t n = 2ll * PP * Tc * (L / 4);
  hipLaunchKernelGGL(draw_noise_slots_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, u_mask, eps_E, tmap,
                     PP, T, Tc, L / 4, (uint32_t)seed_plated, (uint32_t)(seed_plated >> 32), (uint32_t)step,
                     (uint32_t)(step >> 32));
  CRADLE_CHECK_LAUNCH("draw_noise_slots_kernel");
  return CRADLE_OK;
}

__global__ __launch_bounds__(256) void draw_noise_kernel(NoisePlan np, uint32_t plated_lo, uint32_t plated_hi,
                                                         uint32_t local_lo, uint32_t local_hi, uint32_t step_lo,
                                                         uint32_t step_hi, int64_t ch_begin, int64_t ch_end, int which) {
  const int64_t ch = ch_begin + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (ch >= ch_end) return;
  int k = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) k += ch >= np.chunk0[i] ? 1 : 0;
  if (!((which >> k) & 1)) return;
  const int64_t local = ch - np.chunk0[k];
  const Philox ph = k == 3 ? Philox{local_lo, local_hi} : Philox{plated_lo, plated_hi};
  const float4 v = noise_chunk(k, local, ph, step_lo, step_hi);
  float* dst = np.dst[k] + local * 4;
  const int64_t left = np.n[k] - local * 4;
  if (left >= 4 && ((uintptr_t)dst & 15) == 0) {
    *reinterpret_cast<float4*>(dst) = v;
  } else {
    const float vals[4] = {v.x, v.y, v.z, v.w};
    for (int e = 0; e < 4 && e < left; ++e) dst[e] = vals[e];
  }
}

// which: bit k set = draw tensor k of {u_mask, eps_E, eps_qc_E, eps_z}.  A value depends on (seed, step, tensor,
// position) only, so drawing the tensors in separate launches (on different streams) gives the same numbers.
int draw_noise(float* u_mask, float* eps_E, float* eps_qc_E, float* eps_z, int64_t n_mask, int64_t n_qc, int64_t n_z,
               uint64_t seed_plated, uint64_t seed_local, uint64_t step, int which, hipStream_t stream) {
  NoisePlan np;
  np.dst[0] = u_mask; np.dst[1] = eps_E; np.dst[2] = eps_qc_E; np.dst[3] = eps_z;
  np.n[0] = n_mask; np.n[1] = n_mask; np.n[2] = n_qc; np.n[3] = n_z;
  np.chunk0[0] = 0;
  for (int k = 0; k < 4; ++k) np.chunk0[k + 1] = np.chunk0[k] + (np.n[k] + 3) / 4;
  int first = 0, last = 3;
  while (first < 4 && !((which >> first) & 1)) ++first;
  while (last >= 0 && !((which >> last) & 1)) --last;
  if (first > last) return CRADLE_OK;
  const int64_t ch_begin = np.chunk0[first], ch_end = np.chunk0[last + 1];
  if (ch_end == ch_begin) return CRADLE_OK;
  hipLaunchKernelGGL(draw_noise_kernel, dim3((unsigned)((ch_end - ch_begin + 255) / 256)), dim3(256), 0, stream, np,
                     (uint32_t)seed_plated, (uint32_t)(seed_plated >> 32), (uint32_t)seed_local,
                     (uint32_t)(seed_local >> 32), (uint32_t)step, (uint32_t)(step >> 32), ch_begin, ch_end, which);
  CRADLE_CHECK_LAUNCH("draw_noise_kernel");
  return CRADLE_OK;
}

// =====================================================================================
// Dosage plan: dense D (N,T) -> ELL rows (gather) + CSC columns (scatter) + per-treatment counts.
// D reaches the step dense (batch contract, data/utils/perturbation_dataset.py:85-89) but has
// <= 2 non-zeros per row (data/norman/data_module.py:100-102); every matmul(D, .) of the
// reference (model.py:118-122, correlated_normal_guide.py:186-188, loss_modules.py:224-232) is
// evaluated from this plan.  Entry order is (cell, column) ascending, so sums are deterministic.
// =====================================================================================
constexpr int kMaxDosages = 64;  // upper bound of kmax (check_desc in step.hip)

// One workgroup per cell.  Non-zeros are rare (<= kmax of T), so the row is streamed with all 256
// threads, hits are appended to a small LDS list in arrival order and thread 0 sorts the list by
// treatment afterwards: the ELL row is ascending in t whatever the arrival order was.
__global__ __launch_bounds__(256) void dosage_rows_kernel(const float* __restrict__ D, int N, int T, int kmax,
                                                          int* __restrict__ ell_idx, float* __restrict__ ell_val,
                                                          int* __restrict__ tcount, int* __restrict__ flags) {
  __shared__ int cnt;
  __shared__ int idx[kMaxDosages];
  __shared__ float val[kMaxDosages];
  const int n = blockIdx.x;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const float* row = D + (int64_t)n * T;
  // tcount counts the entries that are stored: the CSC columns built from it must match the ELL rows
  // exactly even for a cell with more than kmax dosages (flagged below; the caller raises), or the
  // backward scatter would walk over unwritten column slots
  auto hit = [&](int t, float v) {
    const int pos = atomicAdd(&cnt, 1);
    if (pos < kmax) {
      idx[pos] = t;
      val[pos] = v;
      if (tcount) atomicAdd(&tcount[t], 1);
    }
  };
  if ((T & 3) == 0 && ((uintptr_t)D & 15) == 0) {
    for (int t4 = threadIdx.x; t4 < T / 4; t4 += 256) {
      const float4 v = reinterpret_cast<const float4*>(row)[t4];
      if (v.x != 0.f) hit(4 * t4, v.x);
      if (v.y != 0.f) hit(4 * t4 + 1, v.y);
      if (v.z != 0.f) hit(4 * t4 + 2, v.z);
      if (v.w != 0.f) hit(4 * t4 + 3, v.w);
    }
  } else {
    for (int t = threadIdx.x; t < T; t += 256) {
      const float v = row[t];
      if (v != 0.f) hit(t, v);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int m = min(cnt, kmax);
    for (int i = 1; i < m; ++i) {  // insertion sort by treatment
      const int ti = idx[i];
      const float vi = val[i];
      int j = i - 1;
      for (; j >= 0 && idx[j] > ti; --j) {
        idx[j + 1] = idx[j];
        val[j + 1] = val[j];
      }
      idx[j + 1] = ti;
      val[j + 1] = vi;
    }
    if (cnt > kmax) atomicOr(flags, 1);
  }
  __syncthreads();
  const int m = min(cnt, kmax);
  for (int k = threadIdx.x; k < kmax; k += 256) {
    ell_idx[n * kmax + k] = k < m ? idx[k] : -1;
    ell_val[n * kmax + k] = k < m ? val[k] : 0.f;
  }
}

// Exclusive scans over the treatments in one pass: low word = cells of t (CSC column starts), high
// word = "t has a cell" (slot of every present treatment, ascending; compact axis only).
__global__ __launch_bounds__(1024) void scan_counts_kernel(const int* __restrict__ tcount, int T, int* __restrict__ tstart,
                                                           int Tc, int* __restrict__ tmap, int* __restrict__ tpos,
                                                           int* __restrict__ flags) {
  __shared__ unsigned long long wave_tot[16];
  __shared__ unsigned long long carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0ull;
  __syncthreads();
  for (int c = 0; c < T; c += 1024) {
    const int t = c + threadIdx.x;
    const int cn = t < T ? tcount[t] : 0;
    const unsigned long long v = (unsigned long long)(unsigned)cn | ((unsigned long long)(cn > 0 ? 1u : 0u) << 32);
    unsigned long long inc = v;  // inclusive scan within the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long up = __shfl_up(inc, off, 64);
      if (lane >= off) inc += up;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    unsigned long long base = carry;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    const unsigned long long ex = base + inc - v;
    if (t < T) {
      tstart[t] = (int)(unsigned)(ex & 0xffffffffull);
      if (tmap) {
        const int slot = (int)(ex >> 32);
        if (cn > 0 && slot < Tc) {
          tpos[t] = slot;
          tmap[slot] = t;
        } else {
          tpos[t] = -1;
          if (cn > 0) atomicOr(flags, 1);  // more present treatments than slots: kmax too small
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = base + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) tstart[T] = (int)(unsigned)(carry & 0xffffffffull);
}

// One wave per treatment scans the ELL entries in (cell, slot) order, 64 at a time; the ballot
// prefix keeps the order, so every treatment's cell list is ascending (deterministic sums).
__global__ __launch_bounds__(256) void csc_fill_kernel(const int* __restrict__ ell_idx, const float* __restrict__ ell_val,
                                                       int entries, int kmax, int T, const int* __restrict__ tstart,
                                                       int* __restrict__ csc_cell, float* __restrict__ csc_val) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= T) return;
  int pos = tstart[t];
  const int end = tstart[t + 1];
  for (int c = 0; c < entries && pos < end; c += 64) {
    const int e = c + lane;
    const bool hit = e < entries && ell_idx[e] == t;
    const unsigned long long b = __ballot(hit);
    if (hit) {
      const int q = pos + __popcll(b & ((1ull << lane) - 1ull));
      csc_cell[q] = e / kmax;
      csc_val[q] = ell_val[e];
    }
    pos += __popcll(b);
  }
}

// tcount <- 0, flags <- 0, tmap <- -1 in one launch (three memset nodes cost three boundaries on the plan's stream)
__global__ __launch_bounds__(256) void plan_reset_kernel(int* __restrict__ tcount, int T, int* __restrict__ flags,
                                                         int* __restrict__ tmap, int Tc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < T) tcount[i] = 0;
  if (tmap && i < Tc) tmap[i] = -1;
  if (i == 0) *flags = 0;
}

int build_dosage_plan(const float* D, int N, int T, const DosagePlan& plan, hipStream_t stream) {
  const int m = T > plan.Tc ? T : plan.Tc;
  hipLaunchKernelGGL(plan_reset_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, plan.tcount, T, plan.flags, plan.tmap,
                     plan.Tc);
  hipLaunchKernelGGL(dosage_rows_kernel, dim3(N), dim3(256), 0, stream, D, N, T, plan.kmax, plan.ell_idx, plan.ell_val,
                     plan.tcount, plan.flags);
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, plan.tcount, T, plan.tstart, plan.Tc, plan.tmap,
                     plan.tpos, plan.flags);
  CRADLE_CHECK_LAUNCH("build_dosage_plan");
  return CRADLE_OK;
}

int build_dosage_plan_csc(int N, int T, const DosagePlan& plan, hipStream_t stream) {
  hipLaunchKernelGGL(csc_fill_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, plan.ell_idx, plan.ell_val,
                     N * plan.kmax, plan.kmax, T, plan.tstart, plan.csc_cell, plan.csc_val);
  CRADLE_CHECK_LAUNCH("csc_fill_kernel");
  return CRADLE_OK;
}

// =====================================================================================
// K4 as stand-alone operators (C ABI: cradle_dosage_rows / cradle_gather_sum / cradle_scatter_sum): every
// matmul(D, .) / matmul(qc, .) of the module-level API -- guide.forward (correlated_normal_guide.py:186-194),
// model.forward (model.py:118-122), the plated re-weighting (loss_modules.py:224-232), the ATE estimator's
// per-perturbation offsets (average_treatment_effects.py:157-162,183-188) -- through the <= kmax non-zeros of
// each row instead of a dense (N,T) x (P,T,L) product.  A dense matmul also propagates NaN/Inf from rows
// that no cell selects; the gather does not (vii).
// =====================================================================================
// out[p, n, :] = sum_k val[n, k] * table[p, idx[n, k], :]
template <bool VEC>
__global__ __launch_bounds__(256) void gather_sum_kernel(const float* __restrict__ table, int64_t sp, int64_t st,
                                                         const int* __restrict__ ell_idx, const float* __restrict__ ell_val,
                                                         int kmax, float* __restrict__ out, int64_t op, int64_t on, int L) {
  const int n = blockIdx.x, p = blockIdx.y;
  const float* tb = table + (int64_t)p * sp;
  float* o = out + (int64_t)p * op + (int64_t)n * on;
  const int* ix = ell_idx + (int64_t)n * kmax;
  const float* vl = ell_val + (int64_t)n * kmax;
  if (VEC) {
    for (int l4 = threadIdx.x; l4 < L / 4; l4 += blockDim.x) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < kmax; ++k) {
        const int t = ix[k];
        if (t < 0) break;
        const float v = vl[k];
        const float4 r = reinterpret_cast<const float4*>(tb + (int64_t)t * st)[l4];
        acc.x = fmaf(v, r.x, acc.x); acc.y = fmaf(v, r.y, acc.y); acc.z = fmaf(v, r.z, acc.z); acc.w = fmaf(v, r.w, acc.w);
      }
      reinterpret_cast<float4*>(o)[l4] = acc;
    }
  } else {
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
      float acc = 0.f;
      for (int k = 0; k < kmax; ++k) {
        const int t = ix[k];
        if (t < 0) break;
        acc = fmaf(vl[k], tb[(int64_t)t * st + l], acc);
      }
      o[l] = acc;
    }
  }
}

// The transpose (backward of the gather): dtable[p, t, :] = sum over entries (n, k) with idx[n, k] == t of
// val[n, k] * dout[p, n, :].  One workgroup per (t, p) scans the N*kmax entries in order: deterministic, no
// atomics; the entry list is tiny (a batch of cells) next to the table.
__global__ __launch_bounds__(256) void scatter_sum_kernel(const float* __restrict__ dout, int64_t op, int64_t on,
                                                          const int* __restrict__ ell_idx, const float* __restrict__ ell_val,
                                                          int entries, int kmax, float* __restrict__ dtable, int64_t sp,
                                                          int64_t st, int L) {
  const int t = blockIdx.x, p = blockIdx.y;
  float* dst = dtable + (int64_t)p * sp + (int64_t)t * st;
  const float* src = dout + (int64_t)p * op;
  for (int l = threadIdx.x; l < L; l += blockDim.x) {
    float acc = 0.f;
    for (int e = 0; e < entries; ++e)
      if (ell_idx[e] == t) acc = fmaf(ell_val[e], src[(int64_t)(e / kmax) * on + l], acc);
    dst[l] = acc;
  }
}

}  // namespace cradle

extern "C" {

int cradle_dosage_rows(const float* D, int N, int T, int kmax, int* ell_idx, float* ell_val, int* flags,
                       hipStream_t stream) {
  CRADLE_CHECK_ARG(D && ell_idx && ell_val && flags && N > 0 && T > 0, "dosage_rows: bad arguments");
  CRADLE_CHECK_ARG(kmax >= 1 && kmax <= cradle::kMaxDosages, "dosage_rows: kmax=%d out of range [1,%d]", kmax,
                   cradle::kMaxDosages);
  hipLaunchKernelGGL(cradle::dosage_rows_kernel, dim3(N), dim3(256), 0, stream, D, N, T, kmax, ell_idx, ell_val,
                     (int*)nullptr, flags);
  CRADLE_CHECK_LAUNCH("dosage_rows_kernel");
  return CRADLE_OK;
}

int cradle_gather_sum(const float* table, int64_t stride_p, int64_t stride_t, const int* ell_idx, const float* ell_val,
                      int P, int N, int kmax, int L, float* out, int64_t out_stride_p, int64_t out_stride_n,
                      hipStream_t stream) {
  CRADLE_CHECK_ARG(table && ell_idx && ell_val && out && P > 0 && N > 0 && kmax > 0 && L > 0, "gather_sum: bad arguments");
  CRADLE_CHECK_ARG(P <= 65535, "gather_sum: at most 65535 particles per call");
  const bool vec = L % 4 == 0 && stride_p % 4 == 0 && stride_t % 4 == 0 && out_stride_p % 4 == 0 && out_stride_n % 4 == 0 &&
                   (uintptr_t)table % 16 == 0 && (uintptr_t)out % 16 == 0;
  const int work = vec ? L / 4 : L;
  const int nt = work <= 64 ? 64 : (work <= 128 ? 128 : 256);
  if (vec)
    hipLaunchKernelGGL(cradle::gather_sum_kernel<true>, dim3(N, P), dim3(nt), 0, stream, table, stride_p, stride_t, ell_idx,
                       ell_val, kmax, out, out_stride_p, out_stride_n, L);
  else
    hipLaunchKernelGGL(cradle::gather_sum_kernel<false>, dim3(N, P), dim3(nt), 0, stream, table, stride_p, stride_t, ell_idx,
                       ell_val, kmax, out, out_stride_p, out_stride_n, L);
  CRADLE_CHECK_LAUNCH("gather_sum_kernel");
  return CRADLE_OK;
}

int cradle_scatter_sum(const float* dout, int64_t out_stride_p, int64_t out_stride_n, const int* ell_idx,
                       const float* ell_val, int P, int N, int kmax, int T, int L, float* dtable, int64_t stride_p,
                       int64_t stride_t, hipStream_t stream) {
  CRADLE_CHECK_ARG(dout && ell_idx && ell_val && dtable && P > 0 && N > 0 && kmax > 0 && T > 0 && L > 0,
                   "scatter_sum: bad arguments");
  CRADLE_CHECK_ARG(P <= 65535, "scatter_sum: at most 65535 particles per call");
  const int nt = L <= 64 ? 64 : (L <= 128 ? 128 : 256);
  hipLaunchKernelGGL(cradle::scatter_sum_kernel, dim3(T, P), dim3(nt), 0, stream, dout, out_stride_p, out_stride_n, ell_idx,
                     ell_val, N * kmax, kmax, dtable, stride_p, stride_t, L);
  CRADLE_CHECK_LAUNCH("scatter_sum_kernel");
  return CRADLE_OK;
}

}  // extern "C"

namespace cradle {

// =====================================================================================
// Per-cell weights of loss_modules.py:137-143 and the plate reweighting of :200-232.
//   alpha = self.alpha or qc.sum()/N;  qc <- max over columns if Q > 1
//   recon = alpha (1-qc) logp_good + (1-alpha) qc logp_bad   (qc in {0,1}: one live variant per cell)
//   plated terms: sum_n (cond[n,c] != 0)/obs[c] -> cw[c]
// =====================================================================================
__global__ __launch_bounds__(256) void cell_prep_kernel(const float* __restrict__ qc, const float* __restrict__ qc_obs,
                                                        float alpha_fixed, int P, int N, int Q, CellPrep o,
                                                        int* __restrict__ flags) {
  __shared__ float red[8];
  __shared__ float red2[8];
  float qsum = 0.f, nzc = 0.f;
  bool bad = false;
  for (int n = threadIdx.x; n < N; n += 256) {
    float m = qc[(int64_t)n * Q];
    qsum += m;
    for (int q = 1; q < Q; ++q) {
      const float v = qc[(int64_t)n * Q + q];
      qsum += v;
      m = fmaxf(m, v);
    }
    if (m != 0.f && m != 1.f) bad = true;
    nzc += (m != 0.f) ? 1.f : 0.f;
    o.live[n] = m;
  }
  if (bad) atomicOr(flags, 2);
  const float tot = block_sum<256>(qsum, red);
  const float nzt = block_sum<256>(nzc, red2);
  const float alpha = alpha_fixed >= 0.f ? alpha_fixed : tot / (float)N;
  const float inv_pn = 1.0f / ((float)P * (float)N);
  if (threadIdx.x == 0) o.alpha[0] = alpha;
  __syncthreads();  // o.live written by this block is re-read below
  for (int n = threadIdx.x; n < N; n += 256) {
    const float m = o.live[n];
    o.wrec[n] = (m != 0.f) ? (1.0f - alpha) * m : alpha * (1.0f - m);  // exact for m in {0,1}
    for (int p = 0; p < P; ++p) o.grow[(int64_t)p * N + n] = -o.wrec[n] * inv_pn;
  }
  for (int q = threadIdx.x; q < Q; q += 256) {
    float s = 1.0f / qc_obs[q];
    if (isinf(s)) s = 0.f;
    const float c = nzt * s;
    o.cwq[q] = c;
    o.coefQ[q] = -c * inv_pn;
  }
}

// plate weights of the treatments (loss_modules.py:224-232), by treatment and by slot
__global__ __launch_bounds__(256) void plate_weights_kernel(const float* __restrict__ D_obs, const int* __restrict__ tcount,
                                                            int P, int N, int T, CellPrep o, int Tc,
                                                            const int* __restrict__ tmap) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float inv_pn = 1.0f / ((float)P * (float)N);
  auto weight = [&](int t) {
    float s = 1.0f / D_obs[t];
    if (isinf(s)) s = 0.f;
    return (float)tcount[t] * s;
  };
  if (i < T) {
    const float c = weight(i);
    o.cw[i] = c;
    o.coefT[i] = -c * inv_pn;
  }
  if (i < Tc) {
    const int t = tmap ? tmap[i] : i;
    const float c = t >= 0 ? weight(t) : 0.f;
    o.cwc[i] = c;
    o.coefTc[i] = -c * inv_pn;
  }
}

int cell_prep(const float* qc, const float* D_obs, const float* qc_obs, const int* tcount, float alpha_fixed, int P,
              int N, int T, int Q, const CellPrep& out, int* flags, int Tc, const int* tmap, hipStream_t stream) {
  hipLaunchKernelGGL(cell_prep_kernel, dim3(1), dim3(256), 0, stream, qc, qc_obs, alpha_fixed, P, N, Q, out, flags);
  hipLaunchKernelGGL(plate_weights_kernel, dim3((T + 255) / 256), dim3(256), 0, stream, D_obs, tcount, P, N, T, out, Tc, tmap);
  CRADLE_CHECK_LAUNCH("cell_prep_kernel");
  return CRADLE_OK;
}

// =====================================================================================
// K8 LogStandardizationModule.forward (normalization.py:41-45) for X and cfX in one launch,
// plus the "has a counterfactual" flag of loss_modules.py:145 (cfX.sum(1) != 0).
// =====================================================================================
// One workgroup per row; VEC: 16-byte accesses, two per thread in flight (G % 4 == 0 and aligned pointers).
template <bool VEC>
__global__ __launch_bounds__(256) void log_standardize_kernel(const float* __restrict__ X, const float* __restrict__ cfX,
                                                              const float* __restrict__ mean, const float* __restrict__ scale,
                                                              float* __restrict__ xn, float* __restrict__ hascf, int N, int G,
                                                              int ldo, float* __restrict__ xpad) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* src = r < N ? X + (int64_t)r * G : cfX + (int64_t)(r - N) * G;
  float* dst = xn + (int64_t)r * ldo;
  float s = 0.f;
  if (VEC) {
    const int G4 = G / 4;
    const float4* src4 = reinterpret_cast<const float4*>(src);
    const float4* m4 = reinterpret_cast<const float4*>(mean);
    const float4* s4 = reinterpret_cast<const float4*>(scale);
    float4* dst4 = reinterpret_cast<float4*>(dst);
    for (int g0 = threadIdx.x; g0 < G4; g0 += 512) {
      const int g1 = g0 + 256;
      const bool two = g1 < G4;
      const float4 x0 = src4[g0], mu0 = m4[g0], sc0 = s4[g0];
      const float4 x1 = two ? src4[g1] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 mu1 = two ? m4[g1] : x1, sc1 = two ? s4[g1] : make_float4(1.f, 1.f, 1.f, 1.f);
      s += (x0.x + x0.y) + (x0.z + x0.w) + (x1.x + x1.y) + (x1.z + x1.w);
      dst4[g0] = make_float4(nan_to_num_ref((logf(x0.x + 1.0f) - mu0.x) / sc0.x), nan_to_num_ref((logf(x0.y + 1.0f) - mu0.y) / sc0.y),
                             nan_to_num_ref((logf(x0.z + 1.0f) - mu0.z) / sc0.z), nan_to_num_ref((logf(x0.w + 1.0f) - mu0.w) / sc0.w));
      if (two)
        dst4[g1] = make_float4(nan_to_num_ref((logf(x1.x + 1.0f) - mu1.x) / sc1.x), nan_to_num_ref((logf(x1.y + 1.0f) - mu1.y) / sc1.y),
                               nan_to_num_ref((logf(x1.z + 1.0f) - mu1.z) / sc1.z), nan_to_num_ref((logf(x1.w + 1.0f) - mu1.w) / sc1.w));
    }
  } else {
    // rows of ldo >= G floats, zeros behind the G values; xpad (optional): X itself in the same padded layout
    float* xp = (xpad && r < N) ? xpad + (int64_t)r * ldo : nullptr;
    for (int g = threadIdx.x; g < ldo; g += 256) {
      const bool in = g < G;
      const float x = in ? src[g] : 0.f;
      s += x;
      dst[g] = in ? nan_to_num_ref((logf(x + 1.0f) - mean[g]) / scale[g]) : 0.f;
      if (xp) xp[g] = x;
    }
  }
  if (r >= N) {  // counts are non-negative: the row sum is zero iff every entry is, whatever the order of the sum
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) hascf[r - N] = (s != 0.f) ? 1.f : 0.f;
  }
}

int log_standardize(const float* X, const float* cfX, const float* log_mean, const float* log_scale, float* xn,
                    float* hascf, int N, int G, hipStream_t stream, int ldo, float* xpad) {
  if (ldo <= 0) ldo = G;
  CRADLE_CHECK_ARG(ldo >= G, "log_standardize: output leading dimension %d below n_phenos %d", ldo, G);
  const bool vec = G % 4 == 0 && ldo == G && !xpad && ((uintptr_t)X % 16 == 0) && ((uintptr_t)cfX % 16 == 0) &&
                   ((uintptr_t)log_mean % 16 == 0) && ((uintptr_t)log_scale % 16 == 0) && ((uintptr_t)xn % 16 == 0);
  if (vec)
    hipLaunchKernelGGL(log_standardize_kernel<true>, dim3(2 * N), dim3(256), 0, stream, X, cfX, log_mean, log_scale, xn, hascf,
                       N, G, ldo, xpad);
  else
    hipLaunchKernelGGL(log_standardize_kernel<false>, dim3(2 * N), dim3(256), 0, stream, X, cfX, log_mean, log_scale, xn,
                       hascf, N, G, ldo, xpad);
  CRADLE_CHECK_LAUNCH("log_standardize_kernel");
  return CRADLE_OK;
}

// =====================================================================================
// K3 q(M): relaxed-Bernoulli straight-through sample + Bernoulli log-ratio.
//   gumbel_softmax_bernoulli.py:5-17 over pyro RelaxedBernoulliStraightThrough (pyro-ppl 1.9.1)
//   over torch LogitRelaxedBernoulli.rsample; guide call correlated_normal_guide.py:132-141;
//   prior Bernoulli(logits=logit(p0)) model.py:96; log-ratio loss_modules.py:160-161.
// One wave per (particle, treatment) row; S[row] = sum_l log p(m) - log q(m).
// =====================================================================================
__device__ __forceinline__ float sanitize(float v) { return finitef_(v) ? v : 0.f; }

// V consecutive floats per lane (V = 4: one 16-byte access; V = 1: scalar fallback for odd widths or unaligned
// pointers).  The row kernels below take V as a template parameter: with V = 4 a row of L <= 256 is one pass,
// every load of a lane is issued before the first use, instead of up to four dependent passes of scalar loads.
template <int V>
__device__ __forceinline__ void ldv(const float* __restrict__ p, float (&r)[V]) {
  if constexpr (V == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
  } else {
#pragma unroll
    for (int e = 0; e < V; ++e) r[e] = p[e];
  }
}
template <int V>
__device__ __forceinline__ void stv(float* __restrict__ p, const float (&r)[V]) {
  if constexpr (V == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
#pragma unroll
    for (int e = 0; e < V; ++e) p[e] = r[e];
  }
}
static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int V>
__global__ __launch_bounds__(256) void mask_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ u,
                                                       const float* __restrict__ p0, float tau, float* __restrict__ mask,
                                                       float* __restrict__ soft, float* __restrict__ S, int rows, int T, int Tc,
                                                       const int* __restrict__ tmap, int L) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int j = row % Tc, pp = row / Tc;
  const int t = tmap ? tmap[j] : j;
  if (t < 0) {  // unused slot
    for (int l = lane; l < L; l += 64) {
      mask[(int64_t)row * L + l] = 0.f;
      soft[(int64_t)row * L + l] = 0.f;
    }
    if (lane == 0) S[row] = 0.f;
    return;
  }
  const float* urow = u + ((int64_t)pp * T + t) * L;
  float acc = 0.f;
  for (int l = lane * V; l < L; l += 64 * V) {
    float lgv[V], uv[V], pv[V], mv[V], sv[V];
    ldv<V>(logits + t * L + l, lgv);
    ldv<V>(urow + l, uv);
    ldv<V>(p0 + t * L + l, pv);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float lg = sanitize(lgv[e]);
      const float p = clamp_probs(sigmoidf_(lg));
      const float lq = logf(p) - log1pf(-p);
      const float uu = clamp_probs(uv[e]);
      const float y = (logf(uu) - log1pf(-uu) + lq) / tau;
      const float sft = clamp_probs(sigmoidf_(y));
      const float m = rintf(sft);
      mv[e] = m;
      sv[e] = sft;
      const float pp0 = pv[e];
      const float lp = logf(pp0 / (1.0f - pp0));  // torch.logit
      acc += (m * lp - softplusf_(lp)) - (m * lq - softplusf_(lq));
    }
    stv<V>(mask + (int64_t)row * L + l, mv);
    stv<V>(soft + (int64_t)row * L + l, sv);
  }
  acc = wave_sum(acc);
  if (lane == 0) S[row] = acc;
}

int mask_fwd(const float* logits, const float* u, const float* p_mask_probs, float tau, float* mask, float* soft,
             float* S, int PP, int T, int Tc, const int* tmap, int L, hipStream_t stream) {
  const int rows = PP * Tc;
  const bool vec = L % 4 == 0 && aligned16(logits) && aligned16(u) && aligned16(p_mask_probs) && aligned16(mask) &&
                   aligned16(soft);
  if (vec)
    hipLaunchKernelGGL(mask_fwd_kernel<4>, dim3((rows + 3) / 4), dim3(256), 0, stream, logits, u, p_mask_probs, tau, mask,
                       soft, S, rows, T, Tc, tmap, L);
  else
    hipLaunchKernelGGL(mask_fwd_kernel<1>, dim3((rows + 3) / 4), dim3(256), 0, stream, logits, u, p_mask_probs, tau, mask,
                       soft, S, rows, T, Tc, tmap, L);
  CRADLE_CHECK_LAUNCH("mask_fwd_kernel");
  return CRADLE_OK;
}

// The sample alone, for the module path's inference calls (guide.forward under no_grad: ATE, IWELBO, observation
// sampling): mask[p, t, :] of treatments t0 .. t0 + Ts - 1 from logits (T, L) and uniforms u (P, Ts, L).  Same
// expressions as mask_fwd_kernel: same bits.  One thread per 4 (or 1) elements.
template <int V>
__global__ __launch_bounds__(256) void mask_sample_kernel(const float* __restrict__ logits, const float* __restrict__ u,
                                                          float tau, float* __restrict__ mask, int64_t n, int64_t TsL) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (i >= n) return;
  float lgv[V], uv[V], mv[V];
  ldv<V>(logits + i % TsL, lgv);
  ldv<V>(u + i, uv);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float p = clamp_probs(sigmoidf_(sanitize(lgv[e])));
    const float lq = logf(p) - log1pf(-p);
    const float uu = clamp_probs(uv[e]);
    const float y = (logf(uu) - log1pf(-uu) + lq) / tau;
    mv[e] = rintf(clamp_probs(sigmoidf_(y)));
  }
  stv<V>(mask + i, mv);
}

// out = nan_to_num(mean_raw) + sqrt(nan_to_num(|exp(lv_raw) + 1e-4|)) * eps  (mlp.py:131-141 + Normal.rsample): the
// Gaussian head's draw of the module path's inference calls; gauss_vals / gauss_draw as everywhere else.
template <int V>
__global__ __launch_bounds__(256) void gauss_sample_kernel(const float* __restrict__ mean_raw, int64_t ld_mean,
                                                           const float* __restrict__ lv_raw, int64_t ld_lv,
                                                           const float* __restrict__ eps, float* __restrict__ out,
                                                           int64_t rows, int L) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (i >= rows * L) return;
  const int64_t r = i / L;
  const int l = (int)(i % L);
  float mv[V], lv[V], ev[V], ov[V];
  ldv<V>(mean_raw + r * ld_mean + l, mv);
  ldv<V>(lv_raw + r * ld_lv + l, lv);
  ldv<V>(eps + i, ev);
#pragma unroll
  for (int e = 0; e < V; ++e) ov[e] = gauss_draw(gauss_vals(mv[e], lv[e]), ev[e]);
  stv<V>(out + i, ov);
}

// d loss / d q_mask_logits.  dmask_a/dmask_b: gradients reaching the hard sample from mask*E and
// from the embedding encoder's input; coefT[t] = -cw[t]/(P N) weights the log-ratio term of the
// main pass (particles pp < P).  Straight-through: the hard sample's gradient lands on the soft
// sample.  One thread per (t, l), particles summed in order (deterministic).
__global__ __launch_bounds__(256) void mask_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ p0,
                                                       float tau, const float* __restrict__ mask, const float* __restrict__ soft,
                                                       const float* __restrict__ da, const float* __restrict__ db,
                                                       const float* __restrict__ coefT, float* __restrict__ dlogits, int PP,
                                                       int P, int T, int Tc, const int* __restrict__ tpos, int L) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= T * L) return;
  const int t = i / L, l = i % L;
  const int slot = tpos ? tpos[t] : t;
  const float raw = logits[i];
  if (slot < 0 || !finitef_(raw)) {  // no cell of this treatment in the batch / sanitize_tensor made it a constant
    dlogits[i] = 0.f;
    return;
  }
  const float pr = sigmoidf_(raw);
  const bool p_in = interior(pr);
  const float p = clamp_probs(pr);
  const float lq = logf(p) - log1pf(-p);
  const float pp0 = p0[i];
  const float lp = logf(pp0 / (1.0f - pp0));
  const float ct = coefT[t];
  float g = 0.f;
  for (int pp = 0; pp < PP; ++pp) {
    const int64_t j = ((int64_t)pp * Tc + slot) * L + l;
    const float m = mask[j], s = soft[j];
    float dm = da[j] + db[j];
    if (pp < P) dm += ct * (lp - lq);          // d/dm [log p(m) - log q(m)]
    const float dy = interior(s) ? dm * s * (1.0f - s) : 0.f;
    float dl = dy / tau;                        // through logit(clamp(sigmoid(logits)))
    if (pp < P) dl += ct * (-(m - p));          // d/d logit_q [-log q(m)]
    if (p_in) g += dl;
  }
  dlogits[i] = g;
}

int mask_bwd(const float* logits, const float* p_mask_probs, float tau, const float* mask, const float* soft,
             const float* dmask_a, const float* dmask_b, const float* coefT, float* dlogits, int PP, int P, int T, int Tc,
             const int* tpos, int L, hipStream_t stream) {
  hipLaunchKernelGGL(mask_bwd_kernel, dim3((T * L + 255) / 256), dim3(256), 0, stream, logits, p_mask_probs, tau, mask,
                     soft, dmask_a, dmask_b, coefT, dlogits, PP, P, T, Tc, tpos, L);
  CRADLE_CHECK_LAUNCH("mask_bwd_kernel");
  return CRADLE_OK;
}

// =====================================================================================
// K2 Gaussian heads: BaseGaussianLikelihoodMLP.forward tail (mlp.py:131-141), rsample
// (correlated_normal_guide.py:157,166,202-204) and the Normal log-ratio of loss_modules.py:
// 160-161.  heads row = [mean_raw (L) | log_var_raw (L)].
//   var = nan_to_num(|exp(lv)+1e-4|), mean = nan_to_num(mean_raw), sd = sqrt(var), s = mean + sd*eps
//   log N(ploc,pscale)(s) - log N(mean,sd)(s) = -(s-ploc)^2/(2 pscale^2) - log pscale + eps^2/2 + log sd
// =====================================================================================

// q(E|M): rows = PP*Tc (particle, slot).  Off the critical path: the gather that feeds the encoders
// (gather_offsets_kernel) evaluates the draws it needs itself; this kernel keeps E, sd and the
// log-ratio sums for the loss and the backward pass.
__global__ __launch_bounds__(256) void gaussE_fwd_kernel(const float* __restrict__ heads, const float* __restrict__ eps,
                                                         const float* __restrict__ ploc, const float* __restrict__ pscale,
                                                         float* __restrict__ E, float* __restrict__ sd, float* __restrict__ S,
                                                         int rows, int T, int Tc, const int* __restrict__ tmap, int L) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int j = row % Tc, pp = row / Tc;
  const int t = tmap ? tmap[j] : j;
  if (t < 0) {  // unused slot
    for (int l = lane; l < L; l += 64) {
      E[(int64_t)row * L + l] = 0.f;
      sd[(int64_t)row * L + l] = 1.f;
    }
    if (lane == 0) S[row] = 0.f;
    return;
  }
  const float* erow = eps + ((int64_t)pp * T + t) * L;
  float acc = 0.f;
  for (int l = lane; l < L; l += 64) {
    const GaussVals g = gauss_vals(heads[(int64_t)row * 2 * L + l], heads[(int64_t)row * 2 * L + L + l]);
    const float ep = erow[l];
    const float s = gauss_draw(g, ep);
    E[(int64_t)row * L + l] = s;
    sd[(int64_t)row * L + l] = g.sd;
    const float pl = ploc[t * L + l], ps = pscale[t * L + l];
    const float d = s - pl;
    acc += -(d * d) / (2.0f * ps * ps) - logf(ps) + 0.5f * ep * ep + logf(g.sd);
  }
  acc = wave_sum(acc);
  if (lane == 0) S[row] = acc;
}

int gaussE_fwd(const float* heads, const float* eps, const float* ploc, const float* pscale, float* E, float* sd, float* S,
               int PP, int T, int Tc, const int* tmap, int L, hipStream_t stream) {
  const int rows = PP * Tc;
  hipLaunchKernelGGL(gaussE_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, heads, eps, ploc, pscale, E, sd, S,
                     rows, T, Tc, tmap, L);
  CRADLE_CHECK_LAUNCH("gaussE_fwd_kernel");
  return CRADLE_OK;
}


// Backward of q(E|M) fused with the scatter that produces its incoming gradient: one workgroup per
// (slot, particle) first sums d(mask*E)[pp,t,:] = sum_{cells n of t} D[n,t] * g[pp,n,:] over the CSC
// column of its treatment (g = a1 + a2 for the main pass, b1 for the counterfactual pass), then
// applies the Gaussian head's closed-form backward.  No (2P, Tc, L) gradient buffer in between.
__global__ __launch_bounds__(256) void gaussE_bwd_kernel(const int* __restrict__ tstart, const int* __restrict__ csc_cell,
                                                         const float* __restrict__ csc_val, int entries,
                                                         const float* __restrict__ a1, int64_t ld1, const float* __restrict__ a2,
                                                         int64_t ld2, const float* __restrict__ b1, int64_t ldb1,
                                                         const float* __restrict__ heads, const float* __restrict__ eps,
                                                         const float* __restrict__ mask, const float* __restrict__ ploc,
                                                         const float* __restrict__ pscale, const float* __restrict__ E,
                                                         const float* __restrict__ sd, const float* __restrict__ coefTc,
                                                         float* __restrict__ dheads, float* __restrict__ dmask, int P, int N,
                                                         int T, int Tc, const int* __restrict__ tmap, int L) {
  const int j = blockIdx.x, pp = blockIdx.y;
  const int64_t row = (int64_t)pp * Tc + j;
  const int t = tmap ? tmap[j] : j;
  if (t < 0) {  // unused slot: nothing flows back
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
      dmask[row * L + l] = 0.f;
      dheads[row * 2 * L + l] = 0.f;
      dheads[row * 2 * L + L + l] = 0.f;
    }
    return;
  }
  const int e0 = tstart[t], e1 = min(tstart[t + 1], entries);  // never past the column arrays
  const float coef = pp < P ? coefTc[j] : 0.f;
  for (int l = threadIdx.x; l < L; l += blockDim.x) {
    float dme = 0.f;
    for (int e = e0; e < e1; ++e) {
      const int n = csc_cell[e];
      if ((unsigned)n >= (unsigned)N) continue;  // defensive: an index the plan did not write
      float g;
      if (pp < P) g = a1[((int64_t)pp * N + n) * ld1 + l] + a2[((int64_t)pp * N + n) * ld2 + l];
      else g = b1[((int64_t)(pp - P) * N + n) * ldb1 + l];
      dme = fmaf(csc_val[e], g, dme);
    }
    const int64_t i = row * L + l;
    const GaussVals g = gauss_vals(heads[row * 2 * L + l], heads[row * 2 * L + L + l]);
    // E and its scale are recomputed from the heads (same expressions as gaussE_fwd_kernel) when the caller
    // passes none: the forward bookkeeping kernel then is a leaf that need not have run before this one
    const float ep = eps[((int64_t)pp * T + t) * L + l];
    const float s = E ? E[i] : gauss_draw(g, ep);
    const float sdv = sd ? sd[i] : g.sd;
    const float pl = ploc[t * L + l], ps = pscale[t * L + l];
    const float ds = dme * mask[i] + coef * (-(s - pl) / (ps * ps));
    dmask[i] = dme * s;
    float dmean, dlv;
    gauss_head_grads(g, ds, ep, coef / sdv, dmean, dlv);
    dheads[row * 2 * L + l] = dmean;
    dheads[row * 2 * L + L + l] = dlv;
  }
}

int gaussE_bwd(const DosagePlan& plan, const float* a1, int64_t ld1, const float* a2, int64_t ld2, const float* b1,
               int64_t ldb1, const float* heads, const float* eps, const float* mask, const float* ploc, const float* pscale,
               const float* E, const float* sd, const float* coefTc, float* dheads, float* dmask, int PP, int P, int N, int T,
               int L, hipStream_t stream) {
  const int nt = L <= 64 ? 64 : (L <= 128 ? 128 : 256);  // one pass over the latent columns
  hipLaunchKernelGGL(gaussE_bwd_kernel, dim3(plan.Tc, PP), dim3(nt), 0, stream, plan.tstart, plan.csc_cell, plan.csc_val,
                     N * plan.kmax, a1, ld1, a2, ld2, b1, ldb1, heads, eps, mask, ploc, pscale, E, sd, coefTc, dheads, dmask,
                     P, N, T, plan.Tc, plan.tmap, L);
  CRADLE_CHECK_LAUNCH("gaussE_bwd_kernel");
  return CRADLE_OK;
}

// q(qc_E): the encoder input is the constant ones(Q,L) (correlated_normal_guide.py:74,162), so
// its heads (Q, 2L) are shared by all particles of both guide passes.  qsum[p] = sum_q qc_E[p,q]
// is the bad-variant latent shift of model.py:122 (ones(qc.shape) @ qc_E).
__global__ __launch_bounds__(256) void qce_fwd_kernel(const float* __restrict__ heads, const float* __restrict__ eps,
                                                      const float* __restrict__ ploc, const float* __restrict__ pscale,
                                                      float* __restrict__ qcE, float* __restrict__ sd, float* __restrict__ S,
                                                      float* __restrict__ qsum, int PP, int P, int Q, int L) {
  // one block per particle pp; threads over l; loop q (Q is 1..6)
  __shared__ float red[4];
  const int pp = blockIdx.x;
  for (int q = 0; q < Q; ++q) {
    float acc = 0.f;
    for (int l = threadIdx.x; l < L; l += 256) {
      const GaussVals g = gauss_vals(heads[q * 2 * L + l], heads[q * 2 * L + L + l]);
      const int64_t i = ((int64_t)pp * Q + q) * L + l;
      const float ep = eps[i];
      const float s = g.mean + g.sd * ep;
      qcE[i] = s;
      if (pp == 0) sd[q * L + l] = g.sd;
      const float pl = ploc[q * L + l], ps = pscale[q * L + l];
      const float d = s - pl;
      acc += -(d * d) / (2.0f * ps * ps) - logf(ps) + 0.5f * ep * ep + logf(g.sd);
    }
    acc = block_sum<256>(acc, red);
    if (threadIdx.x == 0) S[pp * Q + q] = acc;
    __syncthreads();
  }
  if (pp < P) {
    for (int l = threadIdx.x; l < L; l += 256) {
      float s = 0.f;
      for (int q = 0; q < Q; ++q) s += qcE[((int64_t)pp * Q + q) * L + l];
      qsum[pp * L + l] = s;
    }
  }
}

int qce_fwd(const float* heads, const float* eps, const float* ploc, const float* pscale, float* qcE, float* sd, float* S,
            float* qsum, int PP, int P, int Q, int L, hipStream_t stream) {
  hipLaunchKernelGGL(qce_fwd_kernel, dim3(PP), dim3(256), 0, stream, heads, eps, ploc, pscale, qcE, sd, S, qsum, PP, P, Q, L);
  CRADLE_CHECK_LAUNCH("qce_fwd_kernel");
  return CRADLE_OK;
}

// dq_enc (PP*Q, L): gradient from the basal encoder's qc offsets; dq_dec (P, L): gradient from the
// decoder's bad-variant shift (same for every q).  Sums over particles in order -> (Q, 2L).
__global__ __launch_bounds__(256) void qce_bwd_kernel(const float* __restrict__ heads, const float* __restrict__ eps,
                                                      const float* __restrict__ ploc, const float* __restrict__ pscale,
                                                      const float* __restrict__ qcE, const float* __restrict__ sd,
                                                      const float* __restrict__ dq_enc, const float* __restrict__ dq_dec,
                                                      const float* __restrict__ coefQ, float* __restrict__ dheads, int PP,
                                                      int P, int Q, int L) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Q * L) return;
  const int q = i / L, l = i % L;
  const GaussVals g = gauss_vals(heads[q * 2 * L + l], heads[q * 2 * L + L + l]);
  const float pl = ploc[i], ps = pscale[i], sdv = sd[i], cq = coefQ[q];
  float gm = 0.f, gv = 0.f;
  for (int pp = 0; pp < PP; ++pp) {
    const int64_t j = ((int64_t)pp * Q + q) * L + l;
    const float s = qcE[j], ep = eps[j];
    float ds = dq_enc[j];
    float coef = 0.f;
    if (pp < P) {
      ds += dq_dec[pp * L + l];
      coef = cq;
      ds += coef * (-(s - pl) / (ps * ps));
    }
    float dmean, dlv;
    gauss_head_grads(g, ds, ep, coef / sdv, dmean, dlv);
    gm += dmean;
    gv += dlv;
  }
  dheads[q * 2 * L + l] = gm;
  dheads[q * 2 * L + L + l] = gv;
}

int qce_bwd(const float* heads, const float* eps, const float* ploc, const float* pscale, const float* qcE,
            const float* sd, const float* dq_enc, const float* dq_dec, const float* coefQ, float* dheads, int PP, int P,
            int Q, int L, hipStream_t stream) {
  hipLaunchKernelGGL(qce_bwd_kernel, dim3((Q * L + 255) / 256), dim3(256), 0, stream, heads, eps, ploc, pscale, qcE, sd,
                     dq_enc, dq_dec, coefQ, dheads, PP, P, Q, L);
  CRADLE_CHECK_LAUNCH("qce_bwd_kernel");
  return CRADLE_OK;
}

// q(z_basal | ...) head + K7 counterfactual KL, one wave per (particle, cell).
//   heads rows: [0,PN) main pass q_z_basal, [PN,2PN) main pass q_z_cf_basal,
//               [2PN,3PN) counterfactual pass q_z_basal        (correlated_normal_guide.py:196-204)
//   z = mean + sd*eps; ltz = sum_l log N(0,1)(z) - log N(mean,sd)(z)            (loss_modules.py:160-161)
//   KL (loss_modules.py:145-151,190-196): on cells with cfX.sum != 0, with 1 = cf pass, 2 = q_z_cf_basal:
//     beta * sum_l mean_p sum_n -0.5 (1 + lv1 - lv2 - (mu1-mu2)^2 - exp(lv1-lv2)),  lv = 2 log sd
//   The KL branches have no other consumer, so their head gradients are written here already.
template <int V>
__global__ __launch_bounds__(256) void zhead_fwd_kernel(const float* __restrict__ heads, const float* __restrict__ eps,
                                                        const float* __restrict__ hascf, float kl_coef, float* __restrict__ z,
                                                        int64_t ldz, float* __restrict__ sd, float* __restrict__ ltz,
                                                        float* __restrict__ klrow, float* __restrict__ dheads, int P, int N,
                                                        int L) {
  const int64_t PN = (int64_t)P * N;
  const int64_t r0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r0 >= PN) return;
  const int n = r0 % N;
  const int64_t r1 = PN + r0, r2 = 2 * PN + r0;
  const bool cf = hascf[n] != 0.f;
  float a_lt = 0.f, a_kl = 0.f;
  for (int l = lane * V; l < L; l += 64 * V) {
    float m0[V], v0[V], ep[V], m1[V], v1[V], m2[V], v2[V];
    ldv<V>(heads + r0 * 2 * L + l, m0);
    ldv<V>(heads + r0 * 2 * L + L + l, v0);
    ldv<V>(eps + r0 * L + l, ep);
    if (cf) {
      ldv<V>(heads + r1 * 2 * L + l, m2);  // q_z_cf_basal
      ldv<V>(heads + r1 * 2 * L + L + l, v2);
      ldv<V>(heads + r2 * 2 * L + l, m1);  // cf pass q_z_basal
      ldv<V>(heads + r2 * 2 * L + L + l, v1);
    }
    float zv[V], sdv[V], dm1[V], dv1[V], dm2[V], dv2[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const GaussVals g = gauss_vals(m0[e], v0[e]);
      const float zz = g.mean + g.sd * ep[e];
      zv[e] = zz;
      sdv[e] = g.sd;
      a_lt += -0.5f * zz * zz + 0.5f * ep[e] * ep[e] + logf(g.sd);
      dm1[e] = dv1[e] = dm2[e] = dv2[e] = 0.f;
      if (cf) {
        const GaussVals g2 = gauss_vals(m2[e], v2[e]);
        const GaussVals g1 = gauss_vals(m1[e], v1[e]);
        // d = lv1 - lv2 = log(var1 / var2).  The two encoder passes see nearly the same input early in training, so
        // d and dmu are small and 1 + d - exp(d) = -(d^2/2 + d^3/6 + ...) is a difference of numbers near 1: written
        // as the reference does, every element carries an fp32 rounding error of ~1e-7 against a value of ~d^2/2, and
        // the sum over cells x latents (the KL is a SUM, loss_modules.py:190-196) sat 6e-5 .. 1.5e-4 off the fp64
        // value.  Same quantity, evaluated without the cancellation (log1p of the relative variance difference, the
        // series of d - expm1(d) for small d).
        const float var1 = g1.sd * g1.sd, var2 = g2.sd * g2.sd;
        const float d = log1pf((var1 - var2) / var2);
        const float dmu = g1.mean - g2.mean;
        const float em1 = expm1f(d);  // ratio - 1
        const float one_d_ratio =
            fabsf(d) < 0.3f
                ? -d * d * (0.5f + d * (1.0f / 6 + d * (1.0f / 24 + d * (1.0f / 120 + d * (1.0f / 720 + d * (1.0f / 5040))))))
                : d - em1;
        a_kl += -0.5f * (one_d_ratio - dmu * dmu);
        // d/d(mean), d/d(lv) then lv = log(exp(lv_raw) + 1e-4): d lv / d lv_raw = e / var
        dm1[e] = g1.mean_ok ? kl_coef * dmu : 0.f;
        dm2[e] = g2.mean_ok ? -kl_coef * dmu : 0.f;
        dv1[e] = g1.var_ok ? kl_coef * 0.5f * em1 * g1.e / var1 : 0.f;
        dv2[e] = g2.var_ok ? kl_coef * (-0.5f) * em1 * g2.e / var2 : 0.f;
      }
    }
    stv<V>(z + r0 * ldz + l, zv);
    stv<V>(sd + r0 * L + l, sdv);
    stv<V>(dheads + r1 * 2 * L + l, dm2);
    stv<V>(dheads + r1 * 2 * L + L + l, dv2);
    stv<V>(dheads + r2 * 2 * L + l, dm1);
    stv<V>(dheads + r2 * 2 * L + L + l, dv1);
  }
  a_lt = wave_sum(a_lt);
  a_kl = wave_sum(a_kl);
  if (lane == 0) {
    ltz[r0] = a_lt;
    klrow[r0] = a_kl;
  }
}

int zhead_fwd(const float* heads, const float* eps, const float* hascf, float beta, float* z, int64_t ldz, float* sd,
              float* ltz, float* klrow, float* dheads, int P, int N, int L, hipStream_t stream) {
  const int64_t rows = (int64_t)P * N;
  const bool vec = L % 4 == 0 && ldz % 4 == 0 && aligned16(heads) && aligned16(eps) && aligned16(z) && aligned16(sd) &&
                   aligned16(dheads);
  if (vec)
    hipLaunchKernelGGL(zhead_fwd_kernel<4>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, heads, eps, hascf,
                       beta / (float)P, z, ldz, sd, ltz, klrow, dheads, P, N, L);
  else
    hipLaunchKernelGGL(zhead_fwd_kernel<1>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, heads, eps, hascf,
                       beta / (float)P, z, ldz, sd, ltz, klrow, dheads, P, N, L);
  CRADLE_CHECK_LAUNCH("zhead_fwd_kernel");
  return CRADLE_OK;
}

// main-pass q_z_basal head gradient: dz from the decoder + the ltz term (weight -1/(P N)).
__global__ __launch_bounds__(256) void zhead_bwd_kernel(const float* __restrict__ heads, const float* __restrict__ eps,
                                                        const float* __restrict__ z, int64_t ldz, const float* __restrict__ sd,
                                                        const float* __restrict__ dz, int64_t lddz, float* __restrict__ dheads,
                                                        int64_t PN, int L, float coef) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= PN * L) return;
  const int64_t r = i / L;
  const int l = i % L;
  const GaussVals g = gauss_vals(heads[r * 2 * L + l], heads[r * 2 * L + L + l]);
  const float ds = dz[r * lddz + l] + coef * (-z[r * ldz + l]);
  float dmean, dlv;
  gauss_head_grads(g, ds, eps[i], coef / sd[i], dmean, dlv);
  dheads[r * 2 * L + l] = dmean;
  dheads[r * 2 * L + L + l] = dlv;
}

int zhead_bwd(const float* heads, const float* eps, const float* z, int64_t ldz, const float* sd, const float* dz,
              int64_t lddz, float* dheads, int P, int N, int L, hipStream_t stream) {
  const int64_t PN = (int64_t)P * N;
  hipLaunchKernelGGL(zhead_bwd_kernel, dim3((unsigned)((PN * L + 255) / 256)), dim3(256), 0, stream, heads, eps, z, ldz, sd,
                     dz, lddz, dheads, PN, L, -1.0f / (float)PN);
  CRADLE_CHECK_LAUNCH("zhead_bwd_kernel");
  return CRADLE_OK;
}

// =====================================================================================
// K4 gather-sum / scatter-add:  off[pp,n,:] = sum_t D[n,t] (mask*E)[pp,t,:]
//   (model.py:121, correlated_normal_guide.py:186-188) and its transpose for the backward.
// Particles pp < P (main pass) go to offA (a column block of the decoder input [z | off]),
// pp >= P (counterfactual pass) to offB.
// =====================================================================================
template <int V>
__global__ __launch_bounds__(64) void gather_offsets_kernel(const float* __restrict__ heads, const float* __restrict__ eps,
                                                            const float* __restrict__ mask, const int* __restrict__ ell_idx,
                                                            const float* __restrict__ ell_val, int kmax, float* __restrict__ offA,
                                                            int64_t ldA, float* __restrict__ offB, int64_t ldB, int P, int N,
                                                            int T, int Tc, const int* __restrict__ tpos, int L) {
  // offB: (2P*N, L) offsets of both guide passes (main rows first), one operand for the encoder GEMMs;
  // offA: the main pass once more, inside the decoder input [z | offset].
  // E[pp,t,l]*mask[pp,t,l] is evaluated here from the Gaussian heads for the <= kmax treatments of the
  // cell (same expression as gaussE_fwd_kernel: same bits), so the encoders do not wait for that kernel.
  // No LDS: a cell reads <= kmax rows of 3L floats once, and the (2P*Tc, 3L) tables (2.5 MB at Norman) sit in L2.
  const int n = blockIdx.x, pp = blockIdx.y;
  float* out = offB + ((int64_t)pp * N + n) * ldB;
  float* out2 = pp < P ? offA + ((int64_t)pp * N + n) * ldA : nullptr;
  for (int l = threadIdx.x * V; l < L; l += 64 * V) {
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    for (int k = 0; k < kmax; ++k) {
      const int t = ell_idx[n * kmax + k];
      if (t < 0) break;
      const int slot = tpos ? tpos[t] : t;
      if (slot < 0) continue;
      const int64_t row = (int64_t)pp * Tc + slot;
      float mr[V], lv[V], ep[V], mk[V];
      ldv<V>(heads + row * 2 * L + l, mr);
      ldv<V>(heads + row * 2 * L + L + l, lv);
      ldv<V>(eps + ((int64_t)pp * T + t) * L + l, ep);
      ldv<V>(mask + row * L + l, mk);
      const float dose = ell_val[n * kmax + k];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const GaussVals g = gauss_vals(mr[e], lv[e]);
        const float sdraw = gauss_draw(g, ep[e]);
        acc[e] = fmaf(dose, sdraw * mk[e], acc[e]);
      }
    }
    stv<V>(out + l, acc);
    if (out2) stv<V>(out2 + l, acc);
  }
}

int gather_offsets(const float* heads, const float* eps, const float* mask, const DosagePlan& plan, float* offA, int64_t ldA,
                   float* offB, int64_t ldB, int PP, int P, int N, int T, int L, hipStream_t stream) {
  const bool vec = L % 4 == 0 && ldA % 4 == 0 && ldB % 4 == 0 && aligned16(heads) && aligned16(eps) && aligned16(mask) &&
                   aligned16(offA) && aligned16(offB);
  if (vec)
    hipLaunchKernelGGL(gather_offsets_kernel<4>, dim3(N, PP), dim3(64), 0, stream, heads, eps, mask, plan.ell_idx,
                       plan.ell_val, plan.kmax, offA, ldA, offB, ldB, P, N, T, plan.Tc, plan.tpos, L);
  else
    hipLaunchKernelGGL(gather_offsets_kernel<1>, dim3(N, PP), dim3(64), 0, stream, heads, eps, mask, plan.ell_idx,
                       plan.ell_val, plan.kmax, offA, ldA, offB, ldB, P, N, T, plan.Tc, plan.tpos, L);
  CRADLE_CHECK_LAUNCH("gather_offsets_kernel");
  return CRADLE_OK;
}

// =====================================================================================
// K5 basal-encoder first layer without materialising [x | offset | qc offset] (P,N,G+2L)
// (correlated_normal_guide.py:196-198): pre = xW[n] + offW[pp,n] + sum_q s_q vq[pp,q] + b, where
// xW = logstd(x) Wx^T is computed once per step for X and cfX, offW = offset W_off^T,
// vq = qc_E W_qc^T.  Three variants, rows of h: v*P*N + p*N + n
//   v=0 main pass q_z_basal (s = qc), v=1 main pass q_z_cf_basal (s = 1-qc),
//   v=2 counterfactual pass q_z_basal (x = cfX, particles P..2P-1, s = 1-qc)
// =====================================================================================
template <int V>
__global__ __launch_bounds__(128) void enc_combine_fwd_kernel(const float* __restrict__ xW, const float* __restrict__ offW,
                                                              const float* __restrict__ vq, const float* __restrict__ bias,
                                                              const float* __restrict__ qc, float* __restrict__ h, int P,
                                                              int N, int Q, int H) {
  const int n = blockIdx.x, p = blockIdx.y, v = blockIdx.z;
  const int pp = v == 2 ? P + p : p;
  const float* xr = xW + (int64_t)(v == 2 ? N + n : n) * H;
  const float* orow = offW + ((int64_t)pp * N + n) * H;
  float* out = h + (((int64_t)v * P + p) * N + n) * H;
  for (int j = threadIdx.x * V; j < H; j += 128 * V) {
    float a[V], o[V], b[V];
    ldv<V>(xr + j, a);
    ldv<V>(orow + j, o);
    ldv<V>(bias + j, b);
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] += o[e];
    for (int q = 0; q < Q; ++q) {
      const float sq = v == 0 ? qc[(int64_t)n * Q + q] : 1.0f - qc[(int64_t)n * Q + q];
      float w[V];
      ldv<V>(vq + ((int64_t)pp * Q + q) * H + j, w);
#pragma unroll
      for (int e = 0; e < V; ++e) a[e] = fmaf(sq, w[e], a[e]);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      a[e] += b[e];
      a[e] = a[e] > 0.f ? a[e] : kLeakySlope * a[e];
    }
    stv<V>(out + j, a);
  }
}

int enc_combine_fwd(const float* xW, const float* offW, const float* vq, const float* bias, const float* qc, float* h,
                    int P, int N, int Q, int H, hipStream_t stream) {
  const bool vec = H % 4 == 0 && aligned16(xW) && aligned16(offW) && aligned16(vq) && aligned16(bias) && aligned16(h);
  if (vec)
    hipLaunchKernelGGL(enc_combine_fwd_kernel<4>, dim3(N, P, 3), dim3(128), 0, stream, xW, offW, vq, bias, qc, h, P, N, Q, H);
  else
    hipLaunchKernelGGL(enc_combine_fwd_kernel<1>, dim3(N, P, 3), dim3(128), 0, stream, xW, offW, vq, bias, qc, h, P, N, Q, H);
  CRADLE_CHECK_LAUNCH("enc_combine_fwd_kernel");
  return CRADLE_OK;
}

// dpre (3PN, H) -> doffW (2P*N, H), dxW (2N, H)
template <int V>
__global__ __launch_bounds__(128) void enc_combine_bwd_kernel(const float* __restrict__ dpre, float* __restrict__ doffW,
                                                              float* __restrict__ dxW, int P, int N, int H) {
  const int n = blockIdx.x;
  const int64_t PN = (int64_t)P * N;
  for (int j = threadIdx.x * V; j < H; j += 128 * V) {
    float sa[V], sb[V];
#pragma unroll
    for (int e = 0; e < V; ++e) sa[e] = sb[e] = 0.f;
    for (int p = 0; p < P; ++p) {
      const int64_t r = (int64_t)p * N + n;
      float a[V], a2[V], b[V];
      ldv<V>(dpre + r * H + j, a);
      ldv<V>(dpre + (PN + r) * H + j, a2);
      ldv<V>(dpre + (2 * PN + r) * H + j, b);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        a[e] += a2[e];
        sa[e] += a[e];
        sb[e] += b[e];
      }
      stv<V>(doffW + r * H + j, a);
      stv<V>(doffW + (PN + r) * H + j, b);
    }
    stv<V>(dxW + (int64_t)n * H + j, sa);
    stv<V>(dxW + (int64_t)(N + n) * H + j, sb);
  }
}

// dvq[pp,q,:] = sum_n s_q[n] dpre[variant rows].  A workgroup owns 64 columns (16 lanes x float4 = one 256-byte
// segment of every row) of one (pp, q); its 16 row slices walk the cells with independent 16-byte loads and meet
// in LDS in a fixed order (deterministic).  VEC = false: the scalar form for H % 4 != 0 (16 columns per workgroup).
template <bool VEC>
__global__ __launch_bounds__(256) void enc_combine_bwd_vq_kernel(const float* __restrict__ dpre, const float* __restrict__ qc,
                                                                 float* __restrict__ dvq, int P, int N, int Q, int H) {
  constexpr int SL = 16;  // row slices per workgroup
  constexpr int W = VEC ? 4 : 1;
  __shared__ float part[SL][16 * W + 1];
  const int q = blockIdx.y % Q, pp = blockIdx.y / Q;
  const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int j = (blockIdx.x * 16 + c) * W;
  const int64_t PN = (int64_t)P * N;
  float acc[W];
#pragma unroll
  for (int e = 0; e < W; ++e) acc[e] = 0.f;
  if (j < H) {
    if (pp < P) {
      const float* a = dpre + ((int64_t)pp * N) * H + j;
      const float* b = a + PN * H;
      int n = sl;
      for (; n + 3 * SL < N; n += 4 * SL) {  // four cells (eight rows) in flight
        float va[4][W], vb[4][W], sq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ldv<W>(a + (int64_t)(n + i * SL) * H, va[i]);
          ldv<W>(b + (int64_t)(n + i * SL) * H, vb[i]);
          sq[i] = qc[(int64_t)(n + i * SL) * Q + q];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < W; ++e) acc[e] += sq[i] * va[i][e] + (1.0f - sq[i]) * vb[i][e];
      }
      for (; n < N; n += SL) {
        float va[W], vb[W];
        ldv<W>(a + (int64_t)n * H, va);
        ldv<W>(b + (int64_t)n * H, vb);
        const float sq = qc[(int64_t)n * Q + q];
#pragma unroll
        for (int e = 0; e < W; ++e) acc[e] += sq * va[e] + (1.0f - sq) * vb[e];
      }
    } else {
      const float* a = dpre + (2 * PN + (int64_t)(pp - P) * N) * H + j;
      int n = sl;
      for (; n + 7 * SL < N; n += 8 * SL) {
        float va[8][W], sq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          ldv<W>(a + (int64_t)(n + i * SL) * H, va[i]);
          sq[i] = qc[(int64_t)(n + i * SL) * Q + q];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int e = 0; e < W; ++e) acc[e] += (1.0f - sq[i]) * va[i][e];
      }
      for (; n < N; n += SL) {
        float va[W];
        ldv<W>(a + (int64_t)n * H, va);
        const float sq = qc[(int64_t)n * Q + q];
#pragma unroll
        for (int e = 0; e < W; ++e) acc[e] += (1.0f - sq) * va[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < W; ++e) part[sl][c * W + e] = acc[e];
  __syncthreads();
  if (sl == 0 && j < H) {
    float t[W];
#pragma unroll
    for (int e = 0; e < W; ++e) {
      t[e] = 0.f;
#pragma unroll
      for (int k = 0; k < SL; ++k) t[e] += part[k][c * W + e];
    }
    stv<W>(dvq + ((int64_t)pp * Q + q) * H + j, t);
  }
}

// which: 1 = the (doffW, dxW) kernel on `stream`, 2 = the dvq kernel on `vq_stream`, 3 = both
int enc_combine_bwd(const float* dpre, const float* qc, float* doffW, float* dxW, float* dvq, int P, int N, int Q, int H,
                    hipStream_t stream, hipStream_t vq_stream, int which) {
  if (which & 1) {
    if (H % 4 == 0 && aligned16(dpre) && aligned16(doffW) && aligned16(dxW))
      hipLaunchKernelGGL(enc_combine_bwd_kernel<4>, dim3(N), dim3(128), 0, stream, dpre, doffW, dxW, P, N, H);
    else
      hipLaunchKernelGGL(enc_combine_bwd_kernel<1>, dim3(N), dim3(128), 0, stream, dpre, doffW, dxW, P, N, H);
  }
  if (which & 2) {
    if (H % 4 == 0 && aligned16(dpre) && aligned16(dvq))
      hipLaunchKernelGGL(enc_combine_bwd_vq_kernel<true>, dim3((H / 4 + 15) / 16, 2 * P * Q), dim3(256), 0, vq_stream, dpre,
                         qc, dvq, P, N, Q, H);
    else
      hipLaunchKernelGGL(enc_combine_bwd_vq_kernel<false>, dim3((H + 15) / 16, 2 * P * Q), dim3(256), 0, vq_stream, dpre,
                         qc, dvq, P, N, Q, H);
  }
  CRADLE_CHECK_LAUNCH("enc_combine_bwd");
  return CRADLE_OK;
}

// =====================================================================================
// small helpers
// =====================================================================================
// out[c, r] = W[r, col0 + c]: the one-hot block of the embedding encoder's first layer
// (correlated_normal_guide.py:147-153) as a (T, H) table that the GEMM epilogue adds by row.
__global__ __launch_bounds__(256) void transpose_cols_kernel(const float* __restrict__ W, int64_t ldw, int rows, int col0,
                                                             int cols, float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? W[(int64_t)r * ldw + col0 + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) out[(int64_t)c * rows + r] = tile[tx][i];
  }
}

int transpose_cols(const float* W, int64_t ldw, int rows, int col0, int cols, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(transpose_cols_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, stream, W, ldw, rows,
                     col0, cols, out);
  CRADLE_CHECK_LAUNCH("transpose_cols_kernel");
  return CRADLE_OK;
}

// out[j, r] = W[r, col0 + tmap[j]] (0 for unused slots): the one-hot block by slot
__global__ __launch_bounds__(256) void gather_cols_kernel(const float* __restrict__ W, int64_t ldw, int rows, int col0,
                                                          const int* __restrict__ tmap, float* __restrict__ out) {
  const int j = blockIdx.x;
  const int t = tmap[j];
  for (int r = threadIdx.x; r < rows; r += 256) out[(int64_t)j * rows + r] = t >= 0 ? W[(int64_t)r * ldw + col0 + t] : 0.f;
}

int gather_cols(const float* W, int64_t ldw, int rows, int col0, int Tc, const int* tmap, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(gather_cols_kernel, dim3(Tc), dim3(256), 0, stream, W, ldw, rows, col0, tmap, out);
  CRADLE_CHECK_LAUNCH("gather_cols_kernel");
  return CRADLE_OK;
}

// dW[h, col0 + t] = sum_groups da[g*Tc + slot(t), h]   (gradient of the one-hot block; 0 for absent treatments)
__global__ __launch_bounds__(256) void sum_groups_to_cols_kernel(const float* __restrict__ da, int groups, int T, int Tc,
                                                                 const int* __restrict__ tpos, int H, float* __restrict__ dW,
                                                                 int64_t lddw, int col0) {
  __shared__ float tile[32][33];
  const int t0 = blockIdx.x * 32, h0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, h = h0 + tx;
    float s = 0.f;
    if (t < T && h < H) {
      const int slot = tpos ? tpos[t] : t;
      if (slot >= 0) {
        const float* src = da + (int64_t)slot * H + h;
        const int64_t gs = (int64_t)Tc * H;
        int g = 0;
        for (; g + 3 < groups; g += 4) {  // four loads in flight, summed in order
          const float v0 = src[g * gs], v1 = src[(g + 1) * gs], v2 = src[(g + 2) * gs], v3 = src[(g + 3) * gs];
          s += v0; s += v1; s += v2; s += v3;
        }
        for (; g < groups; ++g) s += src[g * gs];
      }
    }
    tile[i][tx] = s;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int h = h0 + i, t = t0 + tx;
    if (h < H && t < T) dW[(int64_t)h * lddw + col0 + t] = tile[tx][i];
  }
}

int sum_over_groups_to_cols(const float* da, int groups, int T, int Tc, const int* tpos, int H, float* dW, int64_t lddw,
                            int col0, hipStream_t stream) {
  hipLaunchKernelGGL(sum_groups_to_cols_kernel, dim3((T + 31) / 32, (H + 31) / 32), dim3(256), 0, stream, da, groups, T, Tc,
                     tpos, H, dW, lddw, col0);
  CRADLE_CHECK_LAUNCH("sum_groups_to_cols_kernel");
  return CRADLE_OK;
}

// out[p, :] = sum_n w[n] da[p*N + n, :]   (gradient of the per-particle bad-variant shift)
__global__ __launch_bounds__(256) void weighted_rowgroup_sum_kernel(const float* __restrict__ da, const float* __restrict__ w,
                                                                    int N, int H, float* __restrict__ out) {
  // 16 columns x 16 row slices per block (256 threads: runs on a side stream beside the backward GEMMs)
  constexpr int SL = 16;
  __shared__ float part[SL][17];
  const int p = blockIdx.y;
  const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + c;
  float acc = 0.f;
  if (j < H) {
    const float* src = da + (int64_t)p * N * H + j;
    int n = sl;
    for (; n + 7 * SL < N; n += 8 * SL) {  // eight rows in flight, accumulated in order
      float v[8], wn[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[i] = src[(int64_t)(n + i * SL) * H];
        wn[i] = w[n + i * SL];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = fmaf(wn[i], v[i], acc);
    }
    for (; n < N; n += SL) acc = fmaf(w[n], src[(int64_t)n * H], acc);
  }
  part[sl][c] = acc;
  __syncthreads();
  if (sl == 0 && j < H) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < SL; ++k) t += part[k][c];
    out[(int64_t)p * H + j] = t;
  }
}

int weighted_rowgroup_sum(const float* da, const float* w, int P, int N, int H, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(weighted_rowgroup_sum_kernel, dim3((H + 15) / 16, P), dim3(256), 0, stream, da, w, N, H, out);
  CRADLE_CHECK_LAUNCH("weighted_rowgroup_sum_kernel");
  return CRADLE_OK;
}

// loss_modules.py:179-185: loss = -mean_{p,n}(sum of terms) + beta*KL; metrics = -mean of each term.
// out: [0] loss, [1] reconstruction, [2] mask, [3] E, [4] qc_E, [5] z_basal, [6] cf_KLdivergence
// (256 threads: a single small workgroup finds a free slot beside the GEMMs it shares the chip with; a
// 1024-thread one waited 170 us for a whole CU)
__global__ __launch_bounds__(256) void finalize_loss_kernel(const float* __restrict__ recon, const float* __restrict__ wrec,
                                                             const float* __restrict__ ltz, const float* __restrict__ klrow,
                                                             const float* __restrict__ Smask, const float* __restrict__ SE,
                                                             const float* __restrict__ Sq, const float* __restrict__ cw,
                                                             const float* __restrict__ cwq, float beta, int P, int N, int T,
                                                             int Q, const int* __restrict__ flags, float* __restrict__ out) {
  __shared__ float red[6][4];
  float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int64_t PN = (int64_t)P * N;
  for (int64_t r = threadIdx.x; r < PN; r += 256) {
    a[0] += wrec[r % N] * recon[r];
    a[4] += ltz[r];
    a[5] += klrow[r];
  }
  for (int r = threadIdx.x; r < P * T; r += 256) {
    a[1] += cw[r % T] * Smask[r];
    a[2] += cw[r % T] * SE[r];
  }
  for (int r = threadIdx.x; r < P * Q; r += 256) a[3] += cwq[r % Q] * Sq[r];
  float tot[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) tot[k] = block_sum<256>(a[k], red[k]);
  if (threadIdx.x == 0) {
    const float inv = 1.0f / (float)PN;
    float loss = 0.f;
    for (int k = 0; k < 5; ++k) {
      out[1 + k] = -tot[k] * inv;
      loss += out[1 + k];
    }
    const float kl = beta * tot[5] / (float)P;
    out[6] = kl;
    // an input the step cannot score (flags != 0) must not pass for a loss: NaN until the caller has dealt with it
    const bool rejected = *flags != 0;
    out[0] = rejected ? __int_as_float(0x7fc00000) : loss + kl;
    if (rejected) out[7] += 1.0f;  // sticky count for the host's occasional look (the flags word is per step)
  }
}

int finalize_loss(const float* recon, const float* wrec, const float* ltz, const float* klrow, const float* Smask,
                  const float* SE, const float* Sq, const float* cw, const float* cwq, float beta, int P, int N, int T,
                  int Q, const int* flags, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(finalize_loss_kernel, dim3(1), dim3(256), 0, stream, recon, wrec, ltz, klrow, Smask, SE, Sq, cw, cwq,
                     beta, P, N, T, Q, flags, out);
  CRADLE_CHECK_LAUNCH("finalize_loss_kernel");
  return CRADLE_OK;
}

}  // namespace cradle


extern "C" {

int cradle_mask_sample(const float* logits, const float* u, float tau, float* mask, int P, int Ts, int L,
                       hipStream_t stream) {
  using namespace cradle;
  CRADLE_CHECK_ARG(logits && u && mask && P > 0 && Ts > 0 && L > 0 && tau > 0.f, "mask_sample: bad arguments");
  const int64_t TsL = (int64_t)Ts * L, n = (int64_t)P * TsL;
  if (L % 4 == 0 && aligned16(logits) && aligned16(u) && aligned16(mask))
    hipLaunchKernelGGL(mask_sample_kernel<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, logits, u, tau, mask,
                       n, TsL);
  else
    hipLaunchKernelGGL(mask_sample_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, logits, u, tau, mask, n,
                       TsL);
  CRADLE_CHECK_LAUNCH("mask_sample_kernel");
  return CRADLE_OK;
}

int cradle_gauss_sample(const float* mean_raw, int64_t ld_mean, const float* log_var_raw, int64_t ld_log_var,
                        const float* eps, float* out, int64_t rows, int L, hipStream_t stream) {
  using namespace cradle;
  CRADLE_CHECK_ARG(mean_raw && log_var_raw && eps && out && rows > 0 && L > 0 && ld_mean >= L && ld_log_var >= L,
                   "gauss_sample: bad arguments");
  const int64_t n = rows * L;
  if (L % 4 == 0 && ld_mean % 4 == 0 && ld_log_var % 4 == 0 && aligned16(mean_raw) && aligned16(log_var_raw) &&
      aligned16(eps) && aligned16(out))
    hipLaunchKernelGGL(gauss_sample_kernel<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, mean_raw, ld_mean,
                       log_var_raw, ld_log_var, eps, out, rows, L);
  else
    hipLaunchKernelGGL(gauss_sample_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, mean_raw, ld_mean,
                       log_var_raw, ld_log_var, eps, out, rows, L);
  CRADLE_CHECK_LAUNCH("gauss_sample_kernel");
  return CRADLE_OK;
}

}  // extern "C"
