// Input preparation in front of the hot path, on the device:
//
//  * cradle_log_stats   -- per-gene sums of y = log(x + 1) and y^2 around a per-gene shift over a slab of cells: the
//    train-set statistics of cradle_vae/data/utils/batch_statistics.py:4-23 (batch_log_mean / batch_log_std; the
//    reference walks the matrix twice in host-sized steps) as ONE streaming pass over the stored counts.
//  * cradle_obs_counts_* -- (M != 0).sum(0) of the dosage / qc matrices (perturbation_dataset.py:64-67).
//  * cradle_peer_median -- the counterfactual input of a cell = per-gene lower median over its <= 50 sampled QC-failing
//    peers (reference _get_cfX, cradle_vae/data/norman/data_module.py:175-191: X[idx].median(0) per cell inside a
//    pandas apply); here one launch for all cells of a split.
//
// All three are HBM / L2 streaming kernels over integer-valued count matrices (fp32, uint8, int16 or int32 storage).
#include "common.h"
#include "kernels.h"

namespace cradle {

template <typename S>
struct alignas(4 * sizeof(S)) Vec4 {
  S v[4];
};

// ---------------------------------------------------------------------------------------------- log statistics
// grid (tiles of 256 genes, row slabs).  A wave covers 256 consecutive genes (4 per lane, one
// 4-element vector load per row); the four waves of a workgroup take rows r, r+4, ... of the slab and meet in LDS.
// Sums are kept in fp64: 4 M cells x y^2 <= 1e2 needs 9 digits before the first rounding shows in fp32.
constexpr int kStatGenesPerBlock = 256;

template <typename S>
__global__ __launch_bounds__(256) void log_stats_kernel(const S* __restrict__ X, int64_t rows, int G, int64_t ld,
                                                        const float* __restrict__ shift, int rows_per_slab,
                                                        double* __restrict__ part) {
  __shared__ double red[3][2][4][64];  // [wave - 1][sum kind][element][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g0 = blockIdx.x * kStatGenesPerBlock + lane * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t r1 = r0 + rows_per_slab < rows ? r0 + rows_per_slab : rows;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  float sh[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) sh[e] = (shift && g0 + e < G) ? shift[g0 + e] : 0.f;
  const bool vec = (ld % 4 == 0) && ((uintptr_t)X % (4 * sizeof(S)) == 0) && g0 + 3 < G;
  if (g0 < G) {
    for (int64_t r = r0 + wave; r < r1; r += 4) {
      float x[4] = {0.f, 0.f, 0.f, 0.f};
      const S* row = X + r * ld + g0;
      if (vec) {
        const Vec4<S> t = *reinterpret_cast<const Vec4<S>*>(row);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = (float)t.v[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (g0 + e < G) x[e] = (float)row[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = logf(x[e] + 1.0f) - sh[e];  // torch.log(X + 1) in fp32, as the reference computes it
        s1[e] += (double)y;
        s2[e] += (double)y * (double)y;
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[wave - 1][0][e][lane] = s1[e];
      red[wave - 1][1][e][lane] = s2[e];
    }
  }
  __syncthreads();
  if (wave == 0 && g0 < G) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      for (int w = 0; w < 3; ++w) {  // fixed order: deterministic
        s1[e] += red[w][0][e][lane];
        s2[e] += red[w][1][e][lane];
      }
      if (g0 + e < G) {
        part[((int64_t)blockIdx.y * 2 + 0) * G + g0 + e] = s1[e];
        part[((int64_t)blockIdx.y * 2 + 1) * G + g0 + e] = s2[e];
      }
    }
  }
}

// sums (2, G) (+)= sum over slabs of part (slabs, 2, G), in slab order
__global__ __launch_bounds__(256) void log_stats_reduce_kernel(const double* __restrict__ part, int slabs, int G,
                                                               double* __restrict__ sums, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * G) return;
  double acc = accumulate ? sums[i] : 0.0;
  for (int s = 0; s < slabs; ++s) acc += part[(int64_t)s * 2 * G + i];
  sums[i] = acc;
}

int stats_slabs(int64_t rows, int G) {
  const int tiles = (G + kStatGenesPerBlock - 1) / kStatGenesPerBlock;
  int64_t slabs = (2048 + tiles - 1) / tiles;  // ~2048 workgroups in flight
  if (slabs > (rows + 63) / 64) slabs = (rows + 63) / 64;  // at least 64 rows per slab
  if (slabs < 1) slabs = 1;
  if (slabs > 1024) slabs = 1024;
  return (int)slabs;
}

// ---------------------------------------------------------------------------------------------- observation counts
__global__ __launch_bounds__(256) void obs_counts_dense_kernel(const float* __restrict__ M, int64_t rows, int C,
                                                               unsigned long long* __restrict__ out) {
  // thread = one column of a 256-column tile, rows strided over gridDim.y: coalesced along the columns
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  unsigned long long n = 0;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) n += M[r * C + c] != 0.f;
  if (n) atomicAdd(&out[c], n);
}

__global__ __launch_bounds__(256) void obs_counts_pairs_kernel(const int* __restrict__ idx, const float* __restrict__ val,
                                                               int64_t n_pairs, int T, unsigned long long* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_pairs; i += (int64_t)gridDim.x * 256) {
    const int t = idx[i];
    if (t >= 0 && t < T && val[i] != 0.f) atomicAdd(&out[t], 1ull);
  }
}

// ---------------------------------------------------------------------------------------------- peer median
// One workgroup per cell; a thread owns one gene of a 256-gene tile, keeps its k peer values in an LDS column and finds
// the lower median (the element of rank (k - 1) / 2: what torch.median / the reference return) by bisection on the
// VALUE -- count(x <= mid) against the rank -- between the column's minimum and maximum: k (log2(range) + 2) steps
// instead of a k^2 rank count or a sort, and no step at all for the many genes whose peers all hold the same count
// (zeros, mostly).  Keys are order-preserving 32-bit integers of the stored values, so the result is an element of the
// column, bit for bit, for float storage too.
template <typename S>
__device__ __forceinline__ int32_t to_key(S x) {
  return (int32_t)x;
}
template <>
__device__ __forceinline__ int32_t to_key<float>(float x) {
  const int32_t b = __float_as_int(x);
  return b >= 0 ? b : (int32_t)(0x80000000u - (uint32_t)b);  // monotone; -0.0 -> 0
}
template <typename S>
__device__ __forceinline__ S from_key(int32_t k) {
  return (S)k;
}
template <>
__device__ __forceinline__ float from_key<float>(int32_t k) {
  return __int_as_float(k >= 0 ? k : (int32_t)(0x80000000u - (uint32_t)k));
}

template <typename S>
__global__ __launch_bounds__(256) void peer_median_kernel(const S* __restrict__ src, int64_t src_rows, int G,
                                                          const int* __restrict__ peers, int kmax, S* __restrict__ out,
                                                          int* __restrict__ flags) {
  extern __shared__ int32_t col[];  // [kmax][256]
  __shared__ int prow[256];
  __shared__ int kcell;
  const int n = blockIdx.x;
  if (threadIdx.x == 0) kcell = 0;
  __syncthreads();
  // peers of this cell: the leading entries >= 0 (the list ends at the first negative entry)
  if (threadIdx.x < kmax) {
    const int p = peers[(int64_t)n * kmax + threadIdx.x];
    prow[threadIdx.x] = p;
    if (p >= src_rows) atomicOr(flags, 1);  // a stale peer index: reported, never dereferenced
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int k = 0;
    while (k < kmax && prow[k] >= 0 && prow[k] < src_rows) ++k;
    kcell = k;
  }
  __syncthreads();
  const int k = kcell;
  S* orow = out + (int64_t)n * G;
  if (k == 0) {  // no QC-failing peer: "no counterfactual" = an all-zero row (data_module.py:186-189)
    for (int g = threadIdx.x; g < G; g += 256) orow[g] = (S)0;
    return;
  }
  const int rank = (k - 1) / 2;
  for (int g0 = 0; g0 < G; g0 += 256) {
    const int g = g0 + threadIdx.x;
    int32_t lo = 0x7fffffff, hi = (int32_t)0x80000000;
    if (g < G) {
      for (int i0 = 0; i0 < k; i0 += 8) {  // eight loads in flight per thread (rows of different peers)
        S v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i0 + u < k ? src[(int64_t)prow[i0 + u] * G + g] : (S)0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (i0 + u < k) {
            const int32_t key = to_key<S>(v[u]);
            col[(i0 + u) * 256 + threadIdx.x] = key;
            lo = key < lo ? key : lo;
            hi = key > hi ? key : hi;
          }
        }
      }
      // smallest key v with #(x <= v) >= rank + 1; lo..hi hold it, and it is attained by an element
      while (lo < hi) {
        const int32_t mid = lo + (int32_t)(((uint32_t)hi - (uint32_t)lo) >> 1);
        int cnt = 0;
        for (int i = 0; i < k; ++i) cnt += col[i * 256 + threadIdx.x] <= mid;
        if (cnt >= rank + 1) hi = mid;
        else lo = mid + 1;
      }
      orow[g] = from_key<S>(lo);
    }
  }
}

}  // namespace cradle

using namespace cradle;

extern "C" {

int64_t cradle_log_stats_workspace_doubles(int64_t rows, int G) {
  if (rows <= 0 || G <= 0) return -1;
  return (int64_t)stats_slabs(rows, G) * 2 * G;
}

int cradle_log_stats(const void* X, int x_type, int64_t rows, int G, int64_t ld, const float* shift, double* sums,
                     int accumulate, double* workspace, int64_t workspace_doubles, hipStream_t stream) {
  CRADLE_CHECK_ARG(X && sums && workspace && rows > 0 && G > 0 && ld >= G, "log_stats: bad arguments (rows=%lld G=%d ld=%lld)",
                   (long long)rows, G, (long long)ld);
  CRADLE_CHECK_ARG(x_type >= 0 && x_type <= 3, "log_stats: bad count type %d", x_type);
  const int slabs = stats_slabs(rows, G);
  CRADLE_CHECK_ARG(workspace_doubles >= (int64_t)slabs * 2 * G, "log_stats: workspace has %lld doubles, needs %lld",
                   (long long)workspace_doubles, (long long)slabs * 2 * G);
  const int rows_per_slab = (int)((rows + slabs - 1) / slabs);
  const dim3 grid((G + kStatGenesPerBlock - 1) / kStatGenesPerBlock, slabs);
  switch (x_type) {
    case CRADLE_COUNTS_F32:
      hipLaunchKernelGGL(log_stats_kernel<float>, grid, dim3(256), 0, stream, (const float*)X, rows, G, ld, shift, rows_per_slab, workspace);
      break;
    case CRADLE_COUNTS_U8:
      hipLaunchKernelGGL(log_stats_kernel<uint8_t>, grid, dim3(256), 0, stream, (const uint8_t*)X, rows, G, ld, shift, rows_per_slab, workspace);
      break;
    case CRADLE_COUNTS_I16:
      hipLaunchKernelGGL(log_stats_kernel<int16_t>, grid, dim3(256), 0, stream, (const int16_t*)X, rows, G, ld, shift, rows_per_slab, workspace);
      break;
    default:
      hipLaunchKernelGGL(log_stats_kernel<int32_t>, grid, dim3(256), 0, stream, (const int32_t*)X, rows, G, ld, shift, rows_per_slab, workspace);
      break;
  }
  CRADLE_CHECK_LAUNCH("log_stats_kernel");
  hipLaunchKernelGGL(log_stats_reduce_kernel, dim3((2 * G + 255) / 256), dim3(256), 0, stream, workspace, slabs, G, sums, accumulate);
  CRADLE_CHECK_LAUNCH("log_stats_reduce_kernel");
  return CRADLE_OK;
}

int cradle_obs_counts_dense(const float* M, int64_t rows, int C, int64_t* counts, hipStream_t stream) {
  CRADLE_CHECK_ARG(M && counts && rows >= 0 && C > 0, "obs_counts_dense: bad arguments");
  if (hipMemsetAsync(counts, 0, (size_t)C * 8, stream) != hipSuccess) {
    cradle_set_error("obs_counts_dense: memset failed");
    return CRADLE_ERR_HIP;
  }
  if (rows == 0) return CRADLE_OK;
  const int tiles = (C + 255) / 256;
  int64_t ry = (1024 + tiles - 1) / tiles;
  if (ry > rows) ry = rows;
  hipLaunchKernelGGL(obs_counts_dense_kernel, dim3(tiles, (unsigned)ry), dim3(256), 0, stream, M, rows, C,
                     reinterpret_cast<unsigned long long*>(counts));
  CRADLE_CHECK_LAUNCH("obs_counts_dense_kernel");
  return CRADLE_OK;
}

int cradle_obs_counts_pairs(const int* idx, const float* val, int64_t rows, int kmax, int T, int64_t* counts,
                            hipStream_t stream) {
  CRADLE_CHECK_ARG(idx && val && counts && rows >= 0 && kmax > 0 && T > 0, "obs_counts_pairs: bad arguments");
  if (hipMemsetAsync(counts, 0, (size_t)T * 8, stream) != hipSuccess) {
    cradle_set_error("obs_counts_pairs: memset failed");
    return CRADLE_ERR_HIP;
  }
  const int64_t n = rows * kmax;
  if (n == 0) return CRADLE_OK;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(obs_counts_pairs_kernel, dim3(blocks), dim3(256), 0, stream, idx, val, n, T,
                     reinterpret_cast<unsigned long long*>(counts));
  CRADLE_CHECK_LAUNCH("obs_counts_pairs_kernel");
  return CRADLE_OK;
}

int cradle_peer_median(const void* src, int x_type, int64_t src_rows, int G, const int* peers, int64_t n_cells, int kmax,
                       void* out, int* flags, hipStream_t stream) {
  CRADLE_CHECK_ARG(src && peers && out && flags && src_rows > 0 && G > 0 && n_cells >= 0,
                   "peer_median: bad arguments (src_rows=%lld G=%d n_cells=%lld)", (long long)src_rows, G, (long long)n_cells);
  CRADLE_CHECK_ARG(x_type >= 0 && x_type <= 3, "peer_median: bad count type %d", x_type);
  CRADLE_CHECK_ARG(kmax >= 1 && kmax <= 128, "peer_median: kmax=%d outside [1, 128] (the reference draws at most 50 peers)", kmax);
  CRADLE_CHECK_ARG(src_rows < (1ll << 31) && n_cells < (1ll << 31), "peer_median: more than 2^31 rows");
  if (n_cells == 0) return CRADLE_OK;
  const size_t lds = (size_t)kmax * 256 * 4;
  const dim3 grid((unsigned)n_cells);
#define CRADLE_PM(S)                                                                                                  \
  do {                                                                                                                \
    if (lds > 48 * 1024 &&                                                                                            \
        hipFuncSetAttribute((const void*)peer_median_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != \
            hipSuccess) {                                                                                             \
      cradle_set_error("peer_median: cannot reserve %zu bytes of LDS", lds);                                          \
      return CRADLE_ERR_HIP;                                                                                          \
    }                                                                                                                 \
    hipLaunchKernelGGL(peer_median_kernel<S>, grid, dim3(256), lds, stream, (const S*)src, src_rows, G, peers, kmax,  \
                       (S*)out, flags);                                                                               \
  } while (0)
  switch (x_type) {
    case CRADLE_COUNTS_F32: CRADLE_PM(float); break;
    case CRADLE_COUNTS_U8: CRADLE_PM(uint8_t); break;
    case CRADLE_COUNTS_I16: CRADLE_PM(int16_t); break;
    default: CRADLE_PM(int32_t); break;
  }
#undef CRADLE_PM
  CRADLE_CHECK_LAUNCH("peer_median_kernel");
  return CRADLE_OK;
}

}  // extern "C"
