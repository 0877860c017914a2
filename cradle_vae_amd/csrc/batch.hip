// K0: the batch gather of the device-resident data source.
//
// Reference: the DataLoader over TensorPerturbationDataset / SCRNASeqTensorPerturbationDataset
// (cradle_vae/data/utils/perturbation_dataset.py:44-89: X[idx], D[idx], qc[idx], cfX[idx], library_size[idx], all fp32,
// collated on the host and copied to the device every step).  Here the data set stays on the GPU in its compact form --
// counts as uint8 / int16 / int32 (they are integer-valued), D as <= kmax (treatment, dosage) pairs per cell -- and one
// launch gathers the rows of a shuffled index batch into the dense fp32 tensors of the batch contract; a second
// launch counts the distinct treatments in the batch (cradle_step_desc.max_present).
// HBM-bound: one workgroup per cell streams the cell's two count rows (G values each) and writes 8 G + 4 T bytes.
#include "common.h"
#include "kernels.h"

namespace cradle {

template <typename S>
__device__ __forceinline__ void convert_row(const S* __restrict__ src, float* __restrict__ dst, int G) {
  // 4 values per thread and access where the row allows it (row starts are G * sizeof(S) apart)
  if ((G & 3) == 0 && (((uintptr_t)src) % (4 * sizeof(S))) == 0 && (((uintptr_t)dst) & 15) == 0) {
    struct alignas(4 * sizeof(S)) S4 { S v[4]; };
    const S4* s4 = reinterpret_cast<const S4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int g = threadIdx.x; g < G / 4; g += blockDim.x) {
      const S4 t = s4[g];
      d4[g] = make_float4((float)t.v[0], (float)t.v[1], (float)t.v[2], (float)t.v[3]);
    }
  } else {
    for (int g = threadIdx.x; g < G; g += blockDim.x) dst[g] = (float)src[g];
  }
}

__device__ __forceinline__ void convert_any(const void* src, int type, int64_t row, float* dst, int G) {
  switch (type) {
    case CRADLE_COUNTS_F32: convert_row(reinterpret_cast<const float*>(src) + row * G, dst, G); break;
    case CRADLE_COUNTS_U8: convert_row(reinterpret_cast<const uint8_t*>(src) + row * G, dst, G); break;
    case CRADLE_COUNTS_I16: convert_row(reinterpret_cast<const int16_t*>(src) + row * G, dst, G); break;
    default: convert_row(reinterpret_cast<const int32_t*>(src) + row * G, dst, G); break;
  }
}

__global__ __launch_bounds__(256) void gather_batch_kernel(const cradle_batch_desc d) {
  const int n = blockIdx.x;
  int64_t row = d.pos[n];
  if (d.rows > 0 && (row < 0 || row >= d.rows)) {  // a stale or oversized index: reported, not followed
    if (threadIdx.x == 0 && d.flags) atomicOr(d.flags, 1);
    row = 0;
  }
  convert_any(d.X, d.x_type, row, d.X_out + (int64_t)n * d.G, d.G);
  convert_any(d.cfX, d.cfx_type, row, d.cfX_out + (int64_t)n * d.G, d.G);
  for (int q = threadIdx.x; q < d.Q; q += blockDim.x) d.qc_out[(int64_t)n * d.Q + q] = d.qc[row * d.Q + q];
  if (threadIdx.x == 0) d.lib_out[n] = d.library_size[row];
  float* Drow = d.D_out + (int64_t)n * d.T;
  if (d.d_dense) {
    convert_row(d.d_dense + row * d.T, Drow, d.T);
    return;
  }
  for (int t = threadIdx.x; t < d.T; t += blockDim.x) Drow[t] = 0.f;
  __syncthreads();  // the zeros of this workgroup are in place before its pairs land
  for (int k = threadIdx.x; k < d.kmax; k += blockDim.x) {
    const int t = d.d_idx[row * d.kmax + k];
    if (t >= 0 && t < d.T) Drow[t] = d.d_val[row * d.kmax + k];
  }
}

// (D != 0).any(0).sum() of the gathered batch from the (treatment, dosage) pairs: a bitmap over treatments in LDS,
// one workgroup (N * kmax pairs).
__global__ __launch_bounds__(1024) void count_present_kernel(const cradle_batch_desc d, int words) {
  extern __shared__ unsigned int bits[];
  __shared__ int total;
  for (int w = threadIdx.x; w < words; w += 1024) bits[w] = 0u;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < d.N * d.kmax; i += 1024) {
    int64_t r = d.pos[i / d.kmax];
    if (d.rows > 0 && (r < 0 || r >= d.rows)) r = 0;
    const int64_t e = r * d.kmax + i % d.kmax;
    const int t = d.d_idx[e];
    if (t >= 0 && t < d.T && d.d_val[e] != 0.f) atomicOr(&bits[t >> 5], 1u << (t & 31));
  }
  __syncthreads();
  int c = 0;
  for (int w = threadIdx.x; w < words; w += 1024) c += __popc(bits[w]);
  c = (int)wave_sum((float)c);  // <= 64 * 32 per wave: exact in fp32
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&total, c);
  __syncthreads();
  if (threadIdx.x == 0) *d.present_out = total;
}

// The same from dense stored dosages (N x T elements: 5 M at 512 x 10 000): a workgroup owns 256 treatments, its 1024
// threads walk the batch four cells at a time (256-column row segments: coalesced), and the workgroups' counts meet in
// present_out (cleared by the host side beforehand; integer adds: any order gives the same sum).
__global__ __launch_bounds__(1024) void count_present_dense_kernel(const cradle_batch_desc d) {
  __shared__ int seen[256];
  const int col = threadIdx.x & 255, sub = threadIdx.x >> 8;
  const int t = blockIdx.x * 256 + col;
  if (threadIdx.x < 256) seen[threadIdx.x] = 0;
  __syncthreads();
  bool any = false;
  if (t < d.T) {
    for (int n = sub; n < d.N; n += 4) {
      int64_t r = d.pos[n];
      if (d.rows > 0 && (r < 0 || r >= d.rows)) r = 0;
      any |= d.d_dense[r * d.T + t] != 0.f;
    }
  }
  if (any) seen[col] = 1;  // (every writer stores the same value)
  __syncthreads();
  if (threadIdx.x < 256) {
    const int c = (int)wave_sum((float)seen[threadIdx.x]);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(d.present_out, c);
  }
}

}  // namespace cradle

extern "C" int cradle_gather_batch(const cradle_batch_desc* d, hipStream_t stream) {
  using namespace cradle;
  CRADLE_CHECK_ARG(d, "gather_batch: null descriptor");
  CRADLE_CHECK_ARG(d->N > 0 && d->G > 0 && d->T > 0 && d->Q > 0, "gather_batch: N=%d G=%d T=%d Q=%d must be positive", d->N,
                   d->G, d->T, d->Q);
  CRADLE_CHECK_ARG(d->pos && d->X && d->cfX && d->qc && d->library_size && d->X_out && d->cfX_out && d->qc_out &&
                       d->lib_out && d->D_out,
                   "gather_batch: null pointer argument");
  CRADLE_CHECK_ARG((d->d_dense != nullptr) != (d->d_idx != nullptr && d->d_val != nullptr && d->kmax > 0),
                   "gather_batch: give either dense dosages or (index, value) pairs with kmax > 0");
  CRADLE_CHECK_ARG(d->x_type >= 0 && d->x_type <= 3 && d->cfx_type >= 0 && d->cfx_type <= 3, "gather_batch: bad count type");
  CRADLE_CHECK_ARG(d->rows >= 0, "gather_batch: rows=%lld is negative", (long long)d->rows);
  if (d->flags && hipMemsetAsync(d->flags, 0, sizeof(int), stream) != hipSuccess) {
    cradle_set_error("gather_batch: clearing the flags word failed");
    return CRADLE_ERR_HIP;
  }
  hipLaunchKernelGGL(gather_batch_kernel, dim3(d->N), dim3(256), 0, stream, *d);
  CRADLE_CHECK_LAUNCH("gather_batch_kernel");
  if (d->present_out) {
    if (d->d_dense) {
      if (hipMemsetAsync(d->present_out, 0, sizeof(int), stream) != hipSuccess) {
        cradle_set_error("gather_batch: clearing the treatment count failed");
        return CRADLE_ERR_HIP;
      }
      hipLaunchKernelGGL(count_present_dense_kernel, dim3((d->T + 255) / 256), dim3(1024), 0, stream, *d);
      CRADLE_CHECK_LAUNCH("count_present_dense_kernel");
    } else {
      const int words = (d->T + 31) / 32;
      CRADLE_CHECK_ARG((size_t)words * 4 <= 60 * 1024, "gather_batch: n_treatments=%d too many for the presence bitmap", d->T);
      hipLaunchKernelGGL(count_present_kernel, dim3(1), dim3(1024), (size_t)words * 4, stream, *d, words);
      CRADLE_CHECK_LAUNCH("count_present_kernel");
    }
  }
  return CRADLE_OK;
}
