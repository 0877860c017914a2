// K6b: the same GEMM (gemm.hip: three operand layouts, fused MLP epilogues, split-K) on the bf16 matrix
// cores with split operands -- an OPTIONAL precision mode (cradle_gemm_desc.precision = 1,
// cradle_step_desc.precision = 1), never the default: the reference computes in fp32 and so does the
// headline path.
//
// Every fp32 operand element x is split once, when its tile is staged into LDS, into
//   hi = bf16(x), lo = bf16(x - hi)            (x = hi + lo up to ~2^-17 |x|)
// and a product is evaluated as  a_hi b_hi + a_hi b_lo + a_lo b_hi  with v_mfma_f32_32x32x16_bf16 (fp32
// accumulate); the dropped a_lo b_lo term and the split residuals are ~2^-17 relative to |a||b| each.  Three bf16
// MFMAs of K = 16 replace eight fp32 MFMAs of K = 2 at 1/16 of their cycles per flop: ~5x fewer matrix-pipe
// cycles per output, at a dot-product error of ~1e-6..1e-5 relative instead of ~1e-7.
//
// Tile: (64 TM) x (64 TN), 256 threads = 2 x 2 waves of (32 TM) x (32 TN), K step 32.  LDS holds, per operand
// and stage, a hi and a lo plane of 16-byte slots [c][r ^ 4c] (c = 8-wide k chunk 0..3, r = tile row): a
// fragment read (16 lanes = 16 consecutive rows of one chunk) and both staging patterns below touch every bank
// once.  Staging:
//   K-major operand  [row][k]: 16-byte loads, 8 lanes per row (coalesced), 8-byte LDS stores;
//   MN-major operand [k][row]: 4-byte loads, lanes along the rows (coalesced), eight k values per lane,
//                              16-byte LDS stores -- the transpose happens in registers, no conflicts.
#include <hip/hip_bf16.h>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace cradle {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kBK = 32;  // K step
constexpr int kChunks = kBK / 8;

__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

__device__ __forceinline__ int slot_of(int c, int r, int rows) { return c * rows + (r ^ (4 * c)); }

template <bool KMAJOR, int ROWS>
struct SplitLoader {
  // K-major: ROWS*8 float4 per step over 256 threads; MN-major: ROWS*4 items of 8 floats
  static constexpr int NV = KMAJOR ? ROWS * 8 / 256 : (ROWS * kChunks / 256) * 2;
  static_assert(NV >= 1, "tile too small for 256 threads");
  float4 regs[NV];

  // Range-checked raw buffer loads: an element outside the matrix reads zero (offset pushed past the
  // descriptor's extent), no branch around any load.  Requires 16-byte aligned pointers / leading dimensions
  // and contiguous extents that are multiples of 4 (checked on the host).
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, uint32_t bytes, int row0, int k0, int rows,
                                       int kdim) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, bytes, 0x00020000);
    if (KMAJOR) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int idx = v * 256 + threadIdx.x;
        const int r = row0 + idx / 8, k = k0 + (idx % 8) * 4;
        const uint32_t off = (uint32_t)(((int64_t)r * ld + k) * 4);
        const bool in = r < rows && k < kdim;
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, in ? off : 0x80000000u, 0, 0);
        regs[v] = __builtin_bit_cast(float4, t);
      }
    } else {
#pragma unroll
      for (int it = 0; it < NV / 2; ++it) {
        const int id = it * 256 + threadIdx.x;
        const int c = id / ROWS, r = row0 + id % ROWS;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = k0 + 8 * c + j;
          const uint32_t off = (uint32_t)(((int64_t)k * ld + r) * 4);
          const bool in = r < rows && k < kdim;
          f[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, in ? off : 0x80000000u, 0, 0));
        }
        regs[2 * it] = make_float4(f[0], f[1], f[2], f[3]);
        regs[2 * it + 1] = make_float4(f[4], f[5], f[6], f[7]);
      }
    }
  }

  // ones_row >= 0 (MN-major B of a weight gradient only): logical row whose value is 1 for every k < kdim.
  // SPLIT = false (plain bf16 mode): only the rounded values are staged
  template <bool SPLIT = true>
  __device__ __forceinline__ void store(u32x4* __restrict__ hi_plane, u32x4* __restrict__ lo_plane, int row0, int k0,
                                        int kdim, int ones_row) const {
    if (KMAJOR) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int idx = v * 256 + threadIdx.x;
        const int r = idx / 8, c4 = idx % 8;
        const float x[4] = {regs[v].x, regs[v].y, regs[v].z, regs[v].w};
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          __bf16 a, b;
          split_bf16(x[e], a, b);
          h[e] = a;
          l[e] = b;
        }
        const int s = slot_of(c4 / 2, r, ROWS);
        reinterpret_cast<u32x2*>(hi_plane + s)[c4 & 1] = __builtin_bit_cast(u32x2, h);
        if (SPLIT) reinterpret_cast<u32x2*>(lo_plane + s)[c4 & 1] = __builtin_bit_cast(u32x2, l);
      }
    } else {
#pragma unroll
      for (int it = 0; it < NV / 2; ++it) {
        const int id = it * 256 + threadIdx.x;
        const int c = id / ROWS, r = id % ROWS;
        float x[8] = {regs[2 * it].x,     regs[2 * it].y,     regs[2 * it].z,     regs[2 * it].w,
                      regs[2 * it + 1].x, regs[2 * it + 1].y, regs[2 * it + 1].z, regs[2 * it + 1].w};
        if (ones_row >= 0 && row0 + r == ones_row) {
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = (k0 + 8 * c + j < kdim) ? 1.f : 0.f;
        }
        bf16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          __bf16 a, b;
          split_bf16(x[j], a, b);
          h[j] = a;
          l[j] = b;
        }
        const int s = slot_of(c, r, ROWS);
        hi_plane[s] = __builtin_bit_cast(u32x4, h);
        if (SPLIT) lo_plane[s] = __builtin_bit_cast(u32x4, l);
      }
    }
  }
};

// SPLIT = true: three products per element pair (bf16x3, ~fp32 accuracy); false: one product of the rounded operands
// (plain bf16 inputs, fp32 accumulation: ~3 significant digits per product)
template <bool A_KMAJOR, bool B_KMAJOR, int TM, int TN, bool SPLIT>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(const GemmParams p) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  using LA = SplitLoader<A_KMAJOR, BM>;
  using LB = SplitLoader<B_KMAJOR, BN>;
  constexpr int PA = kChunks * BM, PB = kChunks * BN;  // 16-byte slots per plane
  constexpr int STAGE = 2 * PA + 2 * PB;
  __shared__ u32x4 lds[2 * STAGE];
  auto Ahi = [&](int buf) { return lds + buf * STAGE; };
  auto Alo = [&](int buf) { return lds + buf * STAGE + PA; };
  auto Bhi = [&](int buf) { return lds + buf * STAGE + 2 * PA; };
  auto Blo = [&](int buf) { return lds + buf * STAGE + 2 * PA + PB; };

  // XCD-aware tile order (as in gemm_f32_kernel)
  const int tiles_m = (p.M + BM - 1) / BM;
  const int Np = p.N + (p.extra_col ? 1 : 0);
  const int tiles_n = (Np + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, pos = wg / 8;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  const int tile_m = wg / tiles_n, tile_n = wg % tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * BN;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int k_tiles = (p.K + kBK - 1) / kBK;
  int kt_begin = 0, kt_end = k_tiles;
  if (p.splitk > 1) {
    kt_begin = blockIdx.z * p.k_tiles_per_split;
    kt_end = min(k_tiles, kt_begin + p.k_tiles_per_split);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // register prefetch ring of two K steps, loads issued in ring order (see gemm.hip)
  LA la[2];
  LB lb[2];
  const int ones_row = p.extra_col ? p.N : -1;
  const int ones_b = B_KMAJOR ? -1 : ones_row;
  const int k_limit = min(p.K, kt_end * kBK);
  const int n_steps = (kt_end - kt_begin + 1) / 2 * 2;
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    la[st].load(p.A, p.lda, p.a_bytes, row0, (kt_begin + st) * kBK, p.M, k_limit);
    lb[st].load(p.B, p.ldb, p.b_bytes, col0, (kt_begin + st) * kBK, p.N, k_limit);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int kt = kt_begin; kt < kt_begin + n_steps; kt += 2) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int cur = st;
      la[st].template store<SPLIT>(Ahi(cur), Alo(cur), row0, (kt + st) * kBK, k_limit, -1);
      lb[st].template store<SPLIT>(Bhi(cur), Blo(cur), col0, (kt + st) * kBK, k_limit, ones_b);
      la[st].load(p.A, p.lda, p.a_bytes, row0, (kt + st + 2) * kBK, p.M, k_limit);
      lb[st].load(p.B, p.ldb, p.b_bytes, col0, (kt + st + 2) * kBK, p.N, k_limit);
      __syncthreads();
#pragma unroll
      for (int kg = 0; kg < kBK / 16; ++kg) {
        const int c = kg * 2 + lh;
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int s = slot_of(c, (wm * TM + i) * 32 + li, BM);
          ah[i] = __builtin_bit_cast(bf16x8, Ahi(cur)[s]);
          if (SPLIT) al[i] = __builtin_bit_cast(bf16x8, Alo(cur)[s]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int s = slot_of(c, (wn * TN + j) * 32 + li, BN);
          bh[j] = __builtin_bit_cast(bf16x8, Bhi(cur)[s]);
          if (SPLIT) bl[j] = __builtin_bit_cast(bf16x8, Blo(cur)[s]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (SPLIT) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
      // the buffer written two steps from now is the one read here: a barrier per step is enough, as the
      // store of step st+2 comes after the barrier of step st+1, which every wave reaches after its reads
    }
  }

  // ---- epilogue.  acc element e of lane: row = (e&3) + 8*(e>>2) + 4*lh, col = li
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = col0 + (wn * TN + j) * 32 + li;
      if (col >= Np) continue;
      if (p.splitk > 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (row < p.M) p.partial[(int64_t)blockIdx.z * p.slab + (int64_t)row * Np + col] = acc[i][j][e];
        }
        continue;
      }
      constexpr int CH = 4;
#pragma unroll
      for (int e0 = 0; e0 < 16; e0 += CH) {
        int rows[CH], cols[CH];
        bool ok[CH];
        float v[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          rows[e] = row0 + (wm * TM + i) * 32 + ((e0 + e) & 3) + 8 * ((e0 + e) >> 2) + 4 * lh;
          cols[e] = col;
          ok[e] = rows[e] < p.M;
          v[e] = acc[i][j][e0 + e];
        }
        epilogue_batch<CH>(p, rows, cols, ok, v);
      }
    }
}

// Launch for a prepared GemmParams (splitk, slab, partial, a_bytes, b_bytes filled by gemm_f32).
template <bool SPLIT>
static bool launch_split(int layout, const GemmParams& p, int tm, int tn, hipStream_t s) {
  const int Np = p.N + (p.extra_col ? 1 : 0);
  const int bm = 64 * tm, bn = 64 * tn;
  dim3 grid((unsigned)(((p.M + bm - 1) / bm) * ((Np + bn - 1) / bn)), 1, (unsigned)p.splitk);
#define CRADLE_B3_CASE(TM_, TN_)                                                                                  \
  if (tm == TM_ && tn == TN_) {                                                                                   \
    if (layout == 0) hipLaunchKernelGGL((gemm_bf16x3_kernel<true, true, TM_, TN_, SPLIT>), grid, dim3(256), 0, s, p);    \
    else if (layout == 1) hipLaunchKernelGGL((gemm_bf16x3_kernel<true, false, TM_, TN_, SPLIT>), grid, dim3(256), 0, s, p); \
    else hipLaunchKernelGGL((gemm_bf16x3_kernel<false, false, TM_, TN_, SPLIT>), grid, dim3(256), 0, s, p);              \
    return true;                                                                                                  \
  }
  CRADLE_B3_CASE(2, 2)
  CRADLE_B3_CASE(2, 1)
  CRADLE_B3_CASE(1, 2)
  CRADLE_B3_CASE(1, 1)
#undef CRADLE_B3_CASE
  return false;
}

// precision 1: split (three products); 2: plain bf16 (one)
bool launch_gemm_bf16x3(int layout, const GemmParams& p, int tm, int tn, hipStream_t s) {
  return p.precision == 2 ? launch_split<false>(layout, p, tm, tn, s) : launch_split<true>(layout, p, tm, tn, s);
}

}  // namespace cradle
