// K6: fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, an fmaf
// chain per output element) with the MLP epilogues fused in.
//
// Replaces every torch.nn.Linear on the hot path (reference cradle_vae/models/utils/mlp.py:
// 76-80, 111-119 trunk layers with LeakyReLU + residual, :129-131 Gaussian heads, :153-154
// decoder output layer) and, for the backward pass, autograd's mm kernels.
//
// One kernel template, three operand layouts (no transposed copies are ever made):
//   NT  C[M,N] = A[M,K] . B[N,K]^T      forward of y = x W^T          (x rows, W rows: K contiguous)
//   NN  C[M,N] = A[M,K] . B[K,N]        dgrad   dx = dy W             (W rows: N contiguous)
//   TN  C[M,N] = A[K,M]^T . B[K,N]      wgrad   dW = dy^T x           (both: K is the row index)
// Tiles are staged global -> registers -> LDS (double buffered, one barrier per K step) in
// the layout they have in memory: "K-major" tiles [row][k] are read back with one
// ds_read_b128 per 4 k-values, "MN-major" tiles [k][row] with four ds_read_b32.  The k
// order inside a step is permuted identically for A and B, which a dot product allows.
// 256 threads = 4 waves in a 2x2 arrangement; each wave owns TM x TN tiles of 32x32.
#include "common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace cradle {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// K-major tiles [row][BK] are stored without padding; the 16-byte chunk c of row r sits at chunk
// position c ^ swz(r), swz(r) = (r / (64/BK)) % (BK/4).  A row is BK of the 64 LDS banks, so rows
// r, r + 64/BK, ... start at the same bank and the XOR spreads their chunks: both the ds_write_b128
// of the staging store (BK/4 lanes per row) and the ds_read_b128 of a fragment (one chunk of 16
// consecutive rows) then touch every bank once.  (Padding the rows to BK+4 made the reads
// conflict-free but not the writes: SQ_LDS_BANK_CONFLICT equalled SQ_ACTIVE_INST_LDS.)
template <int BK>
__device__ __forceinline__ int swz_chunk(int r, int c) {
  constexpr int C = BK / 4, WRAP = 64 / BK > 0 ? 64 / BK : 1;
  return c ^ ((r / WRAP) % C);
}

template <bool KMAJOR, int ROWS, int BK, int NT>
struct TileLoader {
  // One operand tile of ROWS x BK floats per K step; every thread moves NV float4.
  static constexpr int NV = ROWS * BK / 4 / NT;
  static_assert(NV >= 1, "tile too small for the thread count");
  float4 regs[NV];

  // row0: first tile row (M or N index), k0: first k.  rows/kdim: logical extents.
  // ones_row: (MN-major only) logical row index whose value is 1 for every k < kdim, or -1.
  // FAST: pointers/leading dimensions 16-byte aligned and the contiguous extent a multiple of 4,
  // so every float4 is either fully inside or fully outside.  The load is a raw buffer load
  // whose byte offset is pushed past the descriptor's range when the float4 is outside: the
  // hardware range check returns zeros, nothing depends on the loaded value until the LDS
  // store, and no branch surrounds a load -- otherwise hipcc waits vmcnt(0) at the joins (or
  // at a data select) and the prefetch ring below degenerates to one latency per K step.
  template <bool FAST>
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, uint32_t bytes, int row0, int k0,
                                       int rows, int kdim) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, bytes, 0x00020000);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = v * NT + threadIdx.x;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      const int r = KMAJOR ? row0 + idx / (BK / 4) : row0 + (idx % (ROWS / 4)) * 4;
      const int k = KMAJOR ? k0 + (idx % (BK / 4)) * 4 : k0 + idx / (ROWS / 4);
      if (FAST) {
        const uint32_t off = KMAJOR ? (uint32_t)(((int64_t)r * ld + k) * 4) : (uint32_t)(((int64_t)k * ld + r) * 4);
        const bool in = r < rows && k < kdim;
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, in ? off : 0x80000000u, 0, 0);
        val = __builtin_bit_cast(float4, t);
      } else if (KMAJOR) {
        if (r < rows) {
          const float* p = P + (int64_t)r * ld + k;
          if (k < kdim) val.x = p[0];
          if (k + 1 < kdim) val.y = p[1];
          if (k + 2 < kdim) val.z = p[2];
          if (k + 3 < kdim) val.w = p[3];
        }
      } else if (k < kdim) {
        const float* p = P + (int64_t)k * ld + r;
        if (r < rows) val.x = p[0];
        if (r + 1 < rows) val.y = p[1];
        if (r + 2 < rows) val.z = p[2];
        if (r + 3 < rows) val.w = p[3];
      }
      regs[v] = val;
    }
  }

  // ones_row >= 0 (MN-major B of a wgrad only): logical row whose value is 1 for every k < kdim --
  // the virtual all-ones column that turns the bias gradient into one more output column.
  __device__ __forceinline__ void store(float* lds, int row0, int k0, int kdim, int ones_row) const {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = v * NT + threadIdx.x;
      if (KMAJOR) {
        const int r = idx / (BK / 4), k4 = idx % (BK / 4);
        *reinterpret_cast<float4*>(lds + r * BK + swz_chunk<BK>(r, k4) * 4) = regs[v];
      } else {
        const int k = idx / (ROWS / 4), r4 = idx % (ROWS / 4);
        float4 val = regs[v];
        if (ones_row >= 0 && k0 + k < kdim) {
          const int r = row0 + r4 * 4;
          val.x = (r == ones_row) ? 1.f : val.x;
          val.y = (r + 1 == ones_row) ? 1.f : val.y;
          val.z = (r + 2 == ones_row) ? 1.f : val.z;
          val.w = (r + 3 == ones_row) ? 1.f : val.w;
        }
        *reinterpret_cast<float4*>(lds + k * ROWS + r4 * 4) = val;
      }
    }
  }
  static constexpr int lds_floats() { return BK * ROWS; }

  // fragment of 4 k-values (k = ks*8 + 4*h + t, t = 0..3) of tile row `r`
  __device__ static __forceinline__ float4 frag(const float* lds, int r, int ks, int h) {
    if (KMAJOR) {
      return *reinterpret_cast<const float4*>(lds + r * BK + swz_chunk<BK>(r, ks * 2 + h) * 4);
    } else {
      const float* p = lds + (ks * 8 + 4 * h) * ROWS + r;
      return make_float4(p[0], p[ROWS], p[2 * ROWS], p[3 * ROWS]);
    }
  }
};

// problem b of a batched launch (GemmParams::batch): its operands, result and split-K partials
__device__ __forceinline__ void batch_offsets(GemmParams& p, int b) {
  p.A += (int64_t)b * p.bs_a;
  p.B += (int64_t)b * p.bs_b;
  p.C += (int64_t)b * p.bs_c;
  if (p.extra_col) p.extra_col += (int64_t)b * p.bs_x;
  if (p.partial) p.partial += (int64_t)b * p.splitk * p.slab;
}

// WK = 2: eight waves per workgroup; waves 4-7 take the second half of every K step of the same
// 2x2 arrangement of 32x32 tiles, so each SIMD holds two waves whose LDS round trips
// (write -> barrier -> fragment read) hide behind the other's MFMA chain; their partial
// accumulators are summed through LDS before the epilogue.
// WS = 1: a single 32x32 wave tile per K group (BM = BN = 32 * T), for problems with too few
// 64x64 tiles to occupy the chip: with WK = 4 the four waves of a workgroup split every K step
// four ways, so a 1050 x 400 x 400 product runs on ~430 workgroups x 4 SIMDs instead of 119.
template <bool A_KMAJOR, bool B_KMAJOR, int TM, int TN, int BK, int D, int WK, bool VEC, int WS = 2>
__global__ __launch_bounds__(64 * WS * WS * WK) void gemm_f32_kernel(const GemmParams p_in) {
  constexpr int BM = WS * TM * 32, BN = WS * TN * 32, NT = 64 * WS * WS * WK;
  GemmParams p = p_in;
  if (p_in.batch > 1) batch_offsets(p, blockIdx.y);
  using LA = TileLoader<A_KMAJOR, BM, BK, NT>;
  using LB = TileLoader<B_KMAJOR, BN, BK, NT>;
  constexpr int STAGE = LA::lds_floats() + LB::lds_floats();
  constexpr int RED = (WK - 1) * WS * WS * TM * TN * 16 * 64;  // K-group reduction scratch (reuses the stages)
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE > RED ? 2 * STAGE : RED];
  auto As = [&](int buf) { return lds + buf * STAGE; };
  auto Bs = [&](int buf) { return lds + buf * STAGE + LA::lds_floats(); };

  // XCD-aware tile order: consecutive workgroups land on different XCDs, so give each XCD a
  // contiguous run of tiles (neighbouring tiles share an A row panel in its L2).
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

  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) % (WS * WS), kgroup = (threadIdx.x >> 6) / (WS * WS);
  const int wm = wave / WS, wn = wave % WS;
  const int li = lane & 31, lh = lane >> 5;

  const int k_tiles = (p.K + BK - 1) / BK;
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

  // Register prefetch ring of D K-steps: the global loads of step kt+D are issued as soon as
  // the registers of step kt have been written to LDS, so D-1 steps of MFMA work cover the
  // L2/HBM latency even with a single wave per SIMD.  The step count is rounded up to a
  // multiple of D; steps past the end load zeros (range check, no branch), which the MFMAs absorb
  // (leaving the last ring turn early instead costs the 64x64 kernels 24 VGPRs: occupancy 8 -> 6).
  LA la[D];
  LB lb[D];
  const int ones_row = p.extra_col ? p.N : -1;
  const int k_limit = min(p.K, kt_end * BK);  // this split's K range ends here
  const int n_steps = (kt_end - kt_begin + D - 1) / D * D;
  const int ones_b = B_KMAJOR ? -1 : ones_row;
  // The ring's loads must be ISSUED in ring order, here as in the loop: the wait before the LDS store of
  // step st is then "all but the (D-1) younger steps' loads" (s_waitcnt vmcnt(2(D-1)NV)).  Left alone, the
  // scheduler interleaves the prologue's loads by address computation, and the wait at the loop head --
  // which has to hold for the prologue's order as well -- becomes vmcnt(0): no prefetch distance at all.
#pragma unroll
  for (int st = 0; st < D; ++st) {
    la[st].template load<VEC>(p.A, p.lda, p.a_bytes, row0, (kt_begin + st) * BK, p.M, k_limit);
    lb[st].template load<VEC>(p.B, p.ldb, p.b_bytes, col0, (kt_begin + st) * BK, p.N, k_limit);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int kt = kt_begin; kt < kt_begin + n_steps; kt += D) {
#pragma unroll
    for (int st = 0; st < D; ++st) {
      const int cur = st & 1;  // D is even: the LDS buffer parity continues across ring turns
      la[st].store(As(cur), row0, (kt + st) * BK, k_limit, -1);
      lb[st].store(Bs(cur), col0, (kt + st) * BK, k_limit, ones_b);
      la[st].template load<VEC>(p.A, p.lda, p.a_bytes, row0, (kt + st + D) * BK, p.M, k_limit);
      lb[st].template load<VEC>(p.B, p.ldb, p.b_bytes, col0, (kt + st + D) * BK, p.N, k_limit);
      __syncthreads();
#pragma unroll
      for (int kq = 0; kq < BK / 8 / WK; ++kq) {
        const int ks = kgroup * (BK / 8 / WK) + kq;
        float4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = LA::frag(As(cur), (wm * TM + i) * 32 + li, ks, lh);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = LB::frag(Bs(cur), (wn * TN + j) * 32 + li, ks, lh);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
          }
      }
    }
  }

  if (WK > 1) {  // sum the K groups: groups 1.. park their accumulators in LDS, group 0 adds them
    constexpr int PER_WAVE = TM * TN * 16 * 64;
    __syncthreads();
    if (kgroup > 0) {
      float* red = lds + (size_t)((kgroup - 1) * WS * WS + wave) * PER_WAVE;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) red[((i * TN + j) * 16 + e) * 64 + lane] = acc[i][j][e];
    }
    __syncthreads();
    if (kgroup > 0) return;
#pragma unroll
    for (int g = 0; g < WK - 1; ++g) {
      const float* red = lds + (size_t)(g * WS * WS + wave) * PER_WAVE;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += red[((i * TN + j) * 16 + e) * 64 + lane];
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
      // CH elements per batch: all 16 at once costs ~80 more VGPRs (a wave of occupancy on the large kernels)
      constexpr int CH = WS == 2 ? 1 : 8;
      // A lane's 16 elements share their column: with CH = 1 the bias is one load per lane, not one per element
      // (element by element, every store waited for its own bias load: 16 dependent L2 round trips per wave)
      GemmParams pe = p;
      float bcol = 0.f;
      if (CH == 1 && p.bias && !p.extra_col) {
        bcol = p.bias[col < p.N ? col : 0];
        pe.bias = nullptr;
      }
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
          v[e] = acc[i][j][e0 + e] + bcol;
        }
        epilogue_batch<CH>(pe, rows, cols, ok, v);
      }
    }
}

// Sums the split-K partial tiles (fixed order: deterministic) and applies the epilogue.  Every
// thread takes four consecutive elements of the flattened (M, Np) result: the slabs are padded to
// a multiple of four floats, so the partials stream in as 16-byte loads whatever Np is.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p_in) {
  GemmParams p = p_in;
  if (p_in.batch > 1) batch_offsets(p, blockIdx.y);
  const int Np = p.N + (p.extra_col ? 1 : 0);
  const int64_t total = (int64_t)p.M * Np;
  const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (idx >= total) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  // eight splits' loads in flight before the first add (the sum itself stays in split order: deterministic, and the
  // same value whatever the batch of eight holds)
  for (int z0 = 0; z0 < p.splitk; z0 += 8) {
    float4 t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      t[i] = z0 + i < p.splitk ? *reinterpret_cast<const float4*>(p.partial + (int64_t)(z0 + i) * p.slab + idx)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (z0 + i < p.splitk) { v.x += t[i].x; v.y += t[i].y; v.z += t[i].z; v.w += t[i].w; }
    }
  }
  int row = (int)(idx / Np), col = (int)(idx % Np);
  int rows[4], cols[4];
  bool ok[4];
  float vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    rows[e] = row; cols[e] = col; ok[e] = idx + e < total;
    if (++col == Np) { col = 0; ++row; }
  }
  epilogue_batch<4>(p, rows, cols, ok, vals);
}

// ws = 2: 2x2 waves per K group (tile 64*tm x 64*tn); ws = 1: one wave per K group (tile 32*tm x 32*tn)
template <bool AK, bool BKM>
static bool launch_cfg(const GemmParams& p, int ws, int tm, int tn, int bk, int depth, int wk, bool vec, dim3 grid,
                       hipStream_t s) {
#define CRADLE_GEMM_CASE(WS_, TM_, TN_, BK_, D_, WK_)                                                          \
  if (ws == WS_ && tm == TM_ && tn == TN_ && bk == BK_ && depth == D_ && wk == WK_) {                          \
    constexpr int NT_ = 64 * WS_ * WS_ * WK_;                                                                  \
    if (vec)                                                                                                   \
      hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM, TM_, TN_, BK_, D_, WK_, true, WS_>), grid, dim3(NT_), 0, s, p);  \
    else                                                                                                       \
      hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM, TM_, TN_, BK_, D_, WK_, false, WS_>), grid, dim3(NT_), 0, s, p); \
    return true;                                                                                               \
  }
  CRADLE_GEMM_CASE(2, 2, 2, 16, 2, 1)
  CRADLE_GEMM_CASE(2, 1, 1, 16, 2, 1)
  CRADLE_GEMM_CASE(2, 1, 1, 16, 4, 1)
  CRADLE_GEMM_CASE(2, 1, 1, 32, 2, 1)
  CRADLE_GEMM_CASE(2, 1, 1, 32, 2, 2)
  CRADLE_GEMM_CASE(2, 1, 1, 32, 4, 2)
  CRADLE_GEMM_CASE(2, 1, 2, 32, 2, 2)
  CRADLE_GEMM_CASE(2, 2, 2, 32, 2, 2)
  CRADLE_GEMM_CASE(1, 1, 1, 32, 2, 4)
  CRADLE_GEMM_CASE(1, 1, 1, 32, 4, 4)
  CRADLE_GEMM_CASE(1, 1, 1, 64, 2, 4)
  CRADLE_GEMM_CASE(1, 1, 1, 64, 4, 4)
  CRADLE_GEMM_CASE(1, 1, 1, 32, 8, 4)
  CRADLE_GEMM_CASE(2, 1, 1, 32, 4, 1)
  CRADLE_GEMM_CASE(1, 1, 2, 32, 2, 4)
  CRADLE_GEMM_CASE(1, 2, 2, 32, 2, 4)
  CRADLE_GEMM_CASE(2, 2, 1, 16, 2, 1)  // 128x64 / 64x128 tiles (64x32, 32x64 per wave): tools/gemm_rect_tiles.py, slower
  CRADLE_GEMM_CASE(2, 1, 2, 16, 2, 1)
#undef CRADLE_GEMM_CASE
  return false;
}

int64_t gemm_workspace_floats(int layout, int M, int N, int K, int extra_col) {
  (void)layout;
  (void)K;
  return (int64_t)32 * (((int64_t)M * (N + (extra_col ? 1 : 0)) + 3) & ~int64_t(3));  // worst case split factor 32
}

// manual override for tuning runs (tools/bench_gemm.py): -1 = automatic
static int g_tune_tm = -1, g_tune_tn = -1, g_tune_bk = -1, g_tune_splitk = -1, g_tune_depth = -1, g_tune_wk = -1,
           g_tune_ws = 2;

// Host-side dispatch.  `workspace` lets the problem split along K (partials + reducer).
int gemm_f32(int layout, GemmParams p, float* workspace, int64_t workspace_floats, hipStream_t stream) {
  CRADLE_CHECK_ARG(layout >= 0 && layout <= 2, "gemm: bad layout %d", layout);
  CRADLE_CHECK_ARG(p.A && p.B && p.C && p.M > 0 && p.N > 0 && p.K > 0, "gemm: bad arguments M=%d N=%d K=%d", p.M, p.N, p.K);
  CRADLE_CHECK_ARG(!p.extra_col || layout == 2, "gemm: the ones-column output needs the TN layout");
  const int Np = p.N + (p.extra_col ? 1 : 0);
  auto tiles_ws = [&](int ws_, int tm_, int tn_) {
    const int bm = 32 * ws_ * tm_, bn = 32 * ws_ * tn_;
    return (int64_t)((p.M + bm - 1) / bm) * ((Np + bn - 1) / bn);
  };
  auto tiles = [&](int tm_, int tn_) { return tiles_ws(2, tm_, tn_); };
  // Heuristics from tools/tune_gemm.py sweeps on MI355X at the ELBO step's shapes:
  //  * 64x64 tiles (four waves, one 32x32 MFMA tile each, BK 16) win at every size tried, up to
  //    10240 x 8000 x 400 (110 TFLOP/s = 87 % of the fp32 matrix peak at the ~1.93 GHz the chip holds
  //    under MFMA load); 128x128 tiles (kept for tuning runs) are 5-35 % slower -- a few hundred
  //    large tiles leave whole rounds of CUs idle.  Long-K problems are split along K until ~2048
  //    workgroups are in flight, never below 16 K-steps per split;
  //  * small problems (up to ~600 tiles of 64x64 and ~5 GFLOP: the embedding / qc encoders, the
  //    hidden layers, most weight gradients) run 15-45 % faster on 32x32 tiles whose four waves
  //    split each 32-wide K step between them: 4x the workgroups and a quarter of the MFMA
  //    chain per wave; split along K towards ~768 workgroups, at least 8 K-steps per split.
  int ws = 2, tm = 1, tn = 1, bk = 16, depth = 2, wk = 1;
  const double flops = 2.0 * p.M * (double)Np * p.K;
  if (tiles(1, 1) <= 600 && flops <= 5e9) { ws = 1; bk = 32; wk = 4; }
  if (g_tune_tm > 0) { ws = g_tune_ws; tm = g_tune_tm; tn = g_tune_tn; bk = g_tune_bk; depth = g_tune_depth; wk = g_tune_wk; }
  const int64_t ntiles = tiles_ws(ws, tm, tn);
  const int batch = max(1, p.batch);
  const int64_t wgs = ntiles * batch;  // workgroups in flight before any K split
  const int k_tiles = (p.K + bk - 1) / bk;
  int splitk = 1;
  if (workspace && ws == 1 && wgs < 768) {
    splitk = (int)min((int64_t)8, (768 + wgs - 1) / wgs);
    splitk = max(1, min(splitk, k_tiles / 8));
  } else if (workspace && ws == 2 && wgs < 2048) {
    splitk = (int)min((int64_t)16, (2048 + wgs - 1) / wgs);
    splitk = max(1, min(splitk, k_tiles / 16));
  }
  if (p.splitk_hint > 0) splitk = min(p.splitk_hint, max(1, k_tiles / 2));
  if (g_tune_splitk > 0) splitk = g_tune_splitk;
  if (!workspace || (uintptr_t)workspace % 16 != 0) splitk = 1;
  p.slab = ((int64_t)p.M * Np + 3) & ~int64_t(3);
  while (splitk > 1 && (int64_t)splitk * batch * p.slab > workspace_floats) --splitk;
  p.k_tiles_per_split = (k_tiles + splitk - 1) / splitk;
  splitk = (k_tiles + p.k_tiles_per_split - 1) / p.k_tiles_per_split;
  p.splitk = splitk;
  p.partial = workspace;
  // Branch-free loader (VEC): 16-byte buffer loads need dword alignment only on gfx950 (measured: an odd leading
  // dimension or pointer offset costs nothing), so what it asks for is that no 16-byte load straddles the end of
  // valid data: along K the extent must be a multiple of 4 (callers may pad K and say how much of it exists with
  // k_valid: the range check returns zeros past it); along M / N of an MN-major operand either the extent is a
  // multiple of 4 or the leading dimension leaves the row readable up to the next multiple (the extra rows /
  // columns this computes are never stored).
  const int Kv = p.k_valid > 0 ? min(p.k_valid, p.K) : p.K;
  const int64_t a_rows = layout == 2 ? Kv : p.M, a_cols = layout == 2 ? p.M : Kv;
  const int64_t b_rows = layout == 0 ? p.N : Kv, b_cols = layout == 0 ? Kv : p.N;
  const int64_t a_bytes = ((a_rows - 1) * p.lda + a_cols) * 4, b_bytes = ((b_rows - 1) * p.ldb + b_cols) * 4;
  p.a_bytes = (uint32_t)a_bytes;
  p.b_bytes = (uint32_t)b_bytes;
  auto row_ok = [](int extent, int64_t ld) { return extent >= 4 && (extent % 4 == 0 || ld >= ((extent + 3) & ~3)); };
  bool aligned = p.K >= 4 && a_bytes < (1ll << 31) && b_bytes < (1ll << 31);
  if (layout == 0) aligned = aligned && p.K % 4 == 0;
  else if (layout == 1) aligned = aligned && p.K % 4 == 0 && row_ok(p.N, p.ldb);
  else aligned = aligned && row_ok(p.M, p.lda) && row_ok(p.N, p.ldb);
  if (batch > 1) {
    CRADLE_CHECK_ARG(!p.bias && !p.rowbias && !p.rowscale && !p.act && !p.res && !p.dact_src && !p.C2 && p.beta == 0.f &&
                         batch <= 65535,
                     "gemm: a batched launch takes plain products only (batch %d)", batch);
  }
  if (p.precision >= 1 && aligned && flops >= 1.0e9 && batch == 1) {
    // split-bf16 kernels: 128 x 128 tiles, 64 in a dimension that pads less that way (400 -> 448 instead of 512);
    // two to four workgroups per CU, so aim at ~1024 of them (tools/bench_bf16x3.py), never below 8 K-steps per split
    auto padded = [](int n, int t) { return (n + 64 * t - 1) / (64 * t) * (64 * t); };
    int tm3 = padded(p.M, 2) <= padded(p.M, 1) ? 2 : 1, tn3 = padded(Np, 2) <= padded(Np, 1) ? 2 : 1;
    if (g_tune_tm > 0) { tm3 = min(g_tune_tm, 2); tn3 = min(g_tune_tn, 2); }
    const int64_t nt3 = (int64_t)(padded(p.M, tm3) / (64 * tm3)) * (padded(Np, tn3) / (64 * tn3));
    const int kt3 = (p.K + 31) / 32;
    int sk = 1;
    if (workspace && (uintptr_t)workspace % 16 == 0 && nt3 < 768) {
      sk = (int)min((int64_t)16, (1024 + nt3 - 1) / nt3);
      sk = max(1, min(sk, kt3 / 8));
    }
    if (p.splitk_hint > 0 && workspace) sk = max(1, min(p.splitk_hint, kt3 / 2));
    if (g_tune_splitk > 0 && workspace) sk = g_tune_splitk;
    while (sk > 1 && (int64_t)sk * p.slab > workspace_floats) --sk;
    p.k_tiles_per_split = (kt3 + sk - 1) / sk;
    p.splitk = (kt3 + p.k_tiles_per_split - 1) / p.k_tiles_per_split;
    CRADLE_CHECK_ARG(launch_gemm_bf16x3(layout, p, tm3, tn3, stream), "gemm: no split-bf16 kernel for tile %dx%d", tm3, tn3);
    CRADLE_CHECK_LAUNCH("gemm_bf16x3_kernel");
    if (p.splitk > 1) {
      const int64_t quads = (p.slab + 3) / 4;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, p);
      CRADLE_CHECK_LAUNCH("splitk_reduce_kernel");
    }
    return CRADLE_OK;
  }
  dim3 grid((unsigned)ntiles, (unsigned)batch, (unsigned)splitk);
  bool ok;
  if (layout == 0) ok = launch_cfg<true, true>(p, ws, tm, tn, bk, depth, wk, aligned, grid, stream);
  else if (layout == 1) ok = launch_cfg<true, false>(p, ws, tm, tn, bk, depth, wk, aligned, grid, stream);
  else ok = launch_cfg<false, false>(p, ws, tm, tn, bk, depth, wk, aligned, grid, stream);
  CRADLE_CHECK_ARG(ok, "gemm: no kernel for ws %d tile %dx%d bk %d depth %d wk %d", ws, tm, tn, bk, depth, wk);
  CRADLE_CHECK_LAUNCH("gemm_f32_kernel");
  if (splitk > 1) {
    const int64_t quads = (p.slab + 3) / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256), (unsigned)batch), dim3(256), 0, stream, p);
    CRADLE_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return CRADLE_OK;
}

}  // namespace cradle

using namespace cradle;

extern "C" {

// Tuning hook (not used by the product path): force tile (tm, tn in {1,2}), K step and split; -1 = auto.
void cradle_gemm_tune(int tm, int tn, int bk, int depth, int wk, int splitk) {
  cradle::g_tune_tm = tm; cradle::g_tune_tn = tn; cradle::g_tune_bk = bk; cradle::g_tune_depth = depth;
  cradle::g_tune_wk = wk; cradle::g_tune_splitk = splitk;
  cradle::g_tune_ws = 2;
  if (wk == 4) cradle::g_tune_ws = 1;  // the four-K-group kernels are the single-wave-tile ones
}

int cradle_gemm_f32(const cradle_gemm_desc* d, float* workspace, int64_t workspace_floats, hipStream_t stream) {
  CRADLE_CHECK_ARG(d, "gemm: null descriptor");
  GemmParams p = {};
  p.A = d->A; p.B = d->B; p.C = d->C; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.M = d->M; p.N = d->N; p.K = d->K; p.beta = d->beta;
  p.bias = d->bias; p.rowbias = d->rowbias; p.ld_rb = d->ld_rowbias; p.rb_mode = d->rowbias_mode; p.rb_n = d->rowbias_n;
  p.rowscale = d->rowscale; p.rs_n = d->rowscale_n;
  p.act = d->act; p.act_out = d->act_out; p.ld_act = d->ld_act_out;
  p.res = d->residual; p.ld_res = d->ld_residual; p.dact_src = d->dact_src; p.ld_ds = d->ld_dact_src;
  p.C2 = d->C2; p.ldc2 = d->ldc2; p.extra_col = d->extra_col;
  CRADLE_CHECK_ARG(d->precision >= 0 && d->precision <= 2, "gemm: precision must be 0 (fp32), 1 (bf16x3) or 2 (bf16), got %d",
                   d->precision);
  p.precision = d->precision;
  CRADLE_CHECK_ARG(d->k_valid >= 0 && d->k_valid <= d->K, "gemm: k_valid=%d outside [0, K=%d]", d->k_valid, d->K);
  p.k_valid = d->k_valid;
  CRADLE_CHECK_ARG(!p.C2 || p.dact_src, "gemm: C2 needs dact_src");
  CRADLE_CHECK_ARG(!p.rowbias || ((p.rb_mode == 1 || p.rb_mode == 2) && p.rb_n > 0), "gemm: bad rowbias mode");
  CRADLE_CHECK_ARG(!p.rowscale || (p.rowbias && p.rs_n > 0), "gemm: rowscale needs rowbias");
  return gemm_f32(d->layout, p, workspace, workspace_floats, stream);
}

// Tuning / benchmarking aid: the same GEMM launched `repeat` times back to back from C, so that a
// Python caller measures kernel time rather than its own per-call overhead.
int cradle_gemm_f32_repeat(const cradle_gemm_desc* d, float* workspace, int64_t workspace_floats, int repeat,
                           hipStream_t stream) {
  for (int i = 0; i < repeat; ++i) {
    int rc = cradle_gemm_f32(d, workspace, workspace_floats, stream);
    if (rc) return rc;
  }
  return CRADLE_OK;
}

int64_t cradle_gemm_workspace_floats(int layout, int M, int N, int K, int extra_col) {
  return gemm_workspace_floats(layout, M, N, K, extra_col);
}

}  // extern "C"
