// K1: library-size negative-binomial (Gamma-Poisson) log-likelihood over genes, fused with
// the row softmax that produces the mean, the gene reduction, and (training) the backward
// pass to the decoder logits and to log_concentration.
//
// Reference chain this replaces (all paths relative to /root/reference):
//   cradle_vae/models/utils/mlp.py:153-166   softmax -> mu = lib*softmax -> rate = clamp(c/(mu+1e-4), 1e-6)
//   pyro GammaPoisson.log_prob (pyro-ppl 1.9.1) called at cradle_vae/models/utils/loss_modules.py:134-135
//   cradle_vae/models/utils/loss_modules.py:143  .sum(-1) over genes
//
// HBM-bound streaming kernel.  One workgroup owns one cell n: the count row X[n,:], the
// per-gene concentration and the per-gene lgamma/digamma terms (which depend only on
// (n,g), not on the particle) stay in registers while the workgroup walks the particle
// rows p of that cell; the next row's logits are prefetched while the current one is
// reduced.  Every logit is read from HBM exactly once and (training) overwritten in place
// by its gradient.
#include "common.h"
#include "nb_math.h"

#include <mutex>
#include <type_traits>

namespace cradle {

// TAIL: the gene count is not a multiple of 4 (Gv < G): the last 16-byte vector of a row is masked.  A separate
// instantiation: the masks cost the aligned case 6 us of its 38 when they are compiled in (register pressure).
// PRE (training step): the particle-independent terms were made beforehand by nb_count_terms_kernel -- the cell's
// sum of L(x,c) is read from pre_lg[n] and psi(c+x) - psi(c) from the cell's own row of dlogc_part, which this
// kernel overwrites with its result at the end.  The lgamma / digamma evaluation (branchy, ~a third of this
// kernel's instructions at five particles per cell) then runs on a side stream during the forward pass.
template <int VPT, int NT, bool BWD, bool TAIL, bool PRE = false>
__global__ __launch_bounds__(NT) void nb_rows_kernel(
    const float* logits, int64_t ld_logits,       // (P*N, ld) rows r = p*N + n
    float* dlogits, int64_t ld_dlogits,           // BWD: gradient rows, may alias logits
    const float* __restrict__ X, int64_t ldx,     // (N, ldx) integer-valued counts
    const float* __restrict__ lib,                // (N) library sizes
    const float* __restrict__ gene_consts,        // (3, G): c, lgamma(c), digamma(c)
    const float* __restrict__ grow,               // BWD: (P*N) d loss / d out[r]
    float* __restrict__ out,                      // (P*N) sum_g log p
    float* __restrict__ dlogc_part,               // BWD: (N*PS, G) partial d loss / d log_c
    int P, int N, int G, int PS, int Gv,    // G: row length run (multiple of 4); Gv <= G: genes that exist
    const float* __restrict__ pre_lg = nullptr) {  // PRE: (N) sum_g L(x[n,g], c[g])
  __shared__ float2 red[6][NT / 64];
  const int n = blockIdx.x;
  const int ps = blockIdx.y;
  const int tid = threadIdx.x;
  const int G4 = G >> 2;

  // ---- per-cell, per-gene state in registers
  float4 x[VPT], c[VPT], dmain[VPT];  // dmain: sum_r grow[r] * d log p / dc
  float lg_local = 0.f;
  float cmin = INFINITY;  // smallest concentration among this thread's genes (decides the rows' clamp-free pass)
  float gsum = 0.f;  // sum of this block's upstream row gradients (scales the digamma part)
  if (BWD)
    for (int p = ps; p < P; p += PS) gsum += grow[(int64_t)p * N + n];
  const float4* X4 = reinterpret_cast<const float4*>(X + (int64_t)n * ldx);
  const float4* C4 = reinterpret_cast<const float4*>(gene_consts);
  const float4* LGC4 = reinterpret_cast<const float4*>(gene_consts + G);
  const float4* PSC4 = reinterpret_cast<const float4*>(gene_consts + 2 * (int64_t)G);
  const float libn = lib[n];
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int k = j * NT + tid;
    if (k < G4) {
      x[j] = TAIL ? mask_gene_tail(X4[k], k, Gv, 0.f) : X4[k];
      c[j] = C4[k];
      cmin = fminf(fminf(fminf(c[j].x, c[j].y), fminf(c[j].z, c[j].w)), cmin);
      float4 dpsi = make_float4(0.f, 0.f, 0.f, 0.f);
      if (PRE) {
        if (BWD) dpsi = reinterpret_cast<const float4*>(dlogc_part + ((int64_t)n * PS + ps) * G)[k];
        dmain[j] = make_float4(gsum * dpsi.x, gsum * dpsi.y, gsum * dpsi.z, gsum * dpsi.w);
        continue;
      }
      const float4 lgc = LGC4[k];
      if (BWD) {
        const float4 psc = PSC4[k];
        lg_local += nb_lgamma_term<true>(x[j].x, c[j].x, lgc.x, psc.x, dpsi.x);
        lg_local += nb_lgamma_term<true>(x[j].y, c[j].y, lgc.y, psc.y, dpsi.y);
        lg_local += nb_lgamma_term<true>(x[j].z, c[j].z, lgc.z, psc.z, dpsi.z);
        lg_local += nb_lgamma_term<true>(x[j].w, c[j].w, lgc.w, psc.w, dpsi.w);
      } else {
        float dummy;
        lg_local += nb_lgamma_term<false>(x[j].x, c[j].x, lgc.x, 0.f, dummy);
        lg_local += nb_lgamma_term<false>(x[j].y, c[j].y, lgc.y, 0.f, dummy);
        lg_local += nb_lgamma_term<false>(x[j].z, c[j].z, lgc.z, 0.f, dummy);
        lg_local += nb_lgamma_term<false>(x[j].w, c[j].w, lgc.w, 0.f, dummy);
      }
      dmain[j] = make_float4(gsum * dpsi.x, gsum * dpsi.y, gsum * dpsi.z, gsum * dpsi.w);
    } else {
      x[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      c[j] = make_float4(1.f, 1.f, 1.f, 1.f);
      dmain[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float LG = PRE ? pre_lg[n] : block_sum<NT>(lg_local, reinterpret_cast<float*>(red[5]));

  // ---- walk this block's particle rows, prefetching one row ahead
  float4 buf_a[VPT], buf_b[VPT];
  auto load_row = [&](int p, float4* dst) {
    const float4* L4 = reinterpret_cast<const float4*>(logits + ((int64_t)p * N + n) * ld_logits);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int k = j * NT + tid;
      dst[j] = (k < G4) ? (TAIL ? mask_gene_tail(L4[k], k, Gv, -INFINITY) : L4[k])
                        : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
  };
  // One row: `cur` holds its logits, the next row's are requested into `nxt` first.  The loop below runs the rows in
  // pairs with the two buffers trading places (no register copies between rows).  rbase: three disjoint LDS rows per
  // row of the walk, alternating between two sets: a fast wave may start the next reduction while a slow one still
  // reads the previous result, never the one after that.
  auto do_row = [&](int p, int rbase, float4* cur, float4* nxt) __attribute__((always_inline)) {
    if (p + PS < P) load_row(p + PS, nxt);

    // row max (v_max3: fmaxf alone compiles to a max plus a canonicalising max per element)
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < VPT; ++j) mx = max3(max3(mx, cur[j].x, cur[j].y), cur[j].z, cur[j].w);
    mx = block_max<NT>(mx, reinterpret_cast<float*>(red[rbase]));
    // exp and row sum: exp(l - mx) = exp2(l * log2(e) - mx * log2(e)), one fused multiply-add per element
    const float mxs = -mx * kLog2e;
    float2 se2 = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      cur[j].x = __builtin_amdgcn_exp2f(fmaf(cur[j].x, kLog2e, mxs));
      cur[j].y = __builtin_amdgcn_exp2f(fmaf(cur[j].y, kLog2e, mxs));
      cur[j].z = __builtin_amdgcn_exp2f(fmaf(cur[j].z, kLog2e, mxs));
      cur[j].w = __builtin_amdgcn_exp2f(fmaf(cur[j].w, kLog2e, mxs));
      se2.x += cur[j].x;
      se2.y += cur[j].y;
      se2.x += cur[j].z;
      se2.y += cur[j].w;
    }
    float se = se2.x + se2.y;
    se = block_sum<NT>(se, reinterpret_cast<float*>(red[rbase + 1]));
    const float inv_se = 1.0f / se;
    const float mu_scale = libn * inv_se;  // mu = e * lib / sum(e)

    const int64_t r = (int64_t)p * N + n;
    const float g_r = BWD ? grow[r] : 0.f;
    float tsum = 0.f, usum = 0.f;
    float4 u[VPT];
    auto main_pass = [&](auto noclamp) {
      constexpr bool NC = decltype(noclamp)::value;
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        const int k = j * NT + tid;
        if (k < G4) {
          float du;
          tsum += nb_main_term<BWD, NC>(cur[j].x * mu_scale, x[j].x, c[j].x, u[j].x, du);
          if (BWD) dmain[j].x = fmaf(g_r, du, dmain[j].x);
          tsum += nb_main_term<BWD, NC>(cur[j].y * mu_scale, x[j].y, c[j].y, u[j].y, du);
          if (BWD) dmain[j].y = fmaf(g_r, du, dmain[j].y);
          tsum += nb_main_term<BWD, NC>(cur[j].z * mu_scale, x[j].z, c[j].z, u[j].z, du);
          if (BWD) dmain[j].z = fmaf(g_r, du, dmain[j].z);
          tsum += nb_main_term<BWD, NC>(cur[j].w * mu_scale, x[j].w, c[j].w, u[j].w, du);
          if (BWD) dmain[j].w = fmaf(g_r, du, dmain[j].w);
          if (BWD) usum += (u[j].x + u[j].y) + (u[j].z + u[j].w);
        }
      }
    };
    // cur <= 1, so no mean of this row exceeds mu_scale: where every lane's smallest c keeps c/(mu + 1e-4) above the
    // clamp of mlp.py:163 (always, for concentrations and library sizes of this world), the wave takes the pass
    // without it.  The test is per wave and row: same path in every run.
    const bool no_clamp = cmin >= kRateMin * 1.0001f * (mu_scale + kMuEps);
    if (__builtin_amdgcn_ballot_w64(!no_clamp) == 0) main_pass(std::true_type{});
    else main_pass(std::false_type{});
    const float2 tu = block_sum2<NT>(tsum, usum, red[rbase + 2]);
    if (tid == 0) out[r] = tu.x + LG;
    if (BWD) {
      const float U = tu.y;
      float4* DL4 = reinterpret_cast<float4*>(dlogits + r * ld_dlogits);
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        const int k = j * NT + tid;
        if (k < G4) {
          float4 d;
          d.x = g_r * fmaf(-cur[j].x * inv_se, U, u[j].x);
          d.y = g_r * fmaf(-cur[j].y * inv_se, U, u[j].y);
          d.z = g_r * fmaf(-cur[j].z * inv_se, U, u[j].z);
          d.w = g_r * fmaf(-cur[j].w * inv_se, U, u[j].w);
          DL4[k] = d;
        }
      }
    }
  };
  if (ps < P) load_row(ps, buf_a);
  for (int p = ps; p < P; p += 2 * PS) {
    do_row(p, 0, buf_a, buf_b);
    if (p + PS < P) do_row(p + PS, 3, buf_b, buf_a);
  }

  if (BWD) {
    float4* DP4 = reinterpret_cast<float4*>(dlogc_part + ((int64_t)n * PS + ps) * G);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int k = j * NT + tid;
      if (k < G4) {
        float4 d;  // chain through c = exp(log_c)
        d.x = c[j].x * dmain[j].x;
        d.y = c[j].y * dmain[j].y;
        d.z = c[j].z * dmain[j].z;
        d.w = c[j].w * dmain[j].w;
        DP4[k] = d;
      }
    }
  }
}

// The particle-independent terms of the likelihood, once per (cell, gene) ahead of nb_rows_kernel<PRE>: the cell's
// sum of L(x,c) -> lg[n], and psi(c+x) - psi(c) -> the cell's PS partial rows of d log_c (each particle split reads
// its own copy before it overwrites the row).  Same thread/gene assignment and summation order as the rows kernel's
// own prologue with NT threads: the two forms give the same bits.
template <int NT, bool TAIL>
__global__ __launch_bounds__(NT) void nb_count_terms_kernel(const float* __restrict__ X, int64_t ldx,
                                                             const float* __restrict__ gene_consts,
                                                             float* __restrict__ dpsi_rows, float* __restrict__ lg, int G,
                                                             int PS, int Gv) {
  __shared__ float2 red[NT / 64];
  const int n = blockIdx.x, tid = threadIdx.x, G4 = G >> 2;
  const float4* X4 = reinterpret_cast<const float4*>(X + (int64_t)n * ldx);
  const float4* C4 = reinterpret_cast<const float4*>(gene_consts);
  const float4* LGC4 = reinterpret_cast<const float4*>(gene_consts + G);
  const float4* PSC4 = reinterpret_cast<const float4*>(gene_consts + 2 * (int64_t)G);
  float lg_local = 0.f;
  for (int k = tid; k < G4; k += NT) {
    const float4 x = TAIL ? mask_gene_tail(X4[k], k, Gv, 0.f) : X4[k];
    const float4 c = C4[k], lgc = LGC4[k], psc = PSC4[k];
    float4 dpsi;
    lg_local += nb_lgamma_term<true>(x.x, c.x, lgc.x, psc.x, dpsi.x);
    lg_local += nb_lgamma_term<true>(x.y, c.y, lgc.y, psc.y, dpsi.y);
    lg_local += nb_lgamma_term<true>(x.z, c.z, lgc.z, psc.z, dpsi.z);
    lg_local += nb_lgamma_term<true>(x.w, c.w, lgc.w, psc.w, dpsi.w);
    for (int ps = 0; ps < PS; ++ps) reinterpret_cast<float4*>(dpsi_rows + ((int64_t)n * PS + ps) * G)[k] = dpsi;
  }
  const float LG = block_sum<NT>(lg_local, reinterpret_cast<float*>(red));
  if (tid == 0) lg[n] = LG;
}

// Per-gene constants from log_concentration: c = exp(log_c), lgamma(c), digamma(c).
// (mlp.py:160 `concentration = torch.exp(self.log_concentration)`.)  G threads, evaluated in
// double so the fp32 constants are correctly rounded.
// Gp >= G: stride of the three rows; the padding genes [G, Gp) get c = 1e-30, lgamma = digamma = 0: with a count of 0
// and a logit of -inf (mu = 0) such a gene adds -c*log1p(1e-6) ~ -1e-36 to a row's log-likelihood and exactly 0 to
// d logits -- the rows kernel runs over Gp genes unchanged when n_phenos is not a multiple of 4.
__global__ void nb_gene_consts_kernel(const float* __restrict__ log_c, float* __restrict__ gc, int G, int Gp) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= Gp) return;
  if (g >= G) {
    gc[g] = 1e-30f;
    gc[Gp + g] = 0.f;
    gc[2 * (int64_t)Gp + g] = 0.f;
    return;
  }
  const float cf = expf(log_c[g]);
  double c = (double)cf;
  // digamma by recurrence to y >= 10 then the asymptotic series
  double y = c, ps = 0.0;
  while (y < 10.0) {
    ps -= 1.0 / y;
    y += 1.0;
  }
  const double iy = 1.0 / y, iy2 = iy * iy;
  ps += log(y) - 0.5 * iy - iy2 * (1.0 / 12 - iy2 * (1.0 / 120 - iy2 * (1.0 / 252 - iy2 * (1.0 / 240))));
  gc[g] = cf;
  gc[Gp + g] = (float)lgamma(c);
  gc[2 * (int64_t)Gp + g] = (float)ps;
}

// Deterministic column sum of the (rows, G) partials written by nb_rows_kernel (G % 4 == 0, rows 16-byte
// aligned).  A workgroup owns 64 columns: 16 lanes x float4 = one 256-byte segment of every row; its 16 row
// slices (four per wave) walk the rows with independent 16-byte loads, eight in flight per lane, and meet in
// LDS in a fixed order.  256-thread workgroups on purpose: this runs on a side stream beside the backward
// pass's GEMMs and has to fit into whatever wave slots they leave free (the 1024-thread form this replaces
// measured 12 us alone and 170 us inside the step, waiting for whole CUs).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, int rows, int G,
                                                      float* __restrict__ out, int accumulate, int valid) {
  __shared__ float4 sm[16][16];
  const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int col4 = blockIdx.x * 16 + c;  // float4 column
  const int G4 = G >> 2;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col4 < G4) {
    const float4* src = reinterpret_cast<const float4*>(part) + col4;
    int r = sl;
    for (; r + 7 * 16 < rows; r += 8 * 16) {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = src[(int64_t)(r + i * 16) * G4];
#pragma unroll
      for (int i = 0; i < 8; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
    for (; r < rows; r += 16) {
      const float4 v = src[(int64_t)r * G4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  sm[sl][c] = acc;
  __syncthreads();
  if (sl == 0 && col4 < G4) {
    float4 t = sm[0][c];
#pragma unroll
    for (int k = 1; k < 16; ++k) { t.x += sm[k][c].x; t.y += sm[k][c].y; t.z += sm[k][c].z; t.w += sm[k][c].w; }
    // columns >= valid are the padding genes of a width that is not a multiple of 4: nothing is read or written there
    const int c0 = col4 * 4;
    if (c0 + 3 < valid) {
      float4* dst = reinterpret_cast<float4*>(out) + col4;
      if (accumulate) { const float4 o = *dst; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
      *dst = t;
    } else {
      const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + e < valid) out[c0 + e] = accumulate ? out[c0 + e] + tv[e] : tv[e];
    }
  }
}

constexpr int kNbMaxGenes = 1024 * 4 * 4;
static int g_nb_tune_nt = -1, g_nb_tune_ps = -1;

static int nb_threads(int G4) {
  if (g_nb_tune_nt > 0) return g_nb_tune_nt;
  return G4 <= 256 * 5 ? 256 : (G4 <= 512 * 4 ? 512 : 1024);  // measured: tools/bench_nb.py
}

template <bool BWD, bool PRE = false>
static int launch_nb_rows(const float* logits, int64_t ld_logits, float* dlogits, int64_t ld_dlogits,
                          const float* X, int64_t ldx, const float* lib, const float* gene_consts,
                          const float* grow, float* out, float* dlogc_part, int P, int N, int G, int PS, int Gv,
                          hipStream_t stream, const float* pre_lg = nullptr) {
  const int G4 = G / 4;
  dim3 grid(N, PS);
#define CRADLE_NB_LAUNCH(VPT, NT)                                                                            \
  do {                                                                                                       \
    if (Gv < G)                                                                                              \
      hipLaunchKernelGGL((nb_rows_kernel<VPT, NT, BWD, true, PRE>), grid, dim3(NT), 0, stream, logits, ld_logits, \
                         dlogits, ld_dlogits, X, ldx, lib, gene_consts, grow, out, dlogc_part, P, N, G, PS, Gv, pre_lg); \
    else                                                                                                     \
      hipLaunchKernelGGL((nb_rows_kernel<VPT, NT, BWD, false, PRE>), grid, dim3(NT), 0, stream, logits, ld_logits, \
                         dlogits, ld_dlogits, X, ldx, lib, gene_consts, grow, out, dlogc_part, P, N, G, PS, Gv, pre_lg); \
  } while (0)
  // thread count / values per thread: as many waves per CU as the register budget allows (all
  // workgroups of a 512-cell batch are then resident at once); override via cradle_nb_tune
  const int nt = nb_threads(G4);
  const int vpt = (G4 + nt - 1) / nt;
  bool launched = true;
  if (nt == 256 && vpt == 1) CRADLE_NB_LAUNCH(1, 256);
  else if (nt == 256 && vpt == 2) CRADLE_NB_LAUNCH(2, 256);
  else if (nt == 256 && vpt == 3) CRADLE_NB_LAUNCH(3, 256);
  else if (nt == 256 && vpt == 4) CRADLE_NB_LAUNCH(4, 256);
  else if (nt == 256 && vpt == 5) CRADLE_NB_LAUNCH(5, 256);
  else if (nt == 512 && vpt == 1) CRADLE_NB_LAUNCH(1, 512);
  else if (nt == 512 && vpt == 2) CRADLE_NB_LAUNCH(2, 512);
  else if (nt == 512 && vpt == 3) CRADLE_NB_LAUNCH(3, 512);
  else if (nt == 512 && vpt == 4) CRADLE_NB_LAUNCH(4, 512);
  else if (nt == 1024 && vpt == 1) CRADLE_NB_LAUNCH(1, 1024);
  else if (nt == 1024 && vpt == 2) CRADLE_NB_LAUNCH(2, 1024);
  else if (nt == 1024 && vpt == 3) CRADLE_NB_LAUNCH(3, 1024);
  else if (nt == 1024 && vpt == 4) CRADLE_NB_LAUNCH(4, 1024);
  else launched = false;
  if (!launched) {
    cradle_set_error("nb_rows: n_genes=%d not supported with %d threads (maximum %d genes)", G, nt, kNbMaxGenes);
    return CRADLE_ERR_UNSUPPORTED;
  }
#undef CRADLE_NB_LAUNCH
  CRADLE_CHECK_LAUNCH("nb_rows_kernel");
  return CRADLE_OK;
}

static int check_common(const void* logits, const void* X, const void* lib, const void* gc, const void* out,
                        int P, int N, int G, int64_t ld_logits, int64_t ldx) {
  CRADLE_CHECK_ARG(logits && X && lib && gc && out, "nb_logprob: null pointer argument");
  CRADLE_CHECK_ARG(P > 0 && N > 0 && G > 0, "nb_logprob: P=%d N=%d G=%d must be positive", P, N, G);
  const int Gp = (G + 3) & ~3;  // rows are run as Gp genes; the tail behind G is masked in the kernel
  CRADLE_CHECK_ARG(ld_logits >= Gp && ld_logits % 4 == 0 && ldx >= Gp && ldx % 4 == 0,
                   "nb_logprob: leading dimensions (%lld, %lld) must be multiples of 4 and >= n_genes rounded up to 4 (%d)",
                   (long long)ld_logits, (long long)ldx, Gp);
  CRADLE_CHECK_ARG(((uintptr_t)logits % 16 == 0) && ((uintptr_t)X % 16 == 0) && ((uintptr_t)gc % 16 == 0),
                   "nb_logprob: logits, X and gene_consts must be 16-byte aligned");
  return CRADLE_OK;
}

}  // namespace cradle

using namespace cradle;

namespace cradle {
// d loss / d log_concentration: column sums of the `rows` = N * particle_split partial rows the rows
// kernel left in `workspace`.  Separate from the rows launch so that the step can run it beside the backward.
// G: row length of the partials (a multiple of 4); valid <= G: genes that exist -- dlog_c holds `valid` floats.
int nb_dlogc_reduce(const float* workspace, int rows, int G, float* dlog_c, int accumulate, hipStream_t stream, int valid) {
  CRADLE_CHECK_ARG(G % 4 == 0 && (uintptr_t)workspace % 16 == 0 && (uintptr_t)dlog_c % 16 == 0,
                   "nb_dlogc_reduce: G must be a multiple of 4, partials and dlog_concentration 16-byte aligned");
  hipLaunchKernelGGL(colsum_kernel, dim3((G / 4 + 15) / 16), dim3(256), 0, stream, workspace, rows, G, dlog_c, accumulate,
                     valid < 0 ? G : valid);
  CRADLE_CHECK_LAUNCH("colsum_kernel");
  return CRADLE_OK;
}

// The training step's form of K1 (kernels.h): nb_count_terms on a side stream during the forward pass, then
// nb_logprob_fwd_bwd_pre where cradle_nb_logprob_fwd_bwd_timed would stand.  Same arguments and checks; `workspace`
// carries psi(c+x) - psi(c) in (the N * particle_split partial rows of d log_c), `lg` (N) the cells' lgamma sums.
int nb_count_terms(const float* X, int64_t ldx, const float* gene_consts, float* workspace, int64_t workspace_floats,
                   float* lg, int P, int N, int G, hipStream_t stream) {
  CRADLE_CHECK_ARG(X && gene_consts && workspace && lg && P > 0 && N > 0 && G > 0, "nb_count_terms: bad arguments");
  const int Gp = (G + 3) & ~3, PS = cradle_nb_particle_split(P, N);
  CRADLE_CHECK_ARG(ldx >= Gp && ldx % 4 == 0 && (uintptr_t)X % 16 == 0 && (uintptr_t)gene_consts % 16 == 0 &&
                       (uintptr_t)workspace % 16 == 0 && workspace_floats >= (int64_t)N * PS * Gp,
                   "nb_count_terms: X / gene_consts / workspace must be 16-byte aligned, ld %% 4 == 0, workspace >= %lld floats",
                   (long long)((int64_t)N * PS * Gp));
  // the thread count of the rows kernel (same gene-to-thread assignment: same summation order)
  const int nt = nb_threads(Gp / 4);
#define CRADLE_CT_LAUNCH(NT)                                                                                          \
  do {                                                                                                                \
    if (G < Gp)                                                                                                       \
      hipLaunchKernelGGL((nb_count_terms_kernel<NT, true>), dim3(N), dim3(NT), 0, stream, X, ldx, gene_consts,        \
                         workspace, lg, Gp, PS, G);                                                                   \
    else                                                                                                              \
      hipLaunchKernelGGL((nb_count_terms_kernel<NT, false>), dim3(N), dim3(NT), 0, stream, X, ldx, gene_consts,       \
                         workspace, lg, Gp, PS, G);                                                                   \
  } while (0)
  if (nt == 256) CRADLE_CT_LAUNCH(256);
  else if (nt == 512) CRADLE_CT_LAUNCH(512);
  else CRADLE_CT_LAUNCH(1024);
#undef CRADLE_CT_LAUNCH
  CRADLE_CHECK_LAUNCH("nb_count_terms_kernel");
  return CRADLE_OK;
}

int nb_logprob_fwd_bwd_pre(const float* logits, int64_t ld_logits, float* dlogits, int64_t ld_dlogits, const float* X,
                           int64_t ldx, const float* lib, const float* gene_consts, const float* grow, float* out,
                           float* workspace, int64_t workspace_floats, const float* lg, int P, int N, int G, void* ev_begin,
                           void* ev_end, hipStream_t stream) {
  int rc = check_common(logits, X, lib, gene_consts, out, P, N, G, ld_logits, ldx);
  if (rc) return rc;
  CRADLE_CHECK_ARG(dlogits && grow && workspace && lg, "nb_logprob_fwd_bwd_pre: null pointer argument");
  const int Gp = (G + 3) & ~3;
  CRADLE_CHECK_ARG(ld_dlogits >= Gp && ld_dlogits % 4 == 0 && (uintptr_t)dlogits % 16 == 0,
                   "nb_logprob_fwd_bwd_pre: dlogits must be 16-byte aligned with ld %% 4 == 0, ld >= n_genes rounded up to 4");
  const int PS = cradle_nb_particle_split(P, N);
  CRADLE_CHECK_ARG(workspace_floats >= (int64_t)N * PS * Gp && (uintptr_t)workspace % 16 == 0,
                   "nb_logprob_fwd_bwd_pre: workspace needs %lld floats, 16-byte aligned (got %lld)",
                   (long long)((int64_t)N * PS * Gp), (long long)workspace_floats);
  if (ev_begin) (void)hipEventRecord((hipEvent_t)ev_begin, stream);
  rc = launch_nb_rows<true, true>(logits, ld_logits, dlogits, ld_dlogits, X, ldx, lib, gene_consts, grow, out, workspace,
                                  P, N, Gp, PS, G, stream, lg);
  if (ev_end) (void)hipEventRecord((hipEvent_t)ev_end, stream);
  return rc;
}

int nb_gene_consts_padded(const float* log_c, float* gene_consts, int G, int Gp, hipStream_t stream) {
  CRADLE_CHECK_ARG(log_c && gene_consts && G > 0 && Gp >= G, "nb_gene_consts: bad arguments");
  hipLaunchKernelGGL(nb_gene_consts_kernel, dim3((Gp + 255) / 256), dim3(256), 0, stream, log_c, gene_consts, G, Gp);
  CRADLE_CHECK_LAUNCH("nb_gene_consts_kernel");
  return CRADLE_OK;
}
}  // namespace cradle

extern "C" {

int cradle_nb_gene_consts(const float* log_c, float* gene_consts, int G, hipStream_t stream) {
  return cradle::nb_gene_consts_padded(log_c, gene_consts, G, (G + 3) & ~3, stream);
}

// tuning hook (tools/bench_nb.py): threads per workgroup (256/512/1024) and particle split; -1 = auto
void cradle_nb_tune(int threads, int particle_split) {
  cradle::g_nb_tune_nt = threads;
  cradle::g_nb_tune_ps = particle_split;
}

int cradle_nb_particle_split(int P, int N) {
  if (cradle::g_nb_tune_ps > 0) return cradle::g_nb_tune_ps < P ? cradle::g_nb_tune_ps : P;
  // enough workgroups to cover the chip (256 CUs x 2) without splitting a cell's particles
  // more than needed: each split re-reads X[n] and redoes the per-cell lgamma terms
  int ps = 1;
  while ((int64_t)N * ps < 512 && ps * 2 <= P) ps *= 2;
  return ps;
}

int cradle_nb_logprob_fwd(const float* logits, int64_t ld_logits, const float* X, int64_t ldx, const float* lib,
                          const float* gene_consts, float* out, int P, int N, int G, hipStream_t stream) {
  int rc = check_common(logits, X, lib, gene_consts, out, P, N, G, ld_logits, ldx);
  if (rc) return rc;
  const int PS = cradle_nb_particle_split(P, N);
  return launch_nb_rows<false>(logits, ld_logits, nullptr, 0, X, ldx, lib, gene_consts, nullptr, out, nullptr, P,
                               N, (G + 3) & ~3, PS, G, stream);
}

// Optional events (cradle_event_create handles) are recorded on `stream` immediately around the
// rows kernel (the dominant launch; the small column sum that follows is outside the bracket).
int cradle_nb_logprob_fwd_bwd_timed(const float* logits, int64_t ld_logits, float* dlogits, int64_t ld_dlogits,
                                    const float* X, int64_t ldx, const float* lib, const float* gene_consts,
                                    const float* grow, float* out, float* dlog_c, int accumulate_dlog_c,
                                    float* workspace, int64_t workspace_floats, int P, int N, int G, void* ev_begin,
                                    void* ev_end, hipStream_t stream) {
  int rc = check_common(logits, X, lib, gene_consts, out, P, N, G, ld_logits, ldx);
  if (rc) return rc;
  CRADLE_CHECK_ARG(dlogits && grow && workspace, "nb_logprob_fwd_bwd: null pointer argument");
  const int Gp = (G + 3) & ~3;
  CRADLE_CHECK_ARG(ld_dlogits >= Gp && ld_dlogits % 4 == 0 && (uintptr_t)dlogits % 16 == 0,
                   "nb_logprob_fwd_bwd: dlogits must be 16-byte aligned with ld %% 4 == 0, ld >= n_genes rounded up to 4");
  const int PS = cradle_nb_particle_split(P, N);
  CRADLE_CHECK_ARG(workspace_floats >= (int64_t)N * PS * Gp && (uintptr_t)workspace % 16 == 0,
                   "nb_logprob_fwd_bwd: workspace needs %lld floats, 16-byte aligned (got %lld)",
                   (long long)((int64_t)N * PS * Gp), (long long)workspace_floats);
  if (ev_begin) (void)hipEventRecord((hipEvent_t)ev_begin, stream);
  rc = launch_nb_rows<true>(logits, ld_logits, dlogits, ld_dlogits, X, ldx, lib, gene_consts, grow, out, workspace,
                            P, N, Gp, PS, G, stream);
  if (ev_end) (void)hipEventRecord((hipEvent_t)ev_end, stream);
  if (rc) return rc;
  if (!dlog_c) return CRADLE_OK;  // internal callers: the reduction follows on another stream (nb_dlogc_reduce)
  return cradle::nb_dlogc_reduce(workspace, N * PS, Gp, dlog_c, accumulate_dlog_c, stream, G);
}

int cradle_nb_logprob_fwd_bwd(const float* logits, int64_t ld_logits, float* dlogits, int64_t ld_dlogits,
                              const float* X, int64_t ldx, const float* lib, const float* gene_consts,
                              const float* grow, float* out, float* dlog_c, int accumulate_dlog_c,
                              float* workspace, int64_t workspace_floats, int P, int N, int G,
                              hipStream_t stream) {
  CRADLE_CHECK_ARG(dlog_c, "nb_logprob_fwd_bwd: null dlog_c");
  return cradle_nb_logprob_fwd_bwd_timed(logits, ld_logits, dlogits, ld_dlogits, X, ldx, lib, gene_consts, grow, out,
                                         dlog_c, accumulate_dlog_c, workspace, workspace_floats, P, N, G, nullptr,
                                         nullptr, stream);
}

int64_t cradle_nb_workspace_floats(int P, int N, int G) {
  return (int64_t)N * cradle_nb_particle_split(P, N) * ((G + 3) & ~3);
}

}  // extern "C"

// ---- the entry point that allocates nothing ------------------------------------------------------
// cradle_nb_logprob_fwd_bwd needs scratch for the per-cell partial rows of d log_concentration
// (N * particle_split * G floats) and takes it from the caller.  Callers that cannot allocate (code under
// HIP graph capture, serving loops, harnesses that forbid hipMalloc) get the same computation with that
// scratch in a file-scope device array of fixed size.  `scratch` is the host-side stand-in of the symbol,
// not a device address: its address on each device is looked up once with hipGetSymbolAddress and only that
// pointer is ever handed to a kernel.
namespace {
constexpr int64_t kStaticScratchFloats = 4ll << 20;  // 16 MiB per GPU: e.g. 512 cells x 8192 genes
static __device__ __align__(256) float scratch[kStaticScratchFloats];
constexpr int kMaxDevices = 64;
float* g_scratch_ptr[kMaxDevices] = {};  // per device: each GPU holds its own copy of the array at its own address
hipError_t g_scratch_err[kMaxDevices] = {};
std::once_flag g_scratch_once[kMaxDevices];

// The address of `scratch` on the current device, looked up once per device (std::call_once: two host threads may
// make a device's first call together).  hipGetSymbolAddress resolves the code object's symbol lazily and is not
// legal under stream capture: a caller that captures calls cradle_nb_static_init() once per device beforehand.
int static_scratch(float** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    cradle_set_error("nb_logprob_fwd_bwd_static: cannot identify the current device (ordinals 0..%d)", kMaxDevices - 1);
    return CRADLE_ERR_HIP;
  }
  std::call_once(g_scratch_once[dev], [dev] {
    float* p = nullptr;
    g_scratch_err[dev] = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    g_scratch_ptr[dev] = p;
  });
  if (g_scratch_err[dev] != hipSuccess || !g_scratch_ptr[dev]) {
    cradle_set_error("nb_logprob_fwd_bwd_static: hipGetSymbolAddress failed: %s", hipGetErrorString(g_scratch_err[dev]));
    return CRADLE_ERR_HIP;
  }
  *out = g_scratch_ptr[dev];
  return CRADLE_OK;
}
}  // namespace

extern "C" {

int64_t cradle_nb_static_scratch_floats(void) { return kStaticScratchFloats; }

// Resolves the scratch array's address on the current device now (the one call of cradle_nb_logprob_fwd_bwd_static's
// path that is not capture-safe).  Optional otherwise: the first call on a device does it by itself.
int cradle_nb_static_init(void) {
  float* p = nullptr;
  return static_scratch(&p);
}

int cradle_nb_logprob_fwd_bwd_static(const float* logits, int64_t ld_logits, float* dlogits, int64_t ld_dlogits,
                                     const float* X, int64_t ldx, const float* lib, const float* gene_consts,
                                     const float* grow, float* out, float* dlog_c, int accumulate_dlog_c, int P, int N,
                                     int G, hipStream_t stream) {
  CRADLE_CHECK_ARG(P > 0 && N > 0 && G > 0, "nb_logprob_fwd_bwd_static: P=%d N=%d G=%d must be positive", P, N, G);
  const int64_t need = cradle_nb_workspace_floats(P, N, G);
  CRADLE_CHECK_ARG(need <= kStaticScratchFloats,
                   "nb_logprob_fwd_bwd_static: N * particle_split * G = %lld floats of scratch exceed the built-in "
                   "%lld (use cradle_nb_logprob_fwd_bwd with a caller workspace)",
                   (long long)need, (long long)kStaticScratchFloats);
  float* ws = nullptr;
  const int rc = static_scratch(&ws);
  if (rc) return rc;
  // every partial row that the column sum reads is rewritten by the rows kernel of this call: nothing has
  // to start at zero, whatever the previous call left behind
  return cradle_nb_logprob_fwd_bwd(logits, ld_logits, dlogits, ld_dlogits, X, ldx, lib, gene_consts, grow, out, dlog_c,
                                   accumulate_dlog_c, ws, kStaticScratchFloats, P, N, G, stream);
}

}  // extern "C"
