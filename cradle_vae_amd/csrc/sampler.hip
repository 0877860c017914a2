// K9: Gamma-Poisson observation sampler fused with the ATE accumulation.
//
// Replaces, per (perturbation, particle) row of decoder logits:
//   cradle_vae/models/utils/mlp.py:153-166      softmax -> mu -> rate, GammaPoisson(c, rate)
//   pyro GammaPoisson.sample (pyro-ppl 1.9.1)   rate' ~ Gamma(c, rate); x ~ Poisson(rate')   (model.py:131)
//   cradle_vae/analysis/average_treatment_effects.py:171-176,190-195
//       method "perturbseq": x <- log2(1e4 * x / sum_g x + 1);  X_sums[t] += sum over particles
// One workgroup per condition (perturbation) t: every thread owns fixed genes, walks the
// particles of that condition and keeps the running sums in registers, so the (P, T, G) sample
// tensor of the reference is never materialised and the sum over particles is deterministic.
// Random numbers: Philox4x32-10 keyed by (seed, stream), counter = (gene, row, draw index): the
// draws do not depend on the launch geometry.  (Bit-parity with torch's generator is
// impossible by construction; parity tests use sampler = MEAN, moments tests the real sampler.)
//
// Two exact samplers of the same distribution.  Marginally x ~ NegBinomial(c, p = rate/(1+rate)):
//   * rate >= 1/4 and pmf(0) = p^c >= e^-15 (mean <= 37: almost every gene of a count matrix): inversion
//     by sequential search from pmf(0) with pmf(k+1) = pmf(k) (k+c)/(k+1) / (1+rate) -- one uniform,
//     one log/exp pair, 1 + x iterations; four genes share one Philox call;
//   * otherwise (rare, high-mean genes; near-Poisson genes with a large concentration, whose pmf(0) is
//     below fp32 resolution) the Gamma -> Poisson composition the reference's sampler performs.
// With workspace, the particles of a condition are split over gridDim.y workgroups whose partial
// sums are added in a fixed order by a second kernel (a single workgroup per condition leaves the
// chip almost empty: ~240 conditions per launch).
#include "common.h"
#include "nb_math.h"
#include "philox.h"

namespace cradle {

// Gamma(shape a, scale 1): Marsaglia-Tsang with the a < 1 boost; bounded rejection loop.
__device__ float sample_gamma(float a, const Philox& ph, uint32_t gene, uint32_t row, uint32_t& draw) {
  const bool small = a < 1.0f;
  const float a0 = a;
  if (small) a += 1.0f;
  const float d = a - 1.0f / 3.0f, c = rsqrtf(9.0f * d);
  float out = d, boost = 1.0f;
  for (int it = 0; it < 16; ++it) {
    const uint4 r = ph(gene, row, draw++, 2u);
    if (it == 0 && small) boost = __powf(u01(r.w), 1.0f / a0);  // the a < 1 boost U^(1/a): the call's fourth word
    const float rad = sqrtf(-2.0f * __logf(u01(r.x)));
    const float x = rad * __cosf(6.283185307179586f * u01(r.y));
    const float t = 1.0f + c * x;
    if (t <= 0.f) continue;
    const float v = t * t * t;
    const float u = u01(r.z);
    const float x2 = x * x;
    // squeeze (Marsaglia & Tsang): accepts ~98 % of the proposals without the two logarithms
    if (u < 1.0f - 0.0331f * x2 * x2 || __logf(u) < 0.5f * x2 + d - d * v + d * __logf(v)) {
      out = d * v;
      break;
    }
  }
  return out * boost;
}

// Poisson(lam), lam >= 10: Hoermann's PTRS (as torch.poisson does).
__device__ float sample_poisson_ptrs(float lam, const Philox& ph, uint32_t gene, uint32_t row, uint32_t& draw) {
  const float slam = sqrtf(lam), loglam = __logf(lam);
  const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
  const float invalpha = 1.1239f + 1.1328f / (b - 3.4f), vr = 0.9277f - 3.6224f / (b - 2.0f);
  for (int it = 0; it < 32; ++it) {
    const uint4 r = ph(gene, row, draw++, 4u);
    const float U = u01(r.x) - 0.5f, V = u01(r.y);
    const float us = 0.5f - fabsf(U);
    const float k = floorf((2.0f * a / us + b) * U + lam + 0.43f);
    if (us >= 0.07f && V <= vr) return k;
    if (k < 0.f || (us < 0.013f && V > us)) continue;
    if (__logf(V) + __logf(invalpha) - __logf(a / (us * us) + b) <= -lam + k * loglam - lgammaf(k + 1.0f)) return k;
  }
  return floorf(lam);
}

// Poisson(lam), lam < kPoisSearchMax: inversion by sequential search from pmf(0) = exp(-lam) >= 4.5e-5, one uniform,
// pmf(k+1) = pmf(k) lam / (k+1) -- four steps per loop turn like sample_nb_inversion below (same table).
constexpr float kPoisSearchMax = 10.0f;
__device__ __forceinline__ float sample_poisson_search(float lam, float u, const float* __restrict__ inv_tab) {
  float pk = __expf(-lam), cdf = pk, k = 0.f;
  if (!(lam > 0.f)) return 0.f;
  const float4* inv4 = reinterpret_cast<const float4*>(inv_tab);
  int ki = 0;
  bool live = u > cdf;
  while (live) {
    const float4 iv = inv4[ki];
    if (u > cdf) { pk *= lam * iv.x; cdf += pk; k += 1.0f; }
    if (u > cdf) { pk *= lam * iv.y; cdf += pk; k += 1.0f; }
    if (u > cdf) { pk *= lam * iv.z; cdf += pk; k += 1.0f; }
    if (u > cdf) { pk *= lam * iv.w; cdf += pk; k += 1.0f; }
    ++ki;
    live = u > cdf && (k <= lam || pk > 1e-9f) && ki < 128;  // past the mode with increments fp32 no longer sees: stop
  }
  return k;
}

// Poisson(lam) where it is met (the path that does not re-queue: more than half of a row's genes deferred)
__device__ float sample_poisson(float lam, const Philox& ph, uint32_t gene, uint32_t row, uint32_t& draw,
                                const float* __restrict__ inv_tab) {
  if (!(lam > 0.f)) return 0.f;
  if (lam < kPoisSearchMax) return sample_poisson_search(lam, u01(ph(gene, row, 0u, 3u).x), inv_tab);
  return sample_poisson_ptrs(lam, ph, gene, row, draw);
}

// The Gamma -> Poisson composition (pyro GammaPoisson.sample).  Not inlined: it is the rare branch
// of the kernel's gene loop, which must stay small enough to unroll (v[], c[] live in registers).
__device__ __noinline__ float sample_gamma_poisson(float c, float rate, Philox ph, uint32_t gene, uint32_t row,
                                                   const float* __restrict__ inv_tab) {
  uint32_t draw = 0;
  const float lam = sample_gamma(c, ph, gene, row, draw) / rate;
  return sample_poisson(lam, ph, gene, row, draw, inv_tab);
}

// First phase of the queued composition: the Gamma draw, and the Poisson draw at once when its rate is 10 or more
// (PTRS).  Returns the count (>= 0), or -(1 + lam) for a rate lam below kPoisSearchMax, which the caller re-queues
// for the search.  Not inlined, like sample_gamma_poisson: its registers must not count against the gene loop's.
__device__ __noinline__ float sample_gamma_stage(float c, float rate, Philox ph, uint32_t gene, uint32_t row) {
  uint32_t draw = 0;
  const float lam = sample_gamma(c, ph, gene, row, draw) / rate;
  if (lam < kPoisSearchMax) return -1.0f - lam;
  return sample_poisson_ptrs(lam, ph, gene, row, draw);
}

// NegBinomial(c, rate) by inversion (sequential search upwards from 0).  Usable where
//   rate >= kInvRateMin            the geometric tail (1+rate)^-k is below fp32 resolution long before
//                                  the iteration cap, and
//   c * log1p(1/rate) <= kInvMaxNegLogP0   pmf(0) = exp(-c log1p(1/rate)) >= 3e-7 is a normal float that the
//                                  recurrence can climb from with full relative precision.  The two
//                                  together bound the mean by 15 * 4 / log(5) = 37, so the cap of 512
//                                  iterations lies > 30 standard deviations out.
// Everything else (large concentration with a mean in the tens or more: pmf(0) tiny or underflowing)
// takes the Gamma -> Poisson composition.  nb_inversion_ok() is the routing predicate.
constexpr float kInvRateMin = 0.25f;
constexpr float kInvMaxNegLogP0 = 15.0f;
__device__ __forceinline__ float nb_log1p_inv_rate(float y) {
  // log1p(y), y = 1/rate: series below 0.03 (log(1+y) would round 1+y first), fast log above
  return y < 0.03f ? y * (1.0f - y * (0.5f - y * (1.0f / 3.0f))) : __logf(1.0f + y);
}
__device__ __forceinline__ bool nb_inversion_ok(float c, float rate) {
  return rate >= kInvRateMin && c * nb_log1p_inv_rate(__frcp_rn(rate)) <= kInvMaxNegLogP0;
}
// inv_tab[k] = 1 / (k + 1), k < 512, in LDS: the search's one reciprocal per step is a quarter-rate instruction
// y = 1/rate in (0, 4], neg_log_p0 = c log1p(1/rate) <= 15: what nb_inversion_ok's test computed already
__device__ __forceinline__ float sample_nb_inversion(float c, float y, float neg_log_p0, float u,
                                                     const float* __restrict__ inv_tab) {
  float pk = __expf(-neg_log_p0);  // pmf(0) = (rate/(1+rate))^c >= exp(-15)
  float cdf = pk, k = 0.f;
  if (u > cdf) {
    const float q = y * __frcp_rn(1.0f + y);  // 1/(1+rate)
    const float cq = c * q;
    const float kmode = (c - 1.0f) * y;       // the pmf rises up to here: never stop the search before it
    // Four steps per turn of the loop: a step is one compare and five operations under the lane's own predicate
    // (pmf(k+1) = pmf(k) (k q + c q) / (k+1)); the four reciprocals come as one 16-byte LDS read, and what does not
    // change the draw -- the test that stops a search whose u lies beyond what the fp32 cdf can reach (past the mode,
    // increments below 1e-9) and the iteration cap -- is looked at once per turn, not once per step.  The loop runs
    // as long as the slowest lane of the wave: its cost per step is what counts.
    const float4* inv4 = reinterpret_cast<const float4*>(inv_tab);
    int ki = 0;
    bool live = true;
    while (live) {
      const float4 iv = inv4[ki];
      if (u > cdf) { pk *= fmaf(k, q, cq) * iv.x; cdf += pk; k += 1.0f; }
      if (u > cdf) { pk *= fmaf(k, q, cq) * iv.y; cdf += pk; k += 1.0f; }
      if (u > cdf) { pk *= fmaf(k, q, cq) * iv.z; cdf += pk; k += 1.0f; }
      if (u > cdf) { pk *= fmaf(k, q, cq) * iv.w; cdf += pk; k += 1.0f; }
      ++ki;
      live = u > cdf && (k <= kmode || pk > 1e-9f) && ki < 128;
    }
  }
  return k;
}

enum { SAMPLER_NB = 0, SAMPLER_MEAN = 1 };
enum { METHOD_MEAN = 0, METHOD_PERTURBSEQ = 1 };

// logits rows r = p * C + t (particle-major, the (P, n, G) layout model.forward flattens).
// TAIL: Gv < G (a gene count that is not a multiple of 4): the last 16-byte vector of a row is masked
// NT threads own VPT 16-byte vectors of genes each.  The per-gene search loops are chains of dependent operations
// (a reciprocal among them) of a length that differs from lane to lane: what hides their latency is waves, and a
// thread's registers grow with VPT (v, c, acc: 12 VPT, plus the unrolled search) -- 230 VGPRs and two waves per
// SIMD at VPT = 6.  So wide rows go to 512 / 1024 threads with VPT <= 2..4 (four waves per SIMD).
template <int VPT, int NT, bool TAIL>
__global__ __launch_bounds__(NT) void nb_sample_accumulate_kernel(const float* __restrict__ logits, int64_t ld,
                                                                   const float* __restrict__ lib, int lib_n,
                                                                   const float* __restrict__ gene_consts, float* __restrict__ sums,
                                                                   float* __restrict__ samples, int P, int C, int G, int method,
                                                                   int sampler, uint32_t seed_lo, uint32_t seed_hi,
                                                                   uint32_t row_offset, float* __restrict__ partials, int Gv) {
  // G: row length run (a multiple of 4; sums, samples and gene_consts have rows of G floats); Gv <= G: genes that exist
  __shared__ float2 red[6][NT / 64];
  // Genes that need the Gamma -> Poisson composition (a few hundred operations, against ~20 for an inversion) are not
  // drawn where they are met -- one such lane would hold its whole wave for every gene slot, and at 10 % of the genes
  // every wave has one -- but queued per row in LDS (the gene's rate goes to slot[g], its index to queue[]) and
  // then drawn by the whole workgroup, one queued gene per lane; the owners collect the counts from slot[g].
  // Draws are keyed by (gene, row), so who draws does not matter.
  extern __shared__ float dyn_lds[];
  float* slot = dyn_lds;                                  // [G]
  int* queue = reinterpret_cast<int*>(dyn_lds + G);       // [G]
  __shared__ int queued, queued_b;
  __shared__ __attribute__((aligned(16))) float inv_tab[512];
  for (int i = threadIdx.x; i < 512; i += NT) inv_tab[i] = 1.0f / (float)(i + 1);  // (visible after the first softmax barrier)
  const int t = blockIdx.x, tid = threadIdx.x;
  const int G4 = G >> 2;
  const Philox ph{seed_lo, seed_hi};
  if (tid == 0) queued = queued_b = 0;  // (the first push comes after the barriers of the first row's softmax)
  const float libt = lib[lib_n == 1 ? 0 : t];
  float4 c[VPT], acc[VPT];
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int k = j * NT + tid;
    c[j] = k < G4 ? reinterpret_cast<const float4*>(gene_consts)[k] : make_float4(1.f, 1.f, 1.f, 1.f);
    acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int it = 0;
  for (int p = blockIdx.y; p < P; p += gridDim.y, ++it) {
    const int64_t r = (int64_t)p * C + t;
    const float4* L4 = reinterpret_cast<const float4*>(logits + r * ld);
    const int rbase = (it & 1) * 3;  // disjoint LDS rows per reduction (see nb_loss.hip)
    float4 v[VPT];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int k = j * NT + tid;
      v[j] = k < G4 ? (TAIL ? mask_gene_tail(L4[k], k, Gv, -INFINITY) : L4[k])
                    : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      mx = fmaxf(fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)), mx);
    }
    mx = block_max<NT>(mx, reinterpret_cast<float*>(red[rbase]));
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      v[j].x = fast_exp(v[j].x - mx); v[j].y = fast_exp(v[j].y - mx);
      v[j].z = fast_exp(v[j].z - mx); v[j].w = fast_exp(v[j].w - mx);
      se += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    se = block_sum<NT>(se, reinterpret_cast<float*>(red[rbase + 1]));
    const float mu_scale = libt / se;
    float xs = 0.f;
    uint32_t deferred = 0;  // bit j*4+q: this thread's gene (j, q) is in the queue
    const uint32_t row = row_offset + (uint32_t)r;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int k = j * NT + tid;
      if (k < G4) {
        float* e = reinterpret_cast<float*>(&v[j]);
        const float* cc = reinterpret_cast<const float*>(&c[j]);
        uint4 u4 = make_uint4(0u, 0u, 0u, 0u);
        if (sampler == SAMPLER_NB) u4 = ph((uint32_t)k, row, 0u, 5u);  // one uniform per gene of this float4
        const uint32_t ur[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float rate = fmaxf(cc[q] * fast_rcp(e[q] * mu_scale + kMuEps), kRateMin);
          const float y = fast_rcp(rate), neg_log_p0 = cc[q] * nb_log1p_inv_rate(y);
          float x;
          if (TAIL && k * 4 + q >= Gv) {
            x = 0.f;  // behind the last gene
          } else if (sampler == SAMPLER_MEAN) {
            x = cc[q] * y;
          } else if (rate >= kInvRateMin && neg_log_p0 <= kInvMaxNegLogP0) {  // == nb_inversion_ok(c, rate)
            x = sample_nb_inversion(cc[q], y, neg_log_p0, u01(ur[q]), inv_tab);
          } else {
            slot[k * 4 + q] = rate;
            queue[atomicAdd(&queued, 1)] = k * 4 + q;
            deferred |= 1u << (j * 4 + q);
            x = 0.f;
          }
          e[q] = x;
          xs += x;
        }
      }
    }
    if (sampler == SAMPLER_NB) {
      __syncthreads();
      const int n_queued = queued;
      if (2 * n_queued <= G) {
        // Phases, so that the lanes of a wave do the same thing: (a) every queued gene's Gamma draw; rates of 10 and
        // more take their Poisson draw (PTRS) at once, the others go to a second queue (filled from the back of
        // queue[]: n_queued entries at the front, at most as many at the back); (b) the second queue's Poisson draws
        // by sequential search.  (One pass with both Poisson forms and the search inside it costs a wave the sum of
        // all branches whenever its lanes differ -- they always do.)
        for (int i = tid; i < n_queued; i += NT) {
          const int g = queue[i];
          const float r = sample_gamma_stage(gene_consts[g], slot[g], ph, (uint32_t)g, row);
          slot[g] = r < 0.f ? -1.0f - r : r;
          if (r < 0.f) queue[G - 1 - atomicAdd(&queued_b, 1)] = g;
        }
        __syncthreads();
        const int n_b = queued_b;
        for (int i = tid; i < n_b; i += NT) {
          const int g = queue[G - 1 - i];
          slot[g] = sample_poisson_search(slot[g], u01(ph((uint32_t)g, row, 0u, 3u).x), inv_tab);
        }
      } else {
        for (int i = tid; i < n_queued; i += NT) {
          const int g = queue[i];
          slot[g] = sample_gamma_poisson(gene_consts[g], slot[g], ph, (uint32_t)g, row, inv_tab);
        }
      }
      __syncthreads();
      if (tid == 0) queued = queued_b = 0;  // the next row pushes only after the two barriers of its softmax
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        float* e = reinterpret_cast<float*>(&v[j]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (deferred & (1u << (j * 4 + q))) {
            e[q] = slot[(j * NT + tid) * 4 + q];
            xs += e[q];
          }
      }
    }
    if (samples) {
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        const int k = j * NT + tid;
        if (k < G4) reinterpret_cast<float4*>(samples + r * (int64_t)G)[k] = v[j];
      }
    }
    if (method == METHOD_PERTURBSEQ) {
      xs = block_sum<NT>(xs, reinterpret_cast<float*>(red[rbase + 2]));
      const float sc = 1e4f / xs;  // 0/0 -> NaN like the reference (:174)
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        acc[j].x += __log2f(v[j].x * sc + 1.0f); acc[j].y += __log2f(v[j].y * sc + 1.0f);
        acc[j].z += __log2f(v[j].z * sc + 1.0f); acc[j].w += __log2f(v[j].w * sc + 1.0f);
      }
    } else {
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        acc[j].x += v[j].x; acc[j].y += v[j].y; acc[j].z += v[j].z; acc[j].w += v[j].w;
      }
    }
  }
  if (partials) {  // this workgroup's share of the particles; summed by partial_sums_kernel
    float4* P4 = reinterpret_cast<float4*>(partials + ((int64_t)blockIdx.y * C + t) * G);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int k = j * NT + tid;
      if (k < G4) P4[k] = acc[j];
    }
    return;
  }
  float4* S4 = reinterpret_cast<float4*>(sums + (int64_t)t * G);
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int k = j * NT + tid;
    if (k < G4) {
      float4 s = S4[k];
      s.x += acc[j].x; s.y += acc[j].y; s.z += acc[j].z; s.w += acc[j].w;
      S4[k] = s;
    }
  }
}

// sums[i] += partials[0][i] + partials[1][i] + ... (fixed order), i over C*G in float4 steps
__global__ __launch_bounds__(256) void partial_sums_kernel(const float* __restrict__ partials, float* __restrict__ sums,
                                                           int64_t n4, int splits) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 a = reinterpret_cast<const float4*>(partials)[i];
  for (int s = 1; s < splits; ++s) {
    const float4 b = reinterpret_cast<const float4*>(partials)[(int64_t)s * n4 + i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  float4 o = reinterpret_cast<float4*>(sums)[i];
  o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
  reinterpret_cast<float4*>(sums)[i] = o;
}

}  // namespace cradle

using namespace cradle;

extern "C" {

static int g_sample_tune_nt = -1;
// tuning hook (tools/bench_sampler.py): threads per workgroup (256 / 512 / 1024); -1 = automatic
void cradle_nb_sample_tune(int threads) { g_sample_tune_nt = threads; }

static int particle_splits(int P, int C) {
  // ~2048 workgroups per launch, at least 4 particles per workgroup
  int s = (2048 + C - 1) / C;
  if (s > P / 4) s = P / 4;
  return s < 1 ? 1 : s;
}

// Floats of workspace that let cradle_nb_sample_accumulate_ws split the particles of every condition
// over several workgroups (0: the launch already fills the chip).
int64_t cradle_nb_sample_workspace_floats(int P, int C, int G) {
  if (P <= 0 || C <= 0 || G <= 0) return 0;
  const int s = particle_splits(P, C);
  return s > 1 ? (int64_t)s * C * ((G + 3) & ~3) : 0;
}

// sums (C, G) += sum over the P particles of transform(x[p, t, :]), x ~ GammaPoisson(c, rate(logits)).
// samples: optional (P*C, G) output of the raw draws (predictor.sample_observations), or NULL.
// lib: (1) or (C) library sizes.  method: 0 raw counts, 1 "perturbseq".  sampler: 0 NB draw, 1 its mean.
// workspace: optional scratch of cradle_nb_sample_workspace_floats(P, C, G) floats (16-byte aligned);
// NULL or too small: one workgroup per condition walks all its particles.
int cradle_nb_sample_accumulate_ws(const float* logits, int64_t ld_logits, const float* library, int library_n,
                                   const float* gene_consts, float* sums, float* samples, int P, int C, int G, int method,
                                   int sampler, uint64_t seed, uint32_t row_offset, float* workspace,
                                   int64_t workspace_floats, hipStream_t stream) {
  CRADLE_CHECK_ARG(logits && library && gene_consts && sums && P > 0 && C > 0 && G > 0, "nb_sample: bad arguments");
  const int Gv = G;
  G = (G + 3) & ~3;  // rows are run as the next multiple of 4 genes: sums / samples / gene_consts rows have that length
  CRADLE_CHECK_ARG(ld_logits % 4 == 0 && ld_logits >= G,
                   "nb_sample: ld must be a multiple of 4 and >= n_genes rounded up to a multiple of 4");
  CRADLE_CHECK_ARG((uintptr_t)logits % 16 == 0 && (uintptr_t)sums % 16 == 0 && (uintptr_t)gene_consts % 16 == 0 &&
                       (uintptr_t)samples % 16 == 0 && (uintptr_t)workspace % 16 == 0,
                   "nb_sample: pointers must be 16-byte aligned");
  CRADLE_CHECK_ARG(library_n == 1 || library_n == C, "nb_sample: library has %d entries, expected 1 or %d", library_n, C);
  CRADLE_CHECK_ARG(method >= 0 && method <= 1 && sampler >= 0 && sampler <= 1, "nb_sample: bad method/sampler");
  CRADLE_CHECK_ARG(C <= 65535 * 64, "nb_sample: too many conditions per launch");
  const int G4 = G / 4;
  const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
  int splits = workspace ? particle_splits(P, C) : 1;
  while (splits > 1 && (int64_t)splits * C * G > workspace_floats) --splits;
  float* partials = splits > 1 ? workspace : nullptr;
#define CRADLE_SAMPLE_LAUNCH(VPT, NT)                                                                                  \
  do {                                                                                                                 \
    if (lds > 48 * 1024) {                                                                                             \
      const void* fn = Gv < G ? (const void*)nb_sample_accumulate_kernel<VPT, NT, true>                                \
                              : (const void*)nb_sample_accumulate_kernel<VPT, NT, false>;                              \
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {               \
        cradle_set_error("nb_sample: cannot reserve %zu bytes of LDS for the deferred-draw queue", lds);               \
        return CRADLE_ERR_HIP;                                                                                         \
      }                                                                                                                \
    }                                                                                                                  \
    if (Gv < G)                                                                                                        \
      hipLaunchKernelGGL((nb_sample_accumulate_kernel<VPT, NT, true>), dim3(C, splits), dim3(NT), lds, stream, logits, \
                         ld_logits, library, library_n, gene_consts, sums, samples, P, C, G, method, sampler, lo, hi,  \
                         row_offset, partials, Gv);                                                                    \
    else                                                                                                               \
      hipLaunchKernelGGL((nb_sample_accumulate_kernel<VPT, NT, false>), dim3(C, splits), dim3(NT), lds, stream, logits, \
                         ld_logits, library, library_n, gene_consts, sums, samples, P, C, G, method, sampler, lo, hi,  \
                         row_offset, partials, Gv);                                                                    \
  } while (0)
  int nt = g_sample_tune_nt > 0 ? g_sample_tune_nt : (G4 <= 512 ? 256 : (G4 <= 1024 ? 512 : 1024));
  const int vpt = (G4 + nt - 1) / nt;
  const size_t lds = sampler == 0 ? (size_t)G * 8 : 0;  // slot[G] + queue[G]; the mean needs neither
  bool launched = true;
  if (nt == 256 && vpt == 1) CRADLE_SAMPLE_LAUNCH(1, 256);
  else if (nt == 256 && vpt == 2) CRADLE_SAMPLE_LAUNCH(2, 256);
  else if (nt == 256 && vpt <= 4) CRADLE_SAMPLE_LAUNCH(4, 256);
  else if (nt == 256 && vpt <= 8) CRADLE_SAMPLE_LAUNCH(8, 256);
  else if (nt == 512 && vpt == 1) CRADLE_SAMPLE_LAUNCH(1, 512);
  else if (nt == 512 && vpt == 2) CRADLE_SAMPLE_LAUNCH(2, 512);
  else if (nt == 512 && vpt == 3) CRADLE_SAMPLE_LAUNCH(3, 512);
  else if (nt == 512 && vpt == 4) CRADLE_SAMPLE_LAUNCH(4, 512);
  else if (nt == 1024 && vpt == 1) CRADLE_SAMPLE_LAUNCH(1, 1024);
  else if (nt == 1024 && vpt == 2) CRADLE_SAMPLE_LAUNCH(2, 1024);
  else if (nt == 1024 && vpt == 3) CRADLE_SAMPLE_LAUNCH(3, 1024);
  else if (nt == 1024 && vpt == 4) CRADLE_SAMPLE_LAUNCH(4, 1024);
  else launched = false;
  if (!launched) {
    cradle_set_error("nb_sample: n_genes=%d not supported with %d threads (maximum 16384 genes)", Gv, nt);
    return CRADLE_ERR_UNSUPPORTED;
  }
#undef CRADLE_SAMPLE_LAUNCH
  CRADLE_CHECK_LAUNCH("nb_sample_accumulate_kernel");
  if (partials) {
    const int64_t n4 = (int64_t)C * G4;
    hipLaunchKernelGGL(partial_sums_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, partials, sums, n4, splits);
    CRADLE_CHECK_LAUNCH("partial_sums_kernel");
  }
  return CRADLE_OK;
}

int cradle_nb_sample_accumulate(const float* logits, int64_t ld_logits, const float* library, int library_n,
                                const float* gene_consts, float* sums, float* samples, int P, int C, int G, int method,
                                int sampler, uint64_t seed, uint32_t row_offset, hipStream_t stream) {
  return cradle_nb_sample_accumulate_ws(logits, ld_logits, library, library_n, gene_consts, sums, samples, P, C, G, method,
                                        sampler, seed, row_offset, nullptr, 0, stream);
}

}  // extern "C"
