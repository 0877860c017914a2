// The fused ELBO step: forward + backward of PerturbationPlatedELBOLossModule.loss
// (reference cradle_vae/models/utils/loss_modules.py:54-187) as one stream-ordered sequence of
// hand-written kernels.  No autograd, no allocation, no host synchronisation: the caller hands
// in parameter / gradient / batch / noise pointers and one workspace, and gets the loss terms
// and every parameter gradient back.  Capturable into a HIP graph.
//
// Algebraic restructurings relative to the reference (i-vii, and viii of DESIGN.md; all exact up to
// fp32 reassociation):
//   (i)   x_good / x_bad are not sampled (model.py:131-132 draws are unused by the loss);
//   (ii)  logstd(X) Wx^T and logstd(cfX) Wx^T are computed once, not once per particle and call;
//   (iii) the counterfactual pass evaluates q_z_basal only (its q_z_cf_basal and z_basal draw are
//         unused, loss_modules.py:146-150);
//   (iv)  qc in {0,1}: only the live decoder variant of each cell is decoded and scored;
//   (v)   the lgamma terms of the likelihood are evaluated per (cell, gene), not per particle;
//   (vi)  the one-hot block of the embedding encoder's first layer is a table lookup;
//   (vii) every matmul(D, .) / matmul(qc, .) is a gather / scatter over <= kmax non-zeros per cell;
//   (viii) when N*kmax < T, q(M) and q(E|M) are evaluated for the treatments present in the batch only
//          (compact treatment axis, kernels.h DosagePlan): the others have plate weight zero.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "kernels.h"

using namespace cradle;

// cradle_step_desc / cradle_mlp_ptrs: the one definition lives in the public header (via common.h)
namespace {

// Schedule experiments (tools only): CRADLE_SCHED is a bit mask read once.
//   2: no side streams at all (everything on the caller's stream)
//   4: q(qc_E) as the chain of GEMM launches instead of the single-workgroup kernels of qc_chain.hip
// cradle_step_debug_stop_after(k): the step returns (all streams joined) right after its k-th checkpoint;
// timing the truncated step in a steady-state loop gives the cumulative cost of its phases, boundaries and
// cross-stream waits included, without a profiler (tools/phase_sweep.py).  0 = run everything.
int g_stop_after = 0;

// CRADLE_PROBE=k (tools only): the event the caller passes as ev_gemm_begin is recorded at forward checkpoint k
// instead of right before the decoder's output layer, so that (ev_gemm_end - ev_gemm_begin) measures, on the
// GPU's own clock and in the steady state, the time from that checkpoint to the end of the output layer
// (k = 0..6), or from backward checkpoint k = 10..16 to the point where all streams have joined.
int probe_at() {
  static const int v = getenv("CRADLE_PROBE") ? atoi(getenv("CRADLE_PROBE")) : 6;
  return v;
}
int sched_flags() {
  static const int v = getenv("CRADLE_SCHED") ? atoi(getenv("CRADLE_SCHED")) : 0;
  return v;
}

struct Arena {
  char* base;
  int64_t off = 0;
  explicit Arena(void* b) : base(reinterpret_cast<char*>(b)) {}
  template <typename T>
  T* take(int64_t count) {
    off = (off + 255) & ~int64_t(255);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * (int64_t)sizeof(T);
    return p;
  }
};

constexpr int64_t kSplitKFloats = 8ll << 20;

struct Work {
  DosagePlan plan;
  CellPrep cp;
  float *hascf, *gene_consts;
  float *mask, *soft, *Smask, *W0tT;
  float *yE[CRADLE_MAX_LAYERS], *rE[CRADLE_MAX_LAYERS], *headsE, *sdE, *E, *SE;
  float *yq[CRADLE_MAX_LAYERS], *rq[CRADLE_MAX_LAYERS], *headsq, *sdq, *qcE, *Sq, *qsum, *ones_q;
  float *xn, *Xp, *xW, *offB, *zoff, *offW, *vq;
  float *yz[CRADLE_MAX_LAYERS], *rz[CRADLE_MAX_LAYERS], *headsz, *sdz, *ltz, *klrow;
  float *wq, *yd[CRADLE_MAX_LAYERS], *rd[CRADLE_MAX_LAYERS], *logits, *recon, *nbws, *nb_lg;
  // backward
  // backward: G ping-pongs on the main stream; the pre-activation gradients da[k] are kept per
  // layer because the weight-gradient GEMMs read them from a side stream
  float *Gd[2], *dad[CRADLE_MAX_LAYERS], *dzoff, *dwq, *dq_dec;
  float *dheadsz, *Gz[2], *daz[CRADLE_MAX_LAYERS], *doffW, *dxW, *dvq, *doffenc, *dq_enc;
  float *dmaskME, *dheadsE, *GE[2], *daE[CRADLE_MAX_LAYERS], *dmask_enc;
  float *dheadsq, *Gq[2], *daq[CRADLE_MAX_LAYERS];
  float* splitk[3];  // one split-K scratch per stream
  int64_t nbws_floats;
  float *u_mask, *eps_E, *eps_qc_E, *eps_z;  // the step's own draws (used when the caller passes none)
};

// Slots of the compact treatment axis (kernels.h, DosagePlan): a batch of N cells with at most kmax
// dosages each cannot hold more than N*kmax distinct treatments.
int compact_slots(const cradle_step_desc& d) {
  int64_t bound = (int64_t)d.N * d.kmax;
  if (d.max_present > 0 && d.max_present < bound) bound = d.max_present;  // the caller has counted (desc.max_present)
  return bound < d.T ? (int)bound : d.T;
}

// Row length of everything K1 touches (logits, counts, gene constants, d log_c partials): n_phenos rounded up to the
// 16-byte vectors that kernel moves.  K1 reads the genes behind n_phenos as count 0, logit -inf (mask_gene_tail) and
// c = 1e-30 (nb_gene_consts_kernel): they add nothing to the likelihood and exact zeros to every gradient.
int padded_genes(int G) { return (G + 3) & ~3; }

int64_t carve(const cradle_step_desc& d, void* base, Work& w) {
  Arena a(base);
  const int P = d.P, N = d.N, T = d.T, L = d.L, Q = d.Q, G = d.G, He = d.He, Hm = d.Hm, Hd = d.Hd;
  const int PP = 2 * P;
  const int Gp = padded_genes(G);
  const int Tc = compact_slots(d);
  const int64_t PN = (int64_t)P * N, PT = (int64_t)PP * Tc;
  w.plan.kmax = d.kmax;
  w.plan.Tc = Tc;
  // (reserved whatever Tc is: the layout then only shrinks when desc.max_present lowers Tc below what the workspace
  // was sized for)
  int* tmap = a.take<int>(Tc);
  int* tpos = a.take<int>(T);
  w.plan.tmap = Tc < T ? tmap : nullptr;
  w.plan.tpos = Tc < T ? tpos : nullptr;
  w.plan.ell_idx = a.take<int>((int64_t)N * d.kmax);
  w.plan.ell_val = a.take<float>((int64_t)N * d.kmax);
  w.plan.tcount = a.take<int>(T);
  w.plan.tstart = a.take<int>(T + 1);
  w.plan.csc_cell = a.take<int>((int64_t)N * d.kmax);
  w.plan.csc_val = a.take<float>((int64_t)N * d.kmax);
  w.plan.flags = d.flags;
  w.cp.alpha = a.take<float>(1);
  w.cp.live = a.take<float>(N);
  w.cp.wrec = a.take<float>(N);
  w.cp.grow = a.take<float>(PN);
  w.cp.cw = a.take<float>(T);
  w.cp.cwq = a.take<float>(Q);
  w.cp.coefT = a.take<float>(T);
  w.cp.coefQ = a.take<float>(Q);
  w.cp.cwc = a.take<float>(Tc);
  w.cp.coefTc = a.take<float>(Tc);
  w.hascf = a.take<float>(N);
  w.gene_consts = a.take<float>(3ll * Gp);
  w.mask = a.take<float>(PT * L);
  w.soft = a.take<float>(PT * L);
  w.Smask = a.take<float>(PT);
  w.W0tT = a.take<float>((int64_t)Tc * Hm);
  for (int k = 0; k < d.emb.n_layers; ++k) {
    w.yE[k] = a.take<float>(PT * Hm);
    w.rE[k] = k ? a.take<float>(PT * Hm) : nullptr;
  }
  w.headsE = a.take<float>(PT * 2 * L);
  w.sdE = a.take<float>(PT * L);
  w.E = a.take<float>(PT * L);
  w.SE = a.take<float>(PT);
  for (int k = 0; k < d.qce.n_layers; ++k) {
    w.yq[k] = a.take<float>((int64_t)Q * Hm);
    w.rq[k] = k ? a.take<float>((int64_t)Q * Hm) : nullptr;
  }
  w.headsq = a.take<float>((int64_t)Q * 2 * L);
  w.sdq = a.take<float>((int64_t)Q * L);
  w.qcE = a.take<float>((int64_t)PP * Q * L);
  w.Sq = a.take<float>((int64_t)PP * Q);
  w.qsum = a.take<float>((int64_t)P * L);
  w.ones_q = a.take<float>((int64_t)Q * L);
  w.xn = a.take<float>(2ll * N * Gp);
  w.Xp = Gp != G ? a.take<float>((int64_t)N * Gp) : nullptr;
  w.xW = a.take<float>(2ll * N * He);
  w.offB = a.take<float>(2 * PN * L);  // offsets of the main pass, then of the counterfactual pass
  w.zoff = a.take<float>(PN * 2 * L);
  w.offW = a.take<float>(2 * PN * He);
  w.vq = a.take<float>((int64_t)PP * Q * He);
  for (int k = 0; k < d.enc.n_layers; ++k) {
    w.yz[k] = a.take<float>(3 * PN * He);
    w.rz[k] = k ? a.take<float>(3 * PN * He) : nullptr;
  }
  w.headsz = a.take<float>(3 * PN * 2 * L);
  w.sdz = a.take<float>(PN * L);
  w.ltz = a.take<float>(PN);
  w.klrow = a.take<float>(PN);
  w.wq = a.take<float>((int64_t)P * Hd);
  for (int k = 0; k < d.dec.n_layers; ++k) {
    w.yd[k] = a.take<float>(PN * Hd);
    w.rd[k] = k ? a.take<float>(PN * Hd) : nullptr;
  }
  w.logits = a.take<float>(PN * Gp);
  w.recon = a.take<float>(PN);
  w.nbws_floats = cradle_nb_workspace_floats(P, N, Gp);
  w.nbws = a.take<float>(w.nbws_floats);
  w.nb_lg = a.take<float>(N);
  for (int i = 0; i < 2; ++i) {
    w.Gd[i] = a.take<float>(PN * Hd);
    w.Gz[i] = a.take<float>(3 * PN * He);
    w.GE[i] = a.take<float>(PT * Hm);
    w.Gq[i] = a.take<float>((int64_t)Q * Hm);
  }
  for (int k = 0; k < d.dec.n_layers; ++k) w.dad[k] = a.take<float>(PN * Hd);
  for (int k = 0; k < d.enc.n_layers; ++k) w.daz[k] = a.take<float>(3 * PN * He);
  for (int k = 0; k < d.emb.n_layers; ++k) w.daE[k] = a.take<float>(PT * Hm);
  for (int k = 0; k < d.qce.n_layers; ++k) w.daq[k] = a.take<float>((int64_t)Q * Hm);
  w.dzoff = a.take<float>(PN * 2 * L);
  w.dwq = a.take<float>((int64_t)P * Hd);
  w.dq_dec = a.take<float>((int64_t)P * L);
  w.dheadsz = a.take<float>(3 * PN * 2 * L);
  w.doffW = a.take<float>(2 * PN * He);
  w.dxW = a.take<float>(2ll * N * He);
  w.dvq = a.take<float>((int64_t)PP * Q * He);
  w.doffenc = a.take<float>(2 * PN * L);
  w.dq_enc = a.take<float>((int64_t)PP * Q * L);
  w.dmaskME = a.take<float>(PT * L);
  w.dheadsE = a.take<float>(PT * 2 * L);
  w.dmask_enc = a.take<float>(PT * L);
  w.dheadsq = a.take<float>((int64_t)Q * 2 * L);
  for (int i = 0; i < 3; ++i) w.splitk[i] = a.take<float>(kSplitKFloats);
  w.u_mask = a.take<float>((int64_t)PP * T * L);
  w.eps_E = a.take<float>((int64_t)PP * T * L);
  w.eps_qc_E = a.take<float>((int64_t)PP * Q * L);
  w.eps_z = a.take<float>(PN * L);
  return (a.off + 255) & ~int64_t(255);
}

// Side streams and fork/join events of the step, created once per device.  The step is a DAG
// (the x Wx^T product, the qc encoder, and every weight-gradient GEMM are off the critical
// path); most of its kernels fill only part of the 256 CUs, so running the independent
// branches on two side streams lets the dispatcher co-schedule them.
constexpr int kMaxEvents = 160;
struct StreamSet {
  bool ready = false;
  hipStream_t side[2];
  hipStream_t marker;  // carries only event waits/records: "all three streams have reached this point"
  hipEvent_t ev[kMaxEvents];
};
StreamSet g_streams[16];

int get_streams(StreamSet** out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
    cradle_set_error("elbo_step: cannot identify the current device");
    return CRADLE_ERR_HIP;
  }
  StreamSet& ss = g_streams[dev];
  if (!ss.ready) {
    for (int i = 0; i < 2; ++i)
      if (hipStreamCreateWithFlags(&ss.side[i], hipStreamNonBlocking) != hipSuccess) {
        cradle_set_error("elbo_step: side stream creation failed");
        return CRADLE_ERR_HIP;
      }
    if (hipStreamCreateWithFlags(&ss.marker, hipStreamNonBlocking) != hipSuccess) {
      cradle_set_error("elbo_step: marker stream creation failed");
      return CRADLE_ERR_HIP;
    }
    // fork/join events order kernels of one device: no system-scope release (a record costs the stream it is
    // made on 3.4 us instead of 4.8 us between two dependent kernels, tools/micro/event_cost.hip)
    const unsigned evflags = hipEventDisableTiming | hipEventDisableSystemFence;
    for (int i = 0; i < kMaxEvents; ++i)
      if (hipEventCreateWithFlags(&ss.ev[i], evflags) != hipSuccess) {
        cradle_set_error("elbo_step: event creation failed");
        return CRADLE_ERR_HIP;
      }
    ss.ready = true;
  }
  *out = &ss;
  return CRADLE_OK;
}

struct Ctx {
  hipStream_t marker = nullptr;
  hipStream_t st[3];  // 0 = the caller's stream (critical path), 1/2 = side streams
  float* splitk[3];
  hipEvent_t* ev;
  int nev = 0;
  int rc = 0;
  hipStream_t s() const { return st[0]; }
  // C[M,N] = A . B(^T) with epilogue options set by the caller in p, on stream q
  int precision = 0;
  void gemm(int layout, GemmParams p, int q = 0) {
    if (rc) return;
    p.precision = precision;
    rc = gemm_f32(layout, p, splitk[q], kSplitKFloats, st[q]);
  }
  // everything enqueued on stream `dst` from now on waits for what `src` holds so far
  void after(int dst, int src) {
    if (rc) return;
    if (nev >= kMaxEvents) {
      cradle_set_error("elbo_step: out of fork/join events");
      rc = CRADLE_ERR_HIP;
      return;
    }
    hipEvent_t e = ev[nev++];
    if (hipEventRecord(e, st[src]) != hipSuccess || hipStreamWaitEvent(st[dst], e, 0) != hipSuccess) {
      cradle_set_error("elbo_step: event record/wait failed");
      rc = CRADLE_ERR_HIP;
    }
  }
  // the two halves of after(): a point on `src` that several later waits can refer to
  hipEvent_t record(int src) {
    if (rc) return nullptr;
    if (nev >= kMaxEvents) {
      cradle_set_error("elbo_step: out of fork/join events");
      rc = CRADLE_ERR_HIP;
      return nullptr;
    }
    hipEvent_t e = ev[nev++];
    if (hipEventRecord(e, st[src]) != hipSuccess) {
      cradle_set_error("elbo_step: event record failed");
      rc = CRADLE_ERR_HIP;
    }
    return e;
  }
  void wait(int dst, hipEvent_t e) {
    if (rc || !e) return;
    if (hipStreamWaitEvent(st[dst], e, 0) != hipSuccess) {
      cradle_set_error("elbo_step: event wait failed");
      rc = CRADLE_ERR_HIP;
    }
  }
};

// records `user_ev` at the point where all three streams have finished what they hold so far
void mark_all(Ctx& c, void* user_ev) {
  if (!user_ev || c.rc || !c.marker) return;
  for (int q = 0; q < 3; ++q) {
    if (c.nev >= kMaxEvents) {
      cradle_set_error("elbo_step: out of fork/join events");
      c.rc = CRADLE_ERR_HIP;
      return;
    }
    hipEvent_t e = c.ev[c.nev++];
    if (hipEventRecord(e, c.st[q]) != hipSuccess || hipStreamWaitEvent(c.marker, e, 0) != hipSuccess) {
      cradle_set_error("elbo_step: event record/wait failed");
      c.rc = CRADLE_ERR_HIP;
      return;
    }
  }
  if (hipEventRecord((hipEvent_t)user_ev, c.marker) != hipSuccess) {
    cradle_set_error("elbo_step: bucket event record failed");
    c.rc = CRADLE_ERR_HIP;
  }
}

GemmParams gp(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
  GemmParams p = {};
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  return p;
}

// Layers 1..n-1 of a ResidualMLP (mlp.py:111-119): y_k = LeakyReLU(y_{k-1} W_k^T + b_k) + y_{k-1}.
void trunk_fwd(Ctx& c, const cradle_mlp_ptrs& m, int M, int H, float* const* y, float* const* r, int q = 0) {
  for (int k = 1; k < m.n_layers; ++k) {
    GemmParams p = gp(y[k - 1], H, m.W[k], H, y[k], H, M, H, H);
    p.bias = m.b[k]; p.act = 1; p.act_out = r[k]; p.ld_act = H; p.res = y[k - 1]; p.ld_res = H;
    c.gemm(GEMM_NT, p, q);
  }
}

// Output layer forward: out = y_last W_head^T + b_head.
void head_fwd(Ctx& c, const cradle_mlp_ptrs& m, int M, int H, int Nout, const float* ylast, float* out, int q = 0,
              int ld_out = 0) {
  GemmParams p = gp(ylast, H, m.W_head, H, out, ld_out > 0 ? ld_out : Nout, M, Nout, H);
  p.bias = m.b_head;
  c.gemm(GEMM_NT, p, q);
}

// Backward through the output layer and trunk layers n-1..1 on stream `q`.  dout: (M, Nout)
// gradient of the output layer's result.  Writes dW_head, db_head, dW[k], db[k] (k >= 1) and
// returns da_0 = dL/d(pre-activation of layer 0) in (M, H).  The data-gradient chain stays on
// `q`; the weight-gradient GEMMs run on the side streams `wq` / `wq2` (pass wq == q to keep everything on
// one stream).  An event record between two dependent kernels costs the recording stream 3-5 us, so the
// chain is not interrupted once per layer: the output layer's weight gradient forks at `ev_dout` (a point on
// `q` where dout exists; null: recorded here); the trunk layers' fork in groups of `layers_per_fork` (0: all
// at one event after the last data gradient).  The event recorded after the last data gradient, if any, is
// handed back in *ev_done (else null) for the caller's own forks.
const float* head_trunk_bwd(Ctx& c, const cradle_mlp_ptrs& m, const cradle_mlp_ptrs& g, int M, int H, int Nout,
                            const float* dout, float* const* y, float* const* r, float* const* Gb, float* const* da,
                            int q, int wq, int wq2, hipEvent_t ev_dout = nullptr, hipEvent_t* ev_done = nullptr,
                            void* ev_head_done = nullptr, int layers_per_fork = 0, int ld_dout = 0,
                            hipEvent_t ev_head_also = nullptr) {
  // ld_dout > Nout: rows of dout padded to a multiple of 4 with zeros (the decoder's d logits)
  if (ld_dout <= 0) ld_dout = Nout;
  const int n = m.n_layers;
  const bool forked = wq != q;
  if (ev_done) *ev_done = nullptr;
  {  // dW_head = dout^T y_last, db_head = colsum(dout)
    GemmParams p = gp(dout, ld_dout, y[n - 1], H, g.W_head, H, Nout, H, M);
    p.extra_col = g.b_head;
    if (forked) {
      if (!ev_dout) ev_dout = c.record(q);
      c.wait(wq, ev_dout);
    }
    c.gemm(GEMM_TN, p, wq);
    // the output layer's gradients have been enqueued, all of them on `wq`: the event goes there -- behind
    // `ev_head_also`, if given: what else the caller counts to the same gradient bucket and makes on another stream
    // (the decoder's d log_concentration: K1's column sum on stream 2)
    if (ev_head_done && ev_head_also) c.wait(wq, ev_head_also);
    if (ev_head_done && !c.rc && hipEventRecord((hipEvent_t)ev_head_done, c.st[wq]) != hipSuccess) {
      cradle_set_error("elbo_step: bucket event record failed");
      c.rc = CRADLE_ERR_HIP;
    }
  }
  int side = wq2;  // the output layer's product went to wq
  auto wgrad = [&](int k, int stream) {  // dW_k = da_k^T y_{k-1}
    GemmParams p = gp(da[k], H, y[k - 1], H, g.W[k], H, H, H, M);
    p.extra_col = g.b[k];
    c.gemm(GEMM_TN, p, stream);
  };
  int pending[CRADLE_MAX_LAYERS], n_pending = 0;
  auto flush = [&]() {  // one event for every weight gradient whose da exists by now
    if (!n_pending) return;
    hipEvent_t e = c.record(q);
    c.wait(wq, e);
    if (wq2 != wq) c.wait(wq2, e);
    // hidden layers have one shape: when their buffers are equally spaced (the workspace and the parameter arena lay
    // them out that way) the pending weight gradients are ONE launch -- grid y = layer -- instead of one each
    bool strided = n_pending > 1 && !(sched_flags() & 8);
    int64_t sa = 0, sb = 0, sc = 0, sx = 0;
    if (strided) {
      const int k0 = pending[0], k1 = pending[1];
      sa = da[k1] - da[k0]; sb = y[k1 - 1] - y[k0 - 1]; sc = g.W[k1] - g.W[k0]; sx = g.b[k1] - g.b[k0];
      for (int i = 2; i < n_pending && strided; ++i) {
        const int ka = pending[i - 1], kb = pending[i];
        strided = da[kb] - da[ka] == sa && y[kb - 1] - y[ka - 1] == sb && g.W[kb] - g.W[ka] == sc && g.b[kb] - g.b[ka] == sx;
      }
    }
    if (strided) {
      const int k0 = pending[0];
      GemmParams p = gp(da[k0], H, y[k0 - 1], H, g.W[k0], H, H, H, M);
      p.extra_col = g.b[k0];
      p.batch = n_pending; p.bs_a = sa; p.bs_b = sb; p.bs_c = sc; p.bs_x = sx;
      c.gemm(GEMM_TN, p, side);
      side = side == wq ? wq2 : wq;
    } else {
      for (int i = 0; i < n_pending; ++i) {
        wgrad(pending[i], side);
        side = side == wq ? wq2 : wq;
      }
    }
    n_pending = 0;
    if (ev_done) *ev_done = e;
  };
  int cur = 0;
  {  // G_{n-1} = dout W_head ; da_{n-1} = G_{n-1} * slope(r_{n-1})
    GemmParams p = gp(dout, ld_dout, m.W_head, H, Gb[cur], H, M, H, ld_dout);
    p.k_valid = Nout;  // W_head has Nout rows: the loader's range check supplies zeros for the padding
    p.dact_src = n > 1 ? r[n - 1] : y[0]; p.ld_ds = H; p.C2 = da[n - 1]; p.ldc2 = H;
    c.gemm(GEMM_NN, p, q);
  }
  for (int k = n - 1; k >= 1; --k) {
    if (!forked) wgrad(k, q);
    else {
      if (ev_done) *ev_done = nullptr;  // a data gradient follows: an earlier event no longer is "the end"
      if (layers_per_fork > 0 && n_pending == layers_per_fork) flush();
      pending[n_pending++] = k;
    }
    {  // G_{k-1} = da_k W_k + G_k ; da_{k-1} = G_{k-1} * slope(r_{k-1})   (r_0 = y_0)
      GemmParams p = gp(da[k], H, m.W[k], H, Gb[cur ^ 1], H, M, H, H);
      p.res = Gb[cur]; p.ld_res = H;
      p.dact_src = k - 1 >= 1 ? r[k - 1] : y[0]; p.ld_ds = H; p.C2 = da[k - 1]; p.ldc2 = H;
      c.gemm(GEMM_NN, p, q);
    }
    cur ^= 1;
  }
  flush();
  return da[0];
}

__global__ void fill_kernel(float* p, int64_t n, float v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

int check_desc(const cradle_step_desc& d) {
  CRADLE_CHECK_ARG(d.P > 0 && d.N > 0 && d.T > 0 && d.L > 0 && d.Q > 0 && d.G > 0 && d.He > 0 && d.Hm > 0 && d.Hd > 0,
                   "elbo_step: all sizes must be positive");
  CRADLE_CHECK_ARG(d.G >= 4, "elbo_step: n_phenos=%d below 4", d.G);
  CRADLE_CHECK_ARG(d.G <= 16384, "elbo_step: n_genes=%d above 16384, the widest row K1 holds in registers", d.G);
  CRADLE_CHECK_ARG(d.kmax >= 1 && d.kmax <= 64, "elbo_step: kmax=%d out of range [1,64]", d.kmax);
  CRADLE_CHECK_ARG(d.max_present >= 0, "elbo_step: max_present=%d is negative", d.max_present);
  const cradle_mlp_ptrs* ms[4] = {&d.emb, &d.qce, &d.enc, &d.dec};
  const cradle_mlp_ptrs* gs[4] = {&d.d_emb, &d.d_qce, &d.d_enc, &d.d_dec};
  for (int i = 0; i < 4; ++i) {
    CRADLE_CHECK_ARG(ms[i]->n_layers >= 1 && ms[i]->n_layers <= CRADLE_MAX_LAYERS, "elbo_step: mlp %d has %d layers", i,
                     ms[i]->n_layers);
    CRADLE_CHECK_ARG(ms[i]->W_head && ms[i]->b_head, "elbo_step: mlp %d has a null output layer", i);
    for (int k = 0; k < ms[i]->n_layers; ++k) {
      CRADLE_CHECK_ARG(ms[i]->W[k] && ms[i]->b[k], "elbo_step: mlp %d layer %d is null", i, k);
      CRADLE_CHECK_ARG(!d.compute_grads || (gs[i]->W[k] && gs[i]->b[k] && gs[i]->W_head && gs[i]->b_head),
                       "elbo_step: gradient pointers of mlp %d layer %d are null", i, k);
    }
  }
  CRADLE_CHECK_ARG(d.q_mask_logits && d.log_concentration && d.p_E_loc && d.p_E_scale && d.p_mask_probs &&
                       d.p_qc_E_loc && d.p_qc_E_scale && d.log_mean && d.log_scale,
                   "elbo_step: null parameter/buffer pointer");
  CRADLE_CHECK_ARG(!d.compute_grads || (d.d_q_mask_logits && d.d_log_concentration), "elbo_step: null gradient pointer");
  CRADLE_CHECK_ARG(d.X && d.cfX && d.D && d.qc && d.library_size && d.D_obs_counts && d.qc_obs_counts,
                   "elbo_step: null batch pointer");
  CRADLE_CHECK_ARG((d.u_mask && d.eps_E && d.eps_qc_E && d.eps_z) || (!d.u_mask && !d.eps_E && !d.eps_qc_E && !d.eps_z),
                   "elbo_step: pass all four noise tensors or none (none: the step draws them itself)");
  CRADLE_CHECK_ARG(d.out && d.flags && d.workspace, "elbo_step: null output/workspace pointer");
  CRADLE_CHECK_ARG((uintptr_t)d.workspace % 256 == 0, "elbo_step: workspace must be 256-byte aligned");
  CRADLE_CHECK_ARG((uintptr_t)d.X % 16 == 0, "elbo_step: X must be 16-byte aligned");
  CRADLE_CHECK_ARG(d.temperature > 0.f, "elbo_step: temperature must be positive");
  CRADLE_CHECK_ARG(d.precision >= 0 && d.precision <= 2, "elbo_step: precision must be 0 (fp32), 1 (bf16x3) or 2 (bf16), got %d",
                   d.precision);
  return CRADLE_OK;
}

}  // namespace

extern "C" {

int64_t cradle_elbo_step_workspace_bytes(const cradle_step_desc* d) {
  if (!d) return -1;
  Work w;
  return carve(*d, nullptr, w);
}

// Where an intermediate of the last step lives inside desc->workspace (for parity tests: the mask and
// the latent draws are compared with the reference's own).  name: "mask" (2P*Tc*L, rows (particle,
// slot)), "E" (same), "qc_E" (2P*Q*L), "z_offset" (P*N rows of [z_basal | D@(E*mask)], ld 2L),
// "logits" (P*N rows of n_phenos rounded up to a multiple of 4; overwritten by d logits when gradients were computed).  Returns the element
// count and the pointer, or -1 for an unknown name.
int64_t cradle_elbo_step_intermediate(const cradle_step_desc* d, const char* name, const float** ptr) {
  if (!d || !name || !ptr || !d->workspace) return -1;
  Work w;
  carve(*d, d->workspace, w);
  const int64_t PP = 2ll * d->P, PN = (int64_t)d->P * d->N, Tc = w.plan.Tc;
  if (!strcmp(name, "mask")) { *ptr = w.mask; return PP * Tc * d->L; }
  if (!strcmp(name, "E")) { *ptr = w.E; return PP * Tc * d->L; }
  if (!strcmp(name, "qc_E")) { *ptr = w.qcE; return PP * d->Q * d->L; }
  if (!strcmp(name, "z_offset")) { *ptr = w.zoff; return PN * 2 * d->L; }
  if (!strcmp(name, "logits")) { *ptr = w.logits; return PN * padded_genes(d->G); }
  // the step's own draws (when the caller passed no noise)
  if (!strcmp(name, "u_mask")) { *ptr = w.u_mask; return PP * d->T * d->L; }
  if (!strcmp(name, "eps_E")) { *ptr = w.eps_E; return PP * d->T * d->L; }
  if (!strcmp(name, "eps_qc_E")) { *ptr = w.eps_qc_E; return PP * d->Q * d->L; }
  if (!strcmp(name, "eps_z")) { *ptr = w.eps_z; return PN * d->L; }
  // diagnostics (tools/grad_errors.py): decoder layer-0 activation and pre-activation gradient, (P*N, Hd)
  if (!strcmp(name, "dec_y0")) { *ptr = w.yd[0]; return PN * d->Hd; }
  if (!strcmp(name, "dec_da0")) { *ptr = w.dad[0]; return PN * d->Hd; }
  return -1;
}

}  // extern "C"

namespace {

int run_step(const cradle_step_desc& d, Work& w, Ctx& c) {
  const int P = d.P, N = d.N, T = d.T, L = d.L, Q = d.Q, G = d.G, He = d.He, Hm = d.Hm, Hd = d.Hd;
  const int Tc = w.plan.Tc;
  const int PP = 2 * P, PN = P * N, PT = PP * Tc;
  const int Gp = padded_genes(G);  // == G for the widths K1 takes as they are
  int rc = 0;
  hipStream_t s = c.st[0], s1 = c.st[1], s2 = c.st[2];
  const float* Wenc0 = d.enc.W[0];
  const int64_t ldenc = (int64_t)G + 2 * L;
  const float* Wd0 = d.dec.W[0];
  const bool own_noise = !d.u_mask;  // check_desc: all four given or none
  const float* nz_u = own_noise ? w.u_mask : d.u_mask;
  const float* nz_E = own_noise ? w.eps_E : d.eps_E;
  const float* nz_q = own_noise ? w.eps_qc_E : d.eps_qc_E;
  const float* nz_z = own_noise ? w.eps_z : d.eps_z;
  const bool qc_fused = qc_chain_supported(Q, L, Hm, PP) && !(sched_flags() & 4);
  const bool nb_pre = d.compute_grads && !(sched_flags() & 16);
#define STEP(call)        \
  do {                    \
    rc = (call);          \
    if (rc) return rc;    \
    if (c.rc) return c.rc; \
  } while (0)

  {
  // ================================================================== forward
  // stream 1: dosage plan (three short launches; stream 0 needs it at the gather, or -- compact axis -- at once),
  //           log-standardise X and cfX, x Wx^T, then what only the loss weights and the backward scatter need
  //           (CSC columns, per-cell weights).
  // stream 2: q(qc_E) -- a chain of ten launches on a handful of rows, the longest path to the join when anything
  //           else shares its stream -- and, last, the per-gene constants K1 reads.
  // stream 0 (caller's): the step's noise, q(M), q(E|M), gathers -- then joins both.
#define PROBE(k) do { if (probe_at() == (k) && d.ev_gemm_begin) (void)hipEventRecord((hipEvent_t)d.ev_gemm_begin, s); } while (0)
  PROBE(0);
  {
    hipEvent_t e = c.record(0);  // one record on the caller's stream for both forks
    c.wait(1, e);
    c.wait(2, e);
  }
  STEP(build_dosage_plan(d.D, N, T, w.plan, s1));
  hipEvent_t ev_plan = c.record(1);
  STEP(log_standardize(d.X, d.cfX, d.log_mean, d.log_scale, w.xn, w.hascf, N, G, s1, Gp, w.Xp));
  {
    // x Wx^T, X and cfX.  Padded width: xn holds zeros behind a row's G values, so whatever the loader reads behind
    // Wx's G columns (the offset block of the same row; nothing past the matrix) meets a zero
    GemmParams p = gp(w.xn, Gp, Wenc0, ldenc, w.xW, He, 2 * N, He, Gp);
    p.k_valid = G;
    p.splitk_hint = 4;  // beside the embedding encoder's chain: 8 us less to the join than the stand-alone heuristic's 2
    c.gemm(GEMM_NT, p, 1);
  }
  hipEvent_t ev_xw = c.record(1);
  STEP(build_dosage_plan_csc(N, T, w.plan, s1));
  STEP(cell_prep(d.qc, d.D_obs_counts, d.qc_obs_counts, w.plan.tcount, d.alpha, P, N, T, Q, w.cp, d.flags, Tc, w.plan.tmap,
                 s1));
  // K1's particle-independent terms (per-gene constants, then the lgamma / digamma terms of every count): behind
  // everything stream 1 owes the encoders, long before the decoder joins it (CRADLE_SCHED bit 16: inside K1 instead)
  STEP(nb_gene_consts_padded(d.log_concentration, w.gene_consts, G, Gp, s1));
  if (nb_pre)
    STEP(nb_count_terms(w.Xp ? w.Xp : d.X, Gp, w.gene_consts, w.nbws, w.nbws_floats, w.nb_lg, P, N, G, s1));
  // the step's own noise: q(qc_E)'s on the stream that reads it, the rest on the caller's -- no event between them
  // Compact treatment axis: u_mask / eps_E are (2P, T, L), but only the rows of the treatments in the batch are ever
  // read -- those are drawn once the slot map exists (below), the same values at the same places
  const bool slot_noise = own_noise && w.plan.tmap && L % 4 == 0;
  if (own_noise) {
    STEP(draw_noise(w.u_mask, w.eps_E, w.eps_qc_E, w.eps_z, (int64_t)PP * T * L, (int64_t)PP * Q * L, (int64_t)PN * L,
                    d.rng_seed_plated, d.rng_seed_local, d.rng_step, slot_noise ? 8 : (1 | 2 | 8), s));
    STEP(draw_noise(w.u_mask, w.eps_E, w.eps_qc_E, w.eps_z, (int64_t)PP * T * L, (int64_t)PP * Q * L, (int64_t)PN * L,
                    d.rng_seed_plated, d.rng_seed_local, d.rng_step, 4, s2));
  }

  hipEvent_t ev_w0t = nullptr;
  if (!w.plan.tmap) {  // one-hot block of the embedding encoder's first layer as a (T, Hm) table: weights only
    STEP(transpose_cols(d.emb.W[0], L + T, Hm, L, T, w.W0tT, s2));
    ev_w0t = c.record(2);
  }
  // q(qc_E): Q rows through the qc encoder.  One single-workgroup launch (qc_chain.hip) when Q is a handful (every
  // shipped configuration: 1); else, or with CRADLE_SCHED bit 2 set, the chain of GEMM launches it replaces
  hipEvent_t ev_vq = nullptr;
  if (qc_fused) {
    STEP(qc_chain_fwd(d.qce, w.yq, w.rq, w.headsq, nz_q, d.p_qc_E_loc, d.p_qc_E_scale, w.qcE, w.sdq, w.Sq, w.qsum, Q, L, Hm,
                      PP, P, s2));
  } else {
    hipLaunchKernelGGL(fill_kernel, dim3((Q * L + 255) / 256), dim3(256), 0, s2, w.ones_q, (int64_t)Q * L, 1.0f);
    {  // constant input ones(Q, L)
      GemmParams p = gp(w.ones_q, L, d.qce.W[0], L, w.yq[0], Hm, Q, Hm, L);
      p.bias = d.qce.b[0]; p.act = 1;
      c.gemm(GEMM_NT, p, 2);
    }
    trunk_fwd(c, d.qce, Q, Hm, w.yq, w.rq, 2);
    head_fwd(c, d.qce, Q, Hm, 2 * L, w.yq[d.qce.n_layers - 1], w.headsq, 2);
    if (c.rc) return c.rc;
    STEP(qce_fwd(w.headsq, nz_q, d.p_qc_E_loc, d.p_qc_E_scale, w.qcE, w.sdq, w.Sq, w.qsum, PP, P, Q, L, s2));
  }
  c.gemm(GEMM_NT, gp(w.qcE, L, Wenc0 + G + L, ldenc, w.vq, He, PP * Q, He, L), 2);  // qc_E Wqc^T
  ev_vq = c.record(2);
  c.gemm(GEMM_NT, gp(w.qsum, L, Wd0 + 2 * L, 3 * L, w.wq, Hd, P, Hd, L), 2);        // bad-variant shift per particle
  hipEvent_t ev_s2 = c.record(2);

  if (w.plan.tmap) c.wait(0, ev_plan);
  if (slot_noise)
    STEP(draw_noise_slots(w.u_mask, w.eps_E, w.plan.tmap, PP, T, Tc, L, d.rng_seed_plated, d.rng_step, s));
  PROBE(1);
  STEP(mask_fwd(d.q_mask_logits, nz_u, d.p_mask_probs, d.temperature, w.mask, w.soft, w.Smask, PP, T, Tc, w.plan.tmap,
                L, s));
  if (w.plan.tmap)
    STEP(gather_cols(d.emb.W[0], L + T, Hm, L, Tc, w.plan.tmap, w.W0tT, s));
  else
    c.wait(0, ev_w0t);
  if (g_stop_after == 1) { c.after(0, 1); c.after(0, 2); return c.rc; }
  {  // layer 0: LeakyReLU(mask Wm^T + W_onehot[:, t] + b)
    GemmParams p = gp(w.mask, L, d.emb.W[0], L + T, w.yE[0], Hm, PT, Hm, L);
    p.bias = d.emb.b[0]; p.act = 1; p.rowbias = w.W0tT; p.ld_rb = Hm; p.rb_mode = 2; p.rb_n = Tc;
    c.gemm(GEMM_NT, p);
  }
  trunk_fwd(c, d.emb, PT, Hm, w.yE, w.rE);
  head_fwd(c, d.emb, PT, Hm, 2 * L, w.yE[d.emb.n_layers - 1], w.headsE);
  if (c.rc) return c.rc;
  if (g_stop_after == 2) { c.after(0, 1); c.after(0, 2); return c.rc; }
  // (E, its scale and the log-ratio sums are needed by the loss only -- the gather below and the backward pass
  // draw the E they need from the heads themselves -- and are made on stream 2 after K1, at a fork that exists anyway)
  if (!w.plan.tmap) c.wait(0, ev_plan);
  PROBE(2);
  STEP(gather_offsets(w.headsE, nz_E, w.mask, w.plan, w.zoff + L, 2 * L, w.offB, L, PP, P, N, T, L, s));  // D @ (mask*E)
  c.gemm(GEMM_NT, gp(w.offB, L, Wenc0 + G, ldenc, w.offW, He, 2 * PN, He, L));  // offsets . Woff^T, both passes
  c.wait(0, ev_xw);  // the encoders need x Wx^T and qc_E Wqc^T; the rest of both side streams is met at the decoder
  c.wait(0, ev_vq);
  if (c.rc) return c.rc;
  if (g_stop_after == 3) { c.after(0, 1); c.after(0, 2); return c.rc; }

  // basal encoder (three variants), z_basal, counterfactual KL
  PROBE(3);
  STEP(enc_combine_fwd(w.xW, w.offW, w.vq, d.enc.b[0], d.qc, w.yz[0], P, N, Q, He, s));
  trunk_fwd(c, d.enc, 3 * PN, He, w.yz, w.rz);
  head_fwd(c, d.enc, 3 * PN, He, 2 * L, w.yz[d.enc.n_layers - 1], w.headsz);
  if (c.rc) return c.rc;
  if (g_stop_after == 4) { c.after(0, 1); c.after(0, 2); return c.rc; }
  PROBE(4);
  STEP(zhead_fwd(w.headsz, nz_z, w.hascf, d.beta, w.zoff, 2 * L, w.sdz, w.ltz, w.klrow, w.dheadsz, P, N, L, s));

  // decoder, live variant per cell (per-cell weights: stream 1; bad-variant shift, gene constants: stream 2)
  c.after(0, 1);
  c.wait(0, ev_s2);
  PROBE(5);
  {
    GemmParams p = gp(w.zoff, 2 * L, Wd0, 3 * L, w.yd[0], Hd, PN, Hd, 2 * L);
    p.bias = d.dec.b[0]; p.act = 1; p.rowbias = w.wq; p.ld_rb = Hd; p.rb_mode = 1; p.rb_n = N;
    p.rowscale = w.cp.live; p.rs_n = N;
    c.gemm(GEMM_NT, p);
  }
  trunk_fwd(c, d.dec, PN, Hd, w.yd, w.rd);
  if (g_stop_after == 5) return c.rc;
  PROBE(6);
  head_fwd(c, d.dec, PN, Hd, G, w.yd[d.dec.n_layers - 1], w.logits, 0, Gp);
  if (d.ev_gemm_end && probe_at() < 10) (void)hipEventRecord((hipEvent_t)d.ev_gemm_end, s);
  if (c.rc) return c.rc;
  if (g_stop_after == 6) return c.rc;
  }
  {
  const float* Xk = w.Xp ? w.Xp : d.X;  // counts in K1's row length (written beside the standardised rows on stream 1)
  if (!d.compute_grads) {
    if (d.ev_k1_begin) (void)hipEventRecord((hipEvent_t)d.ev_k1_begin, s);
    STEP(cradle_nb_logprob_fwd(w.logits, Gp, Xk, Gp, d.library_size, w.gene_consts, w.recon, P, N, G, s));
    if (d.ev_k1_end) (void)hipEventRecord((hipEvent_t)d.ev_k1_end, s);
  } else {  // events bracket the rows kernel only (the launch the roofline in bench.py describes)
    if (nb_pre)
      STEP(nb_logprob_fwd_bwd_pre(w.logits, Gp, w.logits, Gp, Xk, Gp, d.library_size, w.gene_consts, w.cp.grow, w.recon,
                                  w.nbws, w.nbws_floats, w.nb_lg, P, N, G, d.ev_k1_begin, d.ev_k1_end, s));
    else
      STEP(cradle_nb_logprob_fwd_bwd_timed(w.logits, Gp, w.logits, Gp, Xk, Gp, d.library_size, w.gene_consts, w.cp.grow,
                                           w.recon, nullptr, 0, w.nbws, w.nbws_floats, P, N, G, d.ev_k1_begin,
                                           d.ev_k1_end, s));
  }
  }
  // the loss scalars, d log_c and the q(E|M) bookkeeping are leaves of the graph: stream 2, beside the backward
  // pass; the same event forks the output layer's weight gradient to stream 1
  hipEvent_t ev_k1 = c.record(0);
  c.wait(2, ev_k1);
  // (d log_c first: it lies in the first gradient bucket -- output layer + log_concentration -- whose event, when a
  // data-parallel caller asked for one, must not fire before this sum has been written)
  hipEvent_t ev_dlogc = nullptr;
  if (d.compute_grads) {
    STEP(nb_dlogc_reduce(w.nbws, N * cradle_nb_particle_split(P, N), Gp, d.d_log_concentration, 0, s2, G));
    if (d.ev_bucket[0]) ev_dlogc = c.record(2);
  }
  STEP(gaussE_fwd(w.headsE, nz_E, d.p_E_loc, d.p_E_scale, w.E, w.sdE, w.SE, PP, T, Tc, w.plan.tmap, L, s2));
  STEP(finalize_loss(w.recon, w.cp.wrec, w.ltz, w.klrow, w.Smask, w.SE, w.Sq, w.cp.cwc, w.cp.cwq, d.beta, P, N, Tc, Q,
                     d.flags, d.out, s2));
  if (!d.compute_grads) {
    c.after(0, 2);
    return c.rc;
  }

  // ================================================================== backward
  // The data-gradient chain runs on stream 0; the weight-gradient GEMMs are forked to the side streams at as few
  // events as possible (see head_trunk_bwd).  All streams are joined before returning.
  // decoder: logits now hold d loss / d logits
  PROBE(10);
  hipEvent_t ev_dec = nullptr;
  const float* da0d = head_trunk_bwd(c, d.dec, d.d_dec, PN, Hd, G, w.logits, w.yd, w.rd, w.Gd, w.dad, 0, 1, 2, ev_k1,
                                     &ev_dec, d.ev_bucket[0], 0, Gp, ev_dlogc);
  if (c.rc) return c.rc;
  PROBE(11);
  if (!ev_dec) ev_dec = c.record(0);
  c.wait(2, ev_dec);
  {  // dW0[:, :2L] = da0^T [z | off], db0
    GemmParams p = gp(da0d, Hd, w.zoff, 2 * L, d.d_dec.W[0], 3 * L, Hd, 2 * L, PN);
    p.extra_col = d.d_dec.b[0];
    c.gemm(GEMM_TN, p, 2);
  }
  // the bad-variant shift (per particle) only feeds q(qc_E), whose backward lives on stream 2
  STEP(weighted_rowgroup_sum(da0d, w.cp.live, P, N, Hd, w.dwq, s2));
  c.gemm(GEMM_TN, gp(w.dwq, Hd, w.qsum, L, d.d_dec.W[0] + 2 * L, 3 * L, Hd, L, P), 2);  // dW0[:, 2L:]
  mark_all(c, d.ev_bucket[1]);  // every decoder gradient has been enqueued
  c.gemm(GEMM_NN, gp(w.dwq, Hd, Wd0 + 2 * L, 3 * L, w.dq_dec, L, P, L, Hd), 2);         // d qsum
  c.gemm(GEMM_NN, gp(da0d, Hd, Wd0, 3 * L, w.dzoff, 2 * L, PN, 2 * L, Hd));             // [dz | doff]
  if (c.rc) return c.rc;

  // basal encoder
  STEP(zhead_bwd(w.headsz, nz_z, w.zoff, 2 * L, w.sdz, w.dzoff, 2 * L, w.dheadsz, P, N, L, s));
  PROBE(12);
  const float* dpre = head_trunk_bwd(c, d.enc, d.d_enc, 3 * PN, He, 2 * L, w.dheadsz, w.yz, w.rz, w.Gz, w.daz, 0, 1, 2);
  if (c.rc) return c.rc;
  PROBE(13);
  STEP(enc_combine_bwd(dpre, d.qc, w.doffW, w.dxW, w.dvq, P, N, Q, He, s, s2, 1));
  float* dWenc0 = d.d_enc.W[0];
  {
    hipEvent_t e = c.record(0);  // dpre, doffW, dxW exist
    c.wait(1, e);
    c.wait(2, e);
  }
  STEP(enc_combine_bwd(dpre, d.qc, w.doffW, w.dxW, w.dvq, P, N, Q, He, s, s2, 2));  // dvq: feeds dWqc and q(qc_E) only
  {  // dWx = dxW^T logstd([X; cfX]); db0 = colsum(dxW)
    GemmParams p = gp(w.dxW, He, w.xn, Gp, dWenc0, ldenc, He, G, 2 * N);
    p.extra_col = d.d_enc.b[0];
    c.gemm(GEMM_TN, p, 1);
  }
  c.gemm(GEMM_TN, gp(w.doffW, He, w.offB, L, dWenc0 + G, ldenc, He, L, 2 * PN), 1);  // dWoff = doffW^T [offA; offB]
  c.gemm(GEMM_TN, gp(w.dvq, He, w.qcE, L, dWenc0 + G + L, ldenc, He, L, PP * Q), 2);      // dWqc
  mark_all(c, d.ev_bucket[2]);  // every basal-encoder gradient has been enqueued
  c.gemm(GEMM_NN, gp(w.doffW, He, Wenc0 + G, ldenc, w.doffenc, L, 2 * PN, L, He));        // d offsets via encoder
  if (c.rc) return c.rc;

  // q(qc_E): whole backward on stream 2 (tiny, off the critical path; dwq and dvq were made there)
  c.gemm(GEMM_NN, gp(w.dvq, He, Wenc0 + G + L, ldenc, w.dq_enc, L, PP * Q, L, He), 2);    // d qc_E via encoder
  if (qc_fused) {
    STEP(qc_chain_bwd(d.qce, d.d_qce, w.yq, w.rq, w.daq, w.headsq, w.dheadsq, nz_q, d.p_qc_E_loc, d.p_qc_E_scale, w.qcE,
                      w.sdq, w.cp.coefQ, w.dq_enc, w.dq_dec, Q, L, Hm, PP, P, s2));
  } else {
    STEP(qce_bwd(w.headsq, nz_q, d.p_qc_E_loc, d.p_qc_E_scale, w.qcE, w.sdq, w.dq_enc, w.dq_dec, w.cp.coefQ,
                 w.dheadsq, PP, P, Q, L, s2));
    const float* da0q = head_trunk_bwd(c, d.qce, d.d_qce, Q, Hm, 2 * L, w.dheadsq, w.yq, w.rq, w.Gq, w.daq, 2, 2, 2);
    {
      GemmParams p = gp(da0q, Hm, w.ones_q, L, d.d_qce.W[0], L, Hm, L, Q);
      p.extra_col = d.d_qce.b[0];
      c.gemm(GEMM_TN, p, 2);
    }
  }
  if (c.rc) return c.rc;

  // q(E|M), q(M)  (E and its scale are recomputed from the heads: stream 2's bookkeeping kernel is not waited for)
  PROBE(14);
  STEP(gaussE_bwd(w.plan, w.dzoff + L, 2 * L, w.doffenc, L, w.doffenc + (int64_t)PN * L, L, w.headsE, nz_E, w.mask,
                  d.p_E_loc, d.p_E_scale, nullptr, nullptr, w.cp.coefTc, w.dheadsE, w.dmaskME, PP, P, N, T, L, s));
  hipEvent_t ev_emb = nullptr;
  // (the step's tail: nothing else is left to hide a late weight gradient behind, so these fork in twos)
  const float* da0E = head_trunk_bwd(c, d.emb, d.d_emb, PT, Hm, 2 * L, w.dheadsE, w.yE, w.rE, w.GE, w.daE, 0, 1, 1, nullptr,
                                     &ev_emb, nullptr, 2);
  if (c.rc) return c.rc;
  PROBE(15);
  constexpr int y3 = 3;  // bit 0: dW0[:, :L], bit 1: dW0[:, L:] -- on the caller's stream after its last kernel (measured: -16 us)
  if (!ev_emb) ev_emb = c.record(0);
  c.wait(1, ev_emb);
  auto emb_dW0 = [&](int q) {  // dW0[:, :L] = da0^T mask, db0
    GemmParams p = gp(da0E, Hm, w.mask, L, d.d_emb.W[0], L + T, Hm, L, PT);
    p.extra_col = d.d_emb.b[0];
    c.gemm(GEMM_TN, p, q);
  };
  if (!(y3 & 1)) emb_dW0(1);
  if (!(y3 & 2)) STEP(sum_over_groups_to_cols(da0E, PP, T, Tc, w.plan.tpos, Hm, d.d_emb.W[0], L + T, L, s1));            // dW0[:, L:]
  c.gemm(GEMM_NN, gp(da0E, Hm, d.emb.W[0], L + T, w.dmask_enc, L, PT, L, Hm));            // d mask via encoder input
  if (c.rc) return c.rc;
  STEP(mask_bwd(d.q_mask_logits, d.p_mask_probs, d.temperature, w.mask, w.soft, w.dmaskME, w.dmask_enc, w.cp.coefT,
                d.d_q_mask_logits, PP, P, T, Tc, w.plan.tpos, L, s));
  // the caller's stream would only wait from here on: it takes the last leaves itself
  if (y3 & 1) emb_dW0(0);
  if (y3 & 2) STEP(sum_over_groups_to_cols(da0E, PP, T, Tc, w.plan.tpos, Hm, d.d_emb.W[0], L + T, L, s));
  PROBE(16);
  if (probe_at() == 17 || probe_at() == 18) {  // how long the caller's stream waits for stream 1 / stream 2 alone
    if (d.ev_gemm_begin) (void)hipEventRecord((hipEvent_t)d.ev_gemm_begin, s);
    c.after(0, probe_at() == 17 ? 1 : 2);
    if (d.ev_gemm_end) (void)hipEventRecord((hipEvent_t)d.ev_gemm_end, s);
    c.after(0, probe_at() == 17 ? 2 : 1);
  } else {
    c.after(0, 1);
    c.after(0, 2);
    if (d.ev_gemm_end && probe_at() >= 10) (void)hipEventRecord((hipEvent_t)d.ev_gemm_end, s);
  }
  if (c.rc) return c.rc;
#undef PROBE
#undef STEP
  return CRADLE_OK;
}

int prepare(const cradle_step_desc* dp, Work& w, Ctx& c, hipStream_t main) {
  CRADLE_CHECK_ARG(dp, "elbo_step: null descriptor");
  const cradle_step_desc& d = *dp;
  int rc = check_desc(d);
  if (rc) return rc;
  const int64_t need = carve(d, d.workspace, w);
  CRADLE_CHECK_ARG(d.workspace_bytes >= need, "elbo_step: workspace has %lld bytes, needs %lld",
                   (long long)d.workspace_bytes, (long long)need);
  StreamSet* ss = nullptr;
  rc = get_streams(&ss);
  if (rc) return rc;
  c.st[0] = main; c.st[1] = ss->side[0]; c.st[2] = ss->side[1];
  if (sched_flags() & 2) c.st[1] = c.st[2] = main;
  c.marker = ss->marker;
  c.precision = d.precision;
  for (int i = 0; i < 3; ++i) c.splitk[i] = w.splitk[i];
  c.ev = ss->ev;
  return CRADLE_OK;
}

}  // namespace

extern "C" {

void cradle_step_debug_stop_after(int checkpoint) { g_stop_after = checkpoint; }

int cradle_elbo_step(const cradle_step_desc* dp, hipStream_t s) {
  Work w;
  Ctx c;
  int rc = prepare(dp, w, c, s);
  if (rc) return rc;
  return run_step(*dp, w, c);
}

}  // extern "C"
