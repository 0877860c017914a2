// Internal C++ interface between the kernel files and the fused step (not part of the C ABI).
#pragma once
#include "common.h"

namespace cradle {

// ---- gemm.hip
struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  int64_t lda, ldb, ldc;
  int M, N, K;
  float beta;  // C = v + beta * C_old
  const float* bias;      // [N]
  const float* rowbias;   // rowbias[rb_row * ld_rb + col], rb_row = row / rb_n (mode 1) or row % rb_n (mode 2)
  int64_t ld_rb;
  int rb_mode, rb_n;
  const float* rowscale;  // multiplies the rowbias term, indexed by row % rs_n
  int rs_n;
  int act;                // 1: LeakyReLU(0.01)
  float* act_out;         // store of the activation before the residual is added
  int64_t ld_act;
  const float* res;       // residual added after the activation
  int64_t ld_res;
  const float* dact_src;  // second output C2 = v * (dact_src > 0 ? 1 : 0.01)
  int64_t ld_ds;
  float* C2;
  int64_t ldc2;
  float* extra_col;       // TN only: B gets a virtual all-ones column N; its result goes here [M]
  uint32_t a_bytes, b_bytes;  // filled by gemm_f32: operand extents for the buffer descriptors
  int splitk;             // filled by gemm_f32
  float* partial;
  int64_t slab;           // floats between the partial results of consecutive K splits (multiple of 4)
  int k_tiles_per_split;
  int splitk_hint;        // > 0: use this K split instead of the heuristic's (caller knows what runs beside it)
  int k_valid;            // > 0: rows/columns of the K axis that exist in memory (K itself may be padded up to a multiple
                          // of 4 by the caller: the operand extents then end at k_valid and the range check reads zeros)
  int precision;          // 0: fp32 MFMA (default); 1: split-bf16 (3 products) for products large enough to gain
  // > 1: that many independent products of the same shape in ONE launch (grid y): problem b reads A + b * bs_a,
  // B + b * bs_b and writes C + b * bs_c (and extra_col + b * bs_x); strides in floats, any sign.  Plain products only
  // (fp32; of the epilogue features only extra_col): the weight gradients of an MLP's hidden layers.
  int batch;
  int64_t bs_a, bs_b, bs_c, bs_x;
};
enum { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2 };
int gemm_f32(int layout, GemmParams p, float* workspace, int64_t workspace_floats, hipStream_t stream);
int64_t gemm_workspace_floats(int layout, int M, int N, int K, int extra_col);
bool launch_gemm_bf16x3(int layout, const GemmParams& p, int tm, int tn, hipStream_t s);  // gemm_bf16x3.hip

// ---- nb_loss.hip (C ABI entry points, see include/cradle_hip.h)
}  // namespace cradle
extern "C" {
int cradle_nb_gene_consts(const float* log_c, float* gene_consts, int G, hipStream_t stream);
int cradle_nb_particle_split(int P, int N);
int64_t cradle_nb_workspace_floats(int P, int N, int G);
int cradle_nb_logprob_fwd(const float* logits, int64_t ld_logits, const float* X, int64_t ldx, const float* lib,
                          const float* gene_consts, float* out, int P, int N, int G, hipStream_t stream);
int cradle_nb_logprob_fwd_bwd(const float* logits, int64_t ld_logits, float* dlogits, int64_t ld_dlogits,
                              const float* X, int64_t ldx, const float* lib, const float* gene_consts,
                              const float* grow, float* out, float* dlog_c, int accumulate_dlog_c, float* workspace,
                              int64_t workspace_floats, int P, int N, int G, hipStream_t stream);
int cradle_nb_logprob_fwd_bwd_timed(const float* logits, int64_t ld_logits, float* dlogits, int64_t ld_dlogits,
                                    const float* X, int64_t ldx, const float* lib, const float* gene_consts,
                                    const float* grow, float* out, float* dlog_c, int accumulate_dlog_c,
                                    float* workspace, int64_t workspace_floats, int P, int N, int G, void* ev_begin,
                                    void* ev_end, hipStream_t stream);
}
namespace cradle {
int nb_dlogc_reduce(const float* workspace, int rows, int G, float* dlog_c, int accumulate, hipStream_t stream, int valid = -1);
int nb_gene_consts_padded(const float* log_c, float* gene_consts, int G, int Gp, hipStream_t stream);
// K1 of the training step in two launches: the particle-independent lgamma / digamma terms (side stream, forward
// pass), then the rows kernel that reads them (nb_loss.hip)
int nb_count_terms(const float* X, int64_t ldx, const float* gene_consts, float* workspace, int64_t workspace_floats,
                   float* lg, int P, int N, int G, hipStream_t stream);
int nb_logprob_fwd_bwd_pre(const float* logits, int64_t ld_logits, float* dlogits, int64_t ld_dlogits, const float* X,
                           int64_t ldx, const float* lib, const float* gene_consts, const float* grow, float* out,
                           float* workspace, int64_t workspace_floats, const float* lg, int P, int N, int G, void* ev_begin,
                           void* ev_end, hipStream_t stream);

// ---- elementwise.hip : plan / per-cell preparation
struct DosagePlan {
  int* ell_idx;    // (N, kmax) treatment index or -1
  float* ell_val;  // (N, kmax)
  int* tcount;     // (T) cells with a non-zero dosage of t in this batch
  int* tstart;     // (T+1) exclusive scan of tcount
  int* csc_cell;   // (N*kmax) cells of each treatment, ascending
  float* csc_val;  // (N*kmax)
  int* flags;      // bit 0: a cell has more than kmax non-zero dosages; bit 1: qc not in {0,1}
  int kmax;
  // Compact treatment axis: only treatments with a cell in the batch enter the loss or any gradient
  // (their plate weight is zero otherwise, loss_modules.py:224-232), so q(M), q(E|M) are evaluated for
  // the Tc = min(T, N*kmax) slots of tmap only.  Tc == T: identity, tmap/tpos are NULL.
  int Tc;
  int* tmap;       // (Tc) slot -> treatment, ascending, -1 = unused slot
  int* tpos;       // (T) treatment -> slot or -1
};
// first part: ELL rows, counts, scan, compact map (what the forward gathers need); second part: CSC
int build_dosage_plan(const float* D, int N, int T, const DosagePlan& plan, hipStream_t stream);
int build_dosage_plan_csc(int N, int T, const DosagePlan& plan, hipStream_t stream);

struct CellPrep {
  float* alpha;   // (1)
  float* live;    // (N) 1 = the bad (QC-fail) decoder variant is the live one
  float* wrec;    // (N) reconstruction weight of the live variant
  float* grow;    // (P*N) d loss / d recon row = -wrec[n] / (P*N)
  float* cw;      // (T) sum_n (D[n,t] != 0) / D_obs[t]
  float* cwq;     // (Q)
  float* coefT;   // (T) -cw / (P*N)
  float* coefQ;   // (Q)
  float* cwc;     // (Tc) cw by slot (0 for unused slots)
  float* coefTc;  // (Tc) coefT by slot
};
int cell_prep(const float* qc, const float* D_obs, const float* qc_obs, const int* tcount, float alpha_fixed, int P,
              int N, int T, int Q, const CellPrep& out, int* flags, int Tc, const int* tmap, hipStream_t stream);

// u_mask, eps_E (n_mask each), eps_qc_E (n_qc), eps_z (n_z): Philox draws keyed by (seed, step)
int draw_noise_slots(float* u_mask, float* eps_E, const int* tmap, int PP, int T, int Tc, int L, uint64_t seed_plated,
                     uint64_t step, hipStream_t stream);
int draw_noise(float* u_mask, float* eps_E, float* eps_qc_E, float* eps_z, int64_t n_mask, int64_t n_qc, int64_t n_z,
               uint64_t seed_plated, uint64_t seed_local, uint64_t step, int which, hipStream_t stream);

int log_standardize(const float* X, const float* cfX, const float* log_mean, const float* log_scale, float* xn,
                    float* hascf, int N, int G, hipStream_t stream, int ldo = 0, float* xpad = nullptr);

// ---- elementwise.hip : sampling + log-ratio terms
// (rows of mask / E / heads are (particle, slot) pairs: row = pp*Tc + j; noise and per-treatment tables
//  are indexed by the treatment tmap[j])
int mask_fwd(const float* logits, const float* u, const float* p_mask_probs, float tau, float* mask, float* soft,
             float* S, int PP, int T, int Tc, const int* tmap, int L, hipStream_t stream);
int mask_bwd(const float* logits, const float* p_mask_probs, float tau, const float* mask, const float* soft,
             const float* dmask_a, const float* dmask_b, const float* coefT, float* dlogits, int PP, int P, int T,
             int Tc, const int* tpos, int L, hipStream_t stream);
int gaussE_fwd(const float* heads, const float* eps, const float* ploc, const float* pscale, float* E, float* sd, float* S,
               int PP, int T, int Tc, const int* tmap, int L, hipStream_t stream);
// backward of q(E|M) fused with the scatter of its incoming gradient over the cells of each treatment:
// d(mask*E)[pp,t,:] = sum_n D[n,t] * (a1 + a2)[pp,n,:] (pp < P) or b1[pp-P,n,:] (pp >= P)
int gaussE_bwd(const DosagePlan& plan, const float* a1, int64_t ld1, const float* a2, int64_t ld2, const float* b1,
               int64_t ldb1, const float* heads, const float* eps, const float* mask, const float* ploc, const float* pscale,
               const float* E, const float* sd, const float* coefTc, float* dheads, float* dmask, int PP, int P, int N, int T,
               int L, hipStream_t stream);
int qce_fwd(const float* heads, const float* eps, const float* ploc, const float* pscale, float* qcE, float* sd,
            float* S, float* qsum, int PP, int P, int Q, int L, hipStream_t stream);
int qce_bwd(const float* heads, const float* eps, const float* ploc, const float* pscale, const float* qcE,
            const float* sd, const float* dq_enc, const float* dq_dec, const float* coefQ, float* dheads, int PP,
            int P, int Q, int L, hipStream_t stream);
int zhead_fwd(const float* heads, const float* eps, const float* hascf, float beta, float* z, int64_t ldz, float* sd,
              float* ltz, float* klrow, float* dheads, int P, int N, int L, hipStream_t stream);
int zhead_bwd(const float* heads, const float* eps, const float* z, int64_t ldz, const float* sd, const float* dz,
              int64_t lddz, float* dheads, int P, int N, int L, hipStream_t stream);

// ---- elementwise.hip : gathers
int gather_offsets(const float* heads, const float* eps, const float* mask, const DosagePlan& plan, float* offA, int64_t ldA,
                   float* offB, int64_t ldB, int PP, int P, int N, int T, int L, hipStream_t stream);
int enc_combine_fwd(const float* xW, const float* offW, const float* vq, const float* bias, const float* qc, float* h,
                    int P, int N, int Q, int H, hipStream_t stream);
// dvq (the qc-embedding part) is produced on `vq_stream`: its consumers are off the critical path
int enc_combine_bwd(const float* dpre, const float* qc, float* doffW, float* dxW, float* dvq, int P, int N, int Q, int H,
                    hipStream_t stream, hipStream_t vq_stream, int which = 3);

// ---- elementwise.hip : small helpers
int transpose_cols(const float* W, int64_t ldw, int rows, int col0, int cols, float* out, hipStream_t stream);
int gather_cols(const float* W, int64_t ldw, int rows, int col0, int Tc, const int* tmap, float* out, hipStream_t stream);
int sum_over_groups_to_cols(const float* da, int groups, int T, int Tc, const int* tpos, int H, float* dW, int64_t lddw,
                            int col0, hipStream_t stream);
int weighted_rowgroup_sum(const float* da, const float* w, int P, int N, int H, float* out, hipStream_t stream);
int finalize_loss(const float* recon, const float* wrec, const float* ltz, const float* klrow, const float* Smask,
                  const float* SE, const float* Sq, const float* cw, const float* cwq, float beta, int P, int N, int T,
                  int Q, const int* flags, float* out, hipStream_t stream);

// ---- qc_chain.hip : q(qc_E) (Q <= 2 rows through an MLP) as single-workgroup launches
bool qc_chain_supported(int Q, int L, int Hm, int PP);
// every layer, the heads, the draws qc_E + their log-ratio sums S, qsum (what qce_fwd + the GEMM chain before it make)
int qc_chain_fwd(const cradle_mlp_ptrs& m, float* const* y, float* const* r, float* heads, const float* eps, const float* ploc,
                 const float* pscale, float* qcE, float* sd, float* S, float* qsum, int Q, int L, int Hm, int PP, int P,
                 hipStream_t stream);
// the Gaussian head's backward from dq_enc (PP*Q, L) / dq_dec (P, L), the data-gradient chain (da[k] left in HBM), every
// bias gradient; then every weight gradient of the encoder in one more launch
int qc_chain_bwd(const cradle_mlp_ptrs& m, const cradle_mlp_ptrs& g, float* const* y, float* const* r, float* const* da,
                 const float* heads, float* dheads, const float* eps, const float* ploc, const float* pscale, const float* qcE,
                 const float* sd, const float* coefQ, const float* dq_enc, const float* dq_dec, int Q, int L, int Hm, int PP,
                 int P, hipStream_t stream);

// ---- adam.hip
int adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
              float eps, float weight_decay, int step, float max_norm, float grad_scale, float* norm_ws,
              const int* skip_flags, hipStream_t stream);

}  // namespace cradle
