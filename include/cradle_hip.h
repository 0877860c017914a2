/* C ABI of libcradle_hip.so -- MI355X (gfx950) kernels for the CRADLE-VAE ELBO step and
 * ATE predictor.
 *
 * The reference (dmis-lab/CRADLE-VAE) has no native/FFI boundary: its hot path is a chain
 * of PyTorch ops behind the Python classes in cradle_vae/models (resolved by name in
 * cradle_vae/models/utils/perturbation_lightning_module.py:222-242).  Each entry point
 * below therefore cites the reference *Python* lines whose arithmetic it replaces; the
 * Python mirror of the reference classes (package cradle_vae_amd) binds these symbols with
 * ctypes (cradle_vae_amd/hip.py), and INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 data unless stated; the caller owns and
 *    allocates every buffer (inputs, outputs, workspaces); nothing here allocates, frees
 *    or synchronises, so every call can be captured into a HIP graph;
 *  - matrices are row-major with an explicit leading dimension in elements;
 *  - `stream` is the hipStream_t the work is ordered on (0 = the null stream);
 *  - return value: 0 = ok, non-zero = error, text via cradle_last_error() (thread-local);
 *  - all paths in comments are relative to /root/reference.
 */
#ifndef CRADLE_HIP_H
#define CRADLE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* cradle_stream_t; /* == hipStream_t */

#define CRADLE_OK 0
#define CRADLE_ERR_ARG 1
#define CRADLE_ERR_HIP 2
#define CRADLE_ERR_UNSUPPORTED 3

/* ---- library ---------------------------------------------------------------------- */
const char* cradle_last_error(void);
int cradle_abi_version(void);
int cradle_device_count(void);
/* Layout of a descriptor struct ("cradle_step_desc", "cradle_gemm_desc", "cradle_mlp_ptrs") as the library
 * was compiled: out[0] = sizeof, out[1..] = offsetof of every field in declaration order; returns the number
 * of entries written, -1 for an unknown name or a short buffer.  Bindings that mirror the structs check
 * themselves against it at load time (cradle_vae_amd/hip.py). */
int cradle_struct_layout(const char* name, int64_t* out, int max_entries);
/* hipEvent helpers for hosts without their own HIP binding (bench.py) */
int cradle_event_create(void** ev);
int cradle_event_destroy(void* ev);
int cradle_event_elapsed_ms(void* begin, void* end, float* ms);
/* everything enqueued on `stream` from now on waits for `ev` (cradle_step_desc.ev_bucket consumers) */
int cradle_stream_wait_event(cradle_stream_t stream, void* ev);

/* ---- K1: Gamma-Poisson (negative binomial) log-likelihood over genes ----------------
 * Replaces, for one decoder variant (good or bad):
 *   cradle_vae/models/utils/mlp.py:153-166  softmax over genes, mu = library*softmax,
 *       rate = clamp(exp(log_concentration) / (mu + 1e-4), min=1e-6), GammaPoisson(c, rate)
 *   pyro.distributions.GammaPoisson.log_prob (pyro-ppl 1.9.1), called at
 *       cradle_vae/models/utils/loss_modules.py:134-135 and models/utils/predictor.py:158-159
 *   the `.sum(-1)` over genes at loss_modules.py:143 / predictor.py:165.
 * The qc/alpha mixing of loss_modules.py:137-142 is a per-(particle,cell) scalar applied by
 * the caller to `out`.
 *
 * logits  (P*N, ld_logits)  decoder pre-softmax outputs, row r = p*N + n
 * X       (N, ldx)          integer-valued counts;  library (N);  G <= 16384
 * gene_consts (3, Gp)       written by cradle_nb_gene_consts from log_concentration
 * Any gene count G (the reference has no restriction): the kernels move 16-byte vectors, so rows are run as
 * Gp = G rounded up to a multiple of 4 genes -- ld_logits, ld_dlogits and ldx must be multiples of 4 and >= Gp, the
 * rows of gene_consts and of the workspace have Gp floats -- and whatever the memory holds in columns G..Gp-1 never
 * enters the arithmetic (read as count 0, logit -inf; d logits there is written as 0, d log_concentration has G entries).
 * out     (P*N)             sum_g log p(X[n,g] | row r)
 */
int cradle_nb_gene_consts(const float* log_concentration, float* gene_consts, int G, cradle_stream_t stream);

int cradle_nb_logprob_fwd(const float* logits, int64_t ld_logits, const float* X, int64_t ldx,
                          const float* library, const float* gene_consts, float* out, int P, int N, int G,
                          cradle_stream_t stream);

/* Forward + backward in one pass over the logits (training).  `grow` (P*N) is d loss/d out.
 * dlogits (may alias logits: each row is overwritten after it was read) receives
 * d loss/d logits; dlog_concentration (G, 16-byte aligned) receives (accumulate=0) or is incremented by
 * (accumulate=1) d loss/d log_concentration.  workspace: cradle_nb_workspace_floats floats, 16-byte aligned. */
int cradle_nb_logprob_fwd_bwd(const float* logits, int64_t ld_logits, float* dlogits, int64_t ld_dlogits,
                              const float* X, int64_t ldx, const float* library, const float* gene_consts,
                              const float* grow, float* out, float* dlog_concentration, int accumulate,
                              float* workspace, int64_t workspace_floats, int P, int N, int G,
                              cradle_stream_t stream);
/* Same; ev_begin / ev_end (cradle_event_create handles, or NULL) are recorded on `stream`
 * immediately around the rows kernel, the launch bench.py's roofline describes (the small
 * column sum that finishes dlog_concentration runs after ev_end). */
int cradle_nb_logprob_fwd_bwd_timed(const float* logits, int64_t ld_logits, float* dlogits, int64_t ld_dlogits,
                                    const float* X, int64_t ldx, const float* library, const float* gene_consts,
                                    const float* grow, float* out, float* dlog_concentration, int accumulate,
                                    float* workspace, int64_t workspace_floats, int P, int N, int G,
                                    void* ev_begin, void* ev_end, cradle_stream_t stream);
int64_t cradle_nb_workspace_floats(int P, int N, int G);
int cradle_nb_particle_split(int P, int N);
/* cradle_nb_logprob_fwd_bwd without the workspace argument, for callers that cannot allocate one: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.
 * The per-cell partial rows live in a device array built into the library (cradle_nb_static_scratch_floats()
 * floats = 16 MiB, held by every GPU the library runs on until the process ends); a problem that needs more
 * (cradle_nb_workspace_floats(P, N, G) > that) is rejected with CRADLE_ERR_ARG.  Calls no hipMalloc/hipFree.
 * All calls on one GPU share the array: calls that overlap on one GPU (two streams, two threads) race.
 * Results are bit-identical to cradle_nb_logprob_fwd_bwd.  The array's device address is looked up once per device
 * (hipGetSymbolAddress, thread-safe) on the first call there -- that look-up is not legal under stream capture:
 * call cradle_nb_static_init() once per device, with that device current, before capturing. */
int cradle_nb_logprob_fwd_bwd_static(const float* logits, int64_t ld_logits, float* dlogits, int64_t ld_dlogits,
                                     const float* X, int64_t ldx, const float* library, const float* gene_consts,
                                     const float* grow, float* out, float* dlog_concentration, int accumulate,
                                     int P, int N, int G, cradle_stream_t stream);
int64_t cradle_nb_static_scratch_floats(void);
int cradle_nb_static_init(void);
/* tuning hook for tools/bench_nb.py: threads per workgroup (256/512/1024), particle split; -1 = auto */
void cradle_nb_tune(int threads, int particle_split);
void cradle_nb_sample_tune(int threads); /* same for K9 below: 256 / 512 / 1024 threads per workgroup, -1 = automatic */

/* ---- K6: fp32 matrix-core GEMM with fused MLP epilogues ---------------------------------
 * Replaces torch.nn.Linear (+ LeakyReLU + residual) of cradle_vae/models/utils/mlp.py:76-80,
 * 111-119, 129-131, 153-154 and the mm kernels autograd runs for them in backward.
 *   layout 0 (NT): C[M,N] = A[M,K] . B[N,K]^T     y  = x W^T
 *   layout 1 (NN): C[M,N] = A[M,K] . B[K,N]       dx = dy W
 *   layout 2 (TN): C[M,N] = A[K,M]^T . B[K,N]     dW = dy^T x
 * v = acc + bias[col] + rowscale[row % rowscale_n] * rowbias[rb_row, col]
 *       (rb_row = row / rowbias_n if rowbias_mode == 1, row % rowbias_n if 2)
 * if act: v = LeakyReLU_0.01(v);  act_out = v;  v += residual;  C = v + beta * C;
 * C2 = v * (dact_src > 0 ? 1 : 0.01)  (LeakyReLU backward of the previous layer);
 * extra_col (layout 2 only): row sums of A^T, i.e. the bias gradient, as a virtual ones column.
 * Arithmetic: v_mfma_f32_32x32x2_f32, exact fp32 (one rounding per product). */
typedef struct cradle_gemm_desc {
  int layout, M, N, K;
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  float beta;
  int act;
  const float* bias;
  const float* rowbias; int64_t ld_rowbias; int rowbias_mode, rowbias_n;
  const float* rowscale; int rowscale_n;
  float* act_out; int64_t ld_act_out;
  const float* residual; int64_t ld_residual;
  const float* dact_src; int64_t ld_dact_src;
  float* C2; int64_t ldc2;
  float* extra_col;
  int precision;   /* 0: fp32 MFMA (exact fp32, the reference's arithmetic: default).  1: "bf16x3" -- operands split
                      into bf16 hi + lo when a tile is staged, a_hi b_hi + a_hi b_lo + a_lo b_hi on
                      v_mfma_f32_32x32x16_bf16 with fp32 accumulation (~1e-6..1e-5 relative per dot product), used for
                      products of >= 1 GFLOP with 16-byte aligned operands, fp32 otherwise.  2: "bf16" -- the same
                      kernels with the one product of the rounded operands (plain bf16 inputs, fp32 accumulation, ~3
                      digits per product).  Optional modes, no counterpart in the reference (which computes in fp32). */
  int k_valid;     /* 0: the operands hold K rows/columns along the inner dimension.  > 0 (and < K): they hold only
                      k_valid; K is that count rounded up by the caller so that the 16-byte loader applies (K % 4 == 0),
                      and the loader's range check supplies zeros from the end of an operand's last k_valid-long row on.
                      For a K-major operand (A of NT/NN, B of NT) rows other than the last must be readable up to K and
                      hold finite values there (its leading dimension >= K); an MN-major operand (B of NN, both of TN)
                      simply ends after k_valid rows. */
} cradle_gemm_desc;

/* workspace (optional, may be NULL): lets small-output / long-K problems split along K. */
int cradle_gemm_f32(const cradle_gemm_desc* desc, float* workspace, int64_t workspace_floats, cradle_stream_t stream);
int64_t cradle_gemm_workspace_floats(int layout, int M, int N, int K, int extra_col);
/* tuning hook for tools/tune_gemm.py: force the tile (tm, tn in {1,2} x 64), K step (16/32),
 * prefetch depth, K wave-groups per workgroup (1/2) and K split; -1 = automatic.  Not used on
 * the product path. */
void cradle_gemm_tune(int tm, int tn, int bk, int depth, int wk, int splitk);
int cradle_gemm_f32_repeat(const cradle_gemm_desc* desc, float* workspace, int64_t workspace_floats, int repeat,
                           cradle_stream_t stream);

/* ---- K4: matmul(D, .) through the non-zeros of D ------------------------------------------
 * Replaces the dense dosage products of the module-level API: torch.matmul(D, E * mask) and matmul(qc, qc_E) at
 * cradle_vae/models/cradle_vae/guides/correlated_normal_guide.py:186-194 and models/cradle_vae/model.py:118-122,
 * the plated re-weighting matmul of models/utils/loss_modules.py:224-232, and the per-perturbation model calls of
 * analysis/average_treatment_effects.py:157-162,183-188.  D (N, T) has at most a few non-zeros per row
 * (data/norman/data_module.py:100-102).
 *   cradle_dosage_rows : dense D (N, T), contiguous -> ELL rows: ell_idx (N, kmax) treatment index ascending, -1 = none;
 *                        ell_val (N, kmax).  *flags |= 1 if a row holds more than kmax non-zeros (the caller zeroes it
 *                        and falls back to a dense product).
 *   cradle_gather_sum  : out[p, n, :] = sum_k ell_val[n, k] * table[p, ell_idx[n, k], :]     (forward)
 *   cradle_scatter_sum : dtable[p, t, :] = sum_{(n,k): ell_idx[n,k] == t} ell_val[n, k] * dout[p, n, :]  (its transpose,
 *                        for autograd; entries are added in (n, k) order: deterministic)
 * table / dtable: (P, T, L) with element strides (stride_p, stride_t), rows contiguous; out / dout: (P, N, L) with
 * strides (out_stride_p, out_stride_n).  stride_p = 0 broadcasts one table to all particles. */
int cradle_dosage_rows(const float* D, int N, int T, int kmax, int* ell_idx, float* ell_val, int* flags,
                       cradle_stream_t stream);
int cradle_gather_sum(const float* table, int64_t stride_p, int64_t stride_t, const int* ell_idx, const float* ell_val,
                      int P, int N, int kmax, int L, float* out, int64_t out_stride_p, int64_t out_stride_n,
                      cradle_stream_t stream);
int cradle_scatter_sum(const float* dout, int64_t out_stride_p, int64_t out_stride_n, const int* ell_idx,
                       const float* ell_val, int P, int N, int kmax, int T, int L, float* dtable, int64_t stride_p,
                       int64_t stride_t, cradle_stream_t stream);

/* ---- Draws of the module path's inference calls ---------------------------------------------
 * guide.forward under no_grad (the ATE estimator once per particle chunk, IWELBO, observation sampling) draws q(M) and
 * q(E|M), q(z_basal) through ~25 element-wise torch launches per call in the reference's form; these two are the same
 * arithmetic in one launch each (the expressions of the fused step's own kernels: same bits).
 *   cradle_mask_sample : relaxed-Bernoulli straight-through sample, hard value only
 *                        (cradle_vae/models/utils/gumbel_softmax_bernoulli.py:5-17 over pyro RelaxedBernoulliStraightThrough,
 *                        call site guides/correlated_normal_guide.py:132-141): logits (Ts, L) -- NaN/Inf read as 0
 *                        (sanitize_tensor, :212-215) --, uniforms u (P, Ts, L), temperature tau -> mask (P, Ts, L) in {0,1}
 *   cradle_gauss_sample: the Gaussian head's reparameterised draw (models/utils/mlp.py:131-141 + Normal.rsample at
 *                        correlated_normal_guide.py:157,166,202-204): mean_raw, log_var_raw (rows, L) with leading
 *                        dimensions ld_*, eps (rows, L) -> out = nan_to_num(mean) + sqrt(nan_to_num(|exp(lv) + 1e-4|)) * eps */
int cradle_mask_sample(const float* logits, const float* u, float tau, float* mask, int P, int Ts, int L,
                       cradle_stream_t stream);
int cradle_gauss_sample(const float* mean_raw, int64_t ld_mean, const float* log_var_raw, int64_t ld_log_var,
                        const float* eps, float* out, int64_t rows, int L, cradle_stream_t stream);

/* ---- The fused ELBO step ---------------------------------------------------------------
 * Forward + backward of PerturbationPlatedELBOLossModule.loss
 * (cradle_vae/models/utils/loss_modules.py:54-187: guide(X,qc) -> model -> guide(cfX,1-qc),
 * NB reconstruction with qc/alpha mixing :134-143, counterfactual KL :145-151,190-196, plated
 * log-ratio terms :153-177,200-232, loss :179-181) including everything it calls:
 * correlated_normal_guide.py:102-209, model.py:59-129, mlp.py:111-168, normalization.py:41-45,
 * gumbel_softmax_bernoulli.py:5-17 -- and of the autograd backward Lightning runs after it.
 * Kernels on the caller's stream and two library-owned side streams joined back before return (fork/join by
 * events); nothing allocates or synchronises with the host.
 * Restrictions (every shipped config satisfies them): sum_mode=False, mean_field_encoder=False,
 * log_standardize input normalisation, library_nb likelihood, qc entries in {0,1}, at most
 * `kmax` non-zero dosages per cell.  Violations of the last two raise bits in *flags. */
#define CRADLE_MAX_LAYERS 8
typedef struct cradle_mlp_ptrs {
  int n_layers;                 /* trunk layers, mlp.py:58 */
  float* W[CRADLE_MAX_LAYERS];  /* W[k]: (H, K_k) row-major = mlp.layers.{k}.0.weight */
  float* b[CRADLE_MAX_LAYERS];  /* (H) */
  float* W_head;                /* [mean_encoder.weight; log_var_encoder.weight] (2L, H), or
                                   normalized_mean_decoder.0.weight (G, H) for the decoder */
  float* b_head;
} cradle_mlp_ptrs;

typedef struct cradle_step_desc {
  int P, N, T, L, Q, G, He, Hm, Hd, kmax;
  float* q_mask_logits;         /* (T, L) guide.param_dict.q_mask_logits */
  float* d_q_mask_logits;
  cradle_mlp_ptrs emb, d_emb;   /* guide.embedding_encoder, W[0]: (Hm, L+T) */
  cradle_mlp_ptrs qce, d_qce;   /* guide.qc_encoder,        W[0]: (Hm, L)   */
  cradle_mlp_ptrs enc, d_enc;   /* guide.z_basal_encoder,   W[0]: (He, G+2L) */
  cradle_mlp_ptrs dec, d_dec;   /* model.decoder,           W[0]: (Hd, 3L), head (G, Hd) */
  float* log_concentration;     /* (G) */
  float* d_log_concentration;
  const float* p_E_loc; const float* p_E_scale; const float* p_mask_probs;   /* (T, L) model.py:33-39 */
  const float* p_qc_E_loc; const float* p_qc_E_scale;                         /* (Q, L) model.py:41-44 */
  const float* log_mean; const float* log_scale;                              /* (G) normalization.py:38-39 */
  float temperature;            /* guide.gs_temperature */
  const float* X; const float* cfX;   /* (N, G) */
  const float* D;                     /* (N, T) */
  const float* qc;                    /* (N, Q) */
  const float* library_size;          /* (N) */
  const float* D_obs_counts;          /* (T) */
  const float* qc_obs_counts;         /* (Q) */
  /* the random draws of the step: either all four given by the caller (parity tests inject the reference's),
   * or all four NULL: the step then draws them itself into its workspace, Philox4x32-10 keyed by rng_seed_plated
   * (mask, E, qc_E: must be the same on every data-parallel rank) and rng_seed_local (z_basal: per rank) with
   * rng_step in the counter (the caller advances it every step) */
  const float* u_mask;    /* (2P, T, L) uniforms: particles [0,P) main guide pass, [P,2P) counterfactual pass */
  const float* eps_E;     /* (2P, T, L) standard normals */
  const float* eps_qc_E;  /* (2P, Q, L) */
  const float* eps_z;     /* (P, N, L) */
  float beta;
  float alpha;            /* < 0: alpha = qc.sum()/N (loss_modules.py:137-139) */
  int compute_grads;      /* 0: forward only */
  float* out;             /* (8): loss, loss_term_{reconstruction,mask,E,qc_E,z_basal}, cf_KLdivergence, and [7] += 1
                             for every step whose input was rejected (the caller zeroes it; the loss of such a
                             step is NaN) */
  int* flags;             /* (1) written by every step: bit 0 = a cell with more than kmax non-zero dosages (or more
                             treatments present than compact slots), bit 1 = a qc entry outside {0,1}; non-zero =
                             loss and gradients of this step are not valid (hand the word to cradle_adam_step) */
  void* workspace;
  int64_t workspace_bytes;
  void* ev_k1_begin;      /* optional hipEvent_t recorded before / after the K1 launches; NULL = off */
  void* ev_k1_end;
  void* ev_gemm_begin;    /* optional hipEvent_t recorded before / after the largest GEMM of the step (decoder */
  void* ev_gemm_end;      /* output layer forward, 2*P*N*G*Hd flop); NULL = off */
  void* ev_bucket[3];     /* optional hipEvent_t fired as soon as a gradient bucket is final (replaces DDP's
                             bucketed backward hooks under Lightning's ddp strategy, train_norman.py:89-97):
                             [0] the decoder's output layer (d_dec.W_head, d_dec.b_head, d_log_concentration:
                             a third of all parameters, final first), [1] the other decoder gradients,
                             [2] basal-encoder gradients (d_enc.*); the rest is final with the step.
                             NULL = off */
  uint64_t rng_seed_plated, rng_seed_local, rng_step;   /* used when the noise pointers are NULL */
  int precision;          /* cradle_gemm_desc.precision for the step's GEMMs: 0 = fp32 (default), 1 = bf16x3, 2 = bf16 */
  int max_present;        /* 0, or an upper bound on the number of DISTINCT treatments with a non-zero dosage in this
                             batch, when the caller knows one below N*kmax (a loader that has looked at the batch):
                             q(M), q(E|M) and the embedding encoder are evaluated on min(N*kmax, max_present, T) slots
                             -- the reference evaluates all T (model.py:96-117, correlated_normal_guide.py:132-194);
                             treatments absent from the batch contribute closed-form terms only.  A batch with more
                             present treatments than slots raises bit 0 of *flags (nothing is mis-scored). */
} cradle_step_desc;

int64_t cradle_elbo_step_workspace_bytes(const cradle_step_desc* desc); /* only the sizes are read */
/* Inspection (parity tests): where an intermediate of the last step lives inside desc->workspace.
 * name: "mask", "E" (2P*Tc*L each, rows (particle, treatment slot): the guide's draws of
 * correlated_normal_guide.py:132-160), "qc_E" (2P*Q*L), "z_offset" (P*N rows [z_basal | D@(E*mask)], ld 2L:
 * the decoder input of model.py:121), "logits" (P*N*G), and the step's own draws "u_mask", "eps_E" (2P*T*L),
 * "eps_qc_E" (2P*Q*L), "eps_z" (P*N*L).  Returns the element count, or -1. */
int64_t cradle_elbo_step_intermediate(const cradle_step_desc* desc, const char* name, const float** ptr);
int cradle_elbo_step(const cradle_step_desc* desc, cradle_stream_t stream);
/* 1 when cradle_elbo_step runs q(qc_E) (n_qc rows through the qc encoder, correlated_normal_guide.py:162-166) as its
 * single-workgroup chain kernels (three launches) at these sizes -- n_qc Q <= 2, latent width L, the qc encoder's hidden
 * width Hm, P particles --, 0 when as a chain of GEMM launches (same results) */
int cradle_qc_chain_supported(int Q, int L, int Hm, int P);
/* measuring hook for tools/phase_sweep.py (never used on the product path): the step returns after its k-th
 * internal checkpoint with all streams joined (its outputs are then incomplete); 0 = run everything */
void cradle_step_debug_stop_after(int checkpoint);
/* ---- K10: gradient-norm clip + Adam over flat arenas -----------------------------------
 * Replaces Lightning's gradient_clip_val (train_norman.py:96; torch clip_grad_norm_) followed by
 * torch.optim.Adam(lr, weight_decay) (perturbation_lightning_module.py:147-150).
 * g' = grads * grad_scale; coef = max_norm > 0 ? min(1, max_norm / (|g'| + 1e-6)) : 1;
 * then Adam with L2 weight decay on g' * coef.  norm_ws: cradle_adam_workspace_floats() floats;
 * its last element receives |g'|.  skip_flags: NULL, or a device int (cradle_step_desc.flags of the step
 * that produced `grads`): when it is non-zero the update is skipped -- parameters and moments stay as they
 * are -- so gradients of a rejected input never reach the parameters, whenever the host gets to look. */
int cradle_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step, float max_norm,
                     float grad_scale, float* norm_ws, const int* skip_flags, cradle_stream_t stream);
int64_t cradle_adam_workspace_floats(void);
/* The same update with the optimiser work sharded over data-parallel ranks (what Lightning's ddp strategy leaves to
 * every rank in full, train_norman.py:89-97).  The arena is cut into n_total regions (gradient bucket x rank; offsets and
 * lengths in floats, host arrays, offsets multiples of 4).  |g|^2 is defined as the in-order sum of per-region sums, each
 * computed the same way by whoever computes it, so that reduce-scatter + sharded update + all-gather and all-reduce +
 * full update give bit-identical parameters.
 *   cradle_grad_region_sumsq : region_sums[i] = sum_{region i} grads^2 for the n_regions listed regions (ws:
 *                              cradle_adam_regions_workspace_floats() floats)
 *   cradle_adam_step_regions : clip + Adam on the listed regions only; all_sums (device, n_sums = n_total) = the region
 *                              sums of the whole arena (all-gathered over ranks); norm_out (device float[2], may be
 *                              NULL): [0] receives |g'|, [1] is incremented whenever the update is skipped -- skip_flags
 *                              non-zero, or the squared norm not finite
 *   cradle_grad_poison_if_flagged : writes NaN into *element (a float of the gradient arena that is summed over ranks
 *                              after this call) when *flags != 0.  With it a rank whose step rejected its input makes the
 *                              squared norm non-finite on EVERY rank, so all ranks skip the update together: under
 *                              Lightning DDP (train_norman.py:89-97) every rank applies the same update or none */
int cradle_grad_region_sumsq(const float* grads, const int64_t* region_off, const int64_t* region_len, int n_regions,
                             int n_total, float* region_sums, float* ws, cradle_stream_t stream);
int64_t cradle_adam_regions_workspace_floats(void);
int cradle_adam_step_regions(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                             const int64_t* region_off, const int64_t* region_len, int n_regions, const float* all_sums,
                             int n_sums, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                             float max_norm, float grad_scale, float* norm_out, const int* skip_flags,
                             cradle_stream_t stream);
int cradle_grad_poison_if_flagged(float* element, const int* flags, cradle_stream_t stream);

/* ---- K0: batch gather of the device-resident data source ------------------------------------
 * Replaces the DataLoader over TensorPerturbationDataset / SCRNASeqTensorPerturbationDataset
 * (cradle_vae/data/utils/perturbation_dataset.py:44-89: X[idx], D[idx], qc[idx], cfX[idx], library_size[idx] as fp32,
 * collated on the host, copied to the device every step).  The data set stays on the GPU in compact form -- counts as
 * uint8 / int16 / int32 (integer-valued, converted exactly), D as <= kmax (treatment, dosage) pairs per cell (index
 * -1 = unused) or dense -- and one launch gathers the rows pos[0..N) into the dense fp32 tensors of the batch contract.
 * present_out (may be NULL): receives (D_out != 0).any(0).sum(), the number of distinct treatments in the batch
 * (cradle_step_desc.max_present; needs n_treatments <= 491 520). */
enum { CRADLE_COUNTS_F32 = 0, CRADLE_COUNTS_U8 = 1, CRADLE_COUNTS_I16 = 2, CRADLE_COUNTS_I32 = 3 };
typedef struct cradle_batch_desc {
  int N, G, T, Q, kmax;          /* batch rows, genes, treatments, qc columns, pairs per cell (sparse D) */
  int x_type, cfx_type;          /* CRADLE_COUNTS_* of the stored X / cfX */
  const int64_t* pos;            /* (N) rows of the stored tensors to gather (device) */
  const void* X;                 /* (rows, G) */
  const void* cfX;               /* (rows, G) */
  const float* qc;               /* (rows, Q) */
  const float* library_size;     /* (rows) */
  const int* d_idx;              /* (rows, kmax) or NULL */
  const float* d_val;            /* (rows, kmax) or NULL */
  const float* d_dense;          /* (rows, T) or NULL: exactly one of the two forms */
  float* X_out;                  /* (N, G) */
  float* cfX_out;                /* (N, G) */
  float* qc_out;                 /* (N, Q) */
  float* lib_out;                /* (N) */
  float* D_out;                  /* (N, T) */
  int* present_out;              /* (1) or NULL */
  int64_t rows;                  /* rows the stored tensors hold: pos[n] outside [0, rows) is never dereferenced (row 0
                                    is gathered instead) and reported in *flags; 0 = unchecked */
  int* flags;                    /* (1) or NULL: set to 0 by the call, bit 0 = a position was out of range */
} cradle_batch_desc;
int cradle_gather_batch(const cradle_batch_desc* desc, cradle_stream_t stream);

/* ---- input preparation on the device (callers in front of the step) -------------------------
 * cradle_log_stats: the train-set statistics of cradle_vae/data/utils/batch_statistics.py:4-23 (batch_log_mean /
 * batch_log_std, two host-stepped passes there) as one streaming pass: with y = log(x + 1) - shift[g] in fp32,
 *   sums[g] (+)= sum_rows y,   sums[G + g] (+)= sum_rows y^2     (fp64; accumulate != 0 adds to what sums holds, so a
 * matrix can be fed in row slabs).  X: (rows, G) counts of type x_type (CRADLE_COUNTS_*), row stride ld; shift: (G) or
 * NULL (zeros).  shift = log_mean gives batch_log_std's centred sum directly; shift = log(x_0 + 1) of any one row gives
 * mean = shift + S1/n and var = (S2 - S1^2/n)/(n - 1) without cancellation.  workspace:
 * cradle_log_stats_workspace_doubles(rows, G) doubles (partial sums, reduced in a fixed order: deterministic).
 * cradle_obs_counts_dense / _pairs: counts[c] = #(rows with M[r, c] != 0) of a dense (rows, C) matrix, or of a
 * dosage matrix stored as (rows, kmax) (treatment index, dosage) pairs, index -1 = unused -- the D_obs_per_dim /
 * qc_obs_per_dim of cradle_vae/data/utils/perturbation_dataset.py:64-67.
 * cradle_peer_median: out[n, g] = lower median (element of rank (k - 1) / 2, what torch.median returns) over the k
 * leading non-negative entries of peers[n, 0..kmax) of src[peer, g]; k = 0 gives a zero row.  The counterfactual input
 * cfX of cradle_vae/data/norman/data_module.py:175-191 (_get_cfX: X[idx].median(0) over <= 50 sampled QC-failing cells
 * of the cell's treatment).  src, out: x_type storage; flags (device int): bit 0 is set when a peer index is >=
 * src_rows (such a list is cut there, never dereferenced). */
int64_t cradle_log_stats_workspace_doubles(int64_t rows, int G);
int cradle_log_stats(const void* X, int x_type, int64_t rows, int G, int64_t ld, const float* shift, double* sums,
                     int accumulate, double* workspace, int64_t workspace_doubles, cradle_stream_t stream);
int cradle_obs_counts_dense(const float* M, int64_t rows, int C, int64_t* counts, cradle_stream_t stream);
int cradle_obs_counts_pairs(const int* idx, const float* val, int64_t rows, int kmax, int T, int64_t* counts,
                            cradle_stream_t stream);
int cradle_peer_median(const void* src, int x_type, int64_t src_rows, int G, const int* peers, int64_t n_cells, int kmax,
                       void* out, int* flags, cradle_stream_t stream);

/* ---- K9: Gamma-Poisson observation sampler fused with the ATE accumulation ---------------
 * Replaces, per (particle p, condition t) row r = p*C + t of decoder logits:
 *   mlp.py:153-166 (softmax -> mu -> rate) + pyro GammaPoisson.sample at model.py:131
 *   (rate' ~ Gamma(c, rate), x ~ Poisson(rate')) + average_treatment_effects.py:171-176,190-195
 *   (method "perturbseq": log2(1e4*x/sum_g x + 1); X_sums[t] += sum over particles).
 * sums (C, Gp) is accumulated into (+=).  samples: NULL or (P*C, Gp) raw draws; Gp = G rounded up to a multiple of 4
 * (row length of sums, samples, gene_consts; ld_logits >= Gp; columns G..Gp-1 receive zeros)
 * (predictor.py:228-251 sample_observations).  library: 1 or C entries.  method: 0 raw counts,
 * 1 perturbseq.  sampler: 0 Gamma-Poisson draw (Philox4x32-10 keyed by seed, counter = gene,
 * row_offset + r, draw index), 1 the distribution mean c/rate (deterministic, for parity tests). */
int cradle_nb_sample_accumulate(const float* logits, int64_t ld_logits, const float* library, int library_n,
                                const float* gene_consts, float* sums, float* samples, int P, int C, int G,
                                int method, int sampler, uint64_t seed, uint32_t row_offset, cradle_stream_t stream);
/* Same, with scratch that lets the particles of a condition be split over several workgroups (a launch
 * holds only a few hundred conditions; the reference loops over perturbations one at a time,
 * average_treatment_effects.py:150-196).  Partial sums are added in a fixed order: deterministic.
 * The draw is exact for GammaPoisson(c, rate) either way: NegBinomial inversion where rate >= 1/4,
 * the Gamma -> Poisson composition otherwise.  workspace: cradle_nb_sample_workspace_floats(P, C, G)
 * floats, 16-byte aligned, or NULL. */
int cradle_nb_sample_accumulate_ws(const float* logits, int64_t ld_logits, const float* library, int library_n,
                                   const float* gene_consts, float* sums, float* samples, int P, int C, int G,
                                   int method, int sampler, uint64_t seed, uint32_t row_offset, float* workspace,
                                   int64_t workspace_floats, cradle_stream_t stream);
int64_t cradle_nb_sample_workspace_floats(int P, int C, int G);

#ifdef __cplusplus
}
#endif
#endif /* CRADLE_HIP_H */
