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
 * X       (N, ldx)          integer-valued counts;  library (N);  G % 4 == 0, G <= 16384
 * gene_consts (3, G)        written by cradle_nb_gene_consts from log_concentration
 * out     (P*N)             sum_g log p(X[n,g] | row r)
 */
int cradle_nb_gene_consts(const float* log_concentration, float* gene_consts, int G, cradle_stream_t stream);

int cradle_nb_logprob_fwd(const float* logits, int64_t ld_logits, const float* X, int64_t ldx,
                          const float* library, const float* gene_consts, float* out, int P, int N, int G,
                          cradle_stream_t stream);

/* Forward + backward in one pass over the logits (training).  `grow` (P*N) is d loss/d out.
 * dlogits (may alias logits: each row is overwritten after it was read) receives
 * d loss/d logits; dlog_concentration (G) receives (accumulate=0) or is incremented by
 * (accumulate=1) d loss/d log_concentration.  workspace: cradle_nb_workspace_floats floats. */
int cradle_nb_logprob_fwd_bwd(const float* logits, int64_t ld_logits, float* dlogits, int64_t ld_dlogits,
                              const float* X, int64_t ldx, const float* library, const float* gene_consts,
                              const float* grow, float* out, float* dlog_concentration, int accumulate,
                              float* workspace, int64_t workspace_floats, int P, int N, int G,
                              cradle_stream_t stream);
int64_t cradle_nb_workspace_floats(int P, int N, int G);
int cradle_nb_particle_split(int P, int N);

#ifdef __cplusplus
}
#endif
#endif /* CRADLE_HIP_H */
