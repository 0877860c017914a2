// Library-level entry points: version, last-error string.
#include <stdarg.h>
#include <string.h>

#include <stddef.h>

#include "common.h"

static thread_local char g_last_error[512] = "";

void cradle_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* cradle_last_error(void) { return g_last_error; }

int cradle_abi_version(void) { return 9; }

// Layout of the descriptor structs as THIS library was compiled: out[0] = sizeof, out[1..] = offsetof of every
// field in declaration order.  Returns the number of entries (or -1: unknown struct / buffer too small).
// A binding that mirrors the structs (cradle_vae_amd/hip.py: ctypes) compares its own layout with this at
// load time, so that a header edit cannot silently shift fields under an old binding.
int cradle_struct_layout(const char* name, int64_t* out, int max_entries) {
  if (!name || !out) return -1;
  int n = 0;
#define PUT(v)                        \
  do {                                \
    if (n >= max_entries) return -1;  \
    out[n++] = (int64_t)(v);          \
  } while (0)
#define OFF(S, f) PUT(offsetof(S, f))
  if (!strcmp(name, "cradle_mlp_ptrs")) {
    typedef cradle_mlp_ptrs S;
    PUT(sizeof(S));
    OFF(S, n_layers); OFF(S, W); OFF(S, b); OFF(S, W_head); OFF(S, b_head);
    return n;
  }
  if (!strcmp(name, "cradle_batch_desc")) {
    typedef cradle_batch_desc S;
    PUT(sizeof(S));
    OFF(S, N); OFF(S, G); OFF(S, T); OFF(S, Q); OFF(S, kmax); OFF(S, x_type); OFF(S, cfx_type); OFF(S, pos); OFF(S, X);
    OFF(S, cfX); OFF(S, qc); OFF(S, library_size); OFF(S, d_idx); OFF(S, d_val); OFF(S, d_dense); OFF(S, X_out);
    OFF(S, cfX_out); OFF(S, qc_out); OFF(S, lib_out); OFF(S, D_out); OFF(S, present_out); OFF(S, rows); OFF(S, flags);
    return n;
  }
  if (!strcmp(name, "cradle_gemm_desc")) {
    typedef cradle_gemm_desc S;
    PUT(sizeof(S));
    OFF(S, layout); OFF(S, M); OFF(S, N); OFF(S, K); OFF(S, A); OFF(S, lda); OFF(S, B); OFF(S, ldb); OFF(S, C);
    OFF(S, ldc); OFF(S, beta); OFF(S, act); OFF(S, bias); OFF(S, rowbias); OFF(S, ld_rowbias); OFF(S, rowbias_mode);
    OFF(S, rowbias_n); OFF(S, rowscale); OFF(S, rowscale_n); OFF(S, act_out); OFF(S, ld_act_out); OFF(S, residual);
    OFF(S, ld_residual); OFF(S, dact_src); OFF(S, ld_dact_src); OFF(S, C2); OFF(S, ldc2); OFF(S, extra_col);
    OFF(S, precision); OFF(S, k_valid);
    return n;
  }
  if (!strcmp(name, "cradle_step_desc")) {
    typedef cradle_step_desc S;
    PUT(sizeof(S));
    OFF(S, P); OFF(S, N); OFF(S, T); OFF(S, L); OFF(S, Q); OFF(S, G); OFF(S, He); OFF(S, Hm); OFF(S, Hd); OFF(S, kmax);
    OFF(S, q_mask_logits); OFF(S, d_q_mask_logits);
    OFF(S, emb); OFF(S, d_emb); OFF(S, qce); OFF(S, d_qce); OFF(S, enc); OFF(S, d_enc); OFF(S, dec); OFF(S, d_dec);
    OFF(S, log_concentration); OFF(S, d_log_concentration);
    OFF(S, p_E_loc); OFF(S, p_E_scale); OFF(S, p_mask_probs); OFF(S, p_qc_E_loc); OFF(S, p_qc_E_scale);
    OFF(S, log_mean); OFF(S, log_scale); OFF(S, temperature);
    OFF(S, X); OFF(S, cfX); OFF(S, D); OFF(S, qc); OFF(S, library_size); OFF(S, D_obs_counts); OFF(S, qc_obs_counts);
    OFF(S, u_mask); OFF(S, eps_E); OFF(S, eps_qc_E); OFF(S, eps_z);
    OFF(S, beta); OFF(S, alpha); OFF(S, compute_grads);
    OFF(S, out); OFF(S, flags); OFF(S, workspace); OFF(S, workspace_bytes);
    OFF(S, ev_k1_begin); OFF(S, ev_k1_end); OFF(S, ev_gemm_begin); OFF(S, ev_gemm_end); OFF(S, ev_bucket);
    OFF(S, rng_seed_plated); OFF(S, rng_seed_local); OFF(S, rng_step); OFF(S, precision); OFF(S, max_present);
    return n;
  }
#undef OFF
#undef PUT
  return -1;
}

// Number of visible HIP devices, or -1 if the runtime cannot be initialised.  Used by the
// Python loader to fail loudly instead of silently computing elsewhere.
int cradle_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

// HIP-event helpers for callers without a HIP binding of their own (bench.py times single
// kernels of the fused step with them; see cradle_step_desc.ev_k1_begin/end).
int cradle_event_create(void** ev) {
  hipEvent_t e;
  if (!ev || hipEventCreate(&e) != hipSuccess) {
    cradle_set_error("event_create failed");
    return CRADLE_ERR_HIP;
  }
  *ev = (void*)e;
  return CRADLE_OK;
}
int cradle_event_destroy(void* ev) { return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? CRADLE_OK : CRADLE_ERR_HIP; }
// elapsed milliseconds between two recorded events; both must have completed
int cradle_event_elapsed_ms(void* begin, void* end, float* ms) {
  hipError_t e = hipEventElapsedTime(ms, (hipEvent_t)begin, (hipEvent_t)end);
  if (e != hipSuccess) {
    cradle_set_error("event_elapsed_ms: %s", hipGetErrorString(e));
    return CRADLE_ERR_HIP;
  }
  return CRADLE_OK;
}

// makes everything enqueued on `stream` from now on wait for `ev` (no-op if it was never recorded)
int cradle_stream_wait_event(hipStream_t stream, void* ev) {
  if (!ev || hipStreamWaitEvent(stream, (hipEvent_t)ev, 0) != hipSuccess) {
    cradle_set_error("stream_wait_event failed");
    return CRADLE_ERR_HIP;
  }
  return CRADLE_OK;
}

}  // extern "C"
