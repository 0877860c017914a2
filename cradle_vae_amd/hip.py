"""ctypes binding of ``csrc/libcradle_hip.so`` (the C ABI declared in ``include/cradle_hip.h``).

There is no CPU or PyTorch fallback: if the library is missing, or a tensor handed to a
kernel wrapper is not a contiguous fp32 CUDA(HIP) tensor, this module raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

# CRADLE_HIP_LIB: another build of the same library (A/B timing runs of two builds on one GPU box, tools/ only)
_LIB_PATH = os.environ.get("CRADLE_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc",
                                                             "libcradle_hip.so")
_lib: Optional[ctypes.CDLL] = None

c_float_p = ctypes.c_void_p
c_stream = ctypes.c_void_p
i32, i64 = ctypes.c_int, ctypes.c_int64

# name -> (restype, argtypes); must list every symbol include/cradle_hip.h declares
PROTOTYPES = {
    "cradle_last_error": (ctypes.c_char_p, []),
    "cradle_abi_version": (i32, []),
    "cradle_gather_batch": (i32, [ctypes.c_void_p, c_stream]),
    "cradle_device_count": (i32, []),
    "cradle_struct_layout": (i32, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), i32]),
    "cradle_event_create": (i32, [ctypes.POINTER(ctypes.c_void_p)]),
    "cradle_event_destroy": (i32, [ctypes.c_void_p]),
    "cradle_event_elapsed_ms": (i32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "cradle_stream_wait_event": (i32, [c_stream, ctypes.c_void_p]),
    "cradle_nb_gene_consts": (i32, [c_float_p, c_float_p, i32, c_stream]),
    "cradle_nb_logprob_fwd": (i32, [c_float_p, i64, c_float_p, i64, c_float_p, c_float_p, c_float_p, i32, i32, i32, c_stream]),
    "cradle_nb_logprob_fwd_bwd": (
        i32,
        [c_float_p, i64, c_float_p, i64, c_float_p, i64, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, i32,
         c_float_p, i64, i32, i32, i32, c_stream],
    ),
    "cradle_nb_logprob_fwd_bwd_timed": (
        i32,
        [c_float_p, i64, c_float_p, i64, c_float_p, i64, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, i32,
         c_float_p, i64, i32, i32, i32, ctypes.c_void_p, ctypes.c_void_p, c_stream],
    ),
    "cradle_nb_logprob_fwd_bwd_static": (
        i32,
        [c_float_p, i64, c_float_p, i64, c_float_p, i64, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, i32,
         i32, i32, i32, c_stream],
    ),
    "cradle_nb_static_scratch_floats": (i64, []),
    "cradle_nb_static_init": (i32, []),
    "cradle_nb_workspace_floats": (i64, [i32, i32, i32]),
    "cradle_nb_particle_split": (i32, [i32, i32]),
    "cradle_nb_tune": (None, [i32, i32]),
    "cradle_nb_sample_tune": (None, [i32]),
    "cradle_gemm_f32": (i32, [ctypes.c_void_p, c_float_p, i64, c_stream]),
    "cradle_gemm_workspace_floats": (i64, [i32, i32, i32, i32, i32]),
    "cradle_gemm_tune": (None, [i32, i32, i32, i32, i32, i32]),
    "cradle_gemm_f32_repeat": (i32, [ctypes.c_void_p, c_float_p, i64, i32, c_stream]),
    "cradle_dosage_rows": (i32, [c_float_p, i32, i32, i32, ctypes.c_void_p, c_float_p, ctypes.c_void_p, c_stream]),
    "cradle_gather_sum": (i32, [c_float_p, i64, i64, ctypes.c_void_p, c_float_p, i32, i32, i32, i32, c_float_p, i64, i64,
                                c_stream]),
    "cradle_scatter_sum": (i32, [c_float_p, i64, i64, ctypes.c_void_p, c_float_p, i32, i32, i32, i32, i32, c_float_p, i64,
                                 i64, c_stream]),
    "cradle_mask_sample": (i32, [c_float_p, c_float_p, ctypes.c_float, c_float_p, i32, i32, i32, c_stream]),
    "cradle_gauss_sample": (i32, [c_float_p, i64, c_float_p, i64, c_float_p, c_float_p, i64, i32, c_stream]),
    "cradle_elbo_step_workspace_bytes": (i64, [ctypes.c_void_p]),
    "cradle_elbo_step": (i32, [ctypes.c_void_p, c_stream]),
    "cradle_step_debug_stop_after": (None, [i32]),
    "cradle_qc_chain_supported": (i32, [i32, i32, i32, i32]),
    "cradle_elbo_step_intermediate": (i64, [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]),
    "cradle_adam_step": (
        i32,
        [c_float_p, c_float_p, c_float_p, c_float_p, i64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
         ctypes.c_float, ctypes.c_float, i32, ctypes.c_float, ctypes.c_float, c_float_p, ctypes.c_void_p, c_stream],
    ),
    "cradle_adam_workspace_floats": (i64, []),
    "cradle_grad_region_sumsq": (i32, [c_float_p, ctypes.POINTER(i64), ctypes.POINTER(i64), i32, i32, c_float_p, c_float_p,
                                       c_stream]),
    "cradle_adam_regions_workspace_floats": (i64, []),
    "cradle_adam_step_regions": (
        i32,
        [c_float_p, c_float_p, c_float_p, c_float_p, ctypes.POINTER(i64), ctypes.POINTER(i64), i32, c_float_p, i32,
         ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, i32, ctypes.c_float,
         ctypes.c_float, c_float_p, ctypes.c_void_p, c_stream],
    ),
    "cradle_grad_poison_if_flagged": (i32, [c_float_p, ctypes.c_void_p, c_stream]),
    "cradle_log_stats_workspace_doubles": (i64, [i64, i32]),
    "cradle_log_stats": (i32, [ctypes.c_void_p, i32, i64, i32, i64, c_float_p, ctypes.c_void_p, i32, ctypes.c_void_p, i64,
                               c_stream]),
    "cradle_obs_counts_dense": (i32, [c_float_p, i64, i32, ctypes.c_void_p, c_stream]),
    "cradle_obs_counts_pairs": (i32, [ctypes.c_void_p, c_float_p, i64, i32, i32, ctypes.c_void_p, c_stream]),
    "cradle_peer_median": (i32, [ctypes.c_void_p, i32, i64, i32, ctypes.c_void_p, i64, i32, ctypes.c_void_p,
                                 ctypes.c_void_p, c_stream]),
    "cradle_nb_sample_accumulate_ws": (
        i32,
        [c_float_p, i64, c_float_p, i32, c_float_p, c_float_p, c_float_p, i32, i32, i32, i32, i32, ctypes.c_uint64,
         ctypes.c_uint32, c_float_p, i64, c_stream],
    ),
    "cradle_nb_sample_workspace_floats": (i64, [i32, i32, i32]),
    "cradle_nb_sample_accumulate": (
        i32,
        [c_float_p, i64, c_float_p, i32, c_float_p, c_float_p, c_float_p, i32, i32, i32, i32, i32, ctypes.c_uint64,
         ctypes.c_uint32, c_stream],
    ),
}

MAX_LAYERS = 8
ABI_VERSION = 9


class MlpPtrs(ctypes.Structure):
    """struct cradle_mlp_ptrs."""

    _fields_ = [
        ("n_layers", i32),
        ("W", ctypes.c_void_p * MAX_LAYERS),
        ("b", ctypes.c_void_p * MAX_LAYERS),
        ("W_head", ctypes.c_void_p),
        ("b_head", ctypes.c_void_p),
    ]


class StepDesc(ctypes.Structure):
    """struct cradle_step_desc (include/cradle_hip.h); field order is ABI."""

    _fields_ = (
        [(k, i32) for k in ("P", "N", "T", "L", "Q", "G", "He", "Hm", "Hd", "kmax")]
        + [("q_mask_logits", ctypes.c_void_p), ("d_q_mask_logits", ctypes.c_void_p)]
        + [(k, MlpPtrs) for k in ("emb", "d_emb", "qce", "d_qce", "enc", "d_enc", "dec", "d_dec")]
        + [("log_concentration", ctypes.c_void_p), ("d_log_concentration", ctypes.c_void_p)]
        + [(k, ctypes.c_void_p) for k in ("p_E_loc", "p_E_scale", "p_mask_probs", "p_qc_E_loc", "p_qc_E_scale",
                                          "log_mean", "log_scale")]
        + [("temperature", ctypes.c_float)]
        + [(k, ctypes.c_void_p) for k in ("X", "cfX", "D", "qc", "library_size", "D_obs_counts", "qc_obs_counts",
                                          "u_mask", "eps_E", "eps_qc_E", "eps_z")]
        + [("beta", ctypes.c_float), ("alpha", ctypes.c_float), ("compute_grads", i32)]
        + [("out", ctypes.c_void_p), ("flags", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
           ("workspace_bytes", i64), ("ev_k1_begin", ctypes.c_void_p), ("ev_k1_end", ctypes.c_void_p),
           ("ev_gemm_begin", ctypes.c_void_p), ("ev_gemm_end", ctypes.c_void_p), ("ev_bucket", ctypes.c_void_p * 3),
           ("rng_seed_plated", ctypes.c_uint64), ("rng_seed_local", ctypes.c_uint64), ("rng_step", ctypes.c_uint64),
           ("precision", i32), ("max_present", i32)]
    )


class BatchDesc(ctypes.Structure):
    """struct cradle_batch_desc (include/cradle_hip.h); field order is ABI."""

    _fields_ = (
        [(k, i32) for k in ("N", "G", "T", "Q", "kmax", "x_type", "cfx_type")]
        + [(k, ctypes.c_void_p) for k in ("pos", "X", "cfX", "qc", "library_size", "d_idx", "d_val", "d_dense", "X_out",
                                          "cfX_out", "qc_out", "lib_out", "D_out", "present_out")]
        + [("rows", i64), ("flags", ctypes.c_void_p)]
    )


class GemmDesc(ctypes.Structure):
    """struct cradle_gemm_desc (include/cradle_hip.h); field order is ABI."""

    _fields_ = [
        ("layout", i32), ("M", i32), ("N", i32), ("K", i32),
        ("A", ctypes.c_void_p), ("lda", i64),
        ("B", ctypes.c_void_p), ("ldb", i64),
        ("C", ctypes.c_void_p), ("ldc", i64),
        ("beta", ctypes.c_float),
        ("act", i32),
        ("bias", ctypes.c_void_p),
        ("rowbias", ctypes.c_void_p), ("ld_rowbias", i64), ("rowbias_mode", i32), ("rowbias_n", i32),
        ("rowscale", ctypes.c_void_p), ("rowscale_n", i32),
        ("act_out", ctypes.c_void_p), ("ld_act_out", i64),
        ("residual", ctypes.c_void_p), ("ld_residual", i64),
        ("dact_src", ctypes.c_void_p), ("ld_dact_src", i64),
        ("C2", ctypes.c_void_p), ("ldc2", i64),
        ("extra_col", ctypes.c_void_p),
        ("precision", i32),
        ("k_valid", i32),
    ]


class CradleHipError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Loads the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise CradleHipError(
            f"{_LIB_PATH} is missing: build it with `make -C cradle_vae_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path."
        )
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.cradle_abi_version() != ABI_VERSION:
        raise CradleHipError(f"{_LIB_PATH} has ABI version {lib.cradle_abi_version()}, this binding expects "
                             f"{ABI_VERSION}: rebuild it (make -C cradle_vae_amd/csrc)")
    for cname, cls in (("cradle_mlp_ptrs", MlpPtrs), ("cradle_gemm_desc", GemmDesc), ("cradle_step_desc", StepDesc),
                       ("cradle_batch_desc", BatchDesc)):
        mine, theirs = struct_layout(cls), library_struct_layout(lib, cname)
        if mine != theirs:
            raise CradleHipError(f"layout of {cname} differs between include/cradle_hip.h as compiled "
                                 f"({theirs}) and cradle_vae_amd/hip.py ({mine})")
    _lib = lib
    return lib


def struct_layout(cls) -> list:
    """[sizeof, offsetof(field) ...] of a ctypes mirror, in declaration order."""
    return [ctypes.sizeof(cls)] + [getattr(cls, name).offset for name, *_ in cls._fields_]


def library_struct_layout(lib, name: str) -> list:
    buf = (ctypes.c_int64 * 128)()
    n = lib.cradle_struct_layout(name.encode(), buf, 128)
    if n <= 0:
        raise CradleHipError(f"cradle_struct_layout({name!r}) failed")
    return list(buf[:n])


def check(rc: int, what: str):
    if rc != 0:
        raise CradleHipError(f"{what} failed (code {rc}): {load().cradle_last_error().decode()}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise CradleHipError("cradle_vae_amd kernels only run on the GPU; got a CPU tensor (no fallback path)")
    if t.dtype != torch.float32 and t.dtype != torch.int32:
        raise CradleHipError(f"expected an fp32/int32 tensor, got {t.dtype}")
    if not t.is_contiguous():
        raise CradleHipError("expected a contiguous tensor")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class EventPair:
    """Two hipEvent_t handles created through the library (no torch involvement)."""

    def __init__(self):
        lib = load()
        self.begin, self.end = ctypes.c_void_p(), ctypes.c_void_p()
        check(lib.cradle_event_create(ctypes.byref(self.begin)), "event_create")
        check(lib.cradle_event_create(ctypes.byref(self.end)), "event_create")

    def elapsed_ms(self) -> float:
        ms = ctypes.c_float()
        check(load().cradle_event_elapsed_ms(self.begin, self.end, ctypes.byref(ms)), "event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            lib = load()
            lib.cradle_event_destroy(self.begin)
            lib.cradle_event_destroy(self.end)
        except Exception:
            pass
