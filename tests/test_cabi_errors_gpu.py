"""Error behaviour of the C ABI (include/cradle_hip.h): bad arguments return a non-zero status with a
message in cradle_last_error() and launch nothing; the Python wrappers turn that into CradleHipError."""
import ctypes

import pytest
import torch

from cradle_vae_amd import hip, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_gemm_rejects_bad_descriptors():
    lib = hip.load()
    A, B, C = torch.randn(8, 16, device=DEV), torch.randn(4, 16, device=DEV), torch.empty(8, 4, device=DEV)
    d = hip.GemmDesc()
    d.layout, d.M, d.N, d.K = 7, 8, 4, 16  # no such layout
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = A.data_ptr(), 16, B.data_ptr(), 16, C.data_ptr(), 4
    assert lib.cradle_gemm_f32(ctypes.byref(d), None, 0, hip.stream()) != 0
    assert b"layout" in lib.cradle_last_error()
    d.layout, d.K = 0, 0  # empty reduction
    assert lib.cradle_gemm_f32(ctypes.byref(d), None, 0, hip.stream()) != 0
    d.K, d.extra_col = 16, C.data_ptr()  # the bias-gradient column exists for the TN layout only
    assert lib.cradle_gemm_f32(ctypes.byref(d), None, 0, hip.stream()) != 0
    assert b"TN" in lib.cradle_last_error()
    with pytest.raises(hip.CradleHipError):
        ops.gemm(ops.NT, A.cpu(), B)  # CPU tensor: no fallback


def test_sampler_and_adam_reject_bad_arguments():
    lib = hip.load()
    P, C, G = 2, 3, 10  # rows are run as 12 genes: a leading dimension of 10 leaves no room for that
    logits, lib_sz = torch.randn(P * C, 12, device=DEV), torch.ones(1, device=DEV)
    consts, sums = torch.ones(3, 12, device=DEV), torch.zeros(C, 12, device=DEV)
    rc = lib.cradle_nb_sample_accumulate(hip.ptr(logits), 10, hip.ptr(lib_sz), 1, hip.ptr(consts), hip.ptr(sums), None,
                                         P, C, G, 0, 0, 1, 0, hip.stream())
    assert rc != 0 and b"multiple of 4" in lib.cradle_last_error()
    rc = lib.cradle_nb_sample_accumulate(hip.ptr(logits), 12, hip.ptr(lib_sz), 2, hip.ptr(consts), hip.ptr(sums), None,
                                         P, C, 12, 0, 0, 1, 0, hip.stream())
    assert rc != 0 and b"library" in lib.cradle_last_error()  # 1 or C library sizes
    p = torch.zeros(16, device=DEV)
    rc = lib.cradle_adam_step(hip.ptr(p), hip.ptr(p), hip.ptr(p), hip.ptr(p), 16, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 100.0, 1.0,
                              hip.ptr(p), None, hip.stream())
    assert rc != 0  # step counts from 1


def test_step_rejects_bad_descriptors():
    lib = hip.load()
    assert lib.cradle_elbo_step(None, hip.stream()) != 0
    assert b"null descriptor" in lib.cradle_last_error()
    d = hip.StepDesc()
    d.P, d.N, d.T, d.L, d.Q, d.G, d.He, d.Hm, d.Hd, d.kmax = 1, 4, 3, 2, 1, 3, 4, 4, 4, 2  # fewer than 4 genes
    assert lib.cradle_elbo_step(ctypes.byref(d), hip.stream()) != 0
    assert b"below 4" in lib.cradle_last_error()
    d.G, d.kmax = 12, 0
    assert lib.cradle_elbo_step(ctypes.byref(d), hip.stream()) != 0
    assert b"kmax" in lib.cradle_last_error()
    assert lib.cradle_elbo_step_workspace_bytes(None) == -1
    p = ctypes.c_void_p()
    assert lib.cradle_elbo_step_intermediate(ctypes.byref(d), b"mask", ctypes.byref(p)) == -1  # no workspace set
