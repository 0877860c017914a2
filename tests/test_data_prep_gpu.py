"""GPU: the callers in front of the hot path that prepare its inputs on the device against the
reference's own outputs -- the counterfactual-input builder (reference _get_cfX, cradle_vae/data/norman/data_module.py:
175-191; fixture tests/golden/cfx_tiny.npz, written by oracle/make_cfx_golden.py from the reference's function with its
np.random.choice draws recorded) and the train-set log statistics (cradle_vae/data/utils/batch_statistics.py:4-23;
fixture tests/golden/metrics_tiny.npz) and observation counts (perturbation_dataset.py:64-67)."""
import os

import numpy as np
import pytest
import torch

from tests.golden_io import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfx_case(z, sp):
    treatment, split, qc = z["treatment"], z["split"], z["total_qc"]
    rows = np.nonzero(split == sp)[0]
    codes, fail = torch.from_numpy(treatment[rows]), torch.from_numpy(qc[rows] == 1)
    draws = z[f"draws/{sp}"]
    fail_rank_to_row = np.nonzero(fail.numpy())[0]
    peers = {}
    for t in np.unique(treatment[rows]):
        members = np.nonzero(codes.numpy() == t)[0]
        fails_t = np.nonzero((codes.numpy() == t) & fail.numpy())[0]
        if len(fails_t) == 0:
            continue
        k = min(50, len(fails_t))
        rows_drawn = fail_rank_to_row[draws[members, :k]]
        peers[int(t)] = torch.from_numpy(np.searchsorted(fails_t, rows_drawn))
    return rows, codes, fail, peers


@pytest.mark.parametrize("count_dtype", [torch.float32, torch.int16])
def test_counterfactual_builder_on_the_gpu_matches_the_reference(count_dtype):
    """Same draws in, the reference's cfX out -- bit for bit, all three splits, on device tensors (fp32 counts and the
    loader's packed int16 storage)."""
    from cradle_vae_amd.data import build_counterfactual_inputs

    z = np.load(os.path.join(GOLDEN, "cfx_tiny.npz"), allow_pickle=False)
    X_all = torch.from_numpy(z["X"]).to(DEV).to(count_dtype)
    for sp in ("train", "val", "test"):
        rows, codes, fail, peers = _cfx_case(z, sp)
        got = build_counterfactual_inputs(X_all[torch.from_numpy(rows).to(DEV)], codes, fail, peer_index=peers,
                                          reference_source=X_all)
        assert got.is_cuda
        assert torch.equal(got.float().cpu(), torch.from_numpy(z[f"cfX/{sp}"])), sp


def test_counterfactual_builder_on_the_gpu_at_size():
    """Peers drawn on the device (no injected draws), 50 peers, many genes: every output row is the per-gene lower median
    of its recorded peers' rows (torch.median on the CPU as the checker), and cells of a treatment without a QC-failing
    cell get zeros."""
    from cradle_vae_amd.data import build_counterfactual_inputs

    g = torch.Generator().manual_seed(3)
    n, G, T = 3000, 1003, 7
    X = torch.poisson(3.0 * torch.rand(n, G, generator=g)).to(DEV)
    codes = torch.randint(0, T, (n,), generator=g)
    fail = torch.rand(n, generator=g) < 0.3
    fail[codes == 5] = False  # a treatment without failing cells
    fail[(codes == 6).nonzero().squeeze(1)[3:]] = False  # ... and one with three (fewer than n_peers)
    peers = {}
    for t in range(T):
        m, f = int((codes == t).sum()), int(((codes == t) & fail).sum())
        if f:
            peers[t] = torch.randint(0, f, (m, min(50, f)), generator=g)
    got = build_counterfactual_inputs(X, codes, fail, peer_index=peers).cpu()
    Xc = X.cpu()
    for t in range(T):
        members = torch.nonzero(codes == t).squeeze(1)
        fails = torch.nonzero((codes == t) & fail).squeeze(1)
        if not len(fails):
            assert float(got[members].abs().sum()) == 0.0
            continue
        want = Xc[fails[peers[t]]].median(dim=1)[0]
        assert torch.equal(got[members], want), t


def test_log_statistics_on_the_gpu_match_the_reference():
    from cradle_vae_amd.data import batch_log_mean, batch_log_std

    z = np.load(os.path.join(GOLDEN, "metrics_tiny.npz"), allow_pickle=False)
    X = torch.from_numpy(z["X"]).to(DEV)
    for Xd in (X, X.to(torch.int16)):
        lm = batch_log_mean(Xd, step_size=100)
        ls = batch_log_std(Xd, lm, step_size=100)
        assert lm.is_cuda and ls.is_cuda
        np.testing.assert_allclose(lm.cpu().numpy(), z["stats/log_mean"], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(ls.cpu().numpy(), z["stats/log_std"], rtol=2e-6, atol=1e-6)


def test_log_statistics_and_obs_counts_at_size():
    """A count matrix of 40 000 cells x 1003 genes (uint8 storage), one pass: per-gene mean / std of log(x + 1) against
    fp64 numpy, including a gene that is zero everywhere (std 0: the reference lets nan_to_num act on it later) -- and
    the per-treatment / per-qc observation counts of the data set (perturbation_dataset.py:64-67)."""
    from cradle_vae_amd.data import SparseDosage, TensorPerturbationDataset, batch_log_mean, batch_log_std

    g = torch.Generator().manual_seed(5)
    n, G, T = 40000, 1003, 300
    lam = torch.exp(-1.0 + 1.5 * torch.randn(G, generator=g))
    X = torch.poisson(lam.expand(n, G) * torch.exp(0.3 * torch.randn(n, 1, generator=g)), generator=g).clamp_(max=255)
    from cradle_vae_amd.data import log_mean_std

    X[:, 17] = 0
    X[:, 18] = 3  # constant, non-zero
    Xd = X.to(torch.uint8).to(DEV)
    lm = batch_log_mean(Xd)
    ls = batch_log_std(Xd, lm)
    ref = np.log(X.double().numpy() + 1)
    np.testing.assert_allclose(lm.cpu().numpy(), ref.mean(0), rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(ls.cpu().numpy(), ref.std(0, ddof=1), rtol=5e-6, atol=2e-7)
    assert float(ls[17]) == 0.0 and float(lm[17]) == 0.0
    # the single-pass form the data module uses (shifted sums): same statistics, a constant gene has std 0 exactly
    lm1, ls1 = log_mean_std(Xd)
    np.testing.assert_allclose(lm1.cpu().numpy(), ref.mean(0), rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(ls1.cpu().numpy(), ref.std(0, ddof=1), rtol=5e-6, atol=2e-7)
    assert float(ls1[17]) == 0.0 and float(ls1[18]) == 0.0 and abs(float(lm1[18]) - np.log(4.0)) < 1e-6
    # fp32 storage, a width that is not a multiple of 4 and rows fed in slabs (accumulate)
    from cradle_vae_amd import ops

    Xf = X[:5000, :1001].contiguous().to(DEV)
    s = ops.log_stats(Xf[:3000])
    ops.log_stats(Xf[3000:], sums=s)
    np.testing.assert_allclose((s[0] / 5000).cpu().numpy(), ref[:5000, :1001].mean(0), rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(s[1].cpu().numpy(), (ref[:5000, :1001] ** 2).sum(0), rtol=2e-6, atol=1e-6)
    D = torch.zeros(n, T)
    D[torch.arange(n), torch.randint(0, T, (n,), generator=g)] = 1.0
    D[: n // 2, 3] = 0.5
    qc = (torch.rand(n, 2, generator=g) < 0.2).float()
    ds = TensorPerturbationDataset(Xd, SparseDosage.from_dense(D.to(DEV)), qc.to(DEV), cfX=Xd)
    assert torch.equal(ds.get_dosage_obs_per_dim().cpu(), (D != 0).sum(0))
    assert torch.equal(ds.get_qc_obs_per_dim().cpu(), (qc != 0).sum(0))
    dense = TensorPerturbationDataset(Xd, D.to(DEV), qc.to(DEV), cfX=Xd)  # the reference's dense D
    assert torch.equal(dense.get_dosage_obs_per_dim().cpu(), (D != 0).sum(0))
