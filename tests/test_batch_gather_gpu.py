"""K0 (cradle_gather_batch: the device-resident batch source's gather, csrc/batch.hip) against the same loader running
on torch ops (index_select, integer -> fp32, scatter into zeros): bit-equal batches for every storage form -- counts as
uint8 / int16 / int32 / fp32, D as (treatment, dosage) pairs or dense, ragged widths, a last short batch."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dataset(n, G, T, Q, xmax, cfmax, seed, fractional=False):
    from cradle_vae_amd.data import SCRNASeqTensorPerturbationDataset

    g = torch.Generator().manual_seed(seed)
    X = torch.randint(0, xmax + 1, (n, G), generator=g).float()
    X[0, 0] = xmax
    cfX = torch.randint(0, cfmax + 1, (n, G), generator=g).float()
    if fractional:
        X = X + 0.25
    D = torch.zeros(n, T)
    a, b = torch.randint(0, T, (n,), generator=g), torch.randint(0, T, (n,), generator=g)
    D[torch.arange(n), a] = 1.0
    two = torch.rand(n, generator=g) < 0.5
    D[torch.arange(n)[two], b[two]] = 0.5
    D[3] = 0.0  # a control cell: no dosage at all
    qc = (torch.rand(n, Q, generator=g) < 0.2).float()
    return SCRNASeqTensorPerturbationDataset(X.to(DEV), D.to(DEV), qc.to(DEV), cfX.to(DEV))


@pytest.mark.parametrize("G,T,Q,xmax,cfmax,sparse,fractional", [
    (64, 40, 1, 200, 9, True, False),        # uint8 / uint8, pairs
    (301, 37, 2, 30000, 300, True, False),   # int16 / int16, widths that are not multiples of 4
    (128, 2000, 1, 100000, 7, True, False),  # int32 / uint8, more treatments than cells
    (96, 50, 1, 50, 50, False, False),       # dense D as the reference stores it
    (64, 700, 1, 50, 50, False, False),      # dense D, treatments over three workgroups of the presence count
    (100, 33, 1, 50, 5, True, True),         # non-integer X: stays fp32
])
def test_hip_gather_equals_the_torch_loader(G, T, Q, xmax, cfmax, sparse, fractional, monkeypatch):
    from cradle_vae_amd.data import DeviceBatchLoader

    ds = _dataset(150, G, T, Q, xmax, cfmax, seed=G + T, fractional=fractional)
    kw = dict(shuffle=True, seed=3, sparse_dosage=sparse, device=DEV)
    hip_loader = DeviceBatchLoader(ds, 64, **kw)
    assert hip_loader._hip_ok
    got = [dict(b) for b in hip_loader]
    torch.cuda.synchronize()
    monkeypatch.setattr(DeviceBatchLoader, "_hip_ok", property(lambda self: False))
    ref = [dict(b) for b in DeviceBatchLoader(ds, 64, **kw)]
    assert len(got) == len(ref) == 3 and got[-1]["X"].shape[0] == 150 - 128
    for x, y in zip(got, ref):
        assert set(x) == set(y)
        assert x["treatments_present"] == y["treatments_present"] == int((y["D"] != 0).any(0).sum()) > 0
        for k in ("X", "cfX", "D", "qc", "library_size", "idx"):
            assert x[k].dtype == y[k].dtype and x[k].shape == y[k].shape and torch.equal(x[k], y[k]), k
    assert torch.equal(got[0]["X"], ds.X[got[0]["idx"].to(DEV)].float())


def test_a_position_outside_the_stored_rows_is_reported_not_followed():
    """cradle_batch_desc.rows: a stale or oversized index into the stored data set (a loader reused after its data set
    shrank, a corrupted permutation) must not be dereferenced on the device: the gather reads row 0 in its place, raises
    the flag, and the loader turns that into an IndexError when it hands the batch over."""
    import ctypes

    from cradle_vae_amd import hip
    from cradle_vae_amd.data import DeviceBatchLoader

    ds = _dataset(40, 32, 6, 1, 50, 5, seed=1)
    loader = DeviceBatchLoader(ds, 16, device=DEV, prefetch=False)
    good = torch.tensor([3, 5, 39], device=DEV)
    out = loader._gather_hip(good)
    torch.cuda.synchronize()
    assert out["treatments_present"].tolist()[1] == 0 and torch.equal(out["X"], ds.X[good].float())
    bad = torch.tensor([3, 40, -1, 7], device=DEV)  # 40 and -1 do not exist
    out = loader._gather_hip(bad)
    torch.cuda.synchronize()
    assert out["treatments_present"].tolist()[1] == 1
    assert torch.equal(out["X"][[0, 3]], ds.X[torch.tensor([3, 7], device=DEV)].float())
    assert torch.equal(out["X"][1], ds.X[0].float()) and torch.equal(out["X"][2], ds.X[0].float())
    loader._indices = lambda: (torch.tensor([1, 2, 99]), torch.tensor([1, 2, 99]))
    with pytest.raises(IndexError, match="outside the stored data set"):
        list(loader)
