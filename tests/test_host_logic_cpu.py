"""Host-side logic that needs no GPU: the config-built training module, checkpoints in
Lightning's layout, the device batch source (on CPU tensors), namespaces, and the rule that
the product path refuses to compute without the GPU (no CPU fallback)."""
import os

import numpy as np
import pytest
import torch

from tests import golden_io as gio


def _config(cfg, log_mean, log_std):
    from cradle_vae_amd.guide import ObservationNormalizationStatistics

    return dict(
        model="cradle_vae_Model", guide="cradle_vae_CorrelatedNormalGuide", loss_module="cradle_vae_ELBOLossModule",
        predictor="cradle_vae_Predictor", data_module="NormanOODCombinationDataModule",
        data_module_kwargs=dict(n_qc_pass=80, n_qc_fail=20),
        lightning_module_kwargs=dict(lr=3e-4, n_particles=cfg["n_particles"]),
        loss_module_kwargs=dict(beta=cfg["beta"]),
        model_kwargs=dict(
            n_latent=cfg["n_latent"], n_treatments=cfg["n_treatments"], n_phenos=cfg["n_phenos"], n_qc=cfg["n_qc"],
            mask_prior_prob=cfg["mask_prior_prob"], embedding_prior_scale=cfg["embedding_prior_scale"],
            likelihood_key="library_nb", decoder_n_layers=cfg["decoder_n_layers"],
            decoder_n_hidden=cfg["decoder_n_hidden"], sum_mode=False),
        guide_kwargs=dict(
            n_latent=cfg["n_latent"], n_treatments=cfg["n_treatments"], n_phenos=cfg["n_phenos"], n_qc=cfg["n_qc"],
            basal_encoder_n_layers=cfg["basal_encoder_n_layers"], basal_encoder_n_hidden=cfg["basal_encoder_n_hidden"],
            basal_encoder_input_normalization="log_standardize",
            embedding_encoder_n_layers=cfg["embedding_encoder_n_layers"],
            embedding_encoder_n_hidden=cfg["embedding_encoder_n_hidden"],
            x_normalization_stats=ObservationNormalizationStatistics(log_x_mean=log_mean, log_x_std=log_std)),
    )


def _module(case, with_counts=True):
    from cradle_vae_amd.models.utils.perturbation_lightning_module import TrainConfigPerturbationLightningModule

    cfg, p, b = case["cfg"], case["params"], case["batch"]
    config = _config(cfg, p["guide.normalization_module.log_mean"].clone(), p["guide.normalization_module.log_scale"].clone())
    kw = {}
    if with_counts:
        kw = dict(D_obs_counts_train=b["D_obs_counts"], D_obs_counts_val=b["D_obs_counts"], D_obs_counts_test=b["D_obs_counts"],
                  qc_obs_counts_train=b["qc_obs_counts"], qc_obs_counts_val=b["qc_obs_counts"],
                  qc_obs_counts_test=b["qc_obs_counts"])
    return TrainConfigPerturbationLightningModule(config=config, **kw)


def test_models_namespace_matches_reference_exports():
    import cradle_vae_amd.models as m

    for name in ("cradle_vae_Model", "cradle_vae_CorrelatedNormalGuide", "cradle_vae_ELBOLossModule",
                 "cradle_vae_CustomReweightedELBOLossModule", "cradle_vae_IWELBOLossModule", "cradle_vae_Predictor"):
        assert hasattr(m, name), name
    from cradle_vae_amd.analysis.average_treatment_effects import estimate_model_average_treatment_effect  # noqa: F401
    from cradle_vae_amd.models.utils.predictor import PerturbationPlatedPredictor  # noqa: F401


def test_config_module_state_dict_has_reference_keys():
    case = gio.load_elbo("tiny")
    mod = _module(case)
    sd = mod.state_dict()
    # the reference's checkpoint keys: loss_module.{model,guide}.* and the same tensors again under predictor.*
    for k, v in case["params"].items():
        assert f"loss_module.{k}" in sd and f"predictor.{k}" in sd, k
        assert tuple(sd[f"loss_module.{k}"].shape) == tuple(v.shape), k
    assert mod.predictor.model is mod.loss_module.model and mod.predictor.guide is mod.loss_module.guide
    assert mod.loss_module.pos_weight == 0.8 and mod.n_particles == case["cfg"]["n_particles"]
    assert not mod.predictor.model.training  # predictor puts both in eval mode (predictor.py:60-61)
    missing, unexpected = mod.loss_module.load_state_dict(case["params"], strict=True)
    assert not missing and not unexpected
    opt = mod.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["weight_decay"] == 1e-6


def test_obs_count_placeholders_and_condition_values():
    case = gio.load_elbo("tiny")
    mod = _module(case, with_counts=False)
    T, Q = case["cfg"]["n_treatments"], case["cfg"]["n_qc"]
    for split in ("train", "val", "test"):
        assert torch.equal(getattr(mod, f"D_obs_counts_{split}"), -torch.ones(T))
        assert torch.equal(getattr(mod, f"qc_obs_counts_{split}"), -torch.ones(Q))
    cv = mod._get_extra_condition_values(dict(X=1, D=2, qc=3, cfX=4, library_size=5, idx=6))
    assert set(cv) == {"qc", "cfX", "library_size", "idx"}
    with pytest.raises(NotImplementedError):
        mod.get_data_module()


def test_checkpoint_roundtrip(tmp_path):
    from cradle_vae_amd.lightning_module import load_checkpoint, save_checkpoint

    case = gio.load_elbo("tiny")
    mod = _module(case)
    mod.loss_module.load_state_dict(case["params"], strict=True)
    path = os.path.join(tmp_path, "last.ckpt")
    save_checkpoint(mod, path, epoch=3)
    raw = torch.load(path, weights_only=True)  # plain tensors / dicts only
    assert raw["epoch"] == 3 and "loss_module.model.decoder.log_concentration" in raw["state_dict"]
    again = type(mod).load_from_checkpoint(path)
    for (k1, v1), (k2, v2) in zip(mod.state_dict().items(), again.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    assert again.config["loss_module_kwargs"]["beta"] == case["cfg"]["beta"]


def test_no_cpu_fallback():
    from cradle_vae_amd.hip import CradleHipError

    case = gio.load_elbo("tiny")
    mod = _module(case)
    b = case["batch"]
    with pytest.raises(CradleHipError):
        mod.training_step({k: b[k] for k in ("X", "D", "qc", "cfX", "library_size")}, 0)


def test_device_batch_loader_shards_and_gathers():
    from cradle_vae_amd.data import DeviceBatchLoader, SCRNASeqTensorPerturbationDataset

    g = torch.Generator().manual_seed(0)
    n, G, T = 103, 8, 5
    X = torch.poisson(3 * torch.rand(n, G, generator=g))
    cfX = torch.poisson(3 * torch.rand(n, G, generator=g))
    D = (torch.rand(n, T, generator=g) < 0.3).float()
    qc = (torch.rand(n, 1, generator=g) < 0.2).float()
    ds = SCRNASeqTensorPerturbationDataset(X, D, qc, cfX, ids=[f"c{i}" for i in range(n)])
    assert torch.equal(ds.get_dosage_obs_per_dim(), (D != 0).sum(0)) and torch.equal(ds.get_qc_obs_per_dim(), (qc != 0).sum(0))
    s = ds[7]
    assert float(s["library_size"]) == float(X[7].sum()) and torch.equal(s["cfX"], cfX[7])
    assert list(ds.convert_idx_to_ids(np.array([0, 2]))) == ["c0", "c2"]
    # unshuffled, one rank: every cell once, in order
    seen = []
    for b in DeviceBatchLoader(ds, batch_size=16):
        i = b["idx"]
        assert torch.equal(b["X"], X[i]) and torch.equal(b["D"], D[i]) and torch.equal(b["library_size"], X[i].sum(1))
        seen.append(i)
    assert torch.equal(torch.cat(seen), torch.arange(n))
    # shuffled, two ranks: disjoint, equal length, same permutation on both ranks, new order per epoch
    per_rank = []
    for r in range(2):
        ld = DeviceBatchLoader(ds, batch_size=10, shuffle=True, drop_last=True, seed=4, rank=r, world_size=2)
        e0 = torch.cat([b["idx"] for b in ld])
        e1 = torch.cat([b["idx"] for b in ld])
        assert len(e0) == len(ld) * 10 == 50 and not torch.equal(e0, e1)
        per_rank.append(e0)
    assert len(set(per_rank[0].tolist()) & set(per_rank[1].tolist())) == 0
    # evaluation sharding keeps every cell
    ev = [torch.cat([b["idx"] for b in DeviceBatchLoader(ds, 16, rank=r, world_size=2)]) for r in range(2)]
    assert sorted(torch.cat(ev).tolist()) == list(range(n))


def test_sparse_dosage_and_packed_counts_give_the_dense_loaders_batches():
    """D as (index, dosage) pairs and counts as small integers: same batches, bit for bit, as the dense fp32
    storage of the reference's TensorPerturbationDataset (perturbation_dataset.py:44-89), at a fraction of it."""
    from cradle_vae_amd.data import (DeviceBatchLoader, SCRNASeqTensorPerturbationDataset, SparseDosage,
                                     TensorPerturbationDataModule, pack_counts)

    g = torch.Generator().manual_seed(1)
    n, G, T = 203, 12, 300
    X = torch.poisson(40 * torch.rand(n, G, generator=g))
    X[3, 2] = 70000.0  # forces 32-bit storage for X; cfX fits 8 bit
    cfX = torch.poisson(3 * torch.rand(n, G, generator=g))
    cfX[::3] = 0.0
    D = torch.zeros(n, T)
    D[torch.arange(n), torch.randint(0, T, (n,), generator=g)] = 1.0
    two = torch.rand(n, generator=g) < 0.3
    D[torch.arange(n)[two], torch.randint(0, T, (int(two.sum()),), generator=g)] = 0.5  # fractional dosage survives
    D[:5] = 0.0  # control cells
    qc = (torch.rand(n, 1, generator=g) < 0.2).float()
    sp = SparseDosage.from_dense(D)
    assert sp.idx.shape == (n, 2) and torch.equal(sp.dense(), D) and torch.equal(sp[7], D[7]) and torch.equal(sp[2:9], D[2:9])
    assert torch.equal(sp.obs_per_dim(), (D != 0).sum(0)) and sp.max_per_row() == 2
    with pytest.raises(ValueError):
        SparseDosage.from_dense(D, kmax=1)
    assert pack_counts(X).dtype == torch.int32 and pack_counts(cfX).dtype == torch.uint8
    assert pack_counts(X + 0.5).dtype == torch.float32  # not integer-valued: left alone
    dense = SCRNASeqTensorPerturbationDataset(X, D, qc, cfX)
    sparse = SCRNASeqTensorPerturbationDataset(X, sp, qc, cfX)
    assert torch.equal(sparse.get_dosage_obs_per_dim(), dense.get_dosage_obs_per_dim())
    assert sparse.max_dosages_per_cell() == dense.max_dosages_per_cell() == 2 and torch.equal(sparse[11]["D"], D[11])
    a = DeviceBatchLoader(dense, 32, shuffle=True, seed=5, count_dtype=None, sparse_dosage=False)  # as the reference stores it
    b = DeviceBatchLoader(sparse, 32, shuffle=True, seed=5)
    assert b.stored_bytes() < 0.5 * a.stored_bytes()
    for x, y in zip(a, b):
        assert set(x) == set(y)
        assert x["treatments_present"] == y["treatments_present"] == int((x["D"] != 0).any(0).sum())  # a host int
        for k in x:
            if k != "treatments_present":
                assert x[k].dtype == y[k].dtype and torch.equal(x[k], y[k]), k
    # the in-memory data module keeps D sparse throughout
    names = ["ctl" if not r.any() else "+".join(f"g{j}" for j in torch.nonzero(r).squeeze(1).tolist()) for r in D]
    splits = np.where(np.arange(n) % 5 == 0, "val", np.where(np.arange(n) % 7 == 0, "test", "train"))
    dm = TensorPerturbationDataModule(X, sp, qc, names, splits, batch_size=16, control_label="ctl")
    assert isinstance(dm.D, SparseDosage) and isinstance(dm.datasets["train"].D, SparseDosage)
    tr = splits == "train"
    assert torch.equal(dm.get_train_perturbation_obs_counts(), (D[tr] != 0).sum(0))
    bt = next(iter(dm.train_dataloader()))
    assert bt["D"].shape == (16, T) and torch.equal(bt["D"], D[torch.from_numpy(np.nonzero(tr)[0])][
        torch.tensor([list(dm.datasets["train"].ids).index(f"cell{int(i)}") for i in
                      torch.from_numpy(np.nonzero(tr)[0])[bt["idx"]]])])
    some = dm.get_unique_observed_intervention_info().index[:3]
    assert dm.get_unique_observed_intervention_dosages(some).shape == (3, T)


def test_config_semantics_match_the_reference_on_its_demo_yamls():
    """preprocess_config / add_data_info_to_config against the outputs of the reference's own functions
    (train_norman.py:163-227, run by oracle/make_config_golden.py) on its four demo yamls plus a case with the rules
    the demo files do not use: 'a--b' shared keys, shared n_latent, gradient_clip_norm -1, dots beyond one level."""
    import json
    import os

    from cradle_vae_amd.config import add_data_info_to_config, load_config, preprocess_config
    from oracle.make_config_golden import FakeDataModule

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = json.load(open(os.path.join(root, "tests", "golden", "config_demo.json")))
    assert len(cases) == 5
    for name, c in cases.items():
        got = preprocess_config(dict(c["input"]))
        assert json.loads(json.dumps(got)) == c["processed"], name
        if "with_data_info" in c:
            full = add_data_info_to_config(preprocess_config(dict(c["input"])), FakeDataModule())
            assert json.loads(json.dumps(full)) == c["with_data_info"], name
    rules = cases["synthetic_rules"]["processed"]
    assert rules["gradient_clip_norm"] is None and rules["model_kwargs"] == {"n_latent": 7, "x": 3} and rules["g"] == {"h.i": 1}
    # this build's own demo file goes through the same functions
    cfg = preprocess_config(load_config(os.path.join(root, "demo", "cradle_vae_synthetic.yaml")))
    assert cfg["model_kwargs"]["n_latent"] == cfg["guide_kwargs"]["n_latent"] == 32
    assert cfg["data_module"] == "SyntheticPerturbationDataModule" and cfg["data_module_kwargs"]["batch_size"] == 256


def test_compat_alias_package_resolves_reference_import_paths():
    """`import cradle_vae...` with <repo>/compat on the path lands on this build (run in a child
    process so that the alias never shadows the reference for the golden-fixture tooling)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "from cradle_vae import models, data\n"
        "from cradle_vae.models.utils.perturbation_lightning_module import TrainConfigPerturbationLightningModule as T\n"
        "from cradle_vae.models.utils.predictor import PerturbationPlatedPredictor\n"
        "from cradle_vae.models.cradle_vae.guides.correlated_normal_guide import cradle_vae_CorrelatedNormalGuide\n"
        "from cradle_vae.analysis.average_treatment_effects import estimate_model_average_treatment_effect\n"
        "from cradle_vae.data.utils.perturbation_dataset import SCRNASeqTensorPerturbationDataset\n"
        "from cradle_vae.data.utils.perturbation_datamodule import ObservationNormalizationStatistics\n"
        "import cradle_vae_amd.models as m\n"
        "assert models.cradle_vae_Model is m.cradle_vae_Model and T.__module__.startswith('cradle_vae_amd')\n"
        "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "compat"), root]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_counterfactual_builder_matches_per_cell_loop():
    """build_counterfactual_inputs against the reference's per-cell procedure (norman/data_module.py:
    175-191) with the peer draws injected."""
    from cradle_vae_amd.data import build_counterfactual_inputs

    g = torch.Generator().manual_seed(1)
    n, G = 90, 7
    X = torch.poisson(4 * torch.rand(n, G, generator=g))
    codes = torch.randint(0, 4, (n,), generator=g)
    fail = torch.rand(n, generator=g) < 0.3
    fail[codes == 3] = False  # a treatment without QC-failing cells -> zeros
    peers = {}
    for t in range(4):
        n_t, n_f = int((codes == t).sum()), int(((codes == t) & fail).sum())
        if n_f:
            peers[t] = torch.randint(0, n_f, (n_t, min(50, n_f)), generator=g)
    got = build_counterfactual_inputs(X, codes, fail, peer_index=peers, max_rows_per_gather=64)
    want = torch.zeros(n, G)
    for t in range(4):
        members = torch.nonzero(codes == t).squeeze(1)
        fails = torch.nonzero((codes == t) & fail).squeeze(1)
        for j, i in enumerate(members.tolist()):
            if len(fails):
                want[i] = X[fails[peers[t][j]]].median(0)[0]
    assert torch.equal(got, want) and float(got[codes == 3].abs().sum()) == 0.0


def test_counterfactual_builder_matches_the_references_own_output():
    """build_counterfactual_inputs against the output of the reference's ``_get_cfX`` itself
    (norman/data_module.py:175-191, executed by oracle/make_cfx_golden.py with its np.random.choice draws recorded):
    same draws in, same cfX out, for all three splits -- including the reference's indexing of the un-split matrix
    with positions taken inside the split's QC-failing subset."""
    import os

    from cradle_vae_amd.data import build_counterfactual_inputs

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfx_tiny.npz"), allow_pickle=False)
    X_all = torch.from_numpy(z["X"])
    treatment, split, qc = z["treatment"], z["split"], z["total_qc"]
    for sp in ("train", "val", "test"):
        rows = np.nonzero(split == sp)[0]
        codes, fail = torch.from_numpy(treatment[rows]), torch.from_numpy(qc[rows] == 1)
        draws = z[f"draws/{sp}"]  # per cell: positions (within the split's failing cells) np.random.choice returned
        fail_rank_to_row = np.nonzero(fail.numpy())[0]
        peers = {}
        for t in np.unique(treatment[rows]):
            members = np.nonzero(codes.numpy() == t)[0]
            fails_t = np.nonzero((codes.numpy() == t) & fail.numpy())[0]
            if len(fails_t) == 0:
                assert (draws[members] == -1).all()
                continue
            k = min(50, len(fails_t))
            assert (draws[members, :k] >= 0).all() and (draws[members, k:] == -1).all()
            # drawn value = rank among the split's failing cells -> that cell's position in its treatment's fail list
            rows_drawn = fail_rank_to_row[draws[members, :k]]
            assert np.isin(rows_drawn, fails_t).all()  # the reference draws peers of the same treatment ...
            peers[int(t)] = torch.from_numpy(np.searchsorted(fails_t, rows_drawn))
        got = build_counterfactual_inputs(X_all[rows], codes, fail, peer_index=peers, reference_source=X_all)
        assert torch.equal(got, torch.from_numpy(z[f"cfX/{sp}"])), sp
        meant = build_counterfactual_inputs(X_all[rows], codes, fail, peer_index=peers)
        assert not torch.equal(meant, got)  # ... but reads other cells' counts than theirs


def test_synthetic_data_module_interface():
    from cradle_vae_amd.data import SyntheticPerturbationDataModule

    dm = SyntheticPerturbationDataModule(n_cells=500, n_genes=24, n_guides=4, batch_size=50, seed=3)
    n_tr = len(dm.datasets["train"])
    assert n_tr + len(dm.datasets["val"]) + len(dm.datasets["test"]) == 500
    assert torch.equal(dm.get_train_perturbation_obs_counts(), (dm.datasets["train"].D.dense() != 0).sum(0))
    assert dm.get_qc_n()["n_qc_pass"] + dm.get_qc_n()["n_qc_fail"] == 500
    stats = dm.get_x_train_statistics()
    assert stats.log_x_mean.shape == (24,) and bool((stats.log_x_std > 0).all())
    info = dm.get_unique_observed_intervention_info()
    assert "non-targeting" in info.index and set(info.columns) == {"train", "val", "test"}
    names = [x for x in info.index if x != "non-targeting"]
    Dq = dm.get_unique_observed_intervention_dosages(names)
    for name, row in zip(names, Dq):
        assert sorted(f"guide{j}" for j in torch.nonzero(row).squeeze(1).tolist()) == sorted(name.split("+"))
    ate = dm.get_estimated_average_treatment_effects("mean")
    assert ate.X.shape == (len(names), 24) and ate.uns["control"] == "non-targeting"
    b = next(iter(dm.train_dataloader()))
    assert set(b) == {"X", "D", "qc", "cfX", "library_size", "idx", "treatments_present"} and b["X"].shape == (50, 24)


def test_max_dosages_per_cell_reaches_the_loss_module():
    """The data set's bound on non-zero dosages per cell sizes the fused step's dosage lists and its
    compact treatment axis; trainer.fit copies it to the loss module (and a changed bound rebuilds
    the step)."""
    from cradle_vae_amd.data import TensorPerturbationDataset

    D = torch.zeros(7, 5)
    D[0, 1] = 1.0
    D[3, [0, 2, 4]] = torch.tensor([1.0, 0.5, 2.0])
    ds = TensorPerturbationDataset(X=torch.ones(7, 4), D=D, qc=torch.zeros(7, 1))
    assert ds.max_dosages_per_cell() == 3
    empty = TensorPerturbationDataset(X=torch.ones(2, 4), D=torch.zeros(2, 5), qc=torch.zeros(2, 1))
    assert empty.max_dosages_per_cell() == 1  # all-control data: one (unused) slot per cell

    from cradle_vae_amd import trainer

    class _Loader(list):
        dataset = ds

    seen = {}

    class _Stop(Exception):
        pass

    class _LM:
        max_dosages_per_cell = 8

    class _Module:
        loss_module = _LM()
        lr, weight_decay, n_particles, use_scheduler = 1e-3, 0.0, 1, False

    def fake_train_step(lm, **kw):
        seen["kmax"] = lm.max_dosages_per_cell
        raise _Stop()

    orig = trainer.TrainStep
    trainer.TrainStep = fake_train_step
    try:
        with pytest.raises(_Stop):
            trainer.fit(_Module(), _Loader())
    finally:
        trainer.TrainStep = orig
    assert seen["kmax"] == 3


def _layout():
    import json

    with open(os.path.join(gio.GOLDEN, "lightning_layout.json")) as f:
        return json.load(f)


def reference_checkpoint_case(device="cpu"):
    """The state dict the reference's own TrainConfigPerturbationLightningModule accepted (strict) and the ELBO it
    computed from it, with its config (oracle/make_checkpoint_golden.py)."""
    from cradle_vae_amd.guide import ObservationNormalizationStatistics

    z = np.load(os.path.join(gio.GOLDEN, "lightning_layout.npz"), allow_pickle=False)
    lay = _layout()["small_values"]
    config = lay["config"]
    config["guide_kwargs"]["x_normalization_stats"] = ObservationNormalizationStatistics(
        log_x_mean=torch.from_numpy(z["log_x_mean"]), log_x_std=torch.from_numpy(z["log_x_std"]))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    # Lightning writes the predictor's aliases of the same tensors too
    for key, _, _ in lay["state_dict"]:
        if key.startswith("predictor."):
            sd[key] = sd["loss_module." + key[len("predictor."):]]
    sd = {key: sd[key] for key, _, _ in lay["state_dict"]}
    grp = lambda p: {k[len(p) + 1:]: torch.from_numpy(z[k]).to(device) for k in z.files if k.startswith(p + "/")}
    return dict(config=config, state_dict=sd, batch=grp("batch"), noise=grp("noise"), cf_noise=grp("cf_noise"),
                loss=float(z["loss"]), metrics={k: float(v) for k, v in grp("metrics").items()},
                n_particles=int(z["n_particles"]))


@pytest.mark.parametrize("yaml_name", ["cradle_vae_norman.yaml", "cradle_vae_replogle.yaml", "cradle_vae_adamson.yaml",
                                       "cradle_vae_dixit.yaml"])
def test_state_dict_layout_equals_the_reference_modules(yaml_name):
    """Key order, shapes and dtypes of state_dict(), the optimizer and pos_weight of the module built from the
    reference's demo yaml equal those of the reference's own module built from the same config (fixture)."""
    from cradle_vae_amd.config import add_data_info_to_config, preprocess_config
    from cradle_vae_amd.guide import ObservationNormalizationStatistics
    from cradle_vae_amd.models.utils.perturbation_lightning_module import TrainConfigPerturbationLightningModule
    import json

    with open(os.path.join(gio.GOLDEN, "config_demo.json")) as f:
        raw = json.load(f)[yaml_name]["input"]
    lay = _layout()[yaml_name]
    T = [s for k, s, _ in lay["state_dict"] if k == "D_obs_counts_train"][0][0]
    G = [s for k, s, _ in lay["state_dict"] if k == "loss_module.model.decoder.log_concentration"][0][0]

    class DM:
        class _F:
            def __init__(self, n):
                self.shape = (n, 0)

        def get_d_var_info(self):
            return self._F(T)

        def get_x_var_info(self):
            return self._F(G)

        def get_qc_var_info(self):
            return self._F(1)

        def get_x_train_statistics(self):
            return ObservationNormalizationStatistics(log_x_mean=torch.zeros(G), log_x_std=torch.ones(G))

        def get_qc_n(self):
            return {"n_qc_pass": 900, "n_qc_fail": 100}

    config = add_data_info_to_config(preprocess_config(dict(raw)), DM())
    mod = TrainConfigPerturbationLightningModule(config=config)
    mine = [[k, list(v.shape), str(v.dtype)] for k, v in mod.state_dict().items()]
    assert mine == lay["state_dict"]
    assert sum(p.numel() for p in mod.parameters()) == lay["n_parameters"]
    opt = mod.configure_optimizers()
    o = lay["optimizer"]
    assert type(opt).__name__ == o["cls"] and opt.defaults["lr"] == o["lr"] and opt.defaults["eps"] == o["eps"]
    assert opt.defaults["weight_decay"] == o["weight_decay"] and list(opt.defaults["betas"]) == o["betas"]
    assert sum(len(g["params"]) for g in opt.param_groups) == o["n_param_tensors"]
    assert mod.loss_module.pos_weight == lay["pos_weight"] and mod.n_particles == lay["n_particles"]


def test_checkpoint_in_the_references_layout_loads_strictly(tmp_path):
    """A file shaped like Lightning's checkpoint of the reference module ({"state_dict", "hyper_parameters"}) with the
    reference's own key set loads with strict=True and gives back every tensor."""
    from cradle_vae_amd.lightning_module import load_checkpoint
    from cradle_vae_amd.models.utils.perturbation_lightning_module import TrainConfigPerturbationLightningModule

    case = reference_checkpoint_case()
    mod = TrainConfigPerturbationLightningModule(config=case["config"])
    missing, unexpected = mod.load_state_dict(case["state_dict"], strict=True)
    assert not missing and not unexpected
    for k, v in mod.state_dict().items():
        assert torch.equal(v, case["state_dict"][k]), k
    assert mod.predictor.guide is mod.loss_module.guide and mod.predictor.model is mod.loss_module.model


def test_hardware_queue_setting_respects_the_environment():
    """cradle_vae_amd.env: sixteen hardware queues for N > 1 unless the user chose -- a GPU_MAX_HW_QUEUES of 6 or more is
    kept, one below (the step is known to be slow there) is raised with a notice unless CRADLE_HW_QUEUES=keep;
    CRADLE_HW_QUEUES validated against the runtime's 1..32."""
    import pytest

    from cradle_vae_amd.env import set_hw_queues_for_multi_gpu as setq

    env = {"WORLD_SIZE": "1"}
    setq(env)
    assert "GPU_MAX_HW_QUEUES" not in env  # one GPU: HIP's default
    env = {"WORLD_SIZE": "8"}
    setq(env)
    assert env["GPU_MAX_HW_QUEUES"] == "16"
    env = {"WORLD_SIZE": "8", "GPU_MAX_HW_QUEUES": "6"}
    setq(env)
    assert env["GPU_MAX_HW_QUEUES"] == "6"  # a workable setting of the user's is respected
    env = {"WORLD_SIZE": "8", "GPU_MAX_HW_QUEUES": "4", "RANK": "1"}
    setq(env)
    assert env["GPU_MAX_HW_QUEUES"] == "16"  # HIP's own default exported by the host: known slow, raised (rank 0 says so)
    env = {"WORLD_SIZE": "8", "GPU_MAX_HW_QUEUES": "4", "CRADLE_HW_QUEUES": "keep"}
    setq(env)
    assert env["GPU_MAX_HW_QUEUES"] == "4"
    env = {"WORLD_SIZE": "8", "GPU_MAX_HW_QUEUES": "6", "CRADLE_HW_QUEUES": "12", "RANK": "3"}
    setq(env)
    assert env["GPU_MAX_HW_QUEUES"] == "12"  # the explicit choice decides
    env = {"WORLD_SIZE": "2", "CRADLE_HW_QUEUES": "12"}
    setq(env)
    assert env["GPU_MAX_HW_QUEUES"] == "12"
    for bad in ("0", "33", "many"):
        with pytest.raises(ValueError):
            setq({"WORLD_SIZE": "2", "CRADLE_HW_QUEUES": bad})
