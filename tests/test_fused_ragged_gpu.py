"""Gene counts that are not a multiple of 4 (real datasets: whatever the highly-variable-gene filter left) through the
fused HIP step, against the fp64 oracle at the bars of tests/test_fused_fullsize_gpu.py.  The step pads the row
length of everything K1 touches to the next multiple of 4 (padding genes: count 0, logit -inf, concentration 1e-30 --
they add nothing to the likelihood and exact zeros to every gradient) and lets the GEMM loader's range check supply
zeros where an operand ends inside a 16-byte load; the reference has no such limit (any n_phenos), so neither may the
drop-in.  One width per residue mod 4, the last at the Norman widths (split-bf16 mode included)."""
import dataclasses

import pytest
import torch

from tests.test_fused_fullsize_gpu import compare_with_oracle
from tests.test_modular_gpu import DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("base,n_cells,n_phenos,n_latent,seed", [
    ("tiny", 8, 5, 4, 2),           # the smallest widths: one 16-byte vector and a bit
    ("tiny", 5, 7, 3, 3),
    ("tiny", 1, 6, 5, 4),           # a single cell
    ("mid", 64, 301, 16, 5),        # G % 4 == 1, small-tile GEMM kernels
    ("mid", 96, 1202, 15, 6),       # G % 4 == 2, odd latent width: encoder rows start at odd offsets
    ("norman", 256, 5003, 200, 7),  # G % 4 == 3, the bench shape's widths: large-tile kernels, split-K, bf16x3
])
def test_fused_step_with_ragged_gene_count_matches_oracle(base, n_cells, n_phenos, n_latent, seed):
    from cradle_vae_amd import synthetic as syn

    shape = dataclasses.replace(syn.SHAPES[base], name=f"{base}_g{n_phenos}", n_cells=n_cells, n_phenos=n_phenos,
                                n_latent=n_latent)
    precisions = ("f32", "bf16x3") if base == "norman" else ("f32",)
    worst = compare_with_oracle(shape, 2, seed, compact=n_cells * 2 < shape.n_treatments, precisions=precisions)
    print(f"\n{shape.name}: worst relative L2 gradient error / loss error per precision: {worst}")
    if base == "norman":
        assert worst["bf16x3"] > worst["f32"]


def test_padding_genes_leave_no_trace():
    """d logits of the padding columns are exact zeros after a training step, and the gradient of log_concentration
    has exactly n_phenos entries (the float behind it in the arena is untouched)."""
    from cradle_vae_amd import synthetic as syn
    from cradle_vae_amd.trainer import TrainStep
    import bench

    shape = dataclasses.replace(syn.SHAPES["mid"], name="mid_g301", n_phenos=301)
    lm, batch = bench.build(shape, DEV, seed=3)
    st = TrainStep(lm, lr=3e-4, weight_decay=1e-6, gradient_clip_norm=100.0)
    f = st.fused
    off = f.arena.offsets["log_c"]
    f.arena.grad[off + 301: off + 304] = 7.0  # the arena pads segments to 64 floats: these belong to nobody
    out = f.run(compute_grads=True, X=batch["X"], D=batch["D"], D_obs_counts=batch["D_obs_counts"], qc=batch["qc"],
                qc_obs_counts=batch["qc_obs_counts"], cfX=batch["cfX"], library_size=batch["library_size"],
                n_particles=shape.n_particles)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out[:7]).all())
    d = f.intermediate("logits").view(shape.n_particles * shape.n_cells, 304)
    assert torch.equal(d[:, 301:], torch.zeros_like(d[:, 301:]))
    assert float(d[:, :301].abs().max()) > 0
    assert torch.equal(f.arena.grad[off + 301: off + 304], torch.full((3,), 7.0, device=DEV))


def test_widest_supported_gene_axis_and_one_beyond():
    """K1 holds a cell's genes in registers: 16384 is the widest row it takes (4 x 1024 threads x 4).  That width
    matches the oracle; one vector more is refused with a message, by the step as by K1's own entry point."""
    from cradle_vae_amd import hip
    from cradle_vae_amd import synthetic as syn

    shape = dataclasses.replace(syn.SHAPES["tiny"], name="tiny_g16384", n_cells=6, n_phenos=16384)
    worst = compare_with_oracle(shape, 2, 9, compact=False, precisions=("f32",))
    assert worst["f32_loss"] < 1e-5
    import bench

    lm, batch = bench.build(dataclasses.replace(shape, n_phenos=16388), DEV, seed=3)
    with pytest.raises(hip.CradleHipError, match="n_genes"):
        lm._fused_step().run(X=batch["X"], D=batch["D"], D_obs_counts=batch["D_obs_counts"], qc=batch["qc"],
                             qc_obs_counts=batch["qc_obs_counts"], cfX=batch["cfX"], library_size=batch["library_size"],
                             n_particles=shape.n_particles)
    torch.cuda.synchronize()


def test_three_genes_take_the_modular_path_and_say_so():
    """n_phenos < 4 is the one shape the fused step does not run (K1's rows are 16-byte vectors): step = 'auto' takes
    the modular HIP path -- same kernels, autograd between them, ~10x slower -- and warns once, with the reason;
    step = 'fused' raises with the same reason."""
    import warnings

    import bench
    from cradle_vae_amd import synthetic as syn

    shape = dataclasses.replace(syn.SHAPES["tiny"], name="tiny_g3", n_phenos=3)
    lm, batch = bench.build(shape, DEV, seed=3)
    lm.step = "auto"
    kw = dict(X=batch["X"], D=batch["D"], D_obs_counts=batch["D_obs_counts"], qc=batch["qc"],
              qc_obs_counts=batch["qc_obs_counts"], cfX=batch["cfX"],
              condition_values={"library_size": batch["library_size"]}, n_particles=shape.n_particles)
    with pytest.warns(UserWarning, match="n_phenos=3 < 4"):
        loss, _ = lm.loss(**kw)
    assert bool(torch.isfinite(loss))
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # ... once per module
        loss2, _ = lm.loss(**kw)
    assert bool(torch.isfinite(loss2))
    lm.step = "fused"
    with pytest.raises(NotImplementedError, match="n_phenos=3 < 4"):
        lm.loss(**kw)


def test_deeper_mlps_batch_their_weight_gradients():
    """Three hidden layers in every MLP (the shipped configurations have one in the decoder and the basal encoder): the
    hidden layers' weight gradients of each MLP then go out as batched GEMM launches (grid y = layer, csrc/gemm.hip
    GemmParams::batch) with K = P*N and 3*P*N rows and a batched split-K reduction; every gradient against the fp64
    oracle, at a mid and at a wider shape (both tile families)."""
    from cradle_vae_amd import synthetic as syn

    for base, n_cells, hidden, seed in (("mid", 64, 32, 11), ("mid", 256, 192, 12)):
        shape = dataclasses.replace(syn.SHAPES[base], name=f"{base}_deep{hidden}", n_cells=n_cells, decoder_n_layers=3,
                                    basal_encoder_n_layers=3, embedding_encoder_n_layers=5, decoder_n_hidden=hidden,
                                    basal_encoder_n_hidden=hidden, embedding_encoder_n_hidden=hidden)
        worst = compare_with_oracle(shape, 2, seed, compact=False, precisions=("f32",))
        print(f"\n{shape.name}: worst relative L2 gradient error / loss error: {worst}")
