"""Pin the oracle (oracle/fairlora_oracle.py) against golden vectors produced by
the imported reference (tests/golden/make_golden.py).  fp32, rtol 1e-5 (plus a
small atol for near-zero entries): the two sides run the same PyTorch CPU
kernels in a different op order."""
import json
import os

import numpy as np
import pytest
import torch

from fairfedmed_amd import config as C
from fairfedmed_amd import synth
from oracle import fairlora_oracle as O

from tests.golden.make_golden import LAYER_CASES, layer_inputs, rng_tensor, sub


def close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()))
    err = np.abs(a - b)
    bad = err > atol * scale + rtol * np.abs(b)
    assert not bad.any(), f"{what}: max err {err.max():.3e} (scale {scale:.3e}), {bad.sum()} bad"


@pytest.fixture(scope="module")
def unit(golden_dir):
    return np.load(os.path.join(golden_dir, "unit.npz"))


@pytest.fixture(scope="module")
def meta(golden_dir):
    return json.load(open(os.path.join(golden_dir, "meta.json")))


@pytest.mark.parametrize("case", LAYER_CASES, ids=[c[0] for c in LAYER_CASES])
def test_fairlora_layer(unit, case):
    name, L, Bn, fin, fout, r, G, S, hw = case
    x, g, W, bias, A, Sm, Bm, attr = layer_inputs(*case)
    scaling = 2.0 / r
    if hw:
        xin = x.permute(1, 2, 0).reshape(Bn, fin, hw[0], hw[1])
        y = O.fairlora_linear(xin, W.reshape(fout, fin, 1, 1), None, A, Sm, Bm, attr, scaling)
        y = y.reshape(Bn, fout, -1).permute(2, 0, 1)
    else:
        y = O.fairlora_linear(x, W, bias, A, Sm, Bm, attr, scaling)
    dx, dA, dS, dB = O.fairlora_backward(x, g, W, A, Sm, Bm, attr, scaling)
    gy = unit[f"layer.{name}.y"]
    close(y.numpy() if y.numel() <= 65536 else sub(y), gy, rtol=2e-5, atol=2e-6, what="y")
    gdx = unit[f"layer.{name}.dx"]
    close(dx.numpy() if dx.numel() <= 65536 else sub(dx), gdx, rtol=2e-5, atol=2e-6, what="dx")
    close(dA.numpy(), unit[f"layer.{name}.dA"], rtol=2e-5, atol=2e-6, what="dA")
    close(dS.numpy(), unit[f"layer.{name}.dS"], rtol=2e-5, atol=2e-6, what="dS")
    close(dB.numpy(), unit[f"layer.{name}.dB"], rtol=2e-5, atol=2e-6, what="dB")


def test_fairlora_weight_and_global_s(unit):
    """FairLoRALinear.weight(x, attr) (trainers/GLP_OT_SVLoRA.py:425-445; plain one-hot mix) and GLOBAL_S (one more
    vector added to every sample's singular values, :359-363, 418-422, 467-468): the oracle's restatement AND the
    product's module-level weight() (plain tensor algebra, runs on the CPU) against the imported reference layer."""
    import torch.nn as nn
    from fairfedmed_amd.model import FairLoRALinear
    case = LAYER_CASES[0]
    name, L, Bn, fin, fout, r, G, S, hw = case
    x, g, W, bias, A, Sm, Bm, attr = layer_inputs(*case)
    sc = 2.0 / r
    close(O.fairlora_dense_weight(W, A, Sm, Bm, attr, sc, Bn).numpy(), unit[f"layer.{name}.weight_attr"],
          rtol=2e-5, atol=2e-6, what="weight(attr)")
    close(O.fairlora_dense_weight(W, A, Sm, Bm, None, sc, Bn).numpy(), unit[f"layer.{name}.weight_noattr"],
          rtol=2e-5, atol=2e-6, what="weight(None)")
    # GLOBAL_S
    np.testing.assert_allclose(unit[f"layer.{name}.gs.sg_init"], np.linspace(1, 0.1, r, dtype=np.float32), atol=1e-7)
    Sg = torch.from_numpy(unit[f"layer.{name}.gs.Sg"])
    leaves = [t.clone().requires_grad_(True) for t in (x, A, Sm, Bm, Sg)]
    y = O.fairlora_linear(leaves[0], W, bias, leaves[1], leaves[2], leaves[3], attr, sc, S_global=leaves[4])
    y.backward(g)
    close(y.detach().numpy(), unit[f"layer.{name}.gs.y"], rtol=2e-5, atol=2e-6, what="gs.y")
    for t, nm in zip(leaves, ("dx", "dA", "dS", "dB", "dS_global")):
        close(t.grad.numpy(), unit[f"layer.{name}.gs.{nm}"], rtol=2e-5, atol=2e-6, what="gs." + nm)
    close(O.fairlora_dense_weight(W, A, Sm, Bm, attr, sc, Bn, S_global=Sg).numpy(), unit[f"layer.{name}.gs.weight_attr"],
          rtol=2e-5, atol=2e-6, what="gs.weight")
    # the product's module: constructor surface, initial values and weight()
    lin = nn.Linear(fin, fout)
    lin.weight.data, lin.bias.data = W.clone(), bias.clone()
    for gs in (False, True):
        layer = FairLoRALinear(lin, rank=r, alpha=2.0, global_s=gs, num_attrs=G)
        names = [n for n, _ in layer.named_parameters()]
        assert names == ["original_linear.weight", "original_linear.bias", "lora_A.weight", "lora_S.weight"] + \
            (["lora_S_global.weight"] if gs else []) + ["lora_B.weight"]
        if gs:
            assert tuple(layer.lora_S_global.weight.shape) == (r,)
            np.testing.assert_allclose(layer.lora_S_global.weight.detach().numpy(), unit[f"layer.{name}.gs.sg_init"], atol=1e-7)
            layer.lora_S_global.weight.data = Sg.clone()
        layer.lora_A.weight.data, layer.lora_S.weight.data, layer.lora_B.weight.data = A.clone(), Sm.clone(), Bm.clone()
        key = f"layer.{name}.gs.weight_attr" if gs else f"layer.{name}.weight_attr"
        close(layer.weight(x, attr).detach().numpy(), unit[key], rtol=2e-5, atol=2e-6, what="module weight()")
        if not gs:
            close(layer.weight(x, None).detach().numpy(), unit[f"layer.{name}.weight_noattr"], rtol=2e-5, atol=2e-6)
        assert layer.bias() is lin.bias


def test_lora_plain_layer(unit):
    """LoRALinear (plain LoRA) == the FairLoRA product with one group and s = 1; weight() is W + scaling (A B)^T."""
    L, Bn, fin, fout, r = 50, 4, 128, 192, 8
    x, g = rng_tensor("lora_plain.x", (L, Bn, fin)), rng_tensor("lora_plain.g", (L, Bn, fout))
    W = rng_tensor("lora_plain.W", (fout, fin)) * fin ** -0.5
    bias = rng_tensor("lora_plain.b", (fout,)) * 0.1
    A, Bm = rng_tensor("lora_plain.A", (fin, r)) * 0.1, rng_tensor("lora_plain.B", (r, fout))
    ones = torch.ones(1, r)
    y = O.fairlora_linear(x, W, bias, A, ones, Bm, None, 2.0 / r)
    dx, dA, dS, dB = O.fairlora_backward(x, g, W, A, ones, Bm, None, 2.0 / r)
    close(y.numpy(), unit["lora_plain.y"], rtol=2e-5, atol=2e-6, what="y")
    close(dx.numpy(), unit["lora_plain.dx"], rtol=2e-5, atol=2e-6, what="dx")
    close(dA.numpy(), unit["lora_plain.dA"], rtol=2e-5, atol=2e-6, what="dA")
    close(dB.numpy(), unit["lora_plain.dB"], rtol=2e-5, atol=2e-6, what="dB")
    close((W + (2.0 / r) * (A @ Bm).t()).numpy(), unit["lora_plain.weight"], rtol=1e-6, atol=1e-7, what="weight()")


def test_fairlora_backward_matches_autograd():
    case = LAYER_CASES[0]
    name, L, Bn, fin, fout, r, G, S, hw = case
    x, g, W, bias, A, Sm, Bm, attr = layer_inputs(*case)
    xs = [t.double().requires_grad_(True) for t in (x, A, Sm, Bm)]
    y = O.fairlora_linear(xs[0], W.double(), bias.double(), xs[1], xs[2], xs[3], attr, 0.5)
    y.backward(g.double())
    dx, dA, dS, dB = O.fairlora_backward(x.double(), g.double(), W.double(), A.double(), Sm.double(),
                                         Bm.double(), attr, 0.5)
    for got, ref in zip((dx, dA, dS, dB), (t.grad for t in xs)):
        assert torch.allclose(got, ref, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("r", [4, 8, 12, 16, 32])
@pytest.mark.parametrize("G", [2, 3])
def test_lora_s_init(unit, r, G):
    close(synth.lora_s_init(r, G).numpy(), unit[f"s_init.r{r}.g{G}"], rtol=0, atol=1e-7)


def test_lora_s_init_known_rows():
    # SURVEY.md §8(c) (ii)
    s = synth.lora_s_init(8, 3)
    np.testing.assert_allclose(s[0].numpy(), [.5, .3667, .2333, .1, .1, .0733, .0467, .02], atol=1e-4)
    np.testing.assert_allclose(s[1, 4:].numpy(), [.0733, .0467, .02, .1], atol=1e-4)
    s4 = synth.lora_s_init(4, 3)
    for g in range(3):
        np.testing.assert_allclose(s4[g].numpy(), [.5, .1, .1, .02], atol=1e-6)


@pytest.mark.parametrize("case", ["e0_shared", "e3_shared", "e3_plain"])
def test_fedavg_ema(unit, meta, case):
    m = meta[f"fed.{case}"]
    keys = {"a.lora_S.weight": (3, 8), "a.lora_A.weight": (16, 8), "prompt_learner.ctx": (2, 4, 8),
            "frozen.weight": (5, 5), "b.lora_S.weight": (3, 8)}
    w = {u: {k: rng_tensor(f"fed.{case}.{u}.{k}", s) for k, s in keys.items()} for u in range(3)}
    w_g = {k: rng_tensor(f"fed.{case}.g.{k}", s) for k, s in keys.items()}
    res = O.average_weights_ema(w_g, w, m["idxs"], m["n_client"], m["by_attr"], m["epoch"], m["max_epoch"],
                                shared_half_s=m["shared_half_s"])
    for k in keys:
        close(res[k].numpy(), unit[f"fed.{case}.{k}"], rtol=1e-6, atol=1e-7, what=k)


@pytest.mark.parametrize("name", ["n32", "n200"])
def test_auc(unit, meta, name):
    got = O.auc_binary(unit[f"auc.{name}.prob"], unit[f"auc.{name}.y"])
    assert abs(got - meta["auc"][name]) < 1e-12


TINY = {
    "tiny_r4": (C.vit_tiny(rank=4), 8, "random"),
    "tiny_r8g2": (C.vit_tiny(rank=8, num_groups=2), 6, "random"),
    "tiny_refinit": (C.vit_tiny(rank=4), 8, "reference"),
    "tiny3d_r4": (C.vit_tiny_3d(rank=4, dim_per_3d_slice=4), 6, "random"),      # 3D OCT front end (§8 a4)
    "rn_tiny_r4g2": (C.rn_tiny(rank=4, num_groups=2), 6, "random"),             # RN50 trunk (§8 a12)
    "rn_tiny2_r4g2": (C.rn_tiny2(rank=4, num_groups=2), 6, "random"),           # ... with identity-skip Bottlenecks
    "tiny_sched": (C.vit_tiny(rank=4), 8, "random"),                            # 2 epochs x 2 batches, StepLR(2)
    # the other adapter types of apply_lora_to_model (trainers/GLP_OT_SVLoRA.py:516-540) and GLOBAL_S
    "tiny_globals": (C.vit_tiny_lora("FairLoRA", True), 8, "random"),
    "tiny_svlora": (C.vit_tiny_lora("SVLoRA", False), 8, "random"),
    "tiny_svlora_globals": (C.vit_tiny_lora("SVLoRA", True), 8, "random"),
    "tiny_lora": (C.vit_tiny_lora("LoRA", False), 8, "random"),
}


@pytest.mark.parametrize("tag", list(TINY))
def test_tiny_model_step_and_trajectory(golden_dir, meta, tag):
    mcfg, bs, init = TINY[tag]
    gold = np.load(os.path.join(golden_dir, "rn_tiny.npz" if tag.startswith("rn") else
                                "tiny3d.npz" if mcfg.dim_per_3d_slice else "tiny.npz"))
    sd = synth.make_state_dict(mcfg, seed=1, lora_init=init)
    batch = synth.make_batch(mcfg, bs, seed=1234)
    keys = synth.trainable_keys(mcfg)
    assert sum(sd[k].numel() for k in keys) == meta[f"{tag}.trainable_elems"]
    loss, logits, grads = O.loss_and_grads(sd, batch, mcfg, keys)
    close(logits.numpy(), gold[f"{tag}.logits"], rtol=1e-5, atol=1e-6, what="logits")
    assert abs(float(loss) - meta[f"{tag}.loss0"]) <= 1e-5 * abs(meta[f"{tag}.loss0"])
    for k in keys:
        close(grads[k].numpy(), gold[f"{tag}.grad.{k}"], rtol=1e-4, atol=2e-6, what=k)
    # the reference's run wiring: ONE optimizer / scheduler registered under two names, so two optimizer steps per
    # batch and two scheduler steps per epoch end (the goldens are generated with that wiring)
    sc = meta[f"{tag}.sched"]
    assert sc["optimizer_steps_per_batch"] == 2
    opt = O.SgdState()
    sched = O.StepLRState(opt, sc["step_size"])
    for i, ref in enumerate(meta[f"{tag}.traj"]):
        s, _, _ = O.train_step(sd, opt, batch, mcfg, keys, sched=sched, last_batch=(i + 1) % sc["num_batches"] == 0)
        assert abs(s["loss"] - ref["loss"]) <= 1e-5 * abs(ref["loss"]), (s, ref)
        assert abs(s["acc"] - ref["acc"]) < 1e-4
        assert abs(s["auc"] - ref["auc"]) < 1e-9
        assert abs(opt.lr - ref["lr_after"]) <= 1e-12 * ref["lr_after"]
    assert sched.last_epoch == sc["last_epoch"]
    # ResNet trunks: train-mode BatchNorm over 6 samples amplifies fp32 summation-order noise step over step (the
    # deeper rn_tiny2 ends 5e-6 away after its 6 optimizer steps; losses stay within 1e-5)
    post_tol = dict(rtol=1e-4, atol=2e-5) if tag.startswith("rn") else dict(rtol=1e-5, atol=1e-6)
    for k in keys:
        close(sd[k].numpy(), gold[f"{tag}.post.{k}"], what="post." + k, **post_tol)
    for k in synth.buffer_keys(mcfg):                               # BatchNorm running statistics after the steps
        close(sd[k].numpy(), gold[f"{tag}.post.{k}"], rtol=1e-5, atol=1e-6, what="post." + k)


@pytest.mark.parametrize("tag,bs", [("vitb_r8", 8), ("vitb_r8_bs32", 32)])
def test_vitb_step(golden_dir, meta, tag, bs):
    path = os.path.join(golden_dir, "vitb.npz")
    if not os.path.exists(path):
        pytest.skip("vitb.npz not generated")
    gold = np.load(path)
    mcfg = C.vit_b16(rank=8)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    keys = synth.trainable_keys(mcfg)
    assert sum(sd[k].numel() for k in keys) == meta[f"{tag}.trainable_elems"] == 741952
    assert sum(v.numel() for k, v in sd.items() if "token_" not in k) == meta[f"{tag}.total_params"] == 125065793
    assert len(keys) == meta[f"{tag}.trainable_tensors"] == 73
    batch = synth.make_batch(mcfg, bs, seed=1234)
    loss, logits, grads = O.loss_and_grads(sd, batch, mcfg, keys)
    close(logits.numpy(), gold[f"{tag}.logits"], rtol=1e-4, atol=1e-5, what="logits")
    assert abs(float(loss) - meta[f"{tag}.loss0"]) <= 1e-5 * abs(meta[f"{tag}.loss0"])
    for k in keys:
        n = float(grads[k].norm())
        ref = meta[f"{tag}.grad_norms"][k]
        assert abs(n - ref) <= 1e-3 * ref + 1e-9, (k, n, ref)


@pytest.mark.parametrize("ot,top", [("Sinkhorn", 1.0), ("COT", 0.8)])
def test_ot_heads_step_and_trajectory(golden_dir, ot, top):
    """The Sinkhorn / COT logits heads (trainers/GLP_OT_SVLoRA.py:615-675, 713-757): the
    oracle's restatement against the imported reference's logits, loss, gradients and 3-step trajectory."""
    import dataclasses
    import json
    gold = np.load(os.path.join(golden_dir, "ot.npz"))
    meta = json.load(open(os.path.join(golden_dir, "ot.json")))
    tag = f"ot_{ot.lower()}"
    mcfg = dataclasses.replace(C.vit_tiny(rank=4), ot=ot, ot_top_percent=top)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    batch = synth.make_batch(mcfg, 8, seed=1234)
    keys = synth.trainable_keys(mcfg)
    loss, logits, grads = O.loss_and_grads(sd, batch, mcfg, keys)
    close(logits.numpy(), gold[f"{tag}.logits"], rtol=1e-5, atol=1e-6, what="logits")
    assert abs(float(loss) - meta[f"{tag}.loss0"]) <= 1e-5 * abs(meta[f"{tag}.loss0"])
    for k in keys:
        close(grads[k].numpy(), gold[f"{tag}.grad.{k}"], rtol=1e-4, atol=2e-6, what=k)
    opt = O.SgdState()
    for ref in meta[f"{tag}.traj"]:
        s, _, _ = O.train_step(sd, opt, batch, mcfg, keys)
        assert abs(s["loss"] - ref["loss"]) <= 1e-5 * abs(ref["loss"]), (s, ref)
