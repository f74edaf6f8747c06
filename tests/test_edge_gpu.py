"""Edge cases of the hot path on the GPU: the ranks and group counts the reference's scripts actually use (rank 12,
the 'language' / 'ethnicity' attributes), ranks beyond the fused rank-16 path, maximum rank and group count, one
image, error conventions of the C ABI and of the trainer (SURVEY.md §8(b) 'Error conventions')."""
import ctypes as C_
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from fairfedmed_amd import config as C
from fairfedmed_amd import synth

pytestmark = pytest.mark.gpu


def cos(got, ref):
    got = torch.as_tensor(got).double().cpu().flatten()
    ref = torch.as_tensor(ref).double().cpu().flatten()
    return float(torch.dot(got, ref) / (got.norm() * ref.norm()).clamp_min(1e-300))


def rel(got, ref):
    got = torch.as_tensor(got).double().cpu()
    ref = torch.as_tensor(ref).double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("rank,G,bs", [(12, 3, 5),      # scripts/fairfedlora_fairfedmed.sh: LoRA_RANK=12, language (3 groups)
                                       (12, 2, 3),      # ethnicity / gender: 2 groups
                                       (16, 3, 4),      # BASELINE config 4 (3D OCT run): rank 16, last fused rank
                                       (24, 3, 4),      # beyond 16: stand-alone down projection + VALU kernels
                                       (32, 8, 9),      # FFM_MAX_RANK, FFM_MAX_GROUPS
                                       (2, 2, 1)])      # smallest even rank, one image
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_step_vs_oracle_over_ranks_and_groups(rank, G, bs, dtype):
    from oracle import fairlora_oracle as O
    from fairfedmed_amd.engine import FairLoRAEngine
    mcfg = C.vit_tiny(rank=rank, num_groups=G)
    sd = synth.make_state_dict(mcfg, seed=rank, lora_init="random")
    batch = synth.make_batch(mcfg, bs, seed=77 + rank)
    keys = synth.trainable_keys(mcfg)
    eng = FairLoRAEngine(mcfg, sd, dtype=dtype, max_images=bs)
    out = eng.forward_backward(batch["img"].cuda(), batch["attrs"].t()[0].cuda(), batch["label"].cuda())
    loss, logits, grads = O.loss_and_grads(sd, batch, mcfg, keys)
    f32 = dtype == torch.float32
    assert int(out["finite"]) == 1
    assert rel(out["logits"], logits) < (2e-5 if f32 else 3e-2)
    assert abs(float(out["loss"]) - float(loss)) <= (2e-5 if f32 else 1e-2) * abs(float(loss))
    for k in keys:
        g, ref = eng.params.view(k, "grad"), grads[k]
        if float(ref.abs().max()) == 0.0:
            assert float(g.abs().max()) < 1e-12, k
        elif f32:
            assert rel(g, ref) < 2e-3, (k, rel(g, ref))
        else:
            assert cos(g, ref) > 0.985, (k, cos(g, ref))


def test_reference_initialisation_of_the_scripts_rank():
    """'same+cycle' singular values at r = 12, G = 3 (trainers/GLP_OT_SVLoRA.py:402-417): A = 0, so the first step has
    dS = dB = 0 exactly and dA != 0."""
    from fairfedmed_amd.engine import FairLoRAEngine
    mcfg = C.vit_tiny(rank=12, num_groups=3)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
    batch = synth.make_batch(mcfg, 4, seed=5)
    eng = FairLoRAEngine(mcfg, sd, dtype=torch.float32, max_images=4)
    eng.forward_backward(batch["img"].cuda(), batch["attrs"].t()[0].cuda(), batch["label"].cuda())
    for k in synth.trainable_keys(mcfg):
        g = eng.params.view(k, "grad")
        if k.endswith(("lora_S.weight", "lora_B.weight")):
            assert float(g.abs().max()) == 0.0, k
        elif k.endswith("lora_A.weight"):
            assert float(g.abs().max()) > 0.0, k


def test_engine_argument_errors():
    from fairfedmed_amd.engine import FairLoRAEngine
    mcfg = C.vit_tiny(rank=4)
    eng = FairLoRAEngine(mcfg, synth.make_state_dict(mcfg, seed=1), dtype=torch.float32, max_images=4)
    b = synth.make_batch(mcfg, 5, seed=1)
    with pytest.raises(ValueError):
        eng.forward(b["img"].cuda(), None)                            # more images than max_images
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(2, 3, 32, 32, device="cuda"), None)   # wrong resolution
    with pytest.raises(TypeError):
        eng.forward(torch.zeros(2, 3, 64, 64, device="cuda", dtype=torch.float64), None)
    with pytest.raises(TypeError):
        eng.forward(b["img"][:2], None)                               # CPU tensor: there is no CPU path


def test_c_abi_rejects_bad_arguments():
    """Status codes of include/ffm_hip.h: FFM_EINVAL (-1) for bad shapes / alignment / NULL, nothing is launched."""
    from fairfedmed_amd import _lib as L
    lib = L.load()
    x = torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16)
    st = L.stream_ptr()
    a = L.GemmArgs(L.ptr(x), L.ptr(x), L.ptr(x), 64, 64, 32, 64, 64, 64, 0, 0, None, None, None, None, None, None,
                   None, None, None, None, None, None, None, 0, 0, 0.0, 0.0, None)
    assert lib.ffm_gemm_nt(C_.byref(a), L.dtype_code(torch.bfloat16), st) == -1      # K * 2 B is not a multiple of 128 B
    a.K, a.flags = 64, L.EPI_BIAS
    assert lib.ffm_gemm_nt(C_.byref(a), L.dtype_code(torch.bfloat16), st) == -1      # bias flag without a bias pointer
    a.flags = 0
    assert lib.ffm_gemm_nt(C_.byref(a), 7, st) == -1                                  # unknown dtype code
    assert lib.ffm_gemm_nt(C_.byref(a), L.dtype_code(torch.bfloat16), st) == 0
    f = torch.zeros(64, device="cuda")
    assert lib.ffm_layernorm_fwd(None, L.ptr(x), L.ptr(f), L.ptr(f), None, None, 64, 64, 1, st) == -1
    assert lib.ffm_sgd_momentum(L.ptr(f), L.ptr(f), None, 64, C_.c_float(0.1), C_.c_float(0.9), C_.c_float(0.0), 1, st) == -1
    assert lib.ffm_eval_counts(L.ptr(f), L.ptr(f), None, 0, 2, L.ptr(f), st) == -1    # N = 0
    assert lib.ffm_expand_u8(L.ptr(f), L.ptr(f), 1, 1, 6, 1, st) == -1                # HW % 4 != 0
    assert lib.ffm_bn_fwd(L.ptr(x), L.ptr(f), L.ptr(f), L.ptr(f), L.ptr(f), L.ptr(f), L.ptr(f), None, 0, None, L.ptr(x), 64, 64,
                          1, 0, 1, st) == -1                                           # training without scratch
    torch.cuda.synchronize()


def test_trainer_error_conventions_and_single_class_batch():
    """Dassl/dassl/engine/trainer.py:260-262: a non-finite loss raises FloatingPointError; a single-class batch reports
    auc = 1 (trainers/GLP_OT_SVLoRA.py:965-967); unknown LoRA types / datasets raise NotImplementedError."""
    from tests.test_trainer_gpu import make_cfg
    from fairfedmed_amd.trainer import GLP_OT_SVLoRA, SyntheticFedData
    mcfg = C.vit_tiny(rank=4)
    cfg = make_cfg(prec="fp32")
    tr = GLP_OT_SVLoRA(cfg, data=SyntheticFedData(mcfg, 1, 2, 1, 8))
    tr.num_batches, tr.batch_idx = 10, 0
    batch = synth.make_batch(mcfg, 8, seed=3)
    batch["label"] = torch.ones_like(batch["label"])
    s = tr.forward_backward(batch)
    assert s["auc"] == 1.0 and np.isfinite(s["loss"])
    bad = dict(batch)
    bad["img"] = batch["img"].clone()
    bad["img"][0, 0, 0, 0] = float("nan")
    # the summary's values stay on the GPU until read (no host sync inside the step): the non-finite loss raises when
    # the summary is read ...
    lazy = tr.forward_backward(bad)
    with pytest.raises(FloatingPointError, match="Loss is infinite or NaN!"):
        lazy["loss"]
    # ... or at the latest when the local epoch ends ...
    tr2 = GLP_OT_SVLoRA(make_cfg(prec="fp32"), data=SyntheticFedData(mcfg, 1, 2, 1, 8))
    tr2.fed_train_loader_x_dict[0].dataset.batches[1] = bad
    with pytest.raises(FloatingPointError, match="Loss is infinite or NaN!"):
        tr2.run_epoch(idx=0)
    # ... and inside the call, as in the reference, with TRAIN.SYNC_EVERY_STEP
    cfg_s = make_cfg(prec="fp32")
    cfg_s.TRAIN.SYNC_EVERY_STEP = True
    tr3 = GLP_OT_SVLoRA(cfg_s, data=SyntheticFedData(mcfg, 1, 2, 1, 8))
    tr3.num_batches, tr3.batch_idx = 10, 0
    with pytest.raises(FloatingPointError, match="Loss is infinite or NaN!"):
        tr3.forward_backward(bad)
    # the device summary equals the host (numpy / sklearn-equivalent) metrics
    cfg_h = make_cfg(prec="fp32")
    cfg_h.TRAIN.HOST_METRICS = True
    trh = GLP_OT_SVLoRA(cfg_h, data=SyntheticFedData(mcfg, 1, 2, 1, 8))
    trd = GLP_OT_SVLoRA(make_cfg(prec="fp32"), data=SyntheticFedData(mcfg, 1, 2, 1, 8))
    good = synth.make_batch(mcfg, 8, seed=11, signal=0.3)
    for t in (trh, trd):
        t.num_batches, t.batch_idx = 10, 0
    sh, sd_ = trh.forward_backward(good), trd.forward_backward(good)
    assert isinstance(sh, dict) and not isinstance(sd_, dict) and set(sd_) == {"loss", "acc", "auc"}
    assert abs(sh["loss"] - sd_["loss"]) < 1e-9 and abs(sh["acc"] - sd_["acc"]) < 1e-9 and abs(sh["auc"] - sd_["auc"]) < 1e-12
    cfg2 = make_cfg()
    cfg2.TRAINER.GLP_OT_LORA.TYPE = "DoRA"                            # trainers/GLP_OT_SVLoRA.py:533-534
    with pytest.raises(NotImplementedError):
        GLP_OT_SVLoRA(cfg2, data=SyntheticFedData(mcfg, 1, 1, 1, 8))
    cfg3 = make_cfg()
    cfg3.DATASET.NAME = "ImageNet"
    with pytest.raises(NotImplementedError):
        GLP_OT_SVLoRA(cfg3, data=SyntheticFedData(mcfg, 1, 1, 1, 8))


def test_lambda_fairness_changes_the_reported_loss_only():
    """TRAINER.LAMBDA_FAIRNESS != 0 (trainers/GLP_OT_SVLoRA.py:930-948): the confidence-gap term is
    built from detached values, so it moves the reported loss and never the gradients."""
    from oracle import fairlora_oracle as O
    from tests.test_trainer_gpu import make_cfg
    from fairfedmed_amd.trainer import GLP_OT_SVLoRA, SyntheticFedData
    mcfg = C.vit_tiny(rank=4)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    batch = synth.make_batch(mcfg, 8, seed=11)
    keys = synth.trainable_keys(mcfg)
    out = {}
    for lam in (0.0, 0.5):
        cfg = make_cfg(prec="fp32")
        cfg.TRAINER.LAMBDA_FAIRNESS = lam
        cfg.MODEL.STATE_DICT = sd
        tr = GLP_OT_SVLoRA(cfg, data=SyntheticFedData(mcfg, 1, 1, 1, 8))
        tr.num_batches, tr.batch_idx = 10, 0
        s = tr.forward_backward(batch)
        ref_loss, _, _ = O.loss_and_grads(sd, batch, mcfg, keys, lambda_fairness=lam)
        assert abs(s["loss"] - float(ref_loss)) <= 2e-5 * abs(float(ref_loss)), (lam, s["loss"], float(ref_loss))
        out[lam] = (s["loss"], tr.engine.params.grad.clone())
    assert out[0.5][0] > out[0.0][0] and torch.equal(out[0.0][1], out[0.5][1])


def test_disable_attr_runs_one_group_without_attributes():
    """--disable_attr (TRAINER.GLP_OT_LORA.DISABLE_ATTR; trainers/GLP_OT_SVLoRA.py:841, 462): one group, no attribute is
    handed to the model; loss and gradients equal the oracle's with num_groups = 1 and attr = None."""
    from oracle import fairlora_oracle as O
    from tests.test_trainer_gpu import make_cfg
    from fairfedmed_amd.trainer import GLP_OT_SVLoRA, SyntheticFedData
    mcfg = C.vit_tiny(rank=4, num_groups=1)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    batch = synth.make_batch(C.vit_tiny(rank=4), 8, seed=11)          # attrs in 0..2 are present but must be ignored
    cfg = make_cfg(prec="fp32")
    cfg.TRAINER.GLP_OT_LORA.DISABLE_ATTR = True
    cfg.MODEL.STATE_DICT = sd
    tr = GLP_OT_SVLoRA(cfg, data=SyntheticFedData(C.vit_tiny(rank=4), 1, 1, 1, 8))
    assert tr.engine.cfg.lora.num_groups == 1
    tr.num_batches, tr.batch_idx = 10, 0
    p0 = tr.engine.params.flat.clone()
    s = tr.forward_backward(batch)
    keys = synth.trainable_keys(mcfg)
    ref_loss, _, grads = O.loss_and_grads(sd, {"img": batch["img"], "label": batch["label"], "attrs": None}, mcfg, keys)
    assert abs(s["loss"] - float(ref_loss)) <= 2e-5 * abs(float(ref_loss))
    for k in keys:
        g = tr.engine.params.view(k, "grad")
        if float(grads[k].abs().max()) > 0:
            assert rel(g, grads[k]) < 2e-3, k
    assert not torch.equal(p0, tr.engine.params.flat)                  # the SGD step ran


def test_prec_amp_trains_without_the_attribute():
    """TRAINER.GLP_OT.PREC = 'amp' (trainers/GLP_OT_SVLoRA.py:890-898): the training step calls the model without the
    attribute, i.e. with the uniform 1/G group mix; inference still passes it."""
    from oracle import fairlora_oracle as O
    from tests.test_trainer_gpu import make_cfg
    from fairfedmed_amd.trainer import GLP_OT_SVLoRA, SyntheticFedData
    mcfg = C.vit_tiny(rank=4)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    batch = synth.make_batch(mcfg, 8, seed=11)
    cfg = make_cfg(prec="amp")
    cfg.MODEL.STATE_DICT = sd
    tr = GLP_OT_SVLoRA(cfg, data=SyntheticFedData(mcfg, 1, 1, 1, 8))
    tr.num_batches, tr.batch_idx = 10, 0
    s = tr.forward_backward(batch)
    ref_loss, _, _ = O.loss_and_grads(sd, {"img": batch["img"], "label": batch["label"], "attrs": None}, mcfg,
                                      synth.trainable_keys(mcfg))
    with_attr, _, _ = O.loss_and_grads(sd, batch, mcfg, synth.trainable_keys(mcfg))
    assert abs(s["loss"] - float(ref_loss)) <= 2e-5 * abs(float(ref_loss))
    assert abs(float(ref_loss) - float(with_attr)) > 1e-6             # the two mixes do differ on this batch


def test_fp16_gradient_scale_and_overflow_guard():
    """IEEE-half mode: the backward pass runs scaled (x 2^12 on dloss/dlogits, out again on the fp32 gradient buffer:
    ffm_loss_scale / ffm_unscale_check).  (1) the scale leaves no trace in the result: 2^12 and 2^8 give the same gradients
    to rounding, and no scale at all is visibly worse against the oracle on the smallest gradients (half's subnormals) -
    never better; (2) an absurd scale overflows half: the step is marked, its SGD update is SKIPPED (weights and momentum
    bit-identical to before), the scale halves, and the loss flag stays clean - training goes on (round 6; the reference
    has no scaler: Dassl/dassl/engine/trainer.py:339-342)."""
    from fairfedmed_amd.engine import FairLoRAEngine
    from oracle import fairlora_oracle as O
    mcfg = C.vit_tiny(rank=4)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    batch = synth.make_batch(mcfg, 8, seed=1234)
    keys = synth.trainable_keys(mcfg)
    img, attr, label = batch["img"].cuda(), batch["attrs"].t()[0].contiguous().cuda(), batch["label"].cuda()
    _, _, grads = O.loss_and_grads(sd, batch, mcfg, keys)

    def run(scale):
        eng = FairLoRAEngine(mcfg, sd, dtype=torch.float16, max_images=8)
        eng.grad_scale = float(scale)
        out = eng.forward_backward(img, attr, label)
        torch.cuda.synchronize()
        return eng, int(out["finite"]), {k: eng.params.view(k, "grad").clone() for k in keys}

    _, f12, g12 = run(4096.0)
    _, f8, g8 = run(256.0)
    _, f0, g0 = run(1.0)
    assert f12 == 1 and f8 == 1 and f0 == 1
    worst = {1.0: 0.0, 256.0: 0.0, 4096.0: 0.0}
    for k in keys:
        ref = grads[k]
        if float(ref.abs().max()) == 0.0:
            continue
        assert rel(g12[k], g8[k]) < 2e-3, (k, rel(g12[k], g8[k]))
        for s, g in ((1.0, g0), (256.0, g8), (4096.0, g12)):
            worst[s] = max(worst[s], rel(g[k], ref))
    print("fp16 worst gradient error vs the oracle by scale:", worst)
    assert worst[4096.0] <= worst[1.0] * 1.05 + 1e-6 and worst[4096.0] < 1e-2
    eng, fbad, gbad = run(2.0 ** 40)
    assert fbad == 1, "the loss itself is finite: only the gradient overflowed"
    assert not all(bool(torch.isfinite(g).all()) for g in gbad.values())
    assert float(eng.scale_state[2]) == 0.0, "an overflowing gradient scale must mark the step"
    before = (eng.params.flat.clone(), eng.params.momentum.clone())
    eng.sgd_step(1e-2, 0.9, 5e-4, repeats=2)
    torch.cuda.synchronize()
    assert torch.equal(eng.params.flat, before[0]) and torch.equal(eng.params.momentum, before[1]), "the overflowed step must be skipped"
    assert eng.grad_scale == 2.0 ** 39 and eng.overflow_steps() == 1


def test_fp16_overflow_recovers_by_halving_the_device_resident_scale():
    """An injected overflow (scale 2^30 on a tower whose gradients then leave half's range) is skipped step after step, the
    scale halving each time from DEVICE memory - the recorded launch plan is never re-recorded - until the backward pass
    fits; from there on training moves the weights and the trajectory equals that of an engine that started on the scale
    the first one arrived at."""
    from fairfedmed_amd.engine import FairLoRAEngine
    mcfg = C.vit_tiny(rank=4)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    batch = synth.make_batch(mcfg, 8, seed=1234)
    img, attr, label = batch["img"].cuda(), batch["attrs"].t()[0].contiguous().cuda(), batch["label"].cuda()
    a = FairLoRAEngine(mcfg, sd, dtype=torch.float16, max_images=8)
    a.grad_scale = 2.0 ** 30
    start = a.params.flat.clone()
    plans, skipped = None, 0
    for step in range(40):
        out = a.forward_backward(img, attr, label)
        plans = plans or dict(a.step_plans)
        a.sgd_step(1e-2, 0.9, 5e-4, repeats=2)
        assert int(out["finite"]) == 1
        if torch.equal(a.params.flat, start):
            skipped += 1
        else:
            break
    assert 0 < skipped < 40 and a.overflow_steps() == skipped, (skipped, a.overflow_steps())
    assert a.step_plans == plans, "the plan recorded at the first step is still the one replayed"
    assert a.grad_scale == 2.0 ** (30 - skipped)
    b = FairLoRAEngine(mcfg, sd, dtype=torch.float16, max_images=8)
    b.grad_scale = a.grad_scale
    b.forward_backward(img, attr, label)
    b.sgd_step(1e-2, 0.9, 5e-4, repeats=2)
    torch.cuda.synchronize()
    assert b.overflow_steps() == 0 and torch.equal(a.params.flat, b.params.flat)
    with pytest.raises(ValueError):
        a.grad_scale = 0.0


def test_fp16_trainer_keeps_training_through_an_injected_overflow():
    """Through the trainer: one absurd scale in the middle of a local epoch costs the skipped steps and nothing else - no
    FloatingPointError, finite weights at the end, the loss flag untouched."""
    from fairfedmed_amd.registry import build_trainer
    from fairfedmed_amd.trainer import SyntheticFedData
    import fairfedmed_amd.trainer  # noqa: F401
    from tests.test_trainer_gpu import make_cfg
    mcfg = C.vit_tiny(rank=4)
    cfg = make_cfg(prec="fp16", bs=8)
    cfg.DATA = SyntheticFedData(mcfg, 1, train_batches=6, test_batches=1, batch_size=8)
    cfg.MODEL.STATE_DICT = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    tr = build_trainer(cfg)
    tr.engine.grad_scale = 2.0 ** 30
    tr.train(idx=0, global_epoch=0, is_fed=True, is_last_client=True)
    torch.cuda.synchronize()
    assert tr.engine.overflow_steps() >= 1 and bool(torch.isfinite(tr.engine.params.flat).all())
    assert tr.engine.grad_scale < 2.0 ** 30
