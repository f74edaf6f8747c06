"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol
include/ffm_hip.h declares, the registry keeps the reference's error behaviour,
host metrics match the reference's golden values, the product never routes
through the oracle, and the GPU-only entry points fail loudly without a GPU."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from fairfedmed_amd import _lib, config as C, metrics, synth
from fairfedmed_amd.registry import Registry, TRAINER_REGISTRY, build_trainer, check_availability

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ffm_hip.h")).read()
    declared = set(re.findall(r"\bint(?:64_t)?\s+(ffm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes parsed"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in ffm_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert _lib.load().ffm_abi_version() == _lib.ABI_VERSION


def test_every_dtype_entry_point_has_a_half_twin_behind_one_dispatcher():
    """fairfedmed_amd/build.py: every prototype with a `dtype` parameter is defined once by the generated dispatcher
    (public name) and twice below it - `<name>_m` (float32 / bfloat16 / FFM_F32_X3) and `<name>_f16` (FFM_F16, the same
    source compiled with -DFFM_TWIN_F16); FFM_F16 reaches the twin, everything else the main object."""
    from fairfedmed_amd import build as B
    protos = B.api_prototypes()
    typed = [n for _, n, params in protos if any(a == "dtype" for _, a in params)]
    assert len(protos) >= 60 and len(typed) >= 37
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in typed:
        assert hasattr(lib, n) and hasattr(lib, n + "_m") and hasattr(lib, n + "_f16"), n
    disp = open(B.GEN_DISPATCH).read()
    for n in typed:
        assert f"dtype == FFM_F16 ? {n}_f16(" in disp and f": {n}_m(" in disp, n
    # argument validation of both twins without a GPU: null pointers are rejected on every path, never dereferenced
    l = _lib.load()
    for code in (_lib.F32, _lib.BF16, _lib.F16):
        assert l.ffm_layernorm_fwd(None, None, None, None, None, None, 4, 768, code, None) == -1
        assert l.ffm_attention_fwd(None, None, None, 2, 197, 12, 0, code, None) == -1
    assert l.ffm_lora_down_blocks(6304, 768, 8, _lib.F16) == l.ffm_lora_down_blocks(6304, 768, 8, _lib.BF16) > 0
    assert l.ffm_gemm_tiles_m(6304, 3072, 768, 0, 0, _lib.F16, 1) == l.ffm_gemm_tiles_m(6304, 3072, 768, 0, 0, _lib.BF16, 1) > 0
    assert l.ffm_scale_check(None, 1.0, 4, None, None) == -1


def test_argument_validation_needs_no_gpu():
    lib = _lib.load()
    args = _lib.GemmArgs()                          # null pointers
    assert lib.ffm_gemm_nt(ctypes.byref(args), _lib.BF16, None) == -1
    assert lib.ffm_layernorm_fwd(None, None, None, None, None, None, 4, 768, _lib.BF16, None) == -1
    assert lib.ffm_lora_grad_splits(6304) > 0 and lib.ffm_lora_down_blocks(6304, 768, 8, _lib.BF16) > 0


def test_gemm_args_mirror_the_header_field_by_field():
    """fairfedmed_amd/_lib.GemmArgs is the ctypes image of `ffm_gemm_args` (include/ffm_hip.h): same field names in the
    same order, pointers / int32 / float where the header has them (a field appended to one and not the other shifts
    every later pointer silently)."""
    hdr = open(os.path.join(ROOT, "include", "ffm_hip.h")).read()
    body = hdr[hdr.index("typedef struct ffm_gemm_args {"):hdr.index("} ffm_gemm_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"^(const )?(void|float|int32_t)\s*(\*?)\s*(.+)$", decl)
        assert m, decl
        kind = ctypes.c_void_p if m.group(3) else {"float": ctypes.c_float, "int32_t": ctypes.c_int32}[m.group(2)]
        for name in m.group(4).split(","):
            name = name.strip()
            star = name.startswith("*")
            fields.append((name.lstrip("* "), ctypes.c_void_p if star else kind))
    assert [(n, t) for n, t in _lib.GemmArgs._fields_] == fields
    # FFM_EPI_LGRAD: the query answers without a GPU - the 8-wave 208 x 384 tile serves the bench shape (31 row tiles), no
    # kernel serves a shape that tile does not take, and an odd rank is refused
    l = _lib.load()
    fl = _lib.EPI_LORA | _lib.EPI_LORA_KR | _lib.EPI_DGELU | _lib.EPI_RANKOP
    assert l.ffm_gemm_lgrad_rows(6304, 3072, 768, fl, 8, _lib.BF16, 1) == 31 == l.ffm_gemm_lgrad_rows(6304, 3072, 768, fl, 8, _lib.F16, 1)
    assert l.ffm_gemm_lgrad_rows(6304, 3072, 768, fl, 8, _lib.BF16, 0) < 0 and l.ffm_gemm_lgrad_rows(6304, 3072, 768, fl, 6, _lib.BF16, 1) < 0
    assert l.ffm_gemm_lgrad_rows(6304, 768, 3072, fl, 8, _lib.BF16, 1) < 0
    # the LayerNorm-backward folds (ABI 12): the queries the engine decides by.  Bench shape: eight column tiles of the dX
    # product of c_proj leave the row sums; the dX products of c_fc (FairLoRA tile) and of the in-projection (plain tile) apply
    # them; 3D OCT's 19 700 rows are not single-round tiles there; float32 storage is never served; the attention backward
    # leaves its row sums for 97..256 unmasked tokens in 16-bit storage
    f1 = fl | _lib.EPI_LGRAD | _lib.EPI_LNB_STAT
    f2 = _lib.EPI_LORA | _lib.EPI_LORA_KR | _lib.EPI_RANKOP | _lib.EPI_LNB_APPLY
    assert l.ffm_gemm_tiles_n(6304, 3072, 768, f1, 8, _lib.BF16, 1) == 8 == l.ffm_gemm_tiles_n(6304, 3072, 768, f1, 8, _lib.F16, 1)
    assert l.ffm_gemm_tiles_n(6304, 768, 3072, f2, 8, _lib.BF16, 1) == 6 and l.ffm_gemm_tiles_n(6304, 768, 2304, _lib.EPI_LNB_APPLY, 0, _lib.BF16, 1) == 6
    assert l.ffm_gemm_tiles_n(19700, 768, 3072, f2, 16, _lib.BF16, 1) < 0 and l.ffm_gemm_tiles_n(6304, 768, 3072, f2, 8, _lib.F32, 1) < 0
    assert l.ffm_gemm_tiles_n(6304, 768, 3072, f2, 8, _lib.BF16, 0) < 0
    assert l.ffm_attention_bwd_lnstat_ok(197, 0, _lib.BF16) == 1 == l.ffm_attention_bwd_lnstat_ok(256, 0, _lib.F16)
    assert l.ffm_attention_bwd_lnstat_ok(96, 0, _lib.BF16) == 0 == l.ffm_attention_bwd_lnstat_ok(197, 1, _lib.BF16) == l.ffm_attention_bwd_lnstat_ok(197, 0, _lib.F32)


def test_integration_md_stub_mirrors_gemm_args():
    """The ctypes stub a maintainer would paste from INTEGRATION.md lists ffm_gemm_args' fields in the header's order (a stub
    that lags the header shifts every later pointer)."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    body = md[md.index("class GemmArgs(ctypes.Structure):"):md.index("lib.ffm_gemm_nt.argtypes")]
    names = re.findall(r'\("(\w+)",\s*ctypes\.c_(\w+)\)', body)
    kinds = {"void_p": ctypes.c_void_p, "int32": ctypes.c_int32, "float": ctypes.c_float}
    assert [(n, kinds[k]) for n, k in names] == [(n, t) for n, t in _lib.GemmArgs._fields_]


def test_registry_semantics():
    r = Registry("T")

    @r.register()
    class A:
        pass
    assert r.get("A") is A and r.registered_names() == ["A"]
    with pytest.raises(KeyError):
        r.register(A)
    with pytest.raises(KeyError):
        r.get("B")
    with pytest.raises(ValueError, match="GLP_OT_SVLoRA"):
        check_availability("GLP_OT_SVLora", ["GLP_OT_SVLoRA"])
    import fairfedmed_amd.trainer  # noqa: F401  (registers the trainer)
    assert "GLP_OT_SVLoRA" in TRAINER_REGISTRY.registered_names()


def test_metrics_match_reference_goldens(golden_dir):
    unit = np.load(os.path.join(golden_dir, "unit.npz"))
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))
    for name in ("n32", "n200"):
        got = metrics.auc_macro_ovr(unit[f"auc.{name}.prob"], unit[f"auc.{name}.y"])
        assert abs(got - meta["auc"][name]) < 1e-12
    assert metrics.auc_macro_ovr(np.array([[.4, .6], [.3, .7]]), np.array([1, 1])) == 1.0
    assert abs(metrics.macro_f1(np.array([0, 1, 1, 0]), np.array([0, 1, 0, 0]), 2) - (0.8 + 2 / 3) / 2) < 1e-12


def test_manifest_counts_match_reference(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))
    m = synth.manifest(C.vit_b16(rank=8))
    keys = synth.trainable_keys(C.vit_b16(rank=8))
    assert len(keys) == meta["vitb_r8.trainable_tensors"] == 73
    assert sum(int(np.prod(m[k])) for k in keys) == meta["vitb_r8.trainable_elems"] == 741952
    assert sum(int(np.prod(s)) for k, s in m.items() if "token_" not in k) == meta["vitb_r8.total_params"]


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fairfedmed_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_engine_fails_loudly_without_gpu():
    from fairfedmed_amd.engine import FairLoRAEngine
    with pytest.raises(RuntimeError, match="no CPU path"):
        FairLoRAEngine(C.vit_tiny(), {}, max_images=1)
    from fairfedmed_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm_fwd(torch.zeros(4, 128), torch.zeros(4, 128), torch.ones(128), torch.zeros(128))


def test_fairness_scores_vs_reference(golden_dir):
    """Per-group AUC, ES-AUC and between-group disparity against the imported reference functions
    (evaluation/metrics.py:513-552); DPD / EOD against hand-computed rates (fairlearn's published definition)."""
    import json
    import numpy as np
    from fairfedmed_amd import metrics as Mx
    unit = np.load(os.path.join(golden_dir, "unit.npz"))
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))["fair"]
    for name, ref in meta.items():
        prob, y, attr = unit[f"fair.{name}.prob"], unit[f"fair.{name}.y"], unit[f"fair.{name}.attr"]
        assert abs(Mx.auc_macro_ovr(prob, y) - ref["overall"]) < 1e-12
        ga = Mx.group_aucs(prob, y, attr)
        assert np.allclose(ga, ref["group_aucs"], atol=1e-12)
        assert abs(Mx.equity_scaled_auc(prob, y, attr) - ref["es_auc"]) < 1e-12
        assert np.allclose(Mx.between_group_disparity(ga, ref["overall"]), ref["disparity"], atol=1e-12)
        sc = Mx.comprehensive_scores(prob, y, attr[None])
        assert abs(sc["esaucs_by_attrs"][0] - ref["es_auc"]) < 1e-12 and 0.0 <= sc["dpds"][0] <= 1.0
    # DPD / EOD on a case small enough to do by hand: group 0 -> preds 1,1,0,0 labels 1,0,1,0; group 1 -> preds 1,0,0,0 labels 1,1,0,0
    pred = np.array([1, 1, 0, 0, 1, 0, 0, 0])
    lab = np.array([1, 0, 1, 0, 1, 1, 0, 0])
    att = np.array([0, 0, 0, 0, 1, 1, 1, 1])
    assert abs(Mx.demographic_parity_difference(lab, pred, att) - (0.5 - 0.25)) < 1e-12
    # TPR: g0 1/2, g1 1/2 -> gap 0; FPR: g0 1/2, g1 0 -> gap 0.5
    assert abs(Mx.equalized_odds_difference(lab, pred, att) - 0.5) < 1e-12


def test_apply_lora_resnet_rule():
    """ResNet branch of apply_lora_to_model (trainers/GLP_OT_SVLoRA.py:541-573): 1x1 convolutions named conv* under
    image_encoder.layer* become FairLoRALinear, attnpool's nn.Linear plain LoRALinear; 3x3 convolutions, downsample.0
    and everything outside image_encoder stay."""
    import torch.nn as nn
    from fairfedmed_amd.model import FairLoRALinear, LoRALinear, apply_lora_to_model

    class Neck(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1, self.conv2, self.conv3 = nn.Conv2d(64, 64, 1, bias=False), nn.Conv2d(64, 64, 3, bias=False), \
                nn.Conv2d(64, 256, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.downsample = nn.Sequential(nn.AvgPool2d(1), nn.Conv2d(64, 256, 1, bias=False), nn.BatchNorm2d(256))

    class Pool(nn.Module):
        def __init__(self):
            super().__init__()
            self.k_proj, self.q_proj, self.v_proj, self.c_proj = (nn.Linear(128, 128), nn.Linear(128, 128),
                                                                  nn.Linear(128, 128), nn.Linear(128, 64))

    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 32, 3, bias=False)
            self.layer1 = nn.Sequential(Neck())
            self.attnpool = Pool()

    class Model(nn.Module):
        def __init__(self):
            super().__init__()
            self.image_encoder = Enc()
            self.text_encoder = nn.Sequential(nn.Linear(8, 8))

    m = Model()
    apply_lora_to_model(m, True, rank=4, alpha=2.0, lora_type="FairLoRA", num_attrs=2)
    neck = m.image_encoder.layer1[0]
    assert isinstance(neck.conv1, FairLoRALinear) and isinstance(neck.conv3, FairLoRALinear)
    assert tuple(neck.conv1.lora_S.weight.shape) == (2, 4) and tuple(neck.conv3.lora_B.weight.shape) == (4, 256)
    assert isinstance(neck.conv2, nn.Conv2d) and isinstance(neck.downsample[1], nn.Conv2d)
    assert isinstance(m.image_encoder.conv1, nn.Conv2d)
    for n in "kqvc":
        assert isinstance(getattr(m.image_encoder.attnpool, n + "_proj"), LoRALinear)
    assert isinstance(m.text_encoder[0], nn.Linear)
    names = {k for k, p in m.named_parameters() if "lora_" in k}
    assert len(names) == 2 * 3 + 4 * 2
    m2 = Model()
    apply_lora_to_model(m2, False, rank=4, alpha=2.0, lora_type="FairLoRA", num_attrs=2)
    assert not any("lora_" in k for k, _ in m2.named_parameters())


def _brute_counts(prob, y, attr, G):
    """The table ffm_eval_counts produces (include/ffm_hip.h), by brute force in numpy."""
    t = np.zeros((G + 2, 10), dtype=np.int64)
    slot = np.where((attr >= 0) & (attr < G), attr, G)
    pred = prob[:, 1] > prob[:, 0]
    for rows, sel in [(g, slot == g) for g in range(G + 1)] + [(G + 1, np.ones(len(y), bool))]:
        p, yy, pr = prob[sel], y[sel], pred[sel]
        pos, neg = p[yy == 1], p[yy == 0]
        t[rows, 0], t[rows, 1] = len(pos), len(neg)
        t[rows, 2] = (pos[:, None, 1] > neg[None, :, 1]).sum()
        t[rows, 3] = (pos[:, None, 1] == neg[None, :, 1]).sum()
        t[rows, 4] = (neg[None, :, 0] > pos[:, None, 0]).sum()
        t[rows, 5] = (neg[None, :, 0] == pos[:, None, 0]).sum()
        t[rows, 6], t[rows, 7] = (pr & (yy == 1)).sum(), (pr & (yy == 0)).sum()
        t[rows, 8], t[rows, 9] = (~pr & (yy == 0)).sum(), (~pr & (yy == 1)).sum()
    return t


def test_scores_from_counts_equal_the_sort_based_scores():
    """metrics.*_from_counts (fed by the on-device count kernel) == the sort / mid-rank implementations pinned against
    the reference's evaluator, including tied scores, an unknown (-1) attribute value and an absent group."""
    from fairfedmed_amd import metrics as M
    rng = np.random.default_rng(5)
    N = 400
    logits = np.round(rng.normal(size=(N, 2)) * 2, 1)              # coarse grid: many exact ties
    e = np.exp(logits - logits.max(1, keepdims=True))
    prob = (e / e.sum(1, keepdims=True)).astype(np.float32)
    y = rng.integers(0, 2, N)
    a0 = rng.integers(0, 3, N)
    a1 = rng.integers(0, 2, N) * 2                                   # groups 0 and 2 only
    a1[:7] = -1
    attrs = np.stack([a0, a1])
    tables = [_brute_counts(prob, y, a, 8) for a in attrs]
    ref = M.comprehensive_scores(prob, y, attrs)
    got = M.comprehensive_scores_from_counts(tables)
    assert abs(got["overall_auc"] - ref["overall_auc"]) < 1e-12
    for k in ("esaucs_by_attrs", "dpds", "eods"):
        assert np.allclose(got[k], ref[k], rtol=0, atol=1e-12), k
    for a, b in zip(got["aucs_by_attrs"], ref["aucs_by_attrs"]):
        assert a.shape == b.shape and np.allclose(a, b, rtol=0, atol=1e-12)
    assert np.allclose(got["between_group_disparity"], ref["between_group_disparity"], rtol=0, atol=1e-12)
    pred = prob.argmax(-1)
    acc, err, f1, auc = M.basic_from_counts(tables[0])
    assert abs(acc - 100.0 * (pred == y).mean()) < 1e-12 and abs(err - (100 - acc)) < 1e-12
    assert abs(f1 - 100.0 * M.macro_f1(pred, y, 2)) < 1e-12 and abs(auc - 100.0 * M.auc_macro_ovr(prob, y)) < 1e-10
    # a single-class set reports AUC 1 (trainers/GLP_OT_SVLoRA.py:965-967)
    one = _brute_counts(prob, np.ones(N, np.int64), a0, 8)
    res = M.basic_from_counts(one)
    assert res[3] == 100.0
    # macro-F1 runs over the classes present in the labels only (f1_score(labels=np.unique(y_true)),
    # evaluation/evaluator_oph.py:70-75): with one class present it is that class's F1, not half of it
    assert abs(res[2] - 100.0 * M.macro_f1(pred, np.ones(N, np.int64), 2)) < 1e-12
    tp, fn = float((pred == 1).sum()), float((pred == 0).sum())
    assert abs(res[2] - 100.0 * 2 * tp / (2 * tp + fn)) < 1e-12


def test_dpd_eod_against_hand_derived_fairlearn_values():
    """fairlearn is not installable here, so demographic_parity_difference / equalized_odds_difference
    (evaluation/metrics.py:256-279 calls them with sensitive_features = the attribute column, -1 included) are pinned
    against values worked out BY HAND from the package's published definitions:

      selection rate  SR_g  = P(y_hat = 1 | group g)
      DPD                   = max_g SR_g - min_g SR_g
      TPR_g = P(y_hat = 1 | y = 1, g),  FPR_g = P(y_hat = 1 | y = 0, g)   (0 when the group has no such sample:
                              recall_score's zero_division convention, which fairlearn's rates inherit)
      EOD                   = max(max_g TPR_g - min_g TPR_g, max_g FPR_g - min_g FPR_g)

    for the host functions AND for the scores derived from the device evaluator's count table."""
    from fairfedmed_amd import metrics as M

    def table_scores(y, pred, attr, G=4):
        prob = np.stack([1.0 - pred, pred.astype(np.float64)], 1).astype(np.float32)     # argmax reproduces pred
        out = M.comprehensive_scores_from_counts([_brute_counts(prob, y, attr, G)])
        return out["dpds"][0], out["eods"][0]

    # --- case 1: two groups
    #   g0: y_hat 1 1 0 0 | y 1 0 1 0  -> SR 2/4, TPR 1/2, FPR 1/2
    #   g1: y_hat 1 0 0 0 0 | y 1 1 0 0 0 -> SR 1/5, TPR 1/2, FPR 0/3
    pred = np.array([1, 1, 0, 0, 1, 0, 0, 0, 0])
    y = np.array([1, 0, 1, 0, 1, 1, 0, 0, 0])
    a = np.array([0, 0, 0, 0, 1, 1, 1, 1, 1])
    for dpd, eod in ((M.demographic_parity_difference(y, pred, a), M.equalized_odds_difference(y, pred, a)),
                     table_scores(y, pred, a)):
        assert abs(dpd - (2 / 4 - 1 / 5)) < 1e-12                   # 0.3
        assert abs(eod - max(1 / 2 - 1 / 2, 1 / 2 - 0.0)) < 1e-12   # FPR gap 0.5
    # --- case 2: three groups, one of them the "unknown" value -1 (a group like any other for fairlearn)
    #   g-1: y_hat 1 1 1 | y 1 1 0     -> SR 3/3, TPR 2/2, FPR 1/1
    #   g0 : y_hat 0 0 1 0 | y 1 0 0 0 -> SR 1/4, TPR 0/1, FPR 1/3
    #   g2 : y_hat 1 0 | y 1 1         -> SR 1/2, TPR 1/2, FPR 0 (no negatives)
    pred = np.array([1, 1, 1, 0, 0, 1, 0, 1, 0])
    y = np.array([1, 1, 0, 1, 0, 0, 0, 1, 1])
    a = np.array([-1, -1, -1, 0, 0, 0, 0, 2, 2])
    for dpd, eod in ((M.demographic_parity_difference(y, pred, a), M.equalized_odds_difference(y, pred, a)),
                     table_scores(y, pred, a)):
        assert abs(dpd - (1.0 - 1 / 4)) < 1e-12                     # 0.75
        assert abs(eod - max(1.0 - 0.0, 1.0 - 0.0)) < 1e-12         # TPR gap 1 - 0, FPR gap 1 - 0
    # --- case 3: perfectly fair predictor with unequal base rates: equal rates in every group -> both 0
    #   g0: y_hat 1 0 1 0 | y 1 1 0 0 ; g1: y_hat 1 0 1 0 1 0 1 0 | y 1 1 0 0 1 1 0 0  -> SR 1/2, TPR 1/2, FPR 1/2
    pred = np.array([1, 0, 1, 0] + [1, 0, 1, 0, 1, 0, 1, 0])
    y = np.array([1, 1, 0, 0] + [1, 1, 0, 0, 1, 1, 0, 0])
    a = np.array([0] * 4 + [1] * 8)
    for dpd, eod in ((M.demographic_parity_difference(y, pred, a), M.equalized_odds_difference(y, pred, a)),
                     table_scores(y, pred, a)):
        assert dpd == 0.0 and eod == 0.0
    # --- case 4: one group only -> both differences are 0 by definition (max = min)
    a = np.zeros(12, dtype=np.int64)
    assert M.demographic_parity_difference(y, pred, a) == 0.0 and M.equalized_odds_difference(y, pred, a) == 0.0
    assert table_scores(y, pred, a) == (0.0, 0.0)


def test_cli_flags_config_tree_and_scope(tmp_path):
    """fairfedmed_amd.federated_main: the reference's flag names (federated_main.py:791-881), its type=bool quirk
    (any non-empty value is True), config-file merge order, and the scope guard."""
    from fairfedmed_amd import federated_main as FM
    (tmp_path / "ds.yaml").write_text('DATASET:\n  NAME: "FairFedMed"\n')
    (tmp_path / "tr.yaml").write_text('DATALOADER:\n  TRAIN_X:\n    BATCH_SIZE: 32\n  TEST:\n    BATCH_SIZE: 100\n'
                                      'INPUT:\n  SIZE: (224, 224)\n  PIXEL_MEAN: [0.48145466, 0.4578275, 0.40821073]\n'
                                      'OPTIM:\n  LR: 0.5\n  MAX_EPOCH: 7\nMODEL:\n  BACKBONE:\n    NAME: "ViT-B/16"\n')
    argv = ["--root", "DATA/", "--model", "FedOTPLoRA", "--seed", "1", "--num_users", "3", "--frac", "0.8", "--lr", "0.001",
            "--OT", "None", "--gamma", "0.1", "--trainer", "GLP_OT_SVLoRA", "--round", "50", "--stepsize", "200",
            "--input_no_transform", "False", "--attribute_type", "language", "--partition", "noniid-labeldir100",
            "--beta", "0.3", "--n_ctx", "4", "--num_prompt", "2", "--unfreeze_image_encoder", "True", "--lora_rank", "12",
            "--lora_alpha", "2", "--lora_type", "FairLoRA", "--dataset-config-file", str(tmp_path / "ds.yaml"),
            "--config-file", str(tmp_path / "tr.yaml"), "--output-dir", str(tmp_path / "out"), "--shared_half_s", "True",
            "--lambda_fairness", "0.0"]                              # scripts/fairfedlora_fairfedmed.sh
    args = FM.build_parser().parse_args(argv)
    assert args.input_no_transform is True and args.shared_half_s is True and args.lora_local_s is False
    cfg = FM.setup_cfg(args)
    FM.check_scope(args, cfg)
    assert cfg.INPUT.SIZE == (224, 224) and cfg.INPUT.NO_TRANSFORM is True
    assert cfg.OPTIM.LR == 0.001 and cfg.OPTIM.MAX_EPOCH == 1 and cfg.OPTIM.STEPSIZE == 200      # command line wins
    assert cfg.DATALOADER.TRAIN_X.BATCH_SIZE == 32 and cfg.TEST.BATCH_SIZE == 100
    lo = cfg.TRAINER.GLP_OT_LORA
    assert (lo.RANK, lo.ALPHA, lo.TYPE, lo.UNFREEZE_IMAGE_ENCODER) == (12, 2.0, "FairLoRA", True)
    assert cfg.TRAINER.GLP_OT.N == 2 and cfg.TRAINER.GLP_OT.N_CTX == 4 and cfg.DATASET.ATTRIBUTE_TYPE == "language"
    for bad in (["--model", "fedavg"], ["--OT", "Wasserstein"], ["--trainer", "PromptFL"]):
        a = FM.build_parser().parse_args(argv + bad)
        with pytest.raises(NotImplementedError):
            FM.check_scope(a, FM.setup_cfg(a))


def test_missing_or_stale_library_fails_loudly(monkeypatch, tmp_path):
    """No fallback: without libffm_hip.so (or with one of another ABI version) every op raises at load time."""
    from fairfedmed_amd import _lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "libffm_hip.so"))
    with pytest.raises(RuntimeError, match="is missing"):
        L.load()
    monkeypatch.undo()
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "ABI_VERSION", L.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI"):
        L.load()


def test_bench_gpus_flag_builds_a_rank_launch(monkeypatch):
    """bench.py --gpus N outside a launcher starts N torch.distributed.run ranks of itself and relays the exit code."""
    import importlib
    import io
    import bench
    importlib.reload(bench)
    seen = {}

    class FakeProc:
        stdout = io.StringIO('{"metric": "x"}\n')

        def wait(self):
            return 7

    def fake_popen(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        return FakeProc()

    monkeypatch.setattr(bench.subprocess, "Popen", fake_popen)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # more ranks than devices is an error, not a silent one-rank run
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "only 1 GPU" in str(e.value.code)


def _ref_style_cfg(mcfg, attribute="race", dataset="FairFedMed"):
    from types import SimpleNamespace as NS
    return NS(INPUT=NS(SIZE=(mcfg.vision.image_size,) * 2, PIXEL_MEAN=list(mcfg.pixel_mean), PIXEL_STD=list(mcfg.pixel_std)),
              DATASET=NS(NAME=dataset, ATTRIBUTE_TYPE=attribute, MODALITY_TYPE="slo_fundus", DIM_PER_3D_SLICE=0),
              DATALOADER=NS(TRAIN_X=NS(BATCH_SIZE=6)), TEST=NS(BATCH_SIZE=6),
              TRAINER=NS(GLP_OT=NS(N=mcfg.n_prompts, N_CTX=mcfg.n_ctx, PREC="fp32", OT="None", CTX_INIT=False, CSC=False,
                                   CLASS_TOKEN_POSITION="end"),
                         GLP_OT_LORA=NS(RANK=mcfg.lora.rank, ALPHA=mcfg.lora.alpha, TYPE="FairLoRA", GLOBAL_S=False,
                                        UNFREEZE_IMAGE_ENCODER=True, DISABLE_ATTR=False)))


@pytest.mark.parametrize("tag,mk,names,attribute,dataset", [
    ("adapter_vit", lambda: C.vit_tiny(rank=4), ["NOT Glaucoma", "Glaucoma"], "race", "FairFedMed"),
    ("adapter_rn", lambda: C.rn_tiny(rank=4, num_groups=2), ["NOT Pleural Effusion", "Pleural Effusion"], "gender", "FedChexMimic")])
def test_reference_constructor_arguments_to_engine_inputs(golden_dir, tag, mk, names, attribute, dataset):
    """CustomCLIP(cfg, classnames, clip_model)'s host side (fairfedmed_amd/clip_adapter.py): geometry from CLIP's tensor
    shapes, tokenised prompts, token_prefix / token_suffix from CLIP's token embedding, keys and initial adapter values -
    against what the imported reference's CustomCLIP + apply_lora_to_model produced from the same CLIP tensors."""
    from fairfedmed_amd import clip_adapter as A
    unit = np.load(os.path.join(golden_dir, "unit.npz"))
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))
    mcfg = mk()
    clip_model = synth.make_clip_model(mcfg, seed=1)
    torch.manual_seed(0)
    got, sd, toks = A.from_reference_args(_ref_style_cfg(mcfg, attribute, dataset), names, clip_model)
    import dataclasses
    eot = (9, 8) if names[1] == "Glaucoma" else (11, 10)               # EOT positions
    assert got == dataclasses.replace(mcfg, eot=eot)
    assert np.array_equal(toks.numpy(), unit[f"{tag}.tokens"])
    assert list(sd.keys()) == meta[f"{tag}.keys"]
    assert np.array_equal(sd["prompt_learner.token_prefix"].numpy(), unit[f"{tag}.token_prefix"])
    assert np.array_equal(sd["prompt_learner.token_suffix"].numpy(), unit[f"{tag}.token_suffix"])
    assert abs(float(sd["prompt_learner.ctx"].std()) - 0.02) < 0.004 and abs(meta[f"{tag}.ctx_std"] - 0.02) < 0.004
    for k, v in sd.items():
        if k.endswith("lora_A.weight"):
            assert float(v.abs().max()) == 0.0
        elif k.endswith("lora_S.weight"):
            assert torch.equal(v, synth.lora_s_init(mcfg.lora.rank, mcfg.lora.num_groups))
        elif k.endswith("lora_B.weight"):
            assert 0.7 < float(v.std()) < 1.3
    frozen = clip_model.state_dict()
    assert torch.equal(sd["image_encoder.conv1.weight"], frozen["visual.conv1.weight"])
    assert torch.equal(sd["text_encoder.text_projection"], frozen["text_projection"])
    with pytest.raises(NotImplementedError, match="not pinned"):
        A.tokenize_prompts(["Cardiomegaly"], 4)
    # a caller-supplied tokenizer (the reference's clip.tokenize) serves any other class name
    fake = lambda s: torch.tensor([[49406, 343, 343, 343, 343, 1000, 269, 49407] + [0] * 69])
    assert A.tokenize_prompts(["Cardiomegaly"], 4, tokenize=fake).shape == (1, 77)


def test_bench_shapes_select_the_documented_tiles():
    """Kernel selection is host logic (ffm_gemm_tile_shape needs no GPU): at the bench workload's shapes (6 304 token rows,
    bf16, frozen weights packed) the eight GEMMs of a vision block run on the tiles DESIGN section 4.4 describes, each in ONE
    round of <= 256 blocks; at the 3D OCT workload's 19 700 rows the FairLoRA products take the 208 x 384 tile at several
    rounds and the N = 768 products a one-round 240 x 256 tile."""
    import torch
    from fairfedmed_amd import ops
    bf, W, E = torch.bfloat16, 768, _lib
    rk = E.EPI_LORA | E.EPI_RANKOP
    shapes = {
        "qkv": (3 * W, W, E.EPI_BIAS | E.EPI_LNIN, 0), "out": (W, W, E.EPI_BIAS | E.EPI_RESIDUAL | E.EPI_ROWSTATS, 0),
        "c_fc": (4 * W, W, E.EPI_BIAS | E.EPI_GELU | E.EPI_LNIN | rk, 8),
        "c_proj": (W, 4 * W, E.EPI_BIAS | E.EPI_RESIDUAL | E.EPI_ROWSTATS | rk, 8),
        "dx_c_proj": (4 * W, W, E.EPI_LORA_KR | E.EPI_DGELU | rk, 8), "dx_c_fc": (W, 4 * W, E.EPI_LORA_KR | rk, 8),
        "dx_out": (W, W, 0, 0), "dx_qkv": (W, 3 * W, 0, 0)}
    want = {"qkv": (240, 256, 8), "out": (160, 128, 4), "c_fc": (208, 384, 8), "c_proj": (160, 128, 4),
            "dx_c_proj": (208, 384, 8), "dx_c_fc": (160, 128, 4), "dx_out": (160, 128, 4), "dx_qkv": (160, 128, 4)}
    M = 32 * 197
    for name, (N, K, fl, r) in shapes.items():
        cfg, bm, bn, waves = ops.gemm_tile_shape(M, N, K, fl, r, bf, True)
        assert cfg >= 0 and (bm, bn, waves) == want[name], (name, cfg, bm, bn, waves)
        assert ((M + bm - 1) // bm) * (N // bn) <= 256, name                 # one round of tiles
        assert ops.gemm_tile_shape(M, N, K, fl, r, bf, False)[0] == -1       # unpacked weights: the 128 x 128 kernel
        assert ops.gemm_tile_shape(M, N, K, fl & ~(E.EPI_LNIN | E.EPI_ROWSTATS), r, torch.float32, True)[0] == -1   # fp32 mode
    M = 100 * 197
    for name in ("c_fc", "dx_c_proj"):
        N, K, fl, _ = shapes[name]
        cfg, bm, bn, waves = ops.gemm_tile_shape(M, N, K, fl, 16, bf, True)
        assert (bm, bn, waves) == (208, 384, 8) and ((M + bm - 1) // bm) * (N // bn) > 512, name
    assert ops.gemm_tile_shape(M, W, 3 * W, 0, 0, bf, True)[1:] == (240, 256, 8)
