"""Round-loop logic of fairfedmed_amd.federated (the reference's FedOTPLoRA branch, federated_main.py:604-726) on
the host: aggregation against the reference's golden vectors, client sampling, personalisation, and the order in
which a trainer is driven (a stub trainer records the calls)."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from fairfedmed_amd import federated as F


@pytest.fixture(scope="module")
def unit(golden_dir):
    return np.load(os.path.join(golden_dir, "unit.npz"))


@pytest.fixture(scope="module")
def meta(golden_dir):
    return json.load(open(os.path.join(golden_dir, "meta.json")))


@pytest.mark.parametrize("case", ["e0_shared", "e3_shared", "e3_plain"])
def test_average_weights_ema_vs_reference(unit, meta, case):
    from tests.golden.make_golden import rng_tensor
    G, r = 3, 8
    keys = {"a.lora_S.weight": (G, r), "a.lora_A.weight": (16, r), "prompt_learner.ctx": (2, 4, 8),
            "frozen.weight": (5, 5), "b.lora_S.weight": (G, r)}
    m = meta[f"fed.{case}"]
    w = {u: {k: rng_tensor(f"fed.{case}.{u}.{k}", s) for k, s in keys.items()} for u in range(3)}
    w_g = {k: rng_tensor(f"fed.{case}.g.{k}", s) for k, s in keys.items()}
    res = F.average_weights_ema(w_g, w, m["idxs"], m["n_client"], m["by_attr"], m["epoch"], m["max_epoch"],
                                shared_half_s=m["shared_half_s"])
    for k in keys:
        np.testing.assert_allclose(res[k].numpy(), unit[f"fed.{case}.{k}"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("case,idxs,attr", [("all_attr", [0, 1, 2], True), ("two_attr", [2, 0], True), ("all_noattr", [0, 1, 2], False)])
def test_average_weights_plain_vs_reference(golden_dir, case, idxs, attr):
    """utils/fed_utils.py:6-40 under its own name: fixtures from the imported reference (make_golden.py --only-fedavg-plain)"""
    from tests.golden.make_golden import rng_tensor, FEDAVG_PLAIN_KEYS
    gold = np.load(os.path.join(golden_dir, "fedavg_plain.npz"))
    n_client, by_attr = [100, 50, 25], [[50, 30, 20], [10, 20, 20], [5, 5, 15]]
    w = {u: {k: rng_tensor(f"fedp.{case}.{u}.{k}", s) for k, s in FEDAVG_PLAIN_KEYS.items()} for u in range(3)}
    res = F.average_weights(w, idxs, n_client, by_attr if attr else None)
    assert list(res) == list(FEDAVG_PLAIN_KEYS)
    for k in FEDAVG_PLAIN_KEYS:
        np.testing.assert_allclose(res[k].numpy(), gold[f"{case}.{k}"], rtol=1e-6, atol=1e-7)
    if case == "two_attr":
        wl = {u: rng_tensor(f"fedp.list.{u}", (2, 4, 8)) for u in range(3)}
        np.testing.assert_allclose(F.average_weights(wl, [1, 2], n_client, islist=True).numpy(), gold["list"], rtol=1e-6, atol=1e-7)


def test_select_clients_matches_the_reference_rule():
    args = F.FedArgs(num_users=8, frac=0.5)
    assert F.select_clients(0, args, 8) == list(range(8))            # round 0: everybody
    np.random.seed(3)
    got = F.select_clients(1, args, 8)
    np.random.seed(3)
    ref = list(np.random.choice(range(8), max(int(0.5 * 8), 1), replace=False))
    assert got == ref and len(set(got)) == 4
    assert F.select_clients(5, F.FedArgs(num_users=8, idxs_users_train=[2, 5]), 8) == [2, 5]
    assert len(F.select_clients(2, F.FedArgs(num_users=3, frac=0.1), 3)) == 1   # max(int(0.3), 1)


def test_personalize_keeps_local_prompts_and_optionally_lora_s():
    g = {"prompt_learner.ctx": torch.zeros(2, 4, 8), "x.lora_S.weight": torch.zeros(3, 8), "x.lora_A.weight": torch.zeros(4, 8)}
    lctx = {1: torch.ones(1, 4, 8)}
    ls = {1: {"x.lora_S.weight": torch.full((3, 8), 2.0)}}
    a = F.FedArgs(num_users=2, idxs_users_train=[1], avg_prompt=1, num_prompt=2, local_s=False)
    w = F.personalize(g, 1, a, lctx, ls)
    assert float(w["prompt_learner.ctx"][0].abs().sum()) == 0 and float(w["prompt_learner.ctx"][1].min()) == 1
    assert float(w["x.lora_S.weight"].abs().sum()) == 0
    a.local_s = True
    assert float(F.personalize(g, 1, a, lctx, ls)["x.lora_S.weight"].min()) == 2.0
    assert float(F.personalize(g, 0, a, lctx, ls)["prompt_learner.ctx"].abs().sum()) == 0   # client 0 is not listed
    assert float(g["prompt_learner.ctx"].abs().sum()) == 0                                  # the global is not touched


class _StubModel:
    def __init__(self):
        self.sd = {"prompt_learner.ctx": torch.zeros(2, 2, 4), "m.lora_S.weight": torch.ones(3, 4),
                   "m.lora_A.weight": torch.zeros(5, 4)}

    def state_dict(self):
        return self.sd

    def load_state_dict(self, sd, strict=True):
        for k, v in sd.items():
            self.sd[k] = v.clone()


class _StubTrainer:
    """train(idx) adds idx + 1 to every tensor; test(idx) reports the mean of ctx."""

    def __init__(self, users):
        from types import SimpleNamespace as NS
        self.cfg = NS(DATASET=NS(USERS=users, ATTRIBUTE_TYPE="race"))
        self.model = _StubModel()
        ds = lambda n, by: NS(dataset=NS(__len__=lambda: n, count_by_attribute=lambda a: by))
        self.fed_train_loader_x_dict = {}
        for i in range(users):
            d = type("D", (), {"__len__": lambda s, n=10 * (i + 1): n,
                               "count_by_attribute": lambda s, a, i=i: [i + 1, 2, 3]})()
            self.fed_train_loader_x_dict[i] = NS(dataset=d)
        self.calls = []

    def fed_before_train(self):
        self.calls.append("before")

    def fed_after_train(self):
        self.calls.append("after")

    def train(self, idx, global_epoch, is_fed, is_last_client):
        self.calls.append(("train", global_epoch, idx, is_last_client, float(self.model.sd["m.lora_A.weight"].mean())))
        for k in self.model.sd:
            self.model.sd[k] = self.model.sd[k] + (idx + 1)

    def test(self, idx, current_epoch):
        self.calls.append(("test", current_epoch, idx))
        return [float(self.model.sd["prompt_learner.ctx"].mean()), 0.0, 0.0, 0.5]


def test_round_loop_drives_the_trainer_like_the_reference():
    tr = _StubTrainer(3)
    args = F.FedArgs(num_users=3, frac=1.0, round=2, seed=0)
    hist = F.run_fedotplora(tr, args, log=lambda *_: None)
    trains = [c for c in tr.calls if c[0] == "train"]
    assert tr.calls[0] == "before" and tr.calls[-1] == "after"
    # round 0: every client starts from the initial global weights (A mean 0) ...
    assert [c[2] for c in trains[:3]] == [0, 1, 2] and all(c[4] == 0.0 for c in trains[:3])
    assert [c[3] for c in trains[:3]] == [False, False, True]          # is_last_client
    # ... the average is weighted by the client sizes 10, 20, 30: 1/6 * 1 + 2/6 * 2 + 3/6 * 3 = 14/6
    g0 = 14.0 / 6.0
    assert all(abs(c[4] - g0) < 1e-6 for c in trains[3:])              # round 1 starts from the new global
    assert len(hist["acc"]) == 2 and len([c for c in tr.calls if c[0] == "test"]) == 6
    assert abs(hist["acc"][0] - g0) < 1e-6                             # every client is tested on the global
    # lora_S rows use the per-group counts [i+1, 2, 3]: row 0 weight of client i = (i+1)/6
    s = hist["global_weights"]["m.lora_S.weight"]
    assert s.shape == (3, 4) and float(s[1, 0]) != float(s[0, 0])


def test_group_s_rule_ignores_1d_tensors_when_rank_equals_num_groups():
    """GLOBAL_S / SVLoRA keep 1-D [r] tensors whose keys contain 'lora_S'.  With rank == num_groups their
    shape[0] equals G: they must still get the plain n_k / sum n weights, never the [G, r] block treatment
    (which on the flat buffer would read G*r elements, over the neighbouring tensors)."""
    from fairfedmed_amd.fedavg import FedAvgAggregator, element_weights, is_group_s_block
    G = r = 2
    assert is_group_s_block("m.lora_S.weight", (G, r), G)
    assert not is_group_s_block("m.lora_S_global.weight", (r,), G)
    assert not is_group_s_block("m.lora_S.weight", (r,), G)             # SVLoRA's one diagonal
    assert not is_group_s_block("m.lora_A.weight", (G, r), G)
    # flat layout: S_global [r] | lora_B [4, r] | S [G, r]
    offsets = {"m.lora_S_global.weight": (0, (r,)), "m.lora_B.weight": (4, (4, r)), "m.lora_S.weight": (12, (G, r))}
    n_client, by_attr = [10, 30], [[1, 3], [3, 1]]
    w = element_weights(offsets, 16, 0, [0, 1], n_client, by_attr)
    assert torch.allclose(w[:12], torch.full((12,), 0.25))                # plain weights, the neighbours untouched
    assert torch.allclose(w[12:16], torch.tensor([0.25, 0.25, 0.75, 0.75]))
    agg = FedAvgAggregator(torch.zeros(16), offsets, G, r)
    assert agg.s_offsets.tolist() == [12]
    # dict form: the 1-D tensor keeps its shape and the plain weights, with and without shared_half_s
    torch.manual_seed(0)
    ws = {u: {"m.lora_S_global.weight": torch.randn(r), "m.lora_B.weight": torch.randn(4, r),
              "m.lora_S.weight": torch.randn(G, r)} for u in range(2)}
    wg = {k: torch.zeros_like(v) for k, v in ws[0].items()}
    for half in (False, True):
        out = F.average_weights_ema(wg, ws, [0, 1], n_client, by_attr, 0, 4, shared_half_s=half)
        assert out["m.lora_S_global.weight"].shape == (r,)
        exp = 0.25 * ws[0]["m.lora_S_global.weight"] + 0.75 * ws[1]["m.lora_S_global.weight"]
        assert torch.allclose(out["m.lora_S_global.weight"], exp)
        assert torch.allclose(out["m.lora_B.weight"], 0.25 * ws[0]["m.lora_B.weight"] + 0.75 * ws[1]["m.lora_B.weight"])
    s = out["m.lora_S.weight"]
    assert float(s[0, 0]) == float(s[1, 0]) and float(s[0, 1]) != float(s[1, 1])   # first half shared, second not
