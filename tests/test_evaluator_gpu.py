"""On-device evaluator: ffm_eval_counts against a numpy brute force (exact integers) and the
scores derived from it against the sort-based host metrics that are pinned against the reference's evaluator."""
import numpy as np
import pytest
import torch

from fairfedmed_amd import metrics as M
from tests.test_host_cpu import _brute_counts

pytestmark = pytest.mark.gpu


def make(N, G, seed, ties=True, unknown=0):
    rng = np.random.default_rng(seed)
    logits = rng.normal(size=(N, 2)) * 2
    if ties:
        logits = np.round(logits, 1)
    prob = torch.softmax(torch.from_numpy(logits).float(), -1).numpy()
    y = rng.integers(0, 2, N)
    a = rng.integers(0, G, N)
    a[:unknown] = -1
    return prob, y, a


@pytest.mark.parametrize("N,G,unknown", [(1, 2, 0), (37, 3, 0), (256, 2, 5), (1000, 3, 11), (4099, 8, 0)])
def test_eval_counts_exact(N, G, unknown):
    from fairfedmed_amd import ops
    prob, y, a = make(N, G, seed=N, unknown=unknown)
    got = ops.eval_counts(torch.from_numpy(prob).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(a).cuda(), G)
    assert np.array_equal(got.cpu().numpy(), _brute_counts(prob, y, a, G))
    # no attribute column: everything is "unknown", the all-samples row is unchanged
    got2 = ops.eval_counts(torch.from_numpy(prob).cuda(), torch.from_numpy(y).cuda(), None, G).cpu().numpy()
    assert np.array_equal(got2[-1], got.cpu().numpy()[-1]) and np.array_equal(got2[-2], got2[-1])
    assert not got2[:-2].any()


def test_eval_counts_large_is_launch_geometry_independent():
    """20 000 samples (2e8 pairs): the j range is split over grid rows; integer counts leave no rounding to hide in."""
    from fairfedmed_amd import ops
    prob, y, a = make(20000, 3, seed=3, ties=False)
    t = ops.eval_counts(torch.from_numpy(prob).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(a).cuda(), 3)
    t = t.cpu().numpy()
    assert abs(M.basic_from_counts(t)[3] - 100.0 * M.auc_macro_ovr(prob, y)) < 1e-10
    ga = M.group_aucs(prob, y, a)
    for g in range(3):
        assert abs(M._auc_from_row(t[g]) - ga[g]) < 1e-12
    assert t[-1, 0] + t[-1, 1] == 20000 and t[:3, :2].sum() == 20000


def test_trainer_test_device_and_host_metrics_agree():
    """GLP_OT_SVLoRA.test(): the device evaluator reports what the host evaluator reports."""
    from tests.test_trainer_gpu import make_cfg
    from fairfedmed_amd import config as C
    from fairfedmed_amd.trainer import GLP_OT_SVLoRA, SyntheticFedData
    mcfg = C.vit_tiny(rank=4)
    cfg = make_cfg(prec="fp32")
    cfg.TEST.NO_TEST = False
    data = SyntheticFedData(mcfg, num_clients=1, train_batches=2, test_batches=5, batch_size=8, signal=0.4)
    tr = GLP_OT_SVLoRA(cfg, data=data)
    tr.fed_before_train()
    tr.train(idx=0, global_epoch=0, is_fed=True)
    dev = tr.test(idx=0)
    dev_full = dict(tr.last_results)
    cfg.TEST.HOST_METRICS = True
    host = tr.test(idx=0)
    host_full = dict(tr.last_results)
    assert np.allclose(dev, host, rtol=0, atol=1e-9)
    assert set(dev_full) == set(host_full)
    for k in ("overall_auc", "esaucs_by_attrs", "dpds", "eods", "between_group_disparity"):
        assert np.allclose(dev_full[k], host_full[k], rtol=0, atol=1e-12), k
    for a, b in zip(dev_full["aucs_by_attrs"], host_full["aucs_by_attrs"]):
        assert np.allclose(a, b, rtol=0, atol=1e-12)


@pytest.mark.parametrize("N,G,unknown,ties", [(1, 2, 0, True), (37, 3, 0, True), (256, 2, 5, True), (1000, 3, 11, True),
                                              (4099, 8, 0, True), (20000, 3, 7, False), (20000, 2, 0, True)])
def test_eval_counts_sorted_equals_pairs(N, G, unknown, ties):
    """ffm_eval_counts_sorted (O(N log N): sort by score, prefix counts of the negatives, two binary searches per
    positive) gives the integers of the all-pairs kernel, ties and the 'unknown' group included; and of the brute force."""
    from fairfedmed_amd import ops
    prob, y, a = make(N, G, seed=N + G, ties=ties, unknown=unknown)
    args = (torch.from_numpy(prob).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(a).cuda(), G)
    pairs = ops.eval_counts(*args, method="pairs").cpu().numpy()
    srt = ops.eval_counts(*args, method="sort").cpu().numpy()
    assert np.array_equal(srt, pairs)
    if N <= 4099:
        assert np.array_equal(srt, _brute_counts(prob, y, a, G))
    no_attr = ops.eval_counts(args[0], args[1], None, G, method="sort").cpu().numpy()
    assert np.array_equal(no_attr[-1], srt[-1]) and np.array_equal(no_attr[-2], no_attr[-1]) and not no_attr[:-2].any()


def test_eval_counts_200k_samples_sorted():
    """A CheXpert-sized test set: 2 * 10^5 samples = 2 * 10^10 score pairs.  The sorted evaluator's AUC
    equals the host's mid-rank AUC; 'auto' picks it from 32768 samples on."""
    import time
    from fairfedmed_amd import ops
    prob, y, a = make(200000, 3, seed=5, ties=False)
    args = (torch.from_numpy(prob).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(a).cuda(), 3)
    ops.eval_counts(*args)                                                  # warm-up (workspace query, code load)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t = ops.eval_counts(*args)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = t.cpu().numpy()
    assert abs(M.basic_from_counts(t)[3] - 100.0 * M.auc_macro_ovr(prob, y)) < 1e-9
    ga = M.group_aucs(prob, y, a)
    for g in range(3):
        assert abs(M._auc_from_row(t[g]) - ga[g]) < 1e-12
    assert t[-1, 0] + t[-1, 1] == 200000
    print(f"200k-sample evaluator: {dt * 1e3:.2f} ms")
    assert dt < 0.05
