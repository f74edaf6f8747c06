"""RN50 trunk building blocks (csrc/conv.hip) against PyTorch on the GPU: 3x3 convolution as im2col + ffm_gemm_nt and
its input gradient, train-/eval-mode BatchNorm forward and backward, AvgPool2d(2), the attention-pool tokens."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
F = torch.nn.functional
DT = [torch.float32, torch.bfloat16, torch.float16]
IDS = ["f32", "bf16", "f16"]


def tol(dt):
    return 3e-5 if dt == torch.float32 else 1.5e-2


def rel(got, ref):
    got, ref = got.double(), ref.double()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def rnd(*shape, dt=torch.float32, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, device="cuda", generator=g) * scale).to(dt)


@pytest.fixture(scope="module")
def ops():
    from fairfedmed_amd import ops
    return ops


def nhwc(x):      # [B,C,H,W] -> rows [B*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def nchw(r, B, H, W):
    return r.reshape(B, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("dt", DT, ids=IDS)
@pytest.mark.parametrize("B,H,C,Co,stride", [(2, 16, 64, 64, 1), (3, 14, 32, 64, 1), (2, 16, 32, 64, 2)])
def test_conv3x3_forward_and_input_gradient(ops, dt, B, H, C, Co, stride):
    x = rnd(B, C, H, H, dt=dt, seed=1)
    w = rnd(Co, C, 3, 3, dt=dt, scale=(9 * C) ** -0.5, seed=2)
    Ho = ops.conv_out(H, stride)
    ke = 64 if dt != torch.float32 else 32
    Kp = (9 * C + ke - 1) // ke * ke
    cols = torch.empty(B * Ho * Ho, Kp, device="cuda", dtype=dt)
    ops.im2col3x3(nhwc(x), cols, B, H, H, stride)
    wk = torch.zeros(Co, Kp, device="cuda", dtype=dt)
    wk[:, :9 * C] = w.permute(0, 2, 3, 1).reshape(Co, 9 * C)            # [Cout, ky, kx, c]
    y = torch.empty(B * Ho * Ho, Co, device="cuda", dtype=dt)
    ops.gemm_nt(cols, wk, y)
    ref = F.conv2d(x.double(), w.double(), stride=stride, padding=1)
    assert rel(nchw(y, B, Ho, Ho), ref) < tol(dt)
    # dX: dcols = dY W, then the gather
    dy = rnd(B, Co, Ho, Ho, dt=dt, seed=3)
    wt = torch.zeros(Kp, Co, device="cuda", dtype=dt)
    wt[:9 * C] = wk[:, :9 * C].t()
    dcols = torch.empty(B * Ho * Ho, Kp, device="cuda", dtype=dt)
    ops.gemm_nt(nhwc(dy), wt, dcols)
    dx = torch.empty(B * H * H, C, device="cuda", dtype=dt)
    ops.col2im3x3(dcols, dx, B, H, H, stride)
    xr = x.double().requires_grad_()
    F.conv2d(xr, w.double(), stride=stride, padding=1).backward(dy.double())
    assert rel(nchw(dx, B, H, H), xr.grad) < tol(dt) * 2


@pytest.mark.parametrize("dt", DT, ids=IDS)
def test_stem_im2col(ops, dt):
    B, H, Co = 2, 32, 32
    img = torch.rand(B, 3, H, H, device="cuda", generator=torch.Generator("cuda").manual_seed(5)) * 255
    mean3, std3 = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    w = rnd(Co, 3, 3, 3, dt=dt, scale=27 ** -0.5, seed=6)
    Ho = ops.conv_out(H, 2)
    cols = torch.empty(B * Ho * Ho, 64, device="cuda", dtype=dt)
    ops.stem_im2col(img, cols, 2, mean3, std3)
    wk = torch.zeros(Co, 64, device="cuda", dtype=dt)
    wk[:, :27] = w.permute(0, 2, 3, 1).reshape(Co, 27)
    y = torch.empty(B * Ho * Ho, Co, device="cuda", dtype=dt)
    ops.gemm_nt(cols, wk, y)
    xn = (img / 255.0 - torch.tensor(mean3, device="cuda").view(1, 3, 1, 1)) / torch.tensor(std3, device="cuda").view(1, 3, 1, 1)
    ref = F.conv2d(xn.double(), w.double(), stride=2, padding=1)
    assert rel(nchw(y, B, Ho, Ho), ref) < tol(dt)


@pytest.mark.parametrize("dt", DT, ids=IDS)
@pytest.mark.parametrize("rows,C,relu,with_res", [(2 * 16 * 16, 64, True, False), (1000, 256, False, False), (3 * 49, 2048, True, True),
                                                   (6272, 256, True, True), (7168, 128, True, False), (7169, 64, True, True),
                                                   (12000, 256, False, False),
                                                   # round 5: maps of up to 32 768 rows take the folded path (no finalize launch);
                                                   # the rows below pin both sides of its selection rule (csrc/conv.hip: bn_fold_geom)
                                                   (40000, 64, True, True),        # large map: colsum + finalize (4-channel blocks) + apply
                                                   (25088, 128, True, False),      # mid-sized, few strips: stays on the three launches
                                                   (25088, 512, True, True),       # mid-sized, 8 strips x 64 row groups: folded
                                                   (1568, 2048, True, True),       # layer4: 32 strips
                                                   (300, 24, False, False),        # a strip narrower than 64 channels, C % 16 != 0
                                                   (520, 200, True, False)])       # last strip ragged (200 = 3 x 64 + 8)
def test_batchnorm_train_forward_backward_and_eval(ops, dt, rows, C, relu, with_res):
    x = rnd(rows, C, dt=dt, seed=7) * 1.5 + 0.3
    gamma, beta = 1 + 0.1 * rnd(C, seed=8), 0.1 * rnd(C, seed=9)
    rm, rv = 0.05 * rnd(C, seed=10), 1 + 0.1 * rnd(C, seed=11).abs()
    res = rnd(rows, C, dt=dt, seed=12) if with_res else None
    mean, rstd = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    part = torch.empty(ops.bn_blocks(rows) * 2 * C, device="cuda")
    y = torch.empty_like(x)
    rm1, rv1 = rm.clone(), rv.clone()
    ops.bn_fwd(x, gamma, beta, rm1, rv1, mean, rstd, part, y, True, relu, res)
    xr = x.double().requires_grad_()
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    rmr, rvr = rm.double().clone(), rv.double().clone()
    ref = F.batch_norm(xr, rmr, rvr, gr, br, True, 0.1, 1e-5)
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    assert rel(y, ref.detach()) < tol(dt)
    assert rel(rm1, rmr) < 1e-5 and rel(rv1, rvr) < 1e-5
    # backward (through the ReLU mask when there is one)
    dy = rnd(rows, C, dt=dt, seed=13)
    ref.backward(dy.double())
    k12, dg, db = torch.empty(2 * C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dx = torch.empty_like(x)
    gout = torch.full_like(x, float("nan"))
    ops.bn_bwd(dy, y if relu else None, x, gamma, mean, rstd, part, k12, dg, db, dx, g_out=gout)
    t = tol(dt) * (3 if dt == torch.float32 else 1)
    assert rel(dx, xr.grad) < t and rel(dg, gr.grad) < t and rel(db, br.grad) < t
    assert torch.equal(gout, dy * (y > 0) if relu else dy)        # g_out: the ReLU-masked gradient (ffm_relu_bwd's result)
    # eval mode: running statistics, nothing updated
    rm2, rv2 = rm.clone(), rv.clone()
    ops.bn_fwd(x, gamma, beta, rm2, rv2, mean, rstd, None, y, False, False)
    ref_e = F.batch_norm(x.double(), rm.double(), rv.double(), gamma.double(), beta.double(), False, 0.1, 1e-5)
    assert rel(y, ref_e) < tol(dt) and torch.equal(rm2, rm) and torch.equal(rv2, rv)


@pytest.mark.parametrize("dt", DT, ids=IDS)
def test_avgpool_add_relu_and_attnpool_tokens(ops, dt):
    B, H, C = 3, 8, 64
    x = rnd(B, C, H, H, dt=dt, seed=20)
    out = torch.empty(B * 16, C, device="cuda", dtype=dt)
    ops.avgpool2(nhwc(x), out, B, H, H)
    assert rel(nchw(out, B, 4, 4), F.avg_pool2d(x.double(), 2)) < tol(dt)
    dp = rnd(B * 16, C, dt=dt, seed=21)
    dx = torch.empty(B * H * H, C, device="cuda", dtype=dt)
    ops.avgpool2(dp, dx, B, H, H, backward=True)
    xr = x.double().requires_grad_()
    F.avg_pool2d(xr, 2).backward(nchw(dp, B, 4, 4).double())
    assert rel(nchw(dx, B, H, H), xr.grad) < tol(dt)
    a, b = rnd(640, C, dt=dt, seed=22), rnd(640, C, dt=dt, seed=23)
    o = torch.empty_like(a)
    ops.add(a, b, o)
    assert rel(o, a.double() + b.double()) < tol(dt)
    ops.relu_bwd(a, b, o)
    assert rel(o, a.double() * (b.double() > 0)) < 1e-7
    # attention-pool tokens
    HW, E = 4, 128
    feat = rnd(B * HW, E, dt=dt, seed=24)
    pos = rnd(HW + 1, E, dt=dt, scale=0.1, seed=25)
    tok = torch.empty(B * (HW + 1), E, device="cuda", dtype=dt)
    ops.attnpool_tokens(feat, pos, tok, B, HW)
    fr = feat.double().reshape(B, HW, E).requires_grad_()
    ref = torch.cat([fr.mean(1, keepdim=True), fr], 1) + pos.double()
    assert rel(tok.reshape(B, HW + 1, E), ref.detach()) < tol(dt)
    dt_ = rnd(B * (HW + 1), E, dt=dt, seed=26)
    ref.backward(dt_.double().reshape(B, HW + 1, E))
    df = torch.empty_like(feat)
    ops.attnpool_tokens(dt_, None, df, B, HW, backward=True)
    assert rel(df.reshape(B, HW, E), fr.grad) < tol(dt)


def _rows_fwd(w, Kp):
    """[Co, Ci, 3, 3] -> [Co, Kp], k = (ky*3 + kx)*Ci + c."""
    r = w.float().permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    out = torch.zeros(w.shape[0], Kp, device=w.device)
    out[:, :r.shape[1]] = r
    return out


def _rows_bwd(w, Kp):
    """[Co, Ci, 3, 3] -> [Ci, Kp], k = (ky'*3 + kx')*Co + co holding w[co, ci, 2-ky', 2-kx']: dX = conv3x3(dY; w')."""
    r = w.float().flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], -1)
    out = torch.zeros(w.shape[1], Kp, device=w.device)
    out[:, :r.shape[1]] = r
    return out


@pytest.mark.parametrize("dt", DT, ids=IDS)
@pytest.mark.parametrize("B,H,W,C,Co", [(2, 16, 16, 64, 64), (3, 14, 14, 32, 64), (2, 8, 12, 32, 32), (1, 7, 7, 128, 64),
                                        (5, 28, 28, 64, 64), (2, 4, 4, 512, 512), (8, 7, 7, 512, 512),
                                        (4, 14, 14, 256, 256)])
def test_implicit_gemm_conv3x3_forward_and_input_gradient(ops, dt, B, H, W, C, Co):
    """ffm_conv3x3_nhwc: the patches are never materialised; forward and (with the re-ordered weight) the input
    gradient against F.conv2d and its autograd, incl. non-square maps, tile-ragged pixel counts and padded K."""
    kq = 64 if dt != torch.float32 else 32
    rup = lambda v: (v + kq - 1) // kq * kq
    x = rnd(B, C, H, W, dt=dt, seed=1)
    w = rnd(Co, C, 3, 3, dt=dt, scale=1.0 / math.sqrt(9 * C), seed=2)
    zeros = torch.zeros(64, device="cuda", dtype=dt)
    scratch = torch.zeros(8 * B * H * W * max(C, Co), device="cuda")       # split-K kicks in for few tiles and a long K
    xr = x.float().requires_grad_(True)
    ref = F.conv2d(xr, w.float(), padding=1)
    y = torch.empty(B * H * W, Co, device="cuda", dtype=dt)
    ops.conv3x3(nhwc(x), _rows_fwd(w, rup(9 * C)).to(dt), y, B, H, W, zeros)
    assert rel(nchw(y.float(), B, H, W), ref.detach()) < tol(dt)
    y2 = torch.full_like(y, float("nan"))
    ops.conv3x3(nhwc(x), _rows_fwd(w, rup(9 * C)).to(dt), y2, B, H, W, zeros, scratch)
    assert rel(nchw(y2.float(), B, H, W), ref.detach()) < tol(dt)
    g = rnd(B, Co, H, W, dt=dt, seed=3)
    ref.backward(g.float())
    dx = torch.empty(B * H * W, C, device="cuda", dtype=dt)
    ops.conv3x3(nhwc(g), _rows_bwd(w, rup(9 * Co)).to(dt), dx, B, H, W, zeros, scratch)
    assert rel(nchw(dx.float(), B, H, W), xr.grad) < tol(dt)


@pytest.mark.parametrize("dt", DT, ids=IDS)
@pytest.mark.parametrize("M,N,K", [(1000, 256, 64), (128 * 7, 64, 256), (3 * 49, 2048, 512), (5000, 192, 128),
                                   (6272, 1024, 64),      # folded, 49 producer tiles x 16 strips
                                   (40000, 64, 64)])      # 313 producer tiles: the finalize launch (4-channel blocks)
def test_gemm_column_sums_feed_batchnorm(ops, dt, M, N, K):
    """ffm_gemm_args.colstat_part: the 128x128 kernel leaves sum / sum of squares of each row tile's STORED columns
    (ragged last tile included); ffm_bn_fwd(part_rows) on them gives the result of its own pass over the tensor."""
    a, b = rnd(M, K, dt=dt, seed=1), rnd(N, K, dt=dt, scale=K ** -0.5, seed=2)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    t = (M + 127) // 128
    st = torch.full((t, 2, N), float("nan"), device="cuda")
    ops.gemm_nt(a, b, out, colstats=st)
    o = out.double()
    for i in range(t):
        blk = o[i * 128:(i + 1) * 128]
        assert rel(st[i, 0], blk.sum(0)) < 1e-5 and rel(st[i, 1], (blk * blk).sum(0)) < 1e-5
    gamma, beta = 1 + 0.1 * rnd(N, seed=8), 0.1 * rnd(N, seed=9)
    ys, ms, rs, rms, rvs = [], [], [], [], []
    for pr in (0, t):
        mean, rstd = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
        part = st.clone().flatten() if pr else torch.empty(ops.bn_blocks(M) * 2 * N, device="cuda")
        y = torch.empty_like(out)
        ops.bn_fwd(out, gamma, beta, rm, rv, mean, rstd, part, y, True, True, part_rows=pr)
        ys.append(y); ms.append(mean); rs.append(rstd); rms.append(rm); rvs.append(rv)
    assert rel(ms[1], ms[0]) < 1e-5 and rel(rs[1], rs[0]) < 1e-5 and rel(rms[1], rms[0]) < 1e-5 and rel(rvs[1], rvs[0]) < 1e-5
    assert rel(ys[1], ys[0]) < (1e-5 if dt == torch.float32 else 8e-3)


@pytest.mark.parametrize("dt", DT, ids=IDS)
def test_conv3x3_column_sums(ops, dt):
    """ffm_conv3x3_nhwc(colstat_part): partial rows as ffm_conv3x3_colstat_rows says, none when the launch splits K."""
    kq = 64 if dt != torch.float32 else 32
    rup = lambda v: (v + kq - 1) // kq * kq
    zeros = torch.zeros(64, device="cuda", dtype=dt)
    for (B, H, C, Co, sc) in [(5, 28, 64, 64, False), (4, 14, 256, 256, True), (2, 16, 32, 64, True)]:
        x = rnd(B, C, H, H, dt=dt, seed=1)
        w = rnd(Co, C, 3, 3, dt=dt, scale=1.0 / math.sqrt(9 * C), seed=2)
        scratch = torch.zeros(8 * B * H * H * max(C, Co), device="cuda") if sc else None
        M = B * H * H
        y = torch.empty(M, Co, device="cuda", dtype=dt)
        st = torch.full((max((M + 127) // 128, (M + 7) // 8), 2, Co), float("nan"), device="cuda")
        pr = ops.conv3x3(nhwc(x), _rows_fwd(w, rup(9 * C)).to(dt), y, B, H, H, zeros, scratch, colstats=st)
        ref = F.conv2d(x.float(), w.float(), padding=1)
        assert rel(nchw(y.float(), B, H, H), ref) < tol(dt)
        R = 128
        if (B, H, C) == (4, 14, 256):
            R = 8                                                     # 7 x 2 tiles, K = 2304: split over K - the sums leave
            st = st.flatten()[:2 * Co * ((M + 7) // 8)].view(-1, 2, Co)   # with its reduction, 8 rows per partial row (M < 4096)
        assert pr == (M + R - 1) // R
        o = y.double()
        for i in range(pr):
            blk = o[i * R:(i + 1) * R]
            assert rel(st[i, 0], blk.sum(0)) < 1e-5 and rel(st[i, 1], (blk * blk).sum(0)) < 1e-5


def test_narrow_conv_tiles_on_wide_outputs():
    """FFM_CONV_NARROW=<t> (read once per process) sends N = 128 / 256 / 512 convolutions through the 128 x 64 tile kernel as
    several column tiles: the implicit-GEMM and column-sum tests again in a child process with that switch."""
    import os, subprocess, sys
    env = dict(os.environ, FFM_CONV_NARROW="1000000")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(root, "tests", "test_conv_gpu.py"), "-k",
                        "implicit_gemm or column_sums"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dt", DT, ids=IDS)
@pytest.mark.parametrize("M,N,K,relu,lora", [(1000, 256, 64, True, False), (6272, 256, 1024, True, True), (3 * 49, 512, 2048, False, True),
                                            (25088, 128, 512, True, True), (40000, 64, 256, True, True)])
def test_gemm_leaves_batchnorm_backward_sums(ops, dt, M, N, K, relu, lora):
    """FFM_EPI_BNBWD: the dX product whose output is dL/dy of a train-mode BatchNorm (+ ReLU) leaves {sum g, sum g xhat} of
    its row tiles in colstat_part (g = stored output x (ReLU output > 0)), and ffm_bn_bwd(part_rows) on them gives what its
    own pass over dy / x / mask gives - folded small maps and the three-launch large ones alike (clip/model.py:41-60:
    conv3's dX feeds bn2's backward)."""
    r, G = 8, 2
    a, b = rnd(M, K, dt=dt, seed=1), rnd(N, K, dt=dt, scale=K ** -0.5, seed=2)
    x = rnd(M, N, dt=dt, seed=3) * 1.5 + 0.2                       # the BatchNorm's input
    gamma, beta = 1 + 0.1 * rnd(N, seed=4), 0.1 * rnd(N, seed=5)
    mean, rstd = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    y = torch.empty_like(x)
    ops.bn_fwd(x, gamma, beta, rm, rv, mean, rstd, torch.empty(ops.bn_blocks(M) * 2 * N, device="cuda"), y, True, relu)
    mask = y if relu else None
    t = (M + 127) // 128
    st = torch.full((t, 2, N), float("nan"), device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=dt)
    kw = {}
    if lora:
        P, S, lw = rnd(K, r, scale=0.1, seed=6), rnd(G, r, seed=7), rnd(N, r, seed=8)
        attr = torch.randint(0, G, ((M + 48) // 49,), device="cuda", dtype=torch.int32)
        rk = torch.zeros(16, K, device="cuda", dtype=dt)
        ops.PackPlan([(P, False, rk)], dt, "cuda").run()
        ro = ops.RankOp(rk, S, attr, 49, 0.25, 0.7, ts_out=torch.empty(M, r, device="cuda"))
        kw = dict(lw=lw, lw_is_kr=True, rankop=ro)
    # (with the FairLoRA epilogue also the residual form - conv1's dX + the identity path's gradient - and the masked
    # gradient as a second output: what the dX product of a Bottleneck's conv1 leaves for the block in front of it)
    gout = torch.full((M, N), float("nan"), device="cuda", dtype=dt) if lora else None
    if lora:
        kw["res"] = rnd(M, N, dt=dt, seed=9)
    ops.gemm_nt(a, b, out, colstats=st, bnbwd=(x, mask, mean, rstd, gout), **kw)
    out0 = torch.empty_like(out)
    if lora:
        kw["rankop"] = ops.RankOp(rk, S, attr, 49, 0.25, 0.7, ts_out=torch.empty(M, r, device="cuda"))
    ops.gemm_nt(a, b, out0, **kw)
    assert torch.equal(out, out0)                                  # the stored product does not change
    if lora:
        assert torch.equal(gout, out * (y > 0) if relu else out)
    g = out.double() * ((y.double() > 0) if relu else 1.0)
    xh = (x.double() - mean.double()) * rstd.double()
    for i in range(t):
        sl = slice(i * 128, (i + 1) * 128)
        assert rel(st[i, 0], g[sl].sum(0)) < 2e-5 and rel(st[i, 1], (g[sl] * xh[sl]).sum(0)) < 2e-5, i
    res = []
    for pr in (0, t):
        k12, dg, db = torch.empty(2 * N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        dx = torch.empty_like(x)
        part = st.clone().flatten() if pr else torch.empty(ops.bn_blocks(M) * 2 * N, device="cuda")
        ops.bn_bwd(out, mask, x, gamma, mean, rstd, part, k12, dg, db, dx, part_rows=pr)
        res.append((dx, dg, db))
    for u, v in zip(res[0], res[1]):
        assert rel(v, u) < (2e-5 if dt == torch.float32 else 8e-3)
    # a flag combination without a kernel: refused, not silently ignored
    with pytest.raises(RuntimeError):
        ops.gemm_nt(a, b, out, bias=gamma, colstats=st, bnbwd=(x, mask, mean, rstd))


@pytest.mark.parametrize("dt", DT, ids=IDS)
def test_conv3x3_leaves_batchnorm_backward_sums(ops, dt):
    """ffm_conv3x3_nhwc_bnbwd: the implicit-GEMM convolution whose output is dL/dy of a BatchNorm (+ ReLU) - conv2's dX feeding
    bn1's backward in a Bottleneck - leaves {sum g, sum g xhat} per row tile: the 128 x 128 kernel (N = 128), both narrow
    kernels (N = 64 / 32), with and without a ReLU mask, and - for a launch split over K - the reduction kernel behind it."""
    kq = 64 if dt != torch.float32 else 32
    rup = lambda v: (v + kq - 1) // kq * kq
    zeros = torch.zeros(64, device="cuda", dtype=dt)
    for (B, H, C, Co, sc, relu) in [(5, 28, 64, 128, False, True), (5, 28, 64, 64, False, True), (2, 16, 64, 32, True, False),
                                    (3, 12, 32, 32, False, True),
                                    (4, 14, 256, 256, True, True)]:
        xin = rnd(B, C, H, H, dt=dt, seed=1)
        w = rnd(Co, C, 3, 3, dt=dt, scale=1.0 / math.sqrt(9 * C), seed=2)
        scratch = torch.zeros(8 * B * H * H * max(C, Co), device="cuda") if sc else None
        M = B * H * H
        bx = rnd(M, Co, dt=dt, seed=3) * 1.3 + 0.1                      # the BatchNorm's input rows
        mean, rstd = bx.float().mean(0), 1.0 / torch.sqrt(bx.float().var(0, unbiased=False) + 1e-5)
        mask = (rnd(M, Co, dt=dt, seed=4) if relu else None)             # its ReLU output (only the sign matters)
        y, y0 = torch.empty(M, Co, device="cuda", dtype=dt), torch.empty(M, Co, device="cuda", dtype=dt)
        st = torch.full((max((M + 127) // 128, (M + 7) // 8), 2, Co), float("nan"), device="cuda")
        wr = _rows_fwd(w, rup(9 * C)).to(dt)
        pr = ops.conv3x3(nhwc(xin), wr, y, B, H, H, zeros, scratch, colstats=st, bnbwd=(bx, mask, mean, rstd))
        ops.conv3x3(nhwc(xin), wr, y0, B, H, H, zeros, scratch)
        assert torch.equal(y, y0)
        split = (B, H, C) == (4, 14, 256) or ((B, H, C, Co) == (2, 16, 64, 32) and dt == torch.float32)   # (f32: 18 K tiles)
        R = 8 if split else 128                                       # split over K: the sums leave with the reduction (M < 4096)
        assert pr == (M + R - 1) // R
        st = st.flatten()[:2 * Co * pr].view(pr, 2, Co)
        g = y.double() * ((mask.double() > 0) if relu else 1.0)
        xh = (bx.double() - mean.double()) * rstd.double()
        for i in range(pr):
            sl = slice(i * R, (i + 1) * R)
            assert rel(st[i, 0], g[sl].sum(0)) < 2e-5 and rel(st[i, 1], (g[sl] * xh[sl]).sum(0)) < 2e-5, (Co, i)
