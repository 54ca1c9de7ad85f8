"""HIP kernels (through the C ABI) vs the oracle on identical seeded inputs.

fp32 tolerance for the MFMA GEMM paths: the kernels accumulate in fp32 in k order (v_mfma_f32_32x32x2_f32 is a
k-ordered fmaf chain), the oracle in fp64 -> relative error ~1e-6*sqrt(K); asserted as
|gpu - ref| <= 2e-4 * max|ref| + 1e-5 (stated tolerance of this suite).  Integer / index outputs: bit-exact.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    from radnet_hip import lib as L
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test on a machine without a GPU")
    c = L.Context(0)
    ws = torch.empty(128 << 20, dtype=torch.uint8, device="cuda")
    c.check(c.lib.radnet_set_workspace(c.h, ws.data_ptr(), ws.numel()), "ws")
    c._ws = ws
    yield c
    c.close()


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def close(gpu, ref, rtol=2e-4, atol=1e-5):
    ref = np.asarray(ref, dtype=np.float64)
    gpu = np.asarray(gpu, dtype=np.float64)
    tol = rtol * np.abs(ref).max() + atol
    err = np.abs(gpu - ref).max()
    assert err <= tol, "max err %g > tol %g" % (err, tol)


def conv_desc(L, x, w, y, nb, h, wd, c, oh, ow, k, stride, pad, n, ldw, scale=None, shift=None, addend=None, act=0, act_cols=0):
    d = L.ConvDesc()
    d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    d.scale = scale.data_ptr() if scale is not None else None
    d.shift = shift.data_ptr() if shift is not None else None
    d.addend = addend.data_ptr() if addend is not None else None
    d.nb, d.h, d.w_, d.c, d.oh, d.ow = nb, h, wd, c, oh, ow
    d.kh = d.kw = k
    d.stride, d.pad_t, d.pad_l, d.n = stride, pad, pad, n
    d.ldw, d.ldy, d.ld_add, d.act, d.act_cols = ldw, n, n, act, act_cols
    return d


CONV_CASES = [
    # nb, h, w, cin, cout, k, stride, pad, relu, residual
    (1, 19, 23, 64, 64, 1, 1, 0, True, False),
    (1, 19, 23, 64, 256, 1, 1, 0, True, True),
    (1, 19, 23, 64, 64, 3, 1, 1, True, False),
    (1, 21, 30, 256, 128, 1, 2, 0, True, False),
    (1, 38, 63, 256, 256, 3, 1, 1, True, False),       # stage-4 3x3 at full cfg-2 size (split-K path)
    (3, 14, 14, 1024, 512, 1, 2, 0, True, False),      # res5a 2a on 3 RoIs
    (3, 7, 7, 512, 512, 3, 1, 1, True, False),         # res5 3x3 (small M, large K)
    (2, 9, 11, 128, 96, 3, 1, 1, False, True),         # N not a multiple of the tile
    (1, 40, 47, 4, 64, 7, 2, 3, True, False),          # stem: 4-channel padded image
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(ctx, case):
    from radnet_hip import lib as L
    from oracle import dense
    nb, h, w, cin, cout, k, stride, pad, relu, residual = case
    rs = np.random.RandomState(hash(case) % (2 ** 31))
    x = rs.standard_normal((nb, h, w, cin)).astype(np.float32)
    if cin == 4:
        x[..., 3] = 0
    wt = (rs.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rs.standard_normal(cout).astype(np.float32)
    scale = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    oh = (h + 2 * pad - k) // stride + 1
    ow = (w + 2 * pad - k) // stride + 1
    res = rs.standard_normal((nb, oh, ow, cout)).astype(np.float32) if residual else None
    ref = dense.conv2d(x.astype(np.float64), wt.astype(np.float64), None, stride, (pad, pad, pad, pad)) * scale + b
    if residual:
        ref = ref + res
    if relu:
        ref = np.maximum(ref, 0)
    xd, wd, sd, bd = dev(x), dev(wt.reshape(-1, cout)), dev(scale), dev(b)
    y = torch.full((nb, oh, ow, cout), float("nan"), dtype=torch.float32, device="cuda")
    rd = dev(res) if residual else None
    d = conv_desc(L, xd, wd, y, nb, h, w, cin, oh, ow, k, stride, pad, cout, cout, sd, bd, rd, 1 if relu else 0)
    ctx.check(ctx.lib.radnet_conv_fwd(ctx.h, C.byref(d)), "conv_fwd")
    ctx.sync()
    close(y.cpu().numpy(), ref)
    from tolerances import check
    check(y.cpu().numpy(), ref, 2e-4, "conv_fwd %s" % (case,))        # + per output channel and RMS (tests/tolerances.py)


def test_conv_fwd_sigmoid_head_columns(ctx):
    from radnet_hip import lib as L
    from oracle import dense
    rs = np.random.RandomState(5)
    x = rs.standard_normal((1, 10, 13, 512)).astype(np.float32)
    wt = np.zeros((512, 64), np.float32)
    wt[:, :60] = rs.standard_normal((512, 60)) * 0.05
    b = np.zeros(64, np.float32); b[:60] = rs.standard_normal(60) * 0.1
    y = torch.zeros(130, 64, device="cuda")
    xd, wd, bd = dev(x), dev(wt), dev(b)
    d = conv_desc(L, xd, wd, y, 1, 10, 13, 512, 10, 13, 1, 1, 0, 64, 64, None, bd, None, act=2, act_cols=12)
    ctx.check(ctx.lib.radnet_conv_fwd(ctx.h, C.byref(d)), "conv_fwd")
    z = x.reshape(130, 512).astype(np.float64) @ wt + b
    ref = z.copy(); ref[:, :12] = 1 / (1 + np.exp(-z[:, :12]))
    close(y.cpu().numpy(), ref)


DGRAD_CASES = [(2, 7, 7, 512, 512, 3, 1), (2, 7, 7, 512, 2048, 1, 0), (1, 13, 17, 2048, 512, 1, 0), (1, 12, 9, 512, 64, 1, 0), (1, 38, 63, 128, 128, 3, 1)]


def test_autotune_adopts_the_nearest_measured_shape(ctx):
    """radnet_set_autotune(2): the first 1x1 conv at M = 40*50 is measured; the same layer at M = 40*53 (within M/4) adopts
    its launch shape without measuring (one more cached shape, far fewer launches), at M = 40*80 it is measured again.  All
    three results against the oracle."""
    import time
    from radnet_hip import lib as L
    from oracle import dense
    rs = np.random.RandomState(12)
    cin, cout = 192, 160                                   # a shape no other test of this module uses
    wt = (rs.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    wd = dev(wt.reshape(-1, cout))
    n0 = ctx.lib.radnet_tuned_shapes(ctx.h)
    ctx.check(ctx.lib.radnet_set_autotune(ctx.h, 2), "autotune 2")
    took = []
    try:
        for w_ in (50, 53, 80):
            x = rs.standard_normal((1, 40, w_, cin)).astype(np.float32)
            xd = dev(x)
            y = torch.full((1, 40, w_, cout), float("nan"), dtype=torch.float32, device="cuda")
            d = conv_desc(L, xd, wd, y, 1, 40, w_, cin, 40, w_, 1, 1, 0, cout, cout)
            ctx.sync()
            t0 = time.perf_counter()
            ctx.check(ctx.lib.radnet_conv_fwd(ctx.h, C.byref(d)), "conv_fwd")
            ctx.sync()
            took.append(time.perf_counter() - t0)
            close(y.cpu().numpy(), dense.conv2d(x.astype(np.float64), wt.astype(np.float64), None, 1, (0, 0, 0, 0)))
    finally:
        ctx.check(ctx.lib.radnet_set_autotune(ctx.h, 0), "autotune off")
    assert ctx.lib.radnet_tuned_shapes(ctx.h) == n0 + 3
    assert took[1] * 5 < took[0] and took[1] * 5 < took[2], took      # adopted, not measured


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv_dgrad_wgrad(ctx, case):
    """dgrad (with BN scale on dy, residual-path add and the producer's ReLU mask fused) and wgrad + bias colsum."""
    from radnet_hip import lib as L
    from oracle import dense
    nb, h, w, cin, cout, k, pad = case
    rs = np.random.RandomState(sum(case))
    x = np.maximum(rs.standard_normal((nb, h, w, cin)), 0).astype(np.float32)          # post-ReLU producer output
    wt = (rs.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    dy = rs.standard_normal((nb, h, w, cout)).astype(np.float32)
    gs = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    add = rs.standard_normal((nb, h, w, cin)).astype(np.float32)
    dx_ref, dw_ref, db_ref = dense.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), (dy * gs).astype(np.float64), 1, (pad,) * 4)
    dx_ref = (dx_ref + add) * (x > 0)
    xd, wd, dyd, gsd, addd = dev(x), dev(wt.reshape(-1, cout)), dev(dy), dev(gs), dev(add)
    dx = torch.full((nb, h, w, cin), float("nan"), device="cuda")
    dw = torch.full((k * k * cin, cout), float("nan"), device="cuda")
    db = torch.full((cout,), float("nan"), device="cuda")
    d = conv_desc(L, xd, wd, dx, nb, h, w, cin, h, w, k, 1, pad, cout, cout)
    d.dy, d.ld_dy, d.gscale = dyd.data_ptr(), cout, gsd.data_ptr()
    d.dx, d.ld_dx, d.dx_add, d.ld_dx_add, d.dx_mask, d.ld_dx_mask = dx.data_ptr(), cin, addd.data_ptr(), cin, xd.data_ptr(), cin
    d.dw, d.dw_accumulate = dw.data_ptr(), 0
    ctx.check(ctx.lib.radnet_conv_dgrad(ctx.h, C.byref(d)), "dgrad")
    ctx.check(ctx.lib.radnet_conv_wgrad(ctx.h, C.byref(d)), "wgrad")
    ctx.call("radnet_colsum", dyd, nb * h * w, cout, cout, gsd, db, 0)
    ctx.sync()
    close(dx.cpu().numpy(), dx_ref)
    close(dw.cpu().numpy(), dw_ref.reshape(-1, cout))
    close(db.cpu().numpy(), db_ref)
    # accumulate mode adds on top; the bias gradient rides in the same launch when the descriptor names it
    db2 = torch.full((cout,), float("nan"), device="cuda")
    d.db = db2.data_ptr()
    ctx.check(ctx.lib.radnet_conv_wgrad(ctx.h, C.byref(d)), "wgrad")
    close(db2.cpu().numpy(), db_ref)
    d.dw_accumulate = 1
    ctx.check(ctx.lib.radnet_conv_wgrad(ctx.h, C.byref(d)), "wgrad")
    close(dw.cpu().numpy(), 2 * dw_ref.reshape(-1, cout))
    close(db2.cpu().numpy(), 2 * db_ref)


@pytest.mark.parametrize("case", [(3, 7, 7, 512, 512, 3, 1), (3, 7, 7, 2048, 512, 1, 0), (1, 19, 23, 128, 192, 3, 1), (2, 9, 11, 64, 96, 1, 0)])
@pytest.mark.parametrize("shape", [(64, 64, 1), (64, 64, 3), (64, 64, -2), (128, 64, 1), (32, 64, 1), (32, 64, 2), None])
def test_conv_bwd_one_launch_equals_the_two_launches(ctx, case, shape):
    """radnet_conv_bwd (weight gradient + data gradient of a layer from one descriptor): with 64x64 tiles -- un-split, K-split
    with the in-launch reduction, XCD-ordered -- the two problems share ONE launch (conv_bwd_pair_kernel); with other shapes
    (128x64) and with the heuristic choice (None) it issues the two launches.  Either way: dx (+ residual add, ReLU mask), dw,
    db against the oracle, and accumulate mode adds on top."""
    from radnet_hip import lib as L
    from oracle import dense
    nb, h, w, cin, cout, k, pad = case
    rs = np.random.RandomState(sum(case) + 3)
    x = np.maximum(rs.standard_normal((nb, h, w, cin)), 0).astype(np.float32)
    wt = (rs.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    dy = rs.standard_normal((nb, h, w, cout)).astype(np.float32)
    gs = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    add = rs.standard_normal((nb, h, w, cin)).astype(np.float32)
    dx_ref, dw_ref, db_ref = dense.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), (dy * gs).astype(np.float64), 1, (pad,) * 4)
    dx_ref = (dx_ref + add) * (x > 0)
    xd, wd, dyd, gsd, addd = dev(x), dev(wt.reshape(-1, cout)), dev(dy), dev(gs), dev(add)
    dx = torch.full((nb, h, w, cin), float("nan"), device="cuda")
    dw = torch.full((k * k * cin, cout), float("nan"), device="cuda")
    db = torch.full((cout,), float("nan"), device="cuda")
    d = conv_desc(L, xd, wd, dx, nb, h, w, cin, h, w, k, 1, pad, cout, cout)
    d.dy, d.ld_dy, d.gscale = dyd.data_ptr(), cout, gsd.data_ptr()
    d.dx, d.ld_dx, d.dx_add, d.ld_dx_add, d.dx_mask, d.ld_dx_mask = dx.data_ptr(), cin, addd.data_ptr(), cin, xd.data_ptr(), cin
    d.dw, d.dw_accumulate, d.db = dw.data_ptr(), 0, db.data_ptr()
    if shape is not None:
        nk = k * k * cout // 32                                   # K tiles of the dgrad problem
        if (abs(shape[2]) > 1 and nk // abs(shape[2]) < 1) or cin % shape[0]:
            pytest.skip("fewer K tiles than slices / weight-gradient tile does not divide the channels")
        ctx.check(ctx.lib.radnet_force_config(ctx.h, *shape), "force")
    try:
        ctx.check(ctx.lib.radnet_conv_bwd(ctx.h, C.byref(d)), "conv_bwd")
        ctx.sync()
        close(dx.cpu().numpy(), dx_ref)
        close(dw.cpu().numpy(), dw_ref.reshape(-1, cout))
        close(db.cpu().numpy(), db_ref)
        d.dw_accumulate = 1
        dx.fill_(float("nan"))
        ctx.check(ctx.lib.radnet_conv_bwd(ctx.h, C.byref(d)), "conv_bwd")
        ctx.sync()
        close(dx.cpu().numpy(), dx_ref)
        close(dw.cpu().numpy(), 2 * dw_ref.reshape(-1, cout))
        close(db.cpu().numpy(), 2 * db_ref)
    finally:
        ctx.check(ctx.lib.radnet_force_config(ctx.h, 0, 0, 0), "force off")


def test_wgrad_strided_1x1(ctx):
    from radnet_hip import lib as L
    from oracle import dense
    rs = np.random.RandomState(9)
    x = rs.standard_normal((3, 14, 14, 1024)).astype(np.float32)
    wt = np.zeros((1, 1, 1024, 512), np.float32)
    dy = rs.standard_normal((3, 7, 7, 512)).astype(np.float32)
    _, dw_ref, _ = dense.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64), 2, (0,) * 4, need_dx=False)
    xd, wd, dyd = dev(x), dev(wt.reshape(1024, 512)), dev(dy)
    dw = torch.zeros(1024, 512, device="cuda")
    d = conv_desc(L, xd, wd, dw, 3, 14, 14, 1024, 7, 7, 1, 2, 0, 512, 512)
    d.dy, d.ld_dy, d.dw, d.dw_accumulate = dyd.data_ptr(), 512, dw.data_ptr(), 0
    ctx.check(ctx.lib.radnet_conv_wgrad(ctx.h, C.byref(d)), "wgrad")
    close(dw.cpu().numpy(), dw_ref.reshape(1024, 512))


def test_conv_rejects_bad_arguments(ctx):
    from radnet_hip import lib as L
    x = torch.zeros(1, 8, 8, 48, device="cuda"); w = torch.zeros(48, 64, device="cuda"); y = torch.zeros(64, 64, device="cuda")
    d = conv_desc(L, x, w, y, 1, 8, 8, 48, 8, 8, 1, 1, 0, 64, 64)          # 48 channels: not a multiple of 32
    assert ctx.lib.radnet_conv_fwd(ctx.h, C.byref(d)) == -3
    assert b"multiple" in ctx.lib.radnet_last_error(ctx.h)
    d = conv_desc(L, x, w, y, 1, 8, 8, 64, 8, 8, 1, 1, 0, 62, 62)
    assert ctx.lib.radnet_conv_fwd(ctx.h, C.byref(d)) == -1


def test_pools_and_roi_resize(ctx):
    from oracle import dense
    rs = np.random.RandomState(11)
    x = rs.standard_normal((2, 21, 30, 64)).astype(np.float32)
    xd = dev(x)
    y = torch.zeros(2, 10, 14, 64, device="cuda")
    ctx.call("radnet_maxpool_fwd", xd, y, 2, 21, 30, 64, 3, 2)
    assert np.array_equal(y.cpu().numpy(), dense.maxpool_3x3_s2(x))
    y2 = torch.zeros(2, 10, 15, 64, device="cuda")
    ctx.call("radnet_maxpool_fwd", xd, y2, 2, 21, 30, 64, 2, 2)
    assert np.array_equal(y2.cpu().numpy(), dense.maxpool_2x2_s2(x))

    F = rs.standard_normal((1, 38, 63, 1024)).astype(np.float32)
    rois = np.array([[0, 0, 63, 38], [5, 3, 7, 9], [60, 35, 8, 8], [10, 10, 1, 1], [3.9, 2.2, 20.7, 14.1], [20, 5, 14, 28], [62, 37, 1, 1]], np.float32)
    for ps in (14, 7):
        ref = dense.roi_crop_resize(F, rois, ps)
        out = torch.zeros(len(rois), ps, ps, 1024, device="cuda")
        ctx.call("radnet_roi_resize_fwd", dev(F), 38, 63, 1024, dev(rois), len(rois), ps, out)
        close(out.cpu().numpy(), ref, rtol=4e-6, atol=1e-6)   # fp32 lerp; the device contracts a*b+c into one FMA
    dy = rs.standard_normal((len(rois), 7, 7, 1024)).astype(np.float32)
    dF = torch.zeros(1, 38, 63, 1024, device="cuda")
    ctx.call("radnet_roi_resize_bwd", dev(dy), 38, 63, 1024, dev(rois), len(rois), 7, dF)
    close(dF.cpu().numpy(), dense.roi_crop_resize_bwd(F.shape, rois, 7, dy.astype(np.float64)), rtol=1e-5)

    y5 = rs.standard_normal((5, 49, 2048)).astype(np.float32)
    feat = torch.zeros(5, 2048, device="cuda")
    ctx.call("radnet_avgpool_fwd", dev(y5), 5, 49, 2048, feat)
    close(feat.cpu().numpy(), y5.astype(np.float64).mean(1), rtol=1e-6)
    dfeat = rs.standard_normal((5, 2048)).astype(np.float32)
    dx = torch.zeros(5, 49, 2048, device="cuda")
    ctx.call("radnet_avgpool_bwd_relu", dev(dfeat), dev(y5), 5, 49, 2048, dx)
    close(dx.cpu().numpy(), (y5 > 0) * dfeat[:, None, :].astype(np.float64) / 49, rtol=1e-6)


def test_dense_heads_fwd_bwd(ctx):
    from oracle import dense
    rs = np.random.RandomState(12)
    R, nc, nreg = 20, 7, 24
    feat = rs.standard_normal((R, 2048)).astype(np.float32)
    w = np.zeros((2048, 32), np.float32); w[:, :31] = rs.standard_normal((2048, 31)) * 0.02
    b = np.zeros(32, np.float32); b[:31] = rs.standard_normal(31) * 0.1
    pc, pr = torch.zeros(R, nc, device="cuda"), torch.zeros(R, nreg, device="cuda")
    ctx.call("radnet_dense_heads_fwd", dev(feat), R, 2048, dev(w), 32, dev(b), nc, nreg, pc, pr)
    z = feat.astype(np.float64) @ w + b
    close(pc.cpu().numpy(), dense.softmax(z[:, :nc]), rtol=1e-5)
    close(pr.cpu().numpy(), z[:, nc:31], rtol=1e-5)
    dz = rs.standard_normal((R, 31)).astype(np.float32)
    dw, db, dfeat = torch.zeros(2048, 32, device="cuda"), torch.zeros(32, device="cuda"), torch.zeros(R, 2048, device="cuda")
    ctx.call("radnet_dense_heads_bwd", dev(feat), dev(dz), R, 2048, dev(w), 32, 31, dw, db, dfeat, 0)
    close(dw.cpu().numpy()[:, :31], feat.astype(np.float64).T @ dz, rtol=1e-5)
    assert np.all(dw.cpu().numpy()[:, 31] == 0)
    close(db.cpu().numpy()[:31], dz.astype(np.float64).sum(0), rtol=1e-5)
    close(dfeat.cpu().numpy(), dz.astype(np.float64) @ w[:, :31].T, rtol=1e-5)


@pytest.mark.parametrize("mode", [0, 1])
def test_rpn_loss(ctx, mode):
    from oracle import dense
    rs = np.random.RandomState(13)
    A, H, W = 12, 38, 63
    M = H * W
    valid = (rs.uniform(size=(1, H, W, A)) < 0.02).astype(np.float32)
    ov = ((rs.uniform(size=(1, H, W, A)) < 0.5) * valid).astype(np.float32)
    y_cls = np.concatenate([valid, ov], -1)
    y_regr = np.concatenate([np.repeat(ov, 4, -1), (rs.standard_normal((1, H, W, 4 * A)) * 2).astype(np.float32)], -1)
    pred = np.zeros((M, 64), np.float32)
    pred[:, :A] = 1 / (1 + np.exp(-rs.standard_normal((M, A)) * 3))
    pred[:, A:5 * A] = rs.standard_normal((M, 4 * A))
    pred[0, 0] = 1.0; pred[1, 1] = 0.0            # saturated sigmoid outputs
    p = pred[:, :A].reshape(1, H, W, A); r = pred[:, A:5 * A].reshape(1, H, W, 4 * A)
    lc, dp = dense.rpn_loss_cls(y_cls.astype(np.float64), p.astype(np.float64), A, mode == 0)
    lr, dr = dense.smooth_l1_masked(y_regr.astype(np.float64), r.astype(np.float64), 4 * A)
    dz_ref = np.zeros((M, 64))
    dz_ref[:, :A] = (dp * p * (1 - p)).reshape(M, A)
    dz_ref[:, A:5 * A] = dr.reshape(M, 4 * A)
    dz = torch.full((M, 64), float("nan"), device="cuda")
    losses = torch.zeros(2, device="cuda")
    scratch = torch.zeros(8, dtype=torch.float64, device="cuda")
    ctx.call("radnet_rpn_loss", dev(pred), 64, dev(y_cls), dev(y_regr), M, A, mode, dz, 64, losses, scratch)
    got = losses.cpu().numpy()
    assert abs(got[0] - lc) <= 2e-5 * abs(lc) + 1e-7 and abs(got[1] - lr) <= 2e-5 * abs(lr) + 1e-7
    close(dz.cpu().numpy(), dz_ref, rtol=2e-5, atol=1e-9)


def test_det_loss(ctx):
    from oracle import dense
    rs = np.random.RandomState(14)
    R, nc, nreg = 20, 7, 24
    q = dense.softmax(rs.standard_normal((1, R, nc)) * 2).astype(np.float32)
    pr = rs.standard_normal((1, R, nreg)).astype(np.float32)
    cls = rs.randint(0, nc, R)
    Y1 = np.eye(nc, dtype=np.float32)[cls][None]
    lab = np.zeros((R, nreg), np.float32)
    for i, c in enumerate(cls):
        if c != nc - 1:
            lab[i, 4 * c:4 * c + 4] = 1
    Y2 = np.concatenate([lab, (rs.standard_normal((R, nreg)) * 2).astype(np.float32) * lab], -1)[None]
    lc, dq = dense.class_loss_cls(Y1.astype(np.float64), q.astype(np.float64))
    lr, dr = dense.smooth_l1_masked(Y2.astype(np.float64), pr.astype(np.float64), nreg)
    q64 = q[0].astype(np.float64)
    dlog = q64 * (dq[0] - (dq[0] * q64).sum(-1, keepdims=True))
    dz = torch.zeros(R, nc + nreg, device="cuda"); losses = torch.zeros(3, device="cuda")
    ctx.call("radnet_det_loss", dev(q[0]), dev(pr[0]), dev(Y1[0]), dev(Y2[0]), R, nc, nreg, dz, losses)
    got = losses.cpu().numpy()
    assert abs(got[0] - lc) < 1e-5 * abs(lc) + 1e-7 and abs(got[1] - lr) < 1e-5 * abs(lr) + 1e-7
    assert abs(got[2] - dense.categorical_accuracy(Y1, q)) < 1e-6
    close(dz.cpu().numpy()[:, :nc], dlog, rtol=1e-4, atol=1e-8)
    close(dz.cpu().numpy()[:, nc:], dr[0], rtol=1e-5, atol=1e-9)


def test_adam(ctx):
    from oracle import dense
    rs = np.random.RandomState(15)
    n = 4096
    p = rs.standard_normal(n).astype(np.float32); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    pd, md, vd = dev(p), dev(m), dev(v)
    for t in range(1, 5):
        g = rs.standard_normal(n).astype(np.float32) * 0.1
        dense.adam_step(p, g, m, v, t, 5e-5)
        gd = dev(g)
        ctx.call("radnet_adam_step", pd, gd, md, vd, C.c_int64(n), t, C.c_float(5e-5), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-7), C.c_float(1.0),
                 t % 2)
        # zero_grad: the gradient arena is cleared in the same pass (odd t here), left alone otherwise
        assert np.array_equal(gd.cpu().numpy(), np.zeros(n, np.float32) if t % 2 else g)
    assert np.allclose(pd.cpu().numpy(), p, rtol=0, atol=2e-7)
    assert np.allclose(md.cpu().numpy(), m, rtol=1e-5, atol=1e-8)


def test_adam_fused_equals_adam_then_affine_then_filter_transforms(ctx):
    """radnet_adam_step_fused (round 4): one pass = radnet_adam_step over the arena + radnet_affine_vec for the bias range +
    radnet_winograd4_filter for the 3x3 kernels inside it, BIT FOR BIT (the native train step and the scheduler-driven one use either)."""
    from radnet_hip import lib as L
    rs = np.random.RandomState(23)
    layers = [(64, 16, 16), (5000, 64, 32)]                       # (offset, c, n): dense [3][3][c][n] kernels inside the arena
    n = 5000 + 9 * 64 * 32 + 1024
    bias_off, bias_len = n - 256, 128
    base = dict(p=rs.standard_normal(n).astype(np.float32), m=(0.01 * rs.standard_normal(n)).astype(np.float32),
                v=(0.01 * rs.uniform(size=n)).astype(np.float32), g=(0.1 * rs.standard_normal(n)).astype(np.float32))
    scale, t0 = rs.uniform(0.5, 1.5, bias_len).astype(np.float32), rs.standard_normal(bias_len).astype(np.float32)
    sd, td = dev(scale), dev(t0)
    args = lambda t: (C.c_int64(n), t, C.c_float(1e-3), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-7), C.c_float(0.5), 1)

    def run(fused):
        a = {k: dev(v) for k, v in base.items()}
        shift = torch.zeros(bias_len, device="cuda")
        us = [torch.full((36, c, nn), float("nan"), device="cuda") for _, c, nn in layers]
        for t in (1, 2):
            a["g"].copy_(dev(base["g"]) * t)
            if fused:
                arr = (L.AdamWino * len(layers))()
                for k, (off, c, nn) in enumerate(layers):
                    arr[k].off, arr[k].c, arr[k].n, arr[k].u = off, c, nn, us[k].data_ptr()
                ctx.check(ctx.lib.radnet_adam_step_fused(ctx.h, a["p"].data_ptr(), a["g"].data_ptr(), a["m"].data_ptr(), a["v"].data_ptr(), *args(t),
                                                         C.c_int64(bias_off), C.c_int64(bias_len), sd.data_ptr(), td.data_ptr(), shift.data_ptr(), arr, len(layers)),
                          "adam_fused")
            else:
                ctx.call("radnet_adam_step", a["p"], a["g"], a["m"], a["v"], *args(t))
                ctx.call("radnet_affine_vec", shift, sd, a["p"][bias_off:], td, C.c_int64(bias_len))
                for k, (off, c, nn) in enumerate(layers):
                    ctx.call("radnet_winograd4_filter", a["p"][off:], c, nn, nn, us[k])
        torch.cuda.synchronize()
        return [a[k].cpu().numpy() for k in "pmvg"] + [shift.cpu().numpy()] + [u.cpu().numpy() for u in us]

    for name, x, y in zip(("p", "m", "v", "g", "shift", "u0", "u1"), run(True), run(False)):
        assert not np.isnan(x).any(), name
        assert np.array_equal(x, y), (name, int((x != y).sum()), float(np.abs(x - y).max()))


# ---- fp64 glue: bit-exact against vectors produced by the reference itself -------------------------------------
def test_rpn_to_roi_golden_bit_exact(ctx):
    g = load_golden("rpn_to_roi")
    from faster_rcnn.config import Config
    for i in range(int(g["n_cases"])):
        cls, regr = g[f"c{i}_cls"], g[f"c{i}_regr"]
        _, rows, cols, A = cls.shape
        Cc = Config(); Cc.anchor_box_scales = [int(v) for v in g[f"c{i}_scales"]]
        pred = np.zeros((rows * cols, 64), np.float32)
        pred[:, :A] = cls.reshape(-1, A); pred[:, A:5 * A] = regr.reshape(-1, 4 * A)
        awh = np.array([[(s * r[0]) / 16, (s * r[1]) / 16] for s in Cc.anchor_box_scales for r in Cc.anchor_box_ratios], dtype=np.float64)
        mb = int(g[f"c{i}_max"])
        R = torch.zeros(mb, 4, dtype=torch.int64, device="cuda"); Rp = torch.zeros(mb, device="cuda"); Rn = torch.zeros(1, dtype=torch.int32, device="cuda")
        ws = torch.empty(int(ctx.lib.radnet_proposals_ws_bytes(rows * cols * A)), dtype=torch.uint8, device="cuda")
        predd = dev(pred)
        rc = ctx.lib.radnet_rpn_to_roi(ctx.h, predd.data_ptr(), 64, rows, cols, A, awh.ctypes.data_as(C.POINTER(C.c_double)), 4.0, 1,
                                       float(g[f"c{i}_thr"]), mb, R.data_ptr(), Rp.data_ptr(), Rn.data_ptr(), ws.data_ptr())
        ctx.check(rc, "rpn_to_roi")
        n = int(Rn.cpu()[0])
        ref = g[f"c{i}_R"]
        assert n == ref.shape[0], i
        assert np.array_equal(R.cpu().numpy()[:n], ref), i


def test_nms_golden_bit_exact(ctx):
    g = load_golden("nms")
    for i in range(int(g["n_cases"])):
        boxes = g[f"c{i}_boxes"].astype(np.float64); probs = g[f"c{i}_probs"].astype(np.float32)
        n = len(boxes)
        mb = int(g[f"c{i}_max"])
        idx = torch.zeros(mb, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        ws = torch.empty(int(ctx.lib.radnet_proposals_ws_bytes(n)), dtype=torch.uint8, device="cuda")
        ctx.call("radnet_nms", dev(boxes), dev(probs), n, C.c_double(float(g[f"c{i}_thr"])), mb, idx, cnt, ws)
        k = int(cnt.cpu()[0])
        pick = idx.cpu().numpy()[:k]
        assert np.array_equal(boxes[pick].astype("int"), g[f"c{i}_out_boxes"]), i
        assert np.array_equal(g[f"c{i}_probs"][pick], g[f"c{i}_out_probs"]), i
    # malformed box -> the reference asserts; the kernel reports -1
    bad = np.array([[0., 0., 5., 5.], [3., 3., 3., 8.]])
    idx = torch.zeros(10, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = torch.empty(int(ctx.lib.radnet_proposals_ws_bytes(2)), dtype=torch.uint8, device="cuda")
    ctx.call("radnet_nms", dev(bad), dev(np.array([0.5, 0.6], np.float32)), 2, C.c_double(0.5), 10, idx, cnt, ws)
    assert int(cnt.cpu()[0]) == -1
    ctx.call("radnet_nms", None, None, 0, C.c_double(0.5), 10, idx, cnt, ws)
    assert int(cnt.cpu()[0]) == 0


def test_nms_tie_rule(ctx):
    # equal scores: stable ascending sort walked from the end = higher index first (documented tie rule)
    boxes = np.array([[0., 0., 10., 10.], [1., 1., 11., 11.], [50., 50., 60., 60.]])
    probs = np.array([0.5, 0.5, 0.5], np.float32)
    from oracle import glue
    rb, _ = glue.greedy_nms(boxes, probs, 0.5, 300)
    idx = torch.zeros(10, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = torch.empty(int(ctx.lib.radnet_proposals_ws_bytes(3)), dtype=torch.uint8, device="cuda")
    ctx.call("radnet_nms", dev(boxes), dev(probs), 3, C.c_double(0.5), 10, idx, cnt, ws)
    k = int(cnt.cpu()[0])
    assert np.array_equal(boxes[idx.cpu().numpy()[:k]].astype("int"), rb)


def test_anchor_targets_golden(ctx):
    """Labels bit-exact; regression targets bit-exact in the fp32 form the network consumes and within 1 ulp
    (device log vs NumPy log) in fp64."""
    from faster_rcnn.config import Config
    from radnet_hip import engine as E
    g = load_golden("calc_region_props")
    for i in range(int(g["n_cases"])):
        W, H, rw, rh, isz, rseed = (int(v) for v in g[f"c{i}_wh"])
        Cc = Config(); Cc.img_size = isz
        gt = g[f"c{i}_gt_boxes"]; isbg = g[f"c{i}_gt_is_bg"].astype(np.int32)
        ng = len(gt)
        fw, fh = E.feat_len(rw), E.feat_len(rh)
        A = 12
        valid = torch.zeros(A, fh, fw, dtype=torch.uint8, device="cuda"); overlap = torch.zeros_like(valid)
        regr = torch.zeros(fh, fw, 4 * A, dtype=torch.float64, device="cuda")
        best = torch.zeros(max(ng, 1), 4, dtype=torch.int32, device="cuda"); nfor = torch.zeros(max(ng, 1), dtype=torch.int32, device="cuda")
        scratch = torch.zeros(max(ng, 1), dtype=torch.int64, device="cuda")
        sizes = np.array(Cc.anchor_box_scales, np.float64); ratios = np.array(Cc.anchor_box_ratios, np.float64)
        gtd = dev(gt) if ng else None; bgd = dev(isbg) if ng else None
        rc = ctx.lib.radnet_anchor_targets(ctx.h, gtd.data_ptr() if ng else None, bgd.data_ptr() if ng else None, ng, W, H, rw, rh, fw, fh,
                                           sizes.ctypes.data_as(C.POINTER(C.c_double)), 4, ratios.ctypes.data_as(C.POINTER(C.c_double)), 3, 16.0, 0.7,
                                           valid.data_ptr(), overlap.data_ptr(), regr.data_ptr(), best.data_ptr(), nfor.data_ptr(), scratch.data_ptr())
        ctx.check(rc, "anchor_targets")
        v = valid.cpu().numpy(); o = overlap.cpu().numpy()
        np.random.seed(rseed)
        n_pos = E.subsample_valid(v, o)
        assert n_pos == int(g[f"c{i}_n_pos"]), i
        assert np.random.randint(0, 2 ** 31 - 1) == int(g[f"c{i}_rng_after"]), i
        ycls_ref, yregr_ref = g[f"c{i}_y_rpn_cls"], g[f"c{i}_y_rpn_regr"]
        assert np.array_equal(v, ycls_ref[0, :A]), i
        assert np.array_equal(o, ycls_ref[0, A:]), i
        if ng:
            assert np.array_equal(best.cpu().numpy()[:ng], g[f"c{i}_best_anchor"]), i
        rg = regr.cpu().numpy().transpose(2, 0, 1)
        ref = yregr_ref[0, 4 * A:]
        assert np.array_equal(rg.astype(np.float32), ref.astype(np.float32)), i
        assert np.all(np.abs(rg - ref) <= 4.5e-16 * np.maximum(np.abs(ref), 1e-300)), i
        # packed fp32 NHWC training tensors (utils.py:475-478)
        valid.copy_(torch.from_numpy(v))
        ycls = torch.zeros(fh, fw, 2 * A, device="cuda"); yregr = torch.zeros(fh, fw, 8 * A, device="cuda")
        ctx.call("radnet_anchor_targets_pack", valid, overlap, regr, fw, fh, A, C.c_double(4.0), ycls, yregr)
        ref_cls = np.transpose(ycls_ref, (0, 2, 3, 1))[0].astype(np.float32)
        ref_regr = yregr_ref.copy(); ref_regr[:, 4 * A:] *= 4.0
        ref_regr = np.transpose(ref_regr, (0, 2, 3, 1))[0].astype(np.float32)
        assert np.array_equal(ycls.cpu().numpy(), ref_cls), i
        assert np.array_equal(yregr.cpu().numpy(), ref_regr), i


def test_roi_targets_golden(ctx):
    from faster_rcnn.config import Config
    from oracle import glue
    g = load_golden("calc_iou")
    Cc = Config()
    std = np.array(Cc.classifier_regr_std, np.float64)
    for i in range(int(g["n_cases"])):
        R = g[f"c{i}_R"]; gt = g[f"c{i}_gt_boxes"]; gc = g[f"c{i}_gt_cls"].astype(np.int32)
        W, H = (int(v) for v in g[f"c{i}_wh"])
        rw, rh = glue.new_img_size(W, H, Cc.img_size)
        n = len(R)
        keep = torch.zeros(n, dtype=torch.uint8, device="cuda"); cls = torch.zeros(n, dtype=torch.int32, device="cuda")
        box = torch.zeros(n, 4, dtype=torch.int32, device="cuda"); t = torch.zeros(n, 4, dtype=torch.float64, device="cuda")
        iou = torch.zeros(n, dtype=torch.float64, device="cuda")
        Rd, gtd, gcd = dev(R), dev(gt), dev(gc)        # keep the device buffers alive across the call
        rc = ctx.lib.radnet_roi_targets(ctx.h, Rd.data_ptr(), n, gtd.data_ptr(), gcd.data_ptr(), len(gt), W, H, rw, rh, 16.0, 0.1, 0.5,
                                        std.ctypes.data_as(C.POINTER(C.c_double)), 6, keep.data_ptr(), cls.data_ptr(), box.data_ptr(), t.data_ptr(), iou.data_ptr(), None)
        ctx.check(rc, "roi_targets")
        k = keep.cpu().numpy().astype(bool)
        X, Y1, Y2 = g[f"c{i}_X"], g[f"c{i}_Y1"], g[f"c{i}_Y2"]
        assert k.sum() == X.shape[1], i
        assert np.array_equal(box.cpu().numpy()[k], X[0]), i
        assert np.array_equal(cls.cpu().numpy()[k], Y1[0].argmax(-1)), i
        assert np.array_equal(iou.cpu().numpy()[k], g[f"c{i}_ious"]), i
        # pack every kept RoI and compare the fp32 training tensors
        sel = np.nonzero(k)[0].astype(np.int32)
        ro = torch.zeros(len(sel), 4, device="cuda"); y1 = torch.zeros(len(sel), 7, device="cuda"); y2 = torch.zeros(len(sel), 48, device="cuda")
        ctx.call("radnet_roi_batch_pack", dev(sel), len(sel), cls, box, t, 7, 6, ro, y1, y2)
        assert np.array_equal(ro.cpu().numpy(), X[0].astype(np.float32))
        assert np.array_equal(y1.cpu().numpy(), Y1[0].astype(np.float32))
        got, ref = y2.cpu().numpy(), Y2[0].astype(np.float32)
        assert np.array_equal(got[:, :24], ref[:, :24])
        assert np.allclose(got[:, 24:], ref[:, 24:], rtol=2e-7, atol=0)      # log: device vs NumPy, 1 ulp of fp64 -> <= 1 ulp fp32


@pytest.mark.parametrize("form", [2, 4])
@pytest.mark.parametrize("case", [(1, 38, 63, 1024, 512), (20, 7, 7, 512, 512), (2, 9, 12, 64, 128), (1, 75, 125, 128, 128), (1, 5, 3, 32, 64)])
def test_winograd_conv3x3_vs_oracle(ctx, case, form):
    """Winograd F(2x2,3x3) and F(4x4,3x3) paths (filter / input transform, 16 / 36 batched GEMMs, output transform with the
    BN-ReLU epilogue) against the oracle's direct 3x3 'same' convolution at the suite's stated tolerance (2e-4 of the largest
    value); sizes that are not multiples of the tile exercise the partial border tiles."""
    from oracle import dense
    nb, h, w, cin, cout = case
    rs = np.random.RandomState(sum(case))
    x = np.maximum(rs.standard_normal((nb, h, w, cin)), 0).astype(np.float32)
    wt = (rs.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    sc = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rs.standard_normal(cout).astype(np.float32)
    ref = np.maximum(dense.conv2d(x.astype(np.float64), wt.astype(np.float64), None, 1, (1, 1, 1, 1)) * sc + sh, 0)
    T = nb * ((h + form - 1) // form) * ((w + form - 1) // form)
    P, fn = (form + 2) ** 2, "radnet_winograd4_" if form == 4 else "radnet_winograd_"
    U = torch.empty(P, cin, cout, device="cuda")
    V = torch.full((P, T, cin), float("nan"), device="cuda")
    M = torch.empty(P, T, cout, device="cuda")
    y = torch.full((nb, h, w, cout), float("nan"), device="cuda")
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.radnet_set_workspace(ctx.h, ws.data_ptr(), ws.numel()), "ws")
    ctx.call(fn + "filter", dev(wt.reshape(-1, cout)), cin, cout, cout, U)
    ctx.call(fn + "input", dev(x), nb, h, w, cin, V)
    ctx.call("radnet_gemm_batched", V, U, M, P, T, cout, cin)
    ctx.call(fn + "output", M, nb, h, w, cout, dev(sc), dev(sh), 1, y, cout)
    err = np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max()
    print("winograd F(%dx%d) %s: max error %.2e of the largest activation" % (form, form, case, err))
    close(y.cpu().numpy(), ref)
    # the batched GEMM on its own
    mref = np.einsum("ptc,pcn->ptn", V.cpu().numpy().astype(np.float64), U.cpu().numpy().astype(np.float64))
    close(M.cpu().numpy(), mref)


@pytest.mark.parametrize("form", [2, 4])
@pytest.mark.parametrize("case", [(1, 38, 63, 1024, 512), (2, 9, 11, 64, 128)])
def test_winograd_wgrad_vs_oracle(ctx, case, form):
    """Weight gradient of a 3x3 'same' conv in the Winograd domain (dy transform, 16 reduction-over-tiles GEMMs on the
    forward pass's transformed input, inverse filter transform) against the oracle's direct gradient."""
    from oracle import dense
    nb, h, w, cin, cout = case
    rs = np.random.RandomState(sum(case) + 5)
    x = np.maximum(rs.standard_normal((nb, h, w, cin)), 0).astype(np.float32)
    dy = rs.standard_normal((nb, h, w, cout)).astype(np.float32)
    gs = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    wt = np.zeros((3, 3, cin, cout), np.float32)
    _, dw_ref, _ = dense.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), (dy * gs).astype(np.float64), 1, (1, 1, 1, 1), need_dx=False)
    T = nb * ((h + form - 1) // form) * ((w + form - 1) // form)
    P, fn = (form + 2) ** 2, "radnet_winograd4_" if form == 4 else "radnet_winograd_"
    V = torch.empty(P, T, cin, device="cuda")
    dZ = torch.empty(P, T, cout, device="cuda")
    dU = torch.full((P, cin, cout), float("nan"), device="cuda")
    dw = torch.full((9 * cin, cout), float("nan"), device="cuda")
    ctx.call(fn + "input", dev(x), nb, h, w, cin, V)
    ctx.call(fn + "dy", dev(dy), nb, h, w, cout, cout, dev(gs), dZ)
    ctx.call("radnet_wgrad_batched", V, dZ, dU, P, T, cin, cout, 0)
    ctx.call(fn + "filter_grad", dU, cin, cout, cout, dw, 0)
    err = np.abs(dw.cpu().numpy() - dw_ref.reshape(-1, cout)).max() / np.abs(dw_ref).max()
    print("winograd wgrad F(%dx%d) %s: max error %.2e of the largest gradient" % (form, form, case, err))
    close(dw.cpu().numpy(), dw_ref.reshape(-1, cout))
    ctx.call(fn + "filter_grad", dU, cin, cout, cout, dw, 1)          # accumulate: doubles
    close(dw.cpu().numpy(), 2 * dw_ref.reshape(-1, cout))


@pytest.mark.parametrize("groups,idle", [(1, None), (2, None), (2, 1)])
def test_head_tail_fused_equals_separate_kernels(ctx, groups, idle):
    """csrc/head_tail.hip (avg-pool + dense heads + detector losses in one launch) against the three separate kernels it
    replaces on the training step -- which tests/test_gpu_engine.py pins against the oracle."""
    rs = np.random.RandomState(3 + groups)
    rg, nc, nreg, hw, c, ld = 20, 7, 24, 49, 2048, 32
    R = rg * groups
    y5 = np.maximum(rs.standard_normal((R, hw, c)), 0).astype(np.float32)
    w = np.zeros((c, ld), np.float32); w[:, :nc + nreg] = rs.standard_normal((c, nc + nreg)) * 0.02
    b = np.zeros(ld, np.float32); b[:nc + nreg] = rs.standard_normal(nc + nreg) * 0.1
    cls = rs.randint(0, nc, R)
    y1 = np.eye(nc, dtype=np.float32)[cls]
    lab = np.zeros((R, nreg), np.float32)
    for i, k in enumerate(cls):
        if k != nc - 1:
            lab[i, 4 * k:4 * k + 4] = 1
    y2 = np.concatenate([lab, (rs.standard_normal((R, nreg)) * 2).astype(np.float32) * lab], 1)
    d = {k: dev(v) for k, v in dict(y5=y5, w=w, b=b, y1=y1, y2=y2).items()}
    z = lambda *s: torch.full(s, float("nan"), device="cuda")
    # ---- separate kernels
    feat0, pc0, pr0, dz0, dw0, db0, df0, gl0 = z(R, c), z(R, nc), z(R, nreg), z(R, nc + nreg), z(c, ld), z(ld), z(R, c), z(R, hw, c)
    L0 = torch.zeros(groups, 3, device="cuda")
    ctx.call("radnet_avgpool_fwd", d["y5"], R, hw, c, feat0)
    ctx.call("radnet_dense_heads_fwd", feat0, R, c, d["w"], ld, d["b"], nc, nreg, pc0, pr0)
    for g in range(groups):
        o = g * rg
        if idle == g:
            dz0[o:o + rg].zero_()
            continue
        ctx.call("radnet_det_loss", pc0[o:], pr0[o:], d["y1"][o:], d["y2"][o:], rg, nc, nreg, dz0[o:], L0[g])
    ctx.call("radnet_dense_heads_bwd", feat0, dz0, R, c, d["w"], ld, nc + nreg, dw0, db0, df0, 0)
    ctx.call("radnet_avgpool_bwd_relu", df0, d["y5"], R, hw, c, gl0)
    # ---- fused
    feat1, pc1, pr1, dz1, dw1, db1, df1, gl1 = z(R, c), z(R, nc), z(R, nreg), z(R, nc + nreg), z(c, ld), z(ld), z(R, c), z(R, hw, c)
    L1 = torch.zeros(groups, 3, device="cuda")
    scratch = torch.zeros(int(ctx.lib.radnet_head_tail_scratch_bytes(R)), dtype=torch.uint8, device="cuda")
    live = torch.tensor([0 if idle == g else 1 for g in range(groups)], dtype=torch.int32, device="cuda")
    for _ in range(2):                                    # twice: the arrival counter must be back at zero after a launch
        ctx.call("radnet_head_tail_fwd", d["y5"], R, hw, c, d["w"], ld, d["b"], nc, nreg, feat1, pc1, pr1, d["y1"], d["y2"], dz1, L1, groups, live, scratch)
    ctx.call("radnet_dense_heads_bwd", feat1, dz1, R, c, d["w"], ld, nc + nreg, dw1, db1, df1, 0)
    ctx.call("radnet_avgpool_bwd_relu", df1, d["y5"], R, hw, c, gl1)
    ctx.sync()
    close(feat1.cpu().numpy(), feat0.cpu().numpy(), rtol=1e-6, atol=1e-7)        # pooled in four position groups: re-associated
    for a, bb, tol in ((pc1, pc0, 1e-5), (pr1, pr0, 1e-5), (dz1, dz0, 1e-5), (dw1, dw0, 1e-5), (db1, db0, 1e-5), (df1, df0, 1e-5), (gl1, gl0, 1e-5)):
        close(a.cpu().numpy(), bb.cpu().numpy(), rtol=tol, atol=1e-7)
    close(L1.cpu().numpy(), L0.cpu().numpy(), rtol=1e-6, atol=1e-7)
    assert np.all(np.isfinite(dw1.cpu().numpy())) and (idle is None or float(dz1[idle * rg:(idle + 1) * rg].abs().max()) == 0.0)
    # inference form: no targets, no losses
    pc2, pr2, feat2 = z(R, nc), z(R, nreg), z(R, c)
    ctx.call("radnet_head_tail_fwd", d["y5"], R, hw, c, d["w"], ld, d["b"], nc, nreg, feat2, pc2, pr2, None, None, None, None, 1, None, scratch)
    ctx.sync()
    assert np.array_equal(pc2.cpu().numpy(), pc1.cpu().numpy()) and np.array_equal(pr2.cpu().numpy(), pr1.cpu().numpy())



def _rpn_to_roi(ctx, pred, rows, cols, A, awh, thr, mb, rocprim):
    import os
    R = torch.zeros(mb, 4, dtype=torch.int64, device="cuda"); Rp = torch.zeros(mb, device="cuda"); Rn = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = torch.empty(int(ctx.lib.radnet_proposals_ws_bytes(rows * cols * A)), dtype=torch.uint8, device="cuda")
    predd = dev(pred)
    if rocprim:
        os.environ.pop("RADNET_PROPOSALS_SELECT", None)
    else:
        os.environ["RADNET_PROPOSALS_SELECT"] = "1"
    try:
        rc = ctx.lib.radnet_rpn_to_roi(ctx.h, predd.data_ptr(), 64, rows, cols, A, awh.ctypes.data_as(C.POINTER(C.c_double)), 4.0, 1, float(thr), mb,
                                       R.data_ptr(), Rp.data_ptr(), Rn.data_ptr(), ws.data_ptr())
    finally:
        os.environ.pop("RADNET_PROPOSALS_SELECT", None)
    ctx.check(rc, "rpn_to_roi")
    ctx.sync()
    n = int(Rn.cpu()[0])
    return R.cpu().numpy()[:n], Rp.cpu().numpy()[:n]


PROPOSAL_STRESS = [  # rows, cols, A, score kind, regr sigma, thr, max_boxes
    (38, 63, 12, "distinct", 0.5, 0.7, 300),
    (38, 63, 12, "saturated", 0.5, 0.7, 300),      # thousands of scores exactly 1.0: the radix select must cut inside a tie (index digits)
    (38, 63, 12, "few_levels", 0.5, 0.7, 300),     # 5 distinct score values
    (38, 63, 12, "distinct", 0.05, 0.95, 1000),    # picks never reach max_boxes: every band is examined (7 bands)
    (63, 63, 12, "distinct", 0.3, 0.7, 300),       # 47 628 candidates
    (63, 63, 12, "saturated", 1.5, 0.5, 300),
    (10, 12, 9, "distinct", 0.5, 0.9, 300),        # fewer candidates than one band
    (2, 3, 12, "saturated", 0.2, 0.7, 300),
    (38, 63, 12, "tiny", 0.5, 0.7, 300),           # scores ~1e-30 .. 1e-20: exponent digits decide
]


@pytest.mark.parametrize("case", PROPOSAL_STRESS, ids=["%dx%dx%d_%s_thr%s_max%d" % (c[0], c[1], c[2], c[3], c[5], c[6]) for c in PROPOSAL_STRESS])
def test_select_nms_kernel_matches_reference_and_full_sort(ctx, case):
    """radnet_rpn_to_roi's alternative one-workgroup path (RADNET_PROPOSALS_SELECT=1: radix select + LDS sort + integer NMS)
    against (a) the oracle's rpn_to_roi (pinned by the reference's own outputs) where scores are tie-free, and (b) the default
    full-sort path (rocPRIM + fp64 NMS) on adversarial score distributions -- same order rule among equal scores, so the two
    must agree bit for bit."""
    from faster_rcnn.config import Config
    from oracle import glue
    rows, cols, A, kind, sig, thr, mb = case
    rs = np.random.RandomState(rows * 131 + cols + A + len(kind))
    n = rows * cols * A
    if kind == "distinct":
        cls = (rs.permutation(n).astype(np.float32) / np.float32(n))
    elif kind == "saturated":
        cls = np.where(rs.uniform(size=n) < 0.4, np.float32(1.0), rs.uniform(0.0, 1.0, n).astype(np.float32)).astype(np.float32)
    elif kind == "few_levels":
        cls = rs.choice(np.array([0.1, 0.25, 0.5, 0.75, 0.999], np.float32), n)
    else:
        cls = (10.0 ** rs.uniform(-30, -20, n)).astype(np.float32)
    regr = (rs.standard_normal((rows * cols, 4 * A)) * sig * 4.0).astype(np.float32)
    pred = np.zeros((rows * cols, 64), np.float32)
    pred[:, :A] = cls.reshape(rows * cols, A); pred[:, A:5 * A] = regr
    Cc = Config()
    if A == 9:
        Cc.anchor_box_scales = [128, 256, 512]
    awh = np.array([[(s * r[0]) / 16, (s * r[1]) / 16] for s in Cc.anchor_box_scales for r in Cc.anchor_box_ratios], dtype=np.float64)
    R_new, P_new = _rpn_to_roi(ctx, pred, rows, cols, A, awh, thr, mb, rocprim=False)
    R_old, P_old = _rpn_to_roi(ctx, pred, rows, cols, A, awh, thr, mb, rocprim=True)
    assert R_new.shape == R_old.shape and np.array_equal(R_new, R_old) and np.array_equal(P_new, P_old)
    if kind in ("distinct", "tiny") and len(np.unique(cls)) == n:
        ref = glue.rpn_to_roi(pred[:, :A].reshape(1, rows, cols, A), pred[:, A:5 * A].reshape(1, rows, cols, 4 * A), Cc, True, mb, thr)
        assert np.array_equal(R_new, ref)


def test_copy_bytes_between_pinned_host_and_device(ctx):
    """radnet_copy_bytes: the copy kernel behind the step's host<->device transfers -- device <- pinned host, pinned host <- device,
    device <- device; sizes that are not multiples of 16 and pointers that are not 16-byte aligned take the byte path."""
    rs = np.random.RandomState(3)
    for n, off in ((1, 0), (15, 0), (16, 0), (28728, 0), (1800000, 0), (1000, 4), (4099, 1)):
        src = torch.from_numpy(rs.randint(0, 256, n + off).astype(np.uint8)).pin_memory()
        dev_buf = torch.zeros(n + off + 16, dtype=torch.uint8, device="cuda")
        back = torch.zeros(n + off, dtype=torch.uint8).pin_memory()
        ctx.call("radnet_copy_bytes", dev_buf[off:], src[off:], C.c_uint64(n))
        ctx.call("radnet_copy_bytes", back[off:], dev_buf[off:], C.c_uint64(n))
        dev2 = torch.zeros(n, dtype=torch.uint8, device="cuda")
        ctx.call("radnet_copy_bytes", dev2, dev_buf[off:], C.c_uint64(n))
        ctx.sync()
        assert np.array_equal(back.numpy()[off:], src.numpy()[off:]), (n, off)
        assert np.array_equal(dev2.cpu().numpy(), src.numpy()[off:]), (n, off)
        assert int(dev_buf[off + n:].sum()) == 0 and int(back[:off].sum()) == 0          # nothing written outside the range
    assert ctx.lib.radnet_copy_bytes(ctx.h, None, None, C.c_uint64(0)) == 0             # empty copy: no launch, no error


@pytest.mark.parametrize("case", [(3, 14, 14, 1024, 512, 2048, 2), (1, 21, 30, 256, 128, 512, 2), (2, 9, 11, 64, 64, 96, 1)])
def test_conv_fwd_pair_equals_the_two_convolutions(ctx, case):
    """radnet_conv_fwd_pair (round 4: branch2a + shortcut conv of a conv_block, same input, as ONE launch where that measures faster):
    the first call decides (measures the two launches against the paired launch on every tile it has), later calls use the
    decision; both outputs against the oracle either way, ragged N and M edges included."""
    from radnet_hip import lib as L
    from oracle import dense
    nb, h, w, cin, n1, n2, stride = case
    rs = np.random.RandomState(sum(case) + 11)
    x = np.maximum(rs.standard_normal((nb, h, w, cin)), 0).astype(np.float32)
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    outs = []
    descs = []
    keep = [dev(x)]
    for n, relu in ((n1, True), (n2, False)):
        wt = (rs.standard_normal((1, 1, cin, n)) / np.sqrt(cin)).astype(np.float32)
        b = rs.standard_normal(n).astype(np.float32)
        sc = rs.uniform(0.5, 1.5, n).astype(np.float32)
        ref = dense.conv2d(x.astype(np.float64), wt.astype(np.float64), None, stride, (0,) * 4) * sc + b
        ref = np.maximum(ref, 0) if relu else ref
        y = torch.full((nb, oh, ow, n), float("nan"), dtype=torch.float32, device="cuda")
        wd, sd, bd = dev(wt.reshape(-1, n)), dev(sc), dev(b)
        keep += [wd, sd, bd]
        descs.append(conv_desc(L, keep[0], wd, y, nb, h, w, cin, oh, ow, 1, stride, 0, n, n, sd, bd, None, 1 if relu else 0))
        outs.append((y, ref))
    for rep in range(3):
        for y, _ in outs:
            y.fill_(float("nan"))
        ctx.check(ctx.lib.radnet_conv_fwd_pair(ctx.h, C.byref(descs[0]), C.byref(descs[1])), "conv_fwd_pair")
        for y, ref in outs:
            close(y.cpu().numpy(), ref)


@pytest.mark.parametrize("rows", [64, 32])
@pytest.mark.parametrize("case", [(1, 37, 53, 64, 256, True), (2, 19, 23, 64, 256, False), (1, 8, 9, 64, 128, True), (1, 150, 250, 64, 256, True)])
def test_conv_bottleneck_equals_the_separate_convolutions(ctx, tmp_path, case, rows):
    """radnet_conv_bottleneck (round 4: the back of a frozen stage-2 block -- 3x3, 1x1 expand + shortcut, the next block's 1x1 reduce -- as
    ONE launch that passes the two intermediate tiles through LDS): the fused kernel, pinned through a tuning-table entry (kind 33) for
    both of its row counts, against the oracle and against the three separate launches; ragged row edges, two images, N2 = 128 / 256,
    with and without the trailing reduce; the 3x3's own output buffer stays untouched by the fused launch."""
    from radnet_hip import lib as L
    from oracle import dense
    nb, h, w, c, n2, with_next = case
    rs = np.random.RandomState(sum(case[:5]) + rows)
    M = nb * h * w
    x = np.maximum(rs.standard_normal((nb, h, w, c)), 0).astype(np.float32)
    short = rs.standard_normal((nb, h, w, n2)).astype(np.float32)

    def layer(k, cin, n):
        wt = (rs.standard_normal((k, k, cin, n)) / np.sqrt(k * k * cin)).astype(np.float32)
        return wt, rs.uniform(0.5, 1.5, n).astype(np.float32), (0.1 * rs.standard_normal(n)).astype(np.float32)

    wb, sb, bb = layer(3, c, 64)
    wc, scc, bc = layer(1, 64, n2)
    wa, sa, ba = layer(1, n2, 64)
    t2 = np.maximum(dense.conv2d(x.astype(np.float64), wb.astype(np.float64), None, 1, (1, 1, 1, 1)) * sb + bb, 0)
    y_ref = np.maximum(dense.conv2d(t2, wc.astype(np.float64), None, 1, (0,) * 4) * scc + bc + short, 0)
    t_ref = np.maximum(dense.conv2d(y_ref, wa.astype(np.float64), None, 1, (0,) * 4) * sa + ba, 0)
    xd, shd = dev(x), dev(short)
    dv = [dev(a) for a in (wb.reshape(-1, 64), sb, bb, wc.reshape(-1, n2), scc, bc, wa.reshape(-1, 64), sa, ba)]

    def run(fused):
        c2 = L.Context(0)
        c2.check(c2.lib.radnet_set_workspace(c2.h, ctx._ws.data_ptr(), ctx._ws.numel()), "ws")
        c2.check(c2.lib.radnet_set_autotune(c2.h, 1), "autotune")
        table = tmp_path / ("t%d.txt" % fused)
        table.write_text("# radnet tuned GEMM launch shapes v2\n33 %d %d %d %d 9 1 %d 64 %d 0.01 4\n"
                         % (M, n2 * 65536 + (64 if with_next else 0), 9 * c, c, rows, 1 if fused else 2))
        c2.check(c2.lib.radnet_tune_load(c2.h, str(table).encode()), "tune_load")
        t2d = torch.full((nb, h, w, 64), float("nan"), dtype=torch.float32, device="cuda")
        yd = torch.full((nb, h, w, n2), float("nan"), dtype=torch.float32, device="cuda")
        td = torch.full((nb, h, w, 64), float("nan"), dtype=torch.float32, device="cuda")
        db = conv_desc(L, xd, dv[0], t2d, nb, h, w, c, h, w, 3, 1, 1, 64, 64, dv[1], dv[2], None, 1)
        dc = conv_desc(L, t2d, dv[3], yd, nb, h, w, 64, h, w, 1, 1, 0, n2, n2, dv[4], dv[5], shd, 1)
        da = conv_desc(L, yd, dv[6], td, nb, h, w, n2, h, w, 1, 1, 0, 64, 64, dv[7], dv[8], None, 1)
        for rep in range(2):
            c2.check(c2.lib.radnet_conv_bottleneck(c2.h, C.byref(db), C.byref(dc), C.byref(da) if with_next else None), "conv_bottleneck")
        torch.cuda.synchronize()
        out = t2d.cpu().numpy(), yd.cpu().numpy(), td.cpu().numpy()
        c2.close()
        return out

    t2_f, y_f, t_f = run(True)
    t2_s, y_s, t_s = run(False)
    assert np.isnan(t2_f).all() and not np.isnan(t2_s).any()         # fused: the 3x3 output never reaches memory
    close(t2_s, t2)
    close(y_f, y_ref); close(y_s, y_ref)
    np.testing.assert_allclose(y_f, y_s, rtol=0, atol=2e-6 * np.abs(y_ref).max())
    if with_next:
        close(t_f, t_ref); close(t_s, t_ref)
        np.testing.assert_allclose(t_f, t_s, rtol=0, atol=4e-6 * np.abs(t_ref).max())
    else:
        assert np.isnan(t_f).all() and np.isnan(t_s).all()
