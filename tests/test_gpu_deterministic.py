"""Ordered reductions (radnet_set_deterministic, on by default; include/radnet_hip.h).

Every floating-point reduction of a training step that crosses workgroups -- pixel-split weight gradients (conv_wgrad,
wgrad_batched, the wgrad half of conv_bwd) with their bias gradients, radnet_colsum, radnet_roi_resize_bwd over overlapping
RoIs, the loss sums of radnet_rpn_loss -- must return the SAME BITS on every launch, and the right values (oracle).  The
atomics forms (radnet_set_deterministic(ctx, 0)) stay correct to the usual tolerance.  Whole-step consequences (a pipelined
schedule equals the back-to-back one bit for bit) are in test_gpu_engine.py / test_gpu_fullsize.py."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from test_gpu_kernels import close, conv_desc, dev  # noqa: E402


@pytest.fixture()
def ctx():
    from radnet_hip import lib as L
    c = L.Context(0)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    c.check(c.lib.radnet_set_workspace(c.h, ws.data_ptr(), ws.numel()), "ws")
    c._ws = ws
    yield c
    c.lib.radnet_force_config(c.h, 0, 0, 0)
    c.close()


WGRAD_SHAPES = [
    # nb, h, w, cin, cout, k, stride, pad
    (20, 7, 7, 512, 512, 3, 1, 1),           # res5x_branch2b on 20 RoIs: the train step's split weight gradient
    (1, 38, 63, 512, 64, 1, 1, 0),           # fused RPN heads (N = 64)
    (2, 19, 23, 128, 96, 3, 1, 1),           # ragged N
]


@pytest.mark.parametrize("shape", WGRAD_SHAPES)
@pytest.mark.parametrize("tile", [(64, 64), (128, 64), (64, 128), (128, 128)])
def test_split_wgrad_is_reproducible_and_right(ctx, shape, tile):
    from radnet_hip import lib as L
    from oracle import dense
    nb, h, w, cin, cout, k, stride, pad = shape
    if cin % tile[0]:
        pytest.skip("weight-gradient k tile does not divide the channels")
    rs = np.random.RandomState(sum(shape) + 5)
    x = rs.standard_normal((nb, h, w, cin)).astype(np.float32)
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = rs.standard_normal((nb, oh, ow, cout)).astype(np.float32)
    gs = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    wt = np.zeros((k, k, cin, cout), np.float32)
    _, dw_ref, db_ref = dense.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), (dy * gs).astype(np.float64), stride, (pad,) * 4, need_dx=False)
    xd, wd, dyd, gsd = dev(x), dev(wt.reshape(-1, cout)), dev(dy), dev(gs)
    nmt = (nb * oh * ow + 31) // 32
    for splits in (2, 3, 5, 8, 16):
        if splits > nmt // 2:
            continue
        ctx.check(ctx.lib.radnet_force_config(ctx.h, tile[0], tile[1], splits), "force")
        runs = []
        for rep in range(4):
            dw = torch.full((k * k * cin, cout), float("nan"), device="cuda")
            db = torch.full((cout,), float("nan"), device="cuda")
            d = conv_desc(L, xd, wd, dw, nb, h, w, cin, oh, ow, k, stride, pad, cout, cout)
            d.dy, d.ld_dy, d.gscale, d.dw, d.dw_accumulate, d.db = dyd.data_ptr(), cout, gsd.data_ptr(), dw.data_ptr(), 0, db.data_ptr()
            ctx.check(ctx.lib.radnet_conv_wgrad(ctx.h, C.byref(d)), "wgrad %s" % ((tile, splits),))
            runs.append((dw.cpu().numpy(), db.cpu().numpy()))
        close(runs[0][0], dw_ref.reshape(-1, cout))
        close(runs[0][1], db_ref)
        for a, b in runs[1:]:
            assert np.array_equal(a, runs[0][0]) and np.array_equal(b, runs[0][1]), ("not reproducible", tile, splits)
        # accumulate mode: the ordered sum is ADDED to what dw / db hold
        d.dw_accumulate = 1
        ctx.check(ctx.lib.radnet_conv_wgrad(ctx.h, C.byref(d)), "wgrad accumulate")
        close(dw.cpu().numpy(), 2 * dw_ref.reshape(-1, cout))
        close(db.cpu().numpy(), 2 * db_ref)
        # pre-zeroed mode stores
        dw0 = torch.zeros((k * k * cin, cout), device="cuda")
        d.dw, d.dw_accumulate, d.db = dw0.data_ptr(), 2, None
        ctx.check(ctx.lib.radnet_conv_wgrad(ctx.h, C.byref(d)), "wgrad prezeroed")
        assert np.array_equal(dw0.cpu().numpy(), runs[0][0])


def test_atomic_form_still_correct(ctx):
    """radnet_set_deterministic(ctx, 0): the fp32-atomics forms (A/B runs) against the oracle."""
    from radnet_hip import lib as L
    from oracle import dense
    ctx.check(ctx.lib.radnet_set_deterministic(ctx.h, 0), "det off")
    nb, h, w, cin, cout, k, stride, pad = WGRAD_SHAPES[0]
    rs = np.random.RandomState(77)
    x = rs.standard_normal((nb, h, w, cin)).astype(np.float32)
    dy = rs.standard_normal((nb, h, w, cout)).astype(np.float32)
    wt = np.zeros((k, k, cin, cout), np.float32)
    _, dw_ref, db_ref = dense.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64), stride, (pad,) * 4, need_dx=False)
    xd, wd, dyd = dev(x), dev(wt.reshape(-1, cout)), dev(dy)
    ctx.check(ctx.lib.radnet_force_config(ctx.h, 64, 64, 8), "force")
    dw = torch.full((k * k * cin, cout), float("nan"), device="cuda")
    db = torch.full((cout,), float("nan"), device="cuda")
    d = conv_desc(L, xd, wd, dw, nb, h, w, cin, h, w, k, stride, pad, cout, cout)
    d.dy, d.ld_dy, d.dw, d.dw_accumulate, d.db = dyd.data_ptr(), cout, dw.data_ptr(), 0, db.data_ptr()
    ctx.check(ctx.lib.radnet_conv_wgrad(ctx.h, C.byref(d)), "wgrad atomics")
    close(dw.cpu().numpy(), dw_ref.reshape(-1, cout))
    close(db.cpu().numpy(), db_ref)
    out = torch.full((cout,), float("nan"), device="cuda")
    ctx.call("radnet_colsum", dyd, nb * h * w, cout, cout, None, out, 0)
    close(out.cpu().numpy(), db_ref)


def test_wgrad_slabs_larger_than_the_workspace_fall_back(ctx):
    """A forced / loaded split whose partial tiles do not fit the workspace runs with fewer splits instead of failing."""
    from radnet_hip import lib as L
    from oracle import dense
    small = torch.empty(6 << 20, dtype=torch.uint8, device="cuda")           # 3x3 512->512: one split's slabs are 9.4 MB
    ctx.check(ctx.lib.radnet_set_workspace(ctx.h, small.data_ptr(), small.numel()), "ws")
    nb, h, w, cin, cout, k, stride, pad = WGRAD_SHAPES[0]
    rs = np.random.RandomState(78)
    x = rs.standard_normal((nb, h, w, cin)).astype(np.float32)
    dy = rs.standard_normal((nb, h, w, cout)).astype(np.float32)
    wt = np.zeros((k, k, cin, cout), np.float32)
    _, dw_ref, _ = dense.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64), stride, (pad,) * 4, need_dx=False)
    xd, wd, dyd = dev(x), dev(wt.reshape(-1, cout)), dev(dy)
    ctx.check(ctx.lib.radnet_force_config(ctx.h, 64, 64, 8), "force")
    dw = torch.full((k * k * cin, cout), float("nan"), device="cuda")
    d = conv_desc(L, xd, wd, dw, nb, h, w, cin, h, w, k, stride, pad, cout, cout)
    d.dy, d.ld_dy, d.dw, d.dw_accumulate = dyd.data_ptr(), cout, dw.data_ptr(), 0
    ctx.check(ctx.lib.radnet_conv_wgrad(ctx.h, C.byref(d)), "wgrad small workspace")
    close(dw.cpu().numpy(), dw_ref.reshape(-1, cout))


def test_batched_wgrad_reproducible(ctx):
    """radnet_wgrad_batched (Winograd-domain weight gradients): 16 problems, pixel-split, twice the same bits."""
    rs = np.random.RandomState(21)
    P, T, cin, cout = 16, 608, 256, 128
    V = rs.standard_normal((P, T, cin)).astype(np.float32)
    dZ = rs.standard_normal((P, T, cout)).astype(np.float32)
    ref = np.einsum("ptk,ptn->pkn", V.astype(np.float64), dZ.astype(np.float64))
    Vd, dZd = dev(V), dev(dZ)
    ctx.check(ctx.lib.radnet_force_config(ctx.h, 64, 64, 4), "force")
    outs = []
    for rep in range(3):
        dU = torch.full((P, cin, cout), float("nan"), device="cuda")
        ctx.call("radnet_wgrad_batched", Vd, dZd, dU, P, T, cin, cout, 0)
        outs.append(dU.cpu().numpy())
    close(outs[0], ref)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_conv_bwd_pair_reproducible(ctx):
    """The paired launch (dgrad + wgrad, conv_bwd_pair_kernel) with a K-split dgrad AND a pixel-split wgrad: both keep
    their partial tiles in the workspace (start / end) at the same time."""
    from radnet_hip import lib as L
    from oracle import dense
    nb, h, w, cin, cout, k, pad = 20, 7, 7, 512, 512, 3, 1
    rs = np.random.RandomState(31)
    x = np.maximum(rs.standard_normal((nb, h, w, cin)), 0).astype(np.float32)
    wt = (rs.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    dy = rs.standard_normal((nb, h, w, cout)).astype(np.float32)
    dx_ref, dw_ref, db_ref = dense.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64), 1, (pad,) * 4)
    dx_ref = dx_ref * (x > 0)
    xd, wd, dyd = dev(x), dev(wt.reshape(-1, cout)), dev(dy)
    ctx.check(ctx.lib.radnet_force_config(ctx.h, 64, 64, 3), "force")
    outs = []
    for rep in range(3):
        dx = torch.full((nb, h, w, cin), float("nan"), device="cuda")
        dw = torch.full((k * k * cin, cout), float("nan"), device="cuda")
        db = torch.full((cout,), float("nan"), device="cuda")
        d = conv_desc(L, xd, wd, dx, nb, h, w, cin, h, w, k, 1, pad, cout, cout)
        d.dy, d.ld_dy = dyd.data_ptr(), cout
        d.dx, d.ld_dx, d.dx_mask, d.ld_dx_mask = dx.data_ptr(), cin, xd.data_ptr(), cin
        d.dw, d.dw_accumulate, d.db = dw.data_ptr(), 0, db.data_ptr()
        ctx.check(ctx.lib.radnet_conv_bwd(ctx.h, C.byref(d)), "conv_bwd")
        outs.append((dx.cpu().numpy(), dw.cpu().numpy(), db.cpu().numpy()))
    close(outs[0][0], dx_ref)
    close(outs[0][1], dw_ref.reshape(-1, cout))
    close(outs[0][2], db_ref)
    for o in outs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(o, outs[0]))


@pytest.mark.parametrize("m,n", [(2394, 64), (9375, 512), (980, 2048), (100, 64), (300000, 128)])
def test_colsum_reproducible(ctx, m, n):
    rs = np.random.RandomState(m % 1000 + n)
    g = rs.standard_normal((m, n)).astype(np.float32)
    gs = rs.uniform(0.5, 1.5, n).astype(np.float32)
    ref = g.astype(np.float64).sum(0) * gs
    gd, gsd = dev(g), dev(gs)
    outs = []
    for rep in range(3):
        out = torch.full((n,), float("nan"), device="cuda")
        ctx.call("radnet_colsum", gd, m, n, n, gsd, out, 0)
        outs.append(out.cpu().numpy())
    close(outs[0], ref, rtol=2e-5)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    out = torch.ones((n,), device="cuda")
    ctx.call("radnet_colsum", gd, m, n, n, gsd, out, 1)                      # accumulate
    close(out.cpu().numpy(), ref + 1, rtol=2e-5)


def test_roi_resize_bwd_ordered(ctx):
    """Overlapping RoIs (the case where the atomics' order matters): ordered kernel against the oracle, repeated bit for bit,
    and the atomics form against the same oracle."""
    from oracle import dense
    rs = np.random.RandomState(41)
    rois = np.array([[0, 0, 63, 38], [5, 3, 7, 9], [6, 4, 9, 9], [60, 35, 8, 8], [10, 10, 1, 1], [3.9, 2.2, 20.7, 14.1], [20, 5, 14, 28],
                     [62, 37, 1, 1], [5, 3, 7, 9], [0, 0, 0, 5], [30, 20, 2, 3]], np.float32)
    for ps in (7, 14):
        dy = rs.standard_normal((len(rois), ps, ps, 1024)).astype(np.float32)
        ref = dense.roi_crop_resize_bwd((1, 38, 63, 1024), rois, ps, dy.astype(np.float64))
        outs = []
        for rep in range(3):
            dF = torch.zeros(1, 38, 63, 1024, device="cuda")
            ctx.call("radnet_roi_resize_bwd", dev(dy), 38, 63, 1024, dev(rois), len(rois), ps, dF)
            outs.append(dF.cpu().numpy())
        close(outs[0], ref, rtol=1e-5)
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
        dF = torch.ones(1, 38, 63, 1024, device="cuda")                       # adds to what dF holds
        ctx.call("radnet_roi_resize_bwd", dev(dy), 38, 63, 1024, dev(rois), len(rois), ps, dF)
        close(dF.cpu().numpy(), ref + 1, rtol=1e-5)
        ctx.check(ctx.lib.radnet_set_deterministic(ctx.h, 0), "det off")
        dF = torch.zeros(1, 38, 63, 1024, device="cuda")
        ctx.call("radnet_roi_resize_bwd", dev(dy), 38, 63, 1024, dev(rois), len(rois), ps, dF)
        close(dF.cpu().numpy(), ref, rtol=1e-5)
        ctx.check(ctx.lib.radnet_set_deterministic(ctx.h, 1), "det on")


def test_rpn_loss_reproducible(ctx):
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
    pd, yc, yr = dev(pred), dev(y_cls), dev(y_regr)
    outs = []
    for rep in range(4):
        dz = torch.full((M, 64), float("nan"), device="cuda")
        losses = torch.zeros(2, device="cuda")
        scratch = torch.zeros(8, dtype=torch.float64, device="cuda")
        ctx.call("radnet_rpn_loss", pd, 64, yc, yr, M, A, 0, dz, 64, losses, scratch)
        outs.append((losses.cpu().numpy(), dz.cpu().numpy(), scratch.cpu().numpy()[:4]))
    for o in outs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(o, outs[0]))


def test_loaded_split_choice_for_a_batched_launch_falls_back(ctx, tmp_path):
    """a tuning table (shared / loaded) that names a K-split or XCD-ordered unit table for a shape that
    is launched as a BATCH used to fail with a bare -3 (`radnet_program_run failed (-3): ?`).  A batch is one plain grid per
    problem: the launcher now falls back to the un-split shape, and when nothing can run it says why."""
    rs = np.random.RandomState(5)
    P, T, cin, cout = 36, 160, 256, 128
    V = rs.standard_normal((P, T, cin)).astype(np.float32)
    U = rs.standard_normal((P, cin, cout)).astype(np.float32)
    ref = np.einsum("ptk,pkn->ptn", V.astype(np.float64), U.astype(np.float64))
    for splits in (-4, 3):
        path = tmp_path / ("tune%d.txt" % splits)
        path.write_text("# radnet tuned GEMM launch shapes v2\n8 %d %d %d %d 1 %d 64 64 %d 0.010000 4\n" % (T, cout, cin, cin, P, splits))
        ctx.check(ctx.lib.radnet_tune_load(ctx.h, str(path).encode()), "tune_load")
        M = torch.full((P, T, cout), float("nan"), device="cuda")
        ctx.call("radnet_gemm_batched", dev(V), dev(U), M, P, T, cout, cin)
        close(M.cpu().numpy(), ref)
