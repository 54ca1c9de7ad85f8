"""Every launch variant of the conv GEMM kernels (output tile x K slices x workgroup order x waves per workgroup), forced
one by one through radnet_force_config / radnet_force_waves, against the oracle.  The autotuner may pick any of them at run time, so each must be correct on
its own -- including ragged M / N edges, stride 2, 3x3 halos, odd and even K-tile counts per slice."""
import ctypes as C
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from test_gpu_kernels import close, conv_desc, dev  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    from radnet_hip import lib as L
    c = L.Context(0)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    c.check(c.lib.radnet_set_workspace(c.h, ws.data_ptr(), ws.numel()), "ws")
    c._ws = ws
    yield c
    c.lib.radnet_force_config(c.h, 0, 0, 0)
    c.close()


FWD_SHAPES = [
    # nb, h, w, cin, cout, k, stride, pad
    (20, 14, 14, 1024, 512, 1, 2, 0),        # res5a_branch2a on 20 RoIs (M = 980)
    (1, 38, 63, 256, 256, 3, 1, 1),          # stage-4 3x3
    (2, 9, 11, 128, 96, 3, 1, 1),            # ragged N, tiny M
    (1, 21, 30, 64, 256, 1, 1, 0),           # short K (2 K tiles)
    (1, 17, 19, 224, 64, 1, 1, 0),           # 7 K tiles (odd): exercises the two-stage pipeline tail
]
CONFIGS = [(bm, bn, s, wv) for bm, bn in itertools.product((64, 128), (64, 128)) for s in (1, 2, 3, 5, -1, -4) for wv in (4, 8)]
# round 4: 32-row tiles (one wave row, the K tile split between the waves left over), 4 waves only
CONFIGS += [(32, bn, s, 4) for bn in (64, 32) for s in (1, 2, 3, 5, -1, -4)]


@pytest.mark.parametrize("shape", FWD_SHAPES)
def test_fwd_and_dgrad_every_config(ctx, shape):
    from radnet_hip import lib as L
    from oracle import dense
    nb, h, w, cin, cout, k, stride, pad = shape
    rs = np.random.RandomState(sum(shape))
    x = np.maximum(rs.standard_normal((nb, h, w, cin)), 0).astype(np.float32)
    wt = (rs.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rs.standard_normal(cout).astype(np.float32)
    sc = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    ref = np.maximum(dense.conv2d(x.astype(np.float64), wt.astype(np.float64), None, stride, (pad,) * 4) * sc + b, 0)
    xd, wd, sd, bd = dev(x), dev(wt.reshape(-1, cout)), dev(sc), dev(b)
    nk = (k * k * cin + 31) // 32
    do_dgrad = stride == 1
    if do_dgrad:
        dy = rs.standard_normal((nb, oh, ow, cout)).astype(np.float32)
        dx_ref, _, _ = dense.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), (dy * sc).astype(np.float64), 1, (pad,) * 4)
        dx_ref = dx_ref * (x > 0)
        dyd = dev(dy)
        nk_d = (k * k * cout + 31) // 32
    for bm, bn, s, wv in CONFIGS:
        if abs(s) > nk:
            continue
        ctx.check(ctx.lib.radnet_force_config(ctx.h, bm, bn, s), "force")
        ctx.check(ctx.lib.radnet_force_waves(ctx.h, wv), "force waves")
        y = torch.full((nb, oh, ow, cout), float("nan"), dtype=torch.float32, device="cuda")
        d = conv_desc(L, xd, wd, y, nb, h, w, cin, oh, ow, k, stride, pad, cout, cout, sd, bd, None, 1)
        ctx.check(ctx.lib.radnet_conv_fwd(ctx.h, C.byref(d)), "conv_fwd %s" % ((bm, bn, s, wv),))
        close(y.cpu().numpy(), ref)
        if do_dgrad and abs(s) <= nk_d:
            dx = torch.full((nb, h, w, cin), float("nan"), device="cuda")
            d.dy, d.ld_dy, d.gscale = dyd.data_ptr(), cout, sd.data_ptr()
            d.dx, d.ld_dx, d.dx_add, d.dx_mask, d.ld_dx_mask = dx.data_ptr(), cin, None, xd.data_ptr(), cin
            ctx.check(ctx.lib.radnet_conv_dgrad(ctx.h, C.byref(d)), "conv_dgrad %s" % ((bm, bn, s, wv),))
            close(dx.cpu().numpy(), dx_ref)
    ctx.lib.radnet_force_config(ctx.h, 0, 0, 0)
    ctx.lib.radnet_force_waves(ctx.h, 0)


@pytest.mark.parametrize("shape", [(20, 7, 7, 512, 512, 3, 1, 1), (20, 14, 14, 1024, 512, 1, 2, 0), (1, 19, 23, 128, 96, 3, 1, 1)])
def test_wgrad_every_config(ctx, shape):
    from radnet_hip import lib as L
    from oracle import dense
    nb, h, w, cin, cout, k, stride, pad = shape
    rs = np.random.RandomState(sum(shape) + 1)
    x = rs.standard_normal((nb, h, w, cin)).astype(np.float32)
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = rs.standard_normal((nb, oh, ow, cout)).astype(np.float32)
    gs = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    wt = np.zeros((k, k, cin, cout), np.float32)
    _, dw_ref, _ = dense.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), (dy * gs).astype(np.float64), stride, (pad,) * 4, need_dx=False)
    xd, wd, dyd, gsd = dev(x), dev(wt.reshape(-1, cout)), dev(dy), dev(gs)
    nmt = (nb * oh * ow + 31) // 32
    for bmk, bn, s in itertools.product((64, 128), (64, 128), (1, 2, 3, 8)):
        if cin % bmk or s > nmt // 2 and s > 1:
            continue
        ctx.check(ctx.lib.radnet_force_config(ctx.h, bmk, bn, s), "force")
        for mode in (0, 2):
            dw = torch.full((k * k * cin, cout), float("nan") if mode == 0 else 0.0, device="cuda")
            d = conv_desc(L, xd, wd, dw, nb, h, w, cin, oh, ow, k, stride, pad, cout, cout)
            d.dy, d.ld_dy, d.gscale, d.dw, d.dw_accumulate = dyd.data_ptr(), cout, gsd.data_ptr(), dw.data_ptr(), mode
            ctx.check(ctx.lib.radnet_conv_wgrad(ctx.h, C.byref(d)), "wgrad %s" % ((bmk, bn, s, mode),))
            close(dw.cpu().numpy(), dw_ref.reshape(-1, cout))
    ctx.lib.radnet_force_config(ctx.h, 0, 0, 0)


@pytest.mark.parametrize("shape", [(36, 160, 256, 256), (16, 75, 128, 96), (36, 23, 64, 64)])
def test_batched_launches_xcd_contiguous_numbering_changes_no_bit(ctx, shape):
    """slices = -1 on a batched launch (radnet_gemm_batched / radnet_wgrad_batched: the positions of a Winograd layer) only
    renumbers the workgroups -- XCD k runs the k-th eighth of the (problem, tile) list -- so the results are the plain grid's
    bit for bit, for grids that divide by 8 and grids that do not, every tile shape, 4 and 8 waves."""
    batch, T, c, n = shape
    rs = np.random.RandomState(sum(shape))
    V = dev(rs.standard_normal((batch, T, c)).astype(np.float32))
    U = dev(rs.standard_normal((batch, c, n)).astype(np.float32))
    dZ = dev(rs.standard_normal((batch, T, n)).astype(np.float32))
    ref = np.einsum("ptc,pcn->ptn", V.cpu().numpy().astype(np.float64), U.cpu().numpy().astype(np.float64))
    for bm, bn, wv in [(64, 64, 4), (64, 64, 8), (64, 128, 4), (128, 64, 4), (128, 128, 8), (32, 64, 4), (32, 32, 4)]:
        if bn > 64 and n <= 64:
            continue
        out = {}
        for s in (1, -1):
            ctx.check(ctx.lib.radnet_force_config(ctx.h, bm, bn, s), "force")
            ctx.check(ctx.lib.radnet_force_waves(ctx.h, wv), "waves")
            M = torch.zeros(batch, T, n, device="cuda")
            ctx.call("radnet_gemm_batched", V, U, M, batch, T, n, c)
            out[s] = M.cpu().numpy()
        assert np.array_equal(out[1], out[-1]), (shape, bm, bn, wv)
        close(out[-1], ref)
        # round 4, persistent form (|slices| = z > 1 on a batch): z consecutive problems per workgroup as one long K loop --
        # the same sums in the same order, so the same bits; z that does not divide the batch leaves a shorter last group
        if wv == 4 and (bm, bn) in ((64, 64), (32, 64), (64, 128), (32, 32)):
            for z in (2, 3, -4, 5, 7, -12):
                if abs(z) > batch:
                    continue
                ctx.check(ctx.lib.radnet_force_config(ctx.h, bm, bn, z), "force")
                M = torch.full((batch, T, n), float("nan"), device="cuda")
                ctx.call("radnet_gemm_batched", V, U, M, batch, T, n, c)
                assert np.array_equal(M.cpu().numpy(), out[1]), (shape, bm, bn, z)
    ctx.check(ctx.lib.radnet_force_waves(ctx.h, 0), "waves off")
    dref = np.einsum("ptc,ptn->pcn", V.cpu().numpy().astype(np.float64), dZ.cpu().numpy().astype(np.float64))
    for bmk, bn in [(64, 64), (64, 128), (128, 64) if c % 128 == 0 else (64, 64)]:
        if bn > 64 and n <= 64:
            continue
        out = {}
        for s in (1, -1):
            ctx.check(ctx.lib.radnet_force_config(ctx.h, bmk, bn, s), "force")
            dU = torch.zeros(batch, c, n, device="cuda")
            ctx.call("radnet_wgrad_batched", V, dZ, dU, batch, T, c, n, 0)
            out[s] = dU.cpu().numpy()
        assert np.array_equal(out[1], out[-1]), (shape, bmk, bn)
        close(out[-1], dref)
    ctx.lib.radnet_force_config(ctx.h, 0, 0, 0)
