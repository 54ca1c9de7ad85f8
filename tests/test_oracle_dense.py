"""oracle/dense.py (NumPy restatement of the Keras/TF graph) cross-checked against torch CPU ops and
autograd -- an independent second implementation, not the reference (TF is absent: parity unpinned)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import dense

torch.set_num_threads(4)


def t_conv(x, w, b, stride, pad):
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    pt, pl, pb, pr = pad
    xt = TF.pad(xt, (pl, pr, pt, pb))
    wt = torch.from_numpy(w).permute(3, 2, 0, 1)
    return TF.conv2d(xt, wt, None if b is None else torch.from_numpy(b), stride=stride).permute(0, 2, 3, 1)


@pytest.mark.parametrize("k,stride,pad,hw", [(1, 1, (0, 0, 0, 0), (9, 11)), (3, 1, (1, 1, 1, 1), (9, 11)), (1, 2, (0, 0, 0, 0), (14, 14)),
                                             (7, 2, (3, 3, 3, 3), (20, 23)), (1, 2, (0, 0, 0, 0), (9, 12))])
def test_conv_fwd_bwd_vs_torch(k, stride, pad, hw):
    rs = np.random.RandomState(0)
    x = rs.standard_normal((2, hw[0], hw[1], 5))
    w = rs.standard_normal((k, k, 5, 6))
    b = rs.standard_normal(6)
    y = dense.conv2d(x, w, b, stride, pad)
    xt = torch.from_numpy(x).requires_grad_(True)
    wt = torch.from_numpy(w).requires_grad_(True)
    bt = torch.from_numpy(b).requires_grad_(True)
    xp = TF.pad(xt.permute(0, 3, 1, 2), (pad[1], pad[3], pad[0], pad[2]))
    yt = TF.conv2d(xp, wt.permute(3, 2, 0, 1), bt, stride=stride).permute(0, 2, 3, 1)
    assert y.shape == tuple(yt.shape)
    assert np.allclose(y, yt.detach().numpy(), atol=1e-11)
    dy = rs.standard_normal(y.shape)
    yt.backward(torch.from_numpy(dy))
    dx, dw, db = dense.conv2d_bwd(x, w, dy, stride, pad)
    assert np.allclose(dx, xt.grad.numpy(), atol=1e-10)
    assert np.allclose(dw, wt.grad.numpy(), atol=1e-10)
    assert np.allclose(db, bt.grad.numpy(), atol=1e-10)


def test_pools_vs_torch():
    rs = np.random.RandomState(1)
    x = rs.standard_normal((1, 21, 30, 4))
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    assert np.array_equal(dense.maxpool_3x3_s2(x), TF.max_pool2d(xt, 3, 2).permute(0, 2, 3, 1).numpy())
    assert np.array_equal(dense.maxpool_2x2_s2(x), TF.max_pool2d(xt, 2, 2).permute(0, 2, 3, 1).numpy())


def test_resnet_feature_size_matches_formula():
    from oracle import glue
    P = {"conv1": {"kernel": np.zeros((7, 7, 3, 2), np.float32), "bias": np.zeros(2, np.float32)}}
    for L in (600, 800, 1000, 61, 97):
        x = np.zeros((1, L, 64, 3), np.float32)
        y = dense.maxpool_3x3_s2(dense.conv2d(x, P["conv1"]["kernel"], None, 2, (3, 3, 3, 3)))
        n = y.shape[1]
        n = (n - 1) // 2 + 1      # stage 3 stride-2 1x1
        n = (n - 1) // 2 + 1      # stage 4
        assert n == glue.resnet50_feat_len(L)


def test_roi_crop_resize_semantics():
    # TF1 legacy bilinear: src = dst * in/out, no half-pixel shift; hand-checked on a ramp
    H, W = 6, 9
    F = (np.arange(H)[:, None] * 10.0 + np.arange(W)[None, :]).astype(np.float32)[None, :, :, None]
    out = dense.roi_crop_resize(F, np.array([[2, 1, 4, 3]]), 2)          # crop rows 1..3, cols 2..5
    # scale h = 3/2, w = 4/2: samples at y = {0, 1.5}, x = {0, 2}
    exp = np.array([[F[0, 1, 2, 0], F[0, 1, 4, 0]], [(F[0, 2, 2, 0] + F[0, 3, 2, 0]) / 2, (F[0, 2, 4, 0] + F[0, 3, 4, 0]) / 2]])
    assert np.allclose(out[0, :, :, 0], exp)
    # upsampling a 1x1 crop replicates the pixel; RoI sticking out of the map is clamped by the slice
    out = dense.roi_crop_resize(F, np.array([[8, 5, 4, 4]]), 3)
    assert np.all(out == F[0, 5, 8, 0])
    # float RoIs are truncated like K.cast(..., 'int32')
    a = dense.roi_crop_resize(F, np.array([[2.9, 1.2, 4.7, 3.1]]), 2)
    assert np.array_equal(a, dense.roi_crop_resize(F, np.array([[2, 1, 4, 3]]), 2))


def test_roi_crop_resize_bwd_is_adjoint():
    rs = np.random.RandomState(2)
    F = rs.standard_normal((1, 7, 8, 3))
    rois = np.array([[1, 2, 5, 4], [0, 0, 8, 7], [6, 5, 1, 1]])
    y = dense.roi_crop_resize(F, rois, 4)
    dy = rs.standard_normal(y.shape)
    dF = dense.roi_crop_resize_bwd(F.shape, rois, 4, dy)
    G = rs.standard_normal(F.shape)
    # <resize(G), dy> == <G, resize^T(dy)>
    assert np.isclose((dense.roi_crop_resize(G, rois, 4) * dy).sum(), (G * dF).sum())


def _numgrad(f, x, eps=1e-6):
    g = np.zeros_like(x)
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        old = x[i]
        x[i] = old + eps; a = f()
        x[i] = old - eps; b = f()
        x[i] = old
        g[i] = (a - b) / (2 * eps)
    return g


def test_loss_gradients_numeric():
    rs = np.random.RandomState(3)
    A = 3
    valid = (rs.uniform(size=(1, 4, 5, A)) < 0.5).astype(np.float64)
    ov = ((rs.uniform(size=(1, 4, 5, A)) < 0.4) * valid)
    y_cls = np.concatenate([valid, ov], -1)
    p = rs.uniform(0.05, 0.95, (1, 4, 5, A))
    for mode in (True, False):
        val, g = dense.rpn_loss_cls(y_cls, p, A, mode)
        ng = _numgrad(lambda: dense.rpn_loss_cls(y_cls, p, A, mode)[0], p)
        assert np.allclose(g, ng, atol=1e-6), mode
    # textbook mode equals torch BCE
    val, _ = dense.rpn_loss_cls(y_cls, p, A, False)
    tb = TF.binary_cross_entropy(torch.from_numpy(p), torch.from_numpy(ov), reduction="none").numpy()
    assert np.isclose(val, (valid * tb).sum() / (1e-4 * valid.size + valid.sum()))
    # Keras-2 argument order: 16.118*p for negatives, 15.942*(1-p) for positives (fp32 clip asymmetry)
    val2, _ = dense.rpn_loss_cls(y_cls, p, A, True)
    l0 = float(np.log(np.float32(1e-7) / (np.float32(1) - np.float32(1e-7))))
    hi = np.float32(1) - np.float32(1e-7)
    l1 = float(np.log(hi / (np.float32(1) - hi)))
    assert abs(l0 + 16.118) < 1e-3 and abs(l1 - 15.942) < 1e-3
    ce = np.where(ov == 1, l1 * (1 - p) + np.log1p(np.exp(-l1)), -l0 * p + np.log1p(np.exp(l0)))
    assert np.isclose(val2, (valid * ce).sum() / (1e-4 * valid.size + valid.sum()))

    n = 8
    mask = np.repeat((rs.uniform(size=(1, 6, 2)) < 0.5).astype(np.float64), 4, axis=-1)
    tgt = rs.standard_normal((1, 6, n)) * 2
    yt = np.concatenate([mask, tgt], -1)
    pred = rs.standard_normal((1, 6, n))
    val, g = dense.smooth_l1_masked(yt, pred, n)
    assert np.allclose(g, _numgrad(lambda: dense.smooth_l1_masked(yt, pred, n)[0], pred), atol=1e-6)
    ref = TF.smooth_l1_loss(torch.from_numpy(pred), torch.from_numpy(tgt), reduction="none", beta=1.0).numpy()
    assert np.isclose(val, (mask * ref).sum() / (1e-4 * mask.size + mask.sum()))

    q = dense.softmax(rs.standard_normal((1, 5, 7)))
    Y1 = np.eye(7)[rs.randint(0, 7, 5)][None]
    val, g = dense.class_loss_cls(Y1, q)
    assert np.allclose(g, _numgrad(lambda: dense.class_loss_cls(Y1, q)[0], q), atol=1e-6)
    assert np.isclose(val, TF.nll_loss(torch.log(torch.from_numpy(q[0])), torch.from_numpy(Y1[0].argmax(-1))).item())


def test_adam_matches_torch_adam():
    rs = np.random.RandomState(4)
    p = rs.standard_normal(50); p0 = p.copy()
    m = np.zeros(50); v = np.zeros(50)
    pt = torch.from_numpy(p0.copy()).requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=5e-5, betas=(0.9, 0.999), eps=0.0)
    for t in range(1, 6):
        g = rs.standard_normal(50)
        dense.adam_step(p, g, m, v, t, 5e-5, eps=0.0)
        pt.grad = torch.from_numpy(g.copy())
        opt.step()
        assert np.allclose(p, pt.detach().numpy(), rtol=0, atol=1e-12)


# ---- whole sub-graphs against torch autograd (float64, small spatial size) ----------------------
def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _t_cbr(P, T, x, conv, bn, stride=1, pad=0, relu=True, add=None):
    w = T[conv]["kernel"].permute(3, 2, 0, 1)
    z = TF.conv2d(x, w, T[conv]["bias"], stride=stride, padding=pad)
    s = T[bn]["gamma"] / torch.sqrt(T[bn]["var"] + 1e-3)
    y = (z - T[bn]["mean"].view(1, -1, 1, 1)) * s.view(1, -1, 1, 1) + T[bn]["beta"].view(1, -1, 1, 1)
    if add is not None:
        y = y + add
    return torch.relu(y) if relu else y


def _t_block(P, T, x, st, bl, stride, first):
    base = "res%d%s_branch" % (st, bl); bnb = "bn%d%s_branch" % (st, bl)
    a = _t_cbr(P, T, x, base + "2a", bnb + "2a", stride=stride if first else 1)
    b = _t_cbr(P, T, a, base + "2b", bnb + "2b", pad=1)
    sc = _t_cbr(P, T, x, base + "1", bnb + "1", stride=stride, relu=False) if first else x
    return _t_cbr(P, T, b, base + "2c", bnb + "2c", add=sc)


@pytest.fixture(scope="module")
def params64():
    P = dense.init_params(seed=3, dtype=np.float64)
    T = {n: {k: _t(v).clone().requires_grad_(n.startswith(("res5", "rpn", "dense"))) for k, v in d.items()} for n, d in P.items()}
    return P, T


def test_base_and_rpn_forward_vs_torch(params64):
    P, T = params64
    rs = np.random.RandomState(5)
    x = rs.uniform(-120, 130, (1, 70, 90, 3))
    F = dense.base_forward(P, x)
    xt = _t(x).permute(0, 3, 1, 2)
    y = _t_cbr(P, T, TF.pad(xt, (3, 3, 3, 3)), "conv1", "bn_conv1", stride=2)
    y = TF.max_pool2d(y, 3, 2)
    for st, blocks, _, stride in dense.RES_STAGES:
        for bl in blocks:
            y = _t_block(P, T, y, st, bl, stride, bl == "a")
    Ft = y.permute(0, 2, 3, 1).detach().numpy()
    assert F.shape == Ft.shape == (1, 5, 6, 1024)
    assert np.allclose(F, Ft, rtol=1e-9, atol=1e-9)
    assert 0.05 < np.abs(F).mean() < 50          # synthetic init keeps activations O(1)


def test_rpn_and_head_backward_vs_torch_autograd(params64):
    P, T = params64
    rs = np.random.RandomState(6)
    A, nc = 12, 7
    F = np.maximum(rs.standard_normal((1, 6, 7, 1024)), 0)
    # ---- RPN losses
    valid = (rs.uniform(size=(1, 6, 7, A)) < 0.3).astype(np.float64)
    ov = (rs.uniform(size=(1, 6, 7, A)) < 0.3) * valid
    y_cls = np.concatenate([valid, ov], -1)
    y_regr = np.concatenate([np.repeat(ov, 4, -1), rs.standard_normal((1, 6, 7, 4 * A))], -1)
    for mode in (True, False):
        losses, grads = dense.rpn_losses_and_grads(P, F, y_cls, y_regr, A, mode)
        Ft = _t(F).permute(0, 3, 1, 2)
        h = torch.relu(TF.conv2d(Ft, T["rpn_conv1"]["kernel"].permute(3, 2, 0, 1), T["rpn_conv1"]["bias"], padding=1))
        pc = torch.sigmoid(TF.conv2d(h, T["rpn_out_class"]["kernel"].permute(3, 2, 0, 1), T["rpn_out_class"]["bias"])).permute(0, 2, 3, 1)
        pr = TF.conv2d(h, T["rpn_out_regress"]["kernel"].permute(3, 2, 0, 1), T["rpn_out_regress"]["bias"]).permute(0, 2, 3, 1)
        vt, ot = _t(valid), _t(ov)
        if mode:
            l = _t(dense._bce_logits_swapped(ov).astype(np.float64))
            ce = torch.clamp(l, min=0) - l * pc + torch.log1p(torch.exp(-torch.abs(l)))
        else:
            pcl = torch.clamp(pc, 1e-7, 1 - 1e-7)
            ce = TF.binary_cross_entropy(pcl, ot, reduction="none")
        lc = (vt * ce).sum() / (1e-4 + vt).sum()
        mk = _t(y_regr[..., :4 * A]); tg = _t(y_regr[..., 4 * A:])
        lr = (mk * TF.smooth_l1_loss(pr, tg, reduction="none")).sum() / (1e-4 + mk).sum()
        for n in dense.RPN_TRAINABLE:
            for k in T[n]:
                T[n][k].grad = None
        (lc + lr).backward()
        assert np.isclose(losses[1], lc.item()) and np.isclose(losses[2], lr.item())
        for n in dense.RPN_TRAINABLE:
            for k in ("kernel", "bias"):
                assert np.allclose(grads[n][k], T[n][k].grad.numpy(), rtol=1e-7, atol=1e-10), (n, k, mode)

    # ---- classifier head
    rois = np.array([[0, 0, 3, 4], [2, 1, 5, 4], [4, 3, 2, 2]])
    Y1 = np.eye(nc)[[1, 6, 3]][None]
    lab = np.zeros((3, 4 * (nc - 1))); lab[0, 4:8] = 1; lab[2, 12:16] = 1
    Y2 = np.concatenate([lab, rs.standard_normal((3, 4 * (nc - 1))) * lab], -1)[None]
    losses, grads = dense.head_losses_and_grads(P, F, rois, Y1, Y2, nc)
    pooled = dense.roi_crop_resize(F, rois, 14)
    y = _t(pooled).permute(0, 3, 1, 2)
    for bl in "abc":
        y = _t_block(P, T, y, 5, bl, 2, bl == "a")
    feat = y.mean(dim=(2, 3))
    dcn, drn = "dense_class_%d" % nc, "dense_regress_%d" % nc
    logits = feat @ T[dcn]["kernel"] + T[dcn]["bias"]
    pr = feat @ T[drn]["kernel"] + T[drn]["bias"]
    lc = TF.cross_entropy(logits, _t(Y1[0].argmax(-1)))
    mk = _t(Y2[0, :, :24]); tg = _t(Y2[0, :, 24:])
    lr = (mk * TF.smooth_l1_loss(pr, tg, reduction="none")).sum() / (1e-4 + mk).sum()
    names = dense.head_trainable(nc)
    for n in names:
        for k in T[n]:
            T[n][k].grad = None
    (lc + lr).backward()
    assert np.isclose(losses[1], lc.item()) and np.isclose(losses[2], lr.item())
    for n in names:
        for k in ("kernel", "bias"):
            assert np.allclose(grads[n][k], T[n][k].grad.numpy(), rtol=1e-6, atol=1e-9), (n, k)


def test_base_backward_stages_3_4_vs_torch_autograd():
    """cont_train.py trainability: gradients of every stage-3/4 conv given dL/dF, against torch autograd (float64)."""
    P = dense.init_params(seed=3, dtype=np.float64)
    names = dense.s34_trainable()
    assert len(names) == 4 * 3 + 1 + 6 * 3 + 1 and all(n.startswith(("res3", "res4")) for n in names)
    T = {n: {k: _t(v).clone().requires_grad_(n in names) for k, v in d.items()} for n, d in P.items()}
    rs = np.random.RandomState(11)
    x = rs.uniform(-120, 130, (1, 70, 90, 3))
    F, caches = dense.base_forward(P, x, want_cache=True)
    dF = rs.standard_normal(F.shape)
    grads = dense.base_backward(P, caches, dF)
    assert sorted(grads) == sorted(names)
    y = _t_cbr(P, T, TF.pad(_t(x).permute(0, 3, 1, 2), (3, 3, 3, 3)), "conv1", "bn_conv1", stride=2)
    y = TF.max_pool2d(y, 3, 2)
    for st, blocks, _, stride in dense.RES_STAGES:
        for bl in blocks:
            y = _t_block(P, T, y, st, bl, stride, bl == "a")
    (y.permute(0, 2, 3, 1) * _t(dF)).sum().backward()
    for n in names:
        for k in ("kernel", "bias"):
            ref = T[n][k].grad.numpy()
            assert np.allclose(grads[n][k], ref, rtol=1e-8, atol=1e-10 * max(1.0, np.abs(ref).max())), (n, k)
