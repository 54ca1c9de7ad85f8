"""BASELINE.json's full sizes (cfg 2: 1000x600 panel, 12 anchors, 20 RoIs) through size-independent properties -- the
oracle needs minutes at these sizes, so nothing here calls it:
  * linearity: a bias-free conv (+ReLU) commutes with scaling its input by a power of two, exactly (fp32 scaling by 2 is
    exact, relu(2y) = 2 relu(y)) -- on the stem (7x7, 150 000 rows), a stage-2 1x1 (37 101 rows), and rpn_conv1 in both its
    direct and its Winograd form, which must also agree with each other to fp32 re-association error;
  * proposals: scores come out non-increasing, at most max_boxes, and NMS is idempotent on its own output;
  * anchor labels: at most 256 valid anchors after the subsampling, every ground-truth box owns a positive anchor;
  * the pipelined train step equals back-to-back steps at full size (same seeded inputs, same RNG consumption)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from test_gpu_kernels import conv_desc, dev  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    from radnet_hip import lib as L
    c = L.Context(0)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    c.check(c.lib.radnet_set_workspace(c.h, ws.data_ptr(), ws.numel()), "ws")
    c.check(c.lib.radnet_set_autotune(c.h, 1), "tune")
    c._ws = ws
    yield c
    c.close()


@pytest.mark.parametrize("shape", [(1, 600, 1000, 4, 64, 7, 2, 3), (1, 149, 249, 64, 256, 1, 1, 0), (1, 38, 63, 1024, 512, 3, 1, 1)],
                         ids=["stem 7x7 s2", "res2 1x1 64->256", "rpn_conv1 3x3"])
def test_conv_scales_exactly_with_its_input(ctx, shape):
    from radnet_hip import lib as L
    nb, h, w, cin, cout, k, stride, pad = shape
    rs = np.random.RandomState(sum(shape))
    x = torch.from_numpy(rs.standard_normal((nb, h, w, cin)).astype(np.float32)).cuda()
    if cin == 4:
        x[..., 3] = 0                                                   # the padded fourth input channel of the stem
    wt = torch.from_numpy((rs.standard_normal((k * k * cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)).cuda()
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    outs = []
    for scale in (1.0, 2.0):
        y = torch.full((nb, oh, ow, cout), float("nan"), dtype=torch.float32, device="cuda")
        xs = x * scale
        d = conv_desc(L, xs, wt, y, nb, h, w, cin, oh, ow, k, stride, pad, cout, cout, None, None, None, 1)
        ctx.check(ctx.lib.radnet_conv_fwd(ctx.h, C.byref(d)), "conv_fwd")
        outs.append(y)
    assert torch.isfinite(outs[0]).all() and (outs[0] > 0).any()
    assert torch.equal(outs[1], 2.0 * outs[0])
    if k == 3 and stride == 1 and cin % 32 == 0:                        # the same layer as Winograd F(2x2,3x3)
        T = nb * ((h + 1) // 2) * ((w + 1) // 2)
        U = torch.empty(16, cin, cout, device="cuda"); V = torch.empty(16, T, cin, device="cuda"); M = torch.empty(16, T, cout, device="cuda")
        ctx.call("radnet_winograd_filter", wt, cin, cout, cout, U)
        wino = []
        for scale in (1.0, 2.0):
            y = torch.full((nb, h, w, cout), float("nan"), dtype=torch.float32, device="cuda")
            ctx.call("radnet_winograd_input", x * scale, nb, h, w, cin, V)
            ctx.call("radnet_gemm_batched", V, U, M, 16, T, cout, cin)
            ctx.call("radnet_winograd_output", M, nb, h, w, cout, None, None, 1, y, cout)
            wino.append(y)
        assert torch.equal(wino[1], 2.0 * wino[0])
        assert (wino[0] - outs[0]).abs().max() < 1e-4 * outs[0].abs().max()


def test_full_size_proposals_sorted_bounded_idempotent(ctx):
    rows, cols, A = 38, 63, 12
    rs = np.random.RandomState(7)
    pred = np.zeros((rows * cols, 64), np.float32)
    pred[:, :A] = rs.uniform(0, 1, (rows * cols, A)); pred[:, A:5 * A] = rs.standard_normal((rows * cols, 4 * A)) * 0.5
    awh = np.array([[(s * r[0]) / 16, (s * r[1]) / 16] for s in (64, 128, 256, 512) for r in ([1, 1], [1. / np.sqrt(2), 2. / np.sqrt(2)], [2. / np.sqrt(2), 1. / np.sqrt(2)])], dtype=np.float64)
    mb = 300
    R = torch.zeros(mb, 4, dtype=torch.int64, device="cuda"); Rp = torch.zeros(mb, device="cuda"); Rn = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = torch.empty(int(ctx.lib.radnet_proposals_ws_bytes(rows * cols * A)), dtype=torch.uint8, device="cuda")
    predd = dev(pred)
    ctx.check(ctx.lib.radnet_rpn_to_roi(ctx.h, predd.data_ptr(), 64, rows, cols, A, awh.ctypes.data_as(C.POINTER(C.c_double)), 4.0, 1, 0.7, mb,
                                        R.data_ptr(), Rp.data_ptr(), Rn.data_ptr(), ws.data_ptr()), "rpn_to_roi")
    n = int(Rn.cpu()[0])
    assert 0 < n <= mb
    boxes, probs = R.cpu().numpy()[:n], Rp.cpu().numpy()[:n]
    assert np.all(np.diff(probs) <= 0)                                  # picked in descending score order
    assert np.all(boxes[:, 2] > boxes[:, 0]) and np.all(boxes[:, 3] > boxes[:, 1])
    assert boxes.min() >= 0 and boxes[:, 2].max() <= cols and boxes[:, 3].max() <= rows
    # idempotence: no survivor suppresses another one at the same threshold
    bd = torch.from_numpy(boxes.astype(np.float64)).cuda(); pd = torch.from_numpy(probs.astype(np.float32)).cuda()
    idx = torch.zeros(mb, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws2 = torch.empty(int(ctx.lib.radnet_proposals_ws_bytes(n)), dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.radnet_nms(ctx.h, bd.data_ptr(), pd.data_ptr(), n, 0.7, mb, idx.data_ptr(), cnt.data_ptr(), ws2.data_ptr()), "nms")
    assert int(cnt.cpu()[0]) == n and sorted(idx.cpu().numpy()[:n].tolist()) == list(range(n))


@pytest.fixture(scope="module")
def full_step():
    from faster_rcnn.config import Config
    from radnet_hip import synth
    from radnet_hip.engine import FasterRCNNEngine
    from radnet_hip.trainer import TrainStep
    C = Config()
    batches = []
    for i in range(4):
        meta = synth.synthetic_gt(2 + i, n=8, src_w=2000, src_h=1200)
        batches.append([dict(img=synth.synthetic_panel(1 + i, 600, 1000), bboxes=meta["bboxes"], width=2000, height=1200)])
    return C, batches, FasterRCNNEngine, TrainStep, synth


def test_full_size_anchor_labels_properties(full_step):
    C, batches, Engine, TrainStep, synth = full_step
    eng = Engine(C)
    ts = TrainStep(eng)
    np.random.seed(64)
    for b in batches[:2]:
        s = b[0]
        tp = eng.anchor_targets_launch(ts._gt(s), s["width"], s["height"], 1000, 600)
        ycls, yregr, n_pos = eng.anchor_targets_finish(tp)
        y = ycls.cpu().numpy().reshape(38, 63, 2 * eng.A)
        valid, pos = y[..., :eng.A], y[..., eng.A:]
        assert 0 < valid.sum() <= 256                                    # utils.py:777-813: num_regions = 256
        assert (pos * valid).sum() == n_pos and 0 < n_pos <= 128         # valid positives = the count reported, at most half
        assert set(np.unique(valid)) <= {0.0, 1.0} and set(np.unique(pos)) <= {0.0, 1.0}


def test_full_size_pipelined_equals_back_to_back(full_step):
    C, batches, Engine, TrainStep, synth = full_step
    W0 = synth.synthetic_weights(seed=3)
    res, tune = [], None
    for pipelined in (False, True):
        eng = Engine(C)
        if tune is not None:
            eng.load_tuning(tune)
        eng.set_weights(W0)
        np.random.seed(64)
        ts = TrainStep(eng)
        losses = []
        for k, b in enumerate(batches):
            ts.step(b, upcoming=batches[k + 1:k + 4] if pipelined else None)
            losses.append(ts.losses())
        ts.flush()
        res.append((losses, eng.get_weights(), np.random.randint(0, 2 ** 31 - 1), ts.skipped_head_steps))
        if tune is None:
            import tempfile
            tune = tempfile.mktemp(suffix=".txt")
            eng.save_tuning(tune)
    (l0, w0, r0, s0), (l1, w1, r1, s1) = res
    assert r0 == r1 and s0 == s1 == 0
    # ordered reductions (radnet_set_deterministic, default) + one table of launch shapes: bit for bit, every loss, every weight
    for i, (a, b) in enumerate(zip(l0, l1)):
        for key in ("rpn_cls", "rpn_regr", "det_cls", "det_regr", "det_acc"):
            assert a[key] == b[key], (i, key, a[key], b[key])
    for name in w0:
        for k in w0[name]:
            assert np.array_equal(w0[name][k], w1[name][k]), (name, k, float(np.abs(w0[name][k] - w1[name][k]).max()))
