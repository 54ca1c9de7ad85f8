"""cont_train.py trainability (ResNet50 stages 3-4 unfrozen in both models) on the HIP engine against the oracle.

Tolerances: gradients are compared by relative Frobenius error (a pre-activation within fp32 rounding of zero may flip one
ReLU mask entry between two summation orders -- a single such entry moves the max error, not the norm); losses relative
1e-3; the first Adam steps move every weight by ~lr * sign(g), compared where the oracle's move is unambiguous."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def fro_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope="module")
def setup():
    from faster_rcnn.config import Config
    from oracle import dense
    from radnet_hip import synth
    from radnet_hip.engine_cont import ContEngine
    C = Config()
    C.img_size = 300
    P = dense.init_params(seed=3)
    eng = ContEngine(C)
    eng.set_weights(P)
    meta = synth.synthetic_gt(2, n=6, src_w=1000, src_h=600, smin=60, smax=300)
    sample = dict(img=synth.synthetic_panel(1, 300, 500), bboxes=meta["bboxes"], width=1000, height=600)
    return C, P, eng, sample


def test_scatter_strided(setup):
    C, P, eng, _ = setup
    rs = np.random.RandomState(3)
    src = rs.standard_normal((2, 4, 5, 8)).astype(np.float32)
    mask = rs.standard_normal((2, 7, 9, 8)).astype(np.float32)
    ref = np.zeros((2, 7, 9, 8), np.float32)
    ref[:, ::2, ::2][:, :4, :5] = src
    for m in (None, mask):
        dst = torch.full((2, 7, 9, 8), float("nan"), device="cuda")
        md = torch.from_numpy(mask).cuda() if m is not None else None
        eng.ctx.call("radnet_scatter_strided", torch.from_numpy(src).cuda(), 2, 4, 5, 8, 2, 7, 9, md, dst)
        assert np.array_equal(dst.cpu().numpy(), ref * (mask > 0) if m is not None else ref)


def test_rpn_phase_gradients_reach_stages_3_4(setup):
    """RPN loss -> rpn_conv1 dgrad -> dL/dF -> stage 4 -> strided res4a -> stage 3: every conv's weight / bias gradient."""
    from oracle import dense, step as ostep
    C, P, eng, sample = setup
    ot = ostep.OracleTrainerCont(C, copy.deepcopy(P))
    np.random.seed(64)
    y_cls, y_regr = ot.targets(sample)
    x = dense.preprocess_caffe_bgr(sample["img"])
    F, caches = dense.base_forward(P, x, want_cache=True)
    p, r, rc = dense.rpn_forward(P, F)
    _, dp = dense.rpn_loss_cls(y_cls.astype(np.float32), p, ot.A, True)
    _, dr = dense.smooth_l1_masked(y_regr.astype(np.float32), r, 4 * ot.A)
    g_ref, dF = dense.rpn_backward(P, rc, dp, dr, need_dF=True)
    g_ref.update(dense.base_backward(P, caches, dF))

    from radnet_hip.trainer_cont import ContTrainStep
    ts = ContTrainStep(eng)
    H, W = sample["img"].shape[:2]
    tp = eng.anchor_targets_launch(ts._gt(sample), sample["width"], sample["height"], W, H)
    bp = eng.upload_image(sample["img"])
    eng.base_forward(bp)
    rp = eng.rpn_forward(bp)
    np.random.seed(64)
    ycls_d, yregr_d, _ = eng.anchor_targets_finish(tp)
    eng.set_accumulate(rp["bwd"], False, prezeroed=True)
    eng.set_accumulate(bp["bwd34"], False, prezeroed=True)
    eng.rpn_backward(rp, ycls_d, yregr_d)
    eng.s34_backward(bp)
    torch.cuda.synchronize()
    assert fro_err(bp["F"].cpu().numpy(), F) < 1e-4
    assert fro_err(bp["dF"].cpu().numpy().reshape(F.shape), dF * (F > 0)) < 1e-3
    # The gradient of res3a crosses thirty ReLU masks.  Where a pre-activation lands within fp32 rounding of zero the device's mask
    # and the oracle's differ, and WHICH elements do depends on the summation order of the launch shapes in use (round 4: the
    # 32-row tiles moved res3a_branch2a from under 2e-3 to 2.9e-3 against the oracle's own masks).  The strict comparison therefore
    # runs the oracle's backward on the DEVICE's masks (its cached post-ReLU outputs replaced by the device's, which agree to
    # 1e-4); against the oracle's own masks a looser bound stays as a gross-error check.
    caches_dev = copy.deepcopy(caches)
    for co, B in zip(caches_dev[4:], bp["blocks"]):                 # conv1, three stage-2 blocks, then stages 3-4 in order
        for part, buf in (("a", B["a"]), ("b", B["b"]), ("c", B["out"])):
            dev_y = buf.cpu().numpy()
            assert dev_y.shape == co[part]["y"].shape and fro_err(dev_y, co[part]["y"]) < 1e-4
            co[part]["y"] = dev_y
    g_dev = dense.base_backward(P, caches_dev, dF)
    worst = 0.0
    for n in eng.s34_names:
        c = eng.convs[n]
        e = max(fro_err(c.dweight.cpu().numpy(), g_dev[n]["kernel"].reshape(-1, c.cout)), fro_err(c.dbias.cpu().numpy(), g_dev[n]["bias"]))
        worst = max(worst, e)
        assert e < 1e-3, (n, e)
        e_own = max(fro_err(c.dweight.cpu().numpy(), g_ref[n]["kernel"].reshape(-1, c.cout)), fro_err(c.dbias.cpu().numpy(), g_ref[n]["bias"]))
        assert e_own < 1e-2, (n, e_own)
    assert fro_err(eng.convs["rpn_conv1"].dweight.cpu().numpy(), g_ref["rpn_conv1"]["kernel"].reshape(-1, 512)) < 1e-3
    # leave the arenas as a step would: cleared
    eng.adam(eng.rpn_arena)
    eng.adam_s34(0)
    eng.set_weights(P)
    for ar in (eng.rpn_arena, eng.head_arena, eng.s34_arena):
        ar.m.zero_(); ar.v.zero_(); ar.t = 0
    eng.s34_arena.m2.zero_(); eng.s34_arena.v2.zero_(); eng.s34_arena.t2 = 0


def test_cont_train_step_vs_oracle(setup):
    from oracle import step as ostep
    from radnet_hip.trainer_cont import ContTrainStep
    C, P, eng, sample = setup
    np.random.seed(64)
    ts = ContTrainStep(eng)
    ts.capture = []
    ts.step([sample])
    got = ts.losses()
    w_after = eng.get_weights()
    rng_gpu = np.random.randint(0, 2 ** 31 - 1)
    np.random.seed(64)
    ot = ostep.OracleTrainerCont(C, copy.deepcopy(P))
    detail = {}
    ref = ot.step(sample, detail, override_R=ts.capture[0]["R"])
    assert np.random.randint(0, 2 ** 31 - 1) == rng_gpu
    assert ts.capture[0]["sel_kept"] == detail["sel"]
    assert got["n_head"] == 1
    assert abs(got["rpn_cls"] - ref[0]) < 1e-3 * abs(ref[0])
    assert abs(got["rpn_regr"] - ref[1]) < 1e-3 * abs(ref[1]) + 1e-6
    assert abs(got["det_cls"] - ref[2]) < 2e-3 * abs(ref[2])
    assert abs(got["det_regr"] - ref[3]) < 2e-3 * abs(ref[3]) + 1e-5
    lr = 2e-5
    for name in ("rpn_conv1", "res5a_branch2a", "res5c_branch2c", "dense_regress_7"):
        for k in ("kernel", "bias"):
            d_ref, d_gpu = ot.P[name][k] - P[name][k], w_after[name][k] - P[name][k]
            big = np.abs(d_ref) > 0.9 * lr
            assert big.sum() > 0 and np.abs(d_gpu[big] - d_ref[big]).max() < 0.05 * lr, (name, k)
    # shared stage-3/4 weights: two first Adam steps (one per optimizer, own moments): |delta| ~ 2 lr where both agree in sign
    for name in ("res3a_branch2a", "res3d_branch2b", "res4a_branch1", "res4a_branch2a", "res4f_branch2c"):
        for k in ("kernel", "bias"):
            d_ref, d_gpu = ot.P[name][k] - P[name][k], w_after[name][k] - P[name][k]
            big = np.abs(d_ref) > 1.8 * lr
            assert big.sum() > 0, (name, k)
            assert np.mean(np.abs(d_gpu[big] - d_ref[big]) < 0.1 * lr) > 0.995, (name, k)
            assert np.abs(d_gpu).max() <= 2 * lr * 1.0001
    # frozen part untouched
    assert "res2c_branch2c" not in w_after
