"""BASELINE config 5: VGG16 base_model swap, 3 anchor scales x 3 ratios, variable-N RoIs per image.
Device path (radnet_hip.engine_vgg) vs the oracle (oracle/vgg.py, parity unpinned) on identical seeded inputs;
Dropout masks are injected so both sides see the same ones.  fp32 tolerances as in test_gpu_engine.py."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


from tolerances import check  # noqa: E402  (max-norm + per-channel + RMS criteria, tests/tolerances.py)


def rel_err(a, b):
    b = np.asarray(b, dtype=np.float64)
    return np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def setup():
    from faster_rcnn.config import Config
    from oracle import vgg
    from radnet_hip import make_engine
    C = Config()
    C.network = "vgg16"
    C.anchor_box_scales = [128, 256, 512]          # config.py:46: the original 3 scales -> A = 9
    P = vgg.init_params(seed=5, n_anchors=9)
    eng = make_engine(C)
    eng.set_weights(P)
    return C, P, eng


def test_vgg_generators_agree():
    from oracle import vgg
    from radnet_hip import synth
    a, b = vgg.init_params(seed=5), synth.synthetic_weights_vgg16(seed=5)
    assert sorted(a) == sorted(b)
    for n in a:
        for k in a[n]:
            assert np.array_equal(a[n][k], b[n][k]), (n, k)


def test_vgg_rpn_forward_and_proposals(setup):
    from oracle import dense, glue, vgg
    C, P, eng = setup
    assert eng.A == 9 and eng.feat_len(1000) == 62 and eng.feat_len(600) == 37
    img = np.random.RandomState(0).randint(0, 256, (200, 328, 3)).astype(np.uint8)
    F = vgg.base_forward(P, dense.preprocess_caffe_bgr(img))
    p, r, _ = dense.rpn_forward(P, F)
    bp = eng.upload_image(img)
    Fg = eng.base_forward(bp)
    rp = eng.rpn_forward(bp)
    pred = rp["pred"].cpu().numpy()
    assert tuple(Fg.shape) == F.shape == (1, 12, 20, 512)
    assert check(Fg.cpu().numpy(), F, 1e-3) < 1e-3
    assert check(pred[:, :9], p.reshape(-1, 9), 1e-3) < 1e-3 and check(pred[:, 9:45], r.reshape(-1, 36), 1e-3) < 1e-3
    R, Rn = eng.proposals(rp, 0.7, 300)
    n = int(Rn.cpu()[0])
    Rref = glue.rpn_to_roi(pred[:, :9].reshape(1, 12, 20, 9), pred[:, 9:45].reshape(1, 12, 20, 36), C, True, 300, 0.7)
    assert n == len(Rref) and np.array_equal(R.cpu().numpy()[:n], Rref)


@pytest.mark.parametrize("R", [5, 37, 120, 300])
def test_vgg_head_variable_rois(setup, R):
    """Variable-N RoIs (5 / 37 / 120 / 300): forward in inference mode and, up to 120 RoIs (fc1's weight gradient is a
    [25088 x 4096] matrix summed over R rows), the full backward with injected dropout masks."""
    from oracle import vgg
    C, P, eng = setup
    rs = np.random.RandomState(R)
    F = (np.maximum(rs.standard_normal((1, 37, 62, 512)), 0) * 2).astype(np.float32)
    rois = np.stack([rs.randint(0, 50, R), rs.randint(0, 28, R), rs.randint(1, 12, R), rs.randint(1, 9, R)], 1).astype(np.float32)
    Fd = torch.from_numpy(F).cuda()
    hp = eng._plan_head(R, 37, 62, Fd)
    hp["rois"].copy_(torch.from_numpy(rois))
    eng.head_forward(hp, training=False)
    pc, pr, cache = vgg.head_forward(P, F, rois, 7, None)
    assert check(hp["h1"].cpu().numpy(), cache["h1"], 1e-3) < 1e-3
    assert check(hp["pcls"].cpu().numpy(), pc[0], 2e-3) < 2e-3 and check(hp["pregr"].cpu().numpy(), pr[0], 2e-3) < 2e-3
    if R > 120:
        return
    cls = rs.randint(0, 7, R)
    Y1 = np.eye(7, dtype=np.float32)[cls][None]
    lab = np.zeros((R, 24), np.float32)
    for i, c in enumerate(cls):
        if c != 6:
            lab[i, 4 * c:4 * c + 4] = 1
    Y2 = np.concatenate([lab, rs.standard_normal((R, 24)).astype(np.float32) * lab], -1)[None]
    m1 = (rs.uniform(size=(R, 4096)) >= 0.5).astype(np.float32) * 2
    m2 = (rs.uniform(size=(R, 4096)) >= 0.5).astype(np.float32) * 2
    losses, grads = vgg.head_losses_and_grads(P, F, rois, Y1, Y2, 7, (m1, m2))
    hp["y1"].copy_(torch.from_numpy(Y1[0])); hp["y2"].copy_(torch.from_numpy(Y2[0]))
    eng.forced_masks = (m1, m2)
    eng.head_forward(hp, training=True)
    eng.set_accumulate(hp["bwd"], False)
    eng.head_backward(hp, accumulate=False)
    eng.forced_masks = None
    got = eng.det_losses.cpu().numpy()
    assert abs(got[0] - losses[1]) < 2e-3 * abs(losses[1]) and abs(got[1] - losses[2]) < 2e-3 * abs(losses[2]) + 1e-6
    for name in ("fc1", "fc2"):
        c = eng.convs[name]
        assert check(c.dweight.cpu().numpy(), grads[name]["kernel"], 3e-3) < 3e-3, name
        assert check(c.dbias.cpu().numpy(), grads[name]["bias"], 3e-3) < 3e-3, name
    dk = eng.dense_dw.cpu().numpy()
    assert check(dk[:, :7], grads["dense_class_7"]["kernel"], 3e-3) < 3e-3
    assert check(dk[:, 7:31], grads["dense_regress_7"]["kernel"], 3e-3) < 3e-3


def test_vgg_train_step_vs_oracle(setup):
    """Whole iteration (train.py:288-402 order) on the VGG16 engine against oracle.step.OracleTrainerVGG, two steps on two
    samples: anchor-target RNG consumption, RPN losses, proposals bit-exact on the device's own tensors, RoI labelling and
    sampling, detector losses and accuracy with the SAME dropout masks on both sides, and the first-step weight moves of
    rpn_conv1 (C = 512) / rpn_out_* / fc1 / fc2 / dense heads.  (Replaces the round-2 'runs and learns' property test.)"""
    from oracle import glue, step as ostep
    from radnet_hip import synth
    from radnet_hip.trainer import TrainStep
    C, P, eng = setup
    P0 = copy.deepcopy(P)
    eng.set_weights(copy.deepcopy(P0))
    for arena in (eng.rpn_arena, eng.head_arena):
        arena.g.zero_(); arena.m.zero_(); arena.v.zero_(); arena.t = 0
    old_size = C.img_size
    C.img_size = 300
    try:
        samples = []
        for i in range(2):
            meta = synth.synthetic_gt(2 + i, n=6, src_w=1000, src_h=600, smin=60, smax=300)
            samples.append(dict(img=synth.synthetic_panel(1 + i, 300, 500), bboxes=meta["bboxes"], width=1000, height=600))
        mask_rs = np.random.RandomState(99)
        masks = [((mask_rs.uniform(size=(C.n_rois, 4096)) >= 0.5).astype(np.float32) * 2, (mask_rs.uniform(size=(C.n_rois, 4096)) >= 0.5).astype(np.float32) * 2)
                 for _ in samples]
        np.random.seed(64)
        ts = TrainStep(eng)
        ts.capture = []
        got = []
        w_after = None
        for s_, m in zip(samples, masks):
            eng.forced_masks = m
            got.append(ts.step([s_]).losses())
            if w_after is None:
                w_after = eng.get_weights()                 # after the FIRST Adam step of each optimizer
        eng.forced_masks = None
        rng_gpu = int(np.random.randint(0, 2 ** 31 - 1))
        np.random.seed(64)
        ot = ostep.OracleTrainerVGG(C, copy.deepcopy(P0))
        fh, fw = 18, 31
        for k, (s_, m) in enumerate(zip(samples, masks)):
            cap = ts.capture[k]
            pred = cap["pred"]
            Rref = glue.rpn_to_roi(pred[:, :9].reshape(1, fh, fw, 9), pred[:, 9:45].reshape(1, fh, fw, 36), C, True, 300, 0.7)
            assert np.array_equal(cap["R"], Rref)                          # proposals: bit-exact on the device's own outputs
            detail = {}
            ref = ot.step(s_, detail, override_R=cap["R"], masks_fn=lambda R, m=m: m)
            assert ref[2] is not None and got[k]["n_head"] == 1
            assert np.array_equal(cap["cls"][cap["keep"]], detail["Y1"][0].argmax(-1)) and cap["sel_kept"] == detail["sel"]
            assert abs(got[k]["rpn_cls"] - ref[0]) < 1e-3 * abs(ref[0]) and abs(got[k]["rpn_regr"] - ref[1]) < 1e-3 * abs(ref[1]) + 1e-6
            assert abs(got[k]["det_cls"] - ref[2]) < 3e-3 * abs(ref[2]) and abs(got[k]["det_regr"] - ref[3]) < 3e-3 * abs(ref[3]) + 1e-5
            assert abs(got[k]["det_acc"] - ref[4]) < 1e-6
            if k == 0:
                g_first = (detail["g_rpn"], detail["g_head"])
                P_first = copy.deepcopy(ot.P)
        assert rng_gpu == int(np.random.randint(0, 2 ** 31 - 1))
        # weights after the first Adam step of each optimizer: |delta| <= lr; equal to the oracle's where the gradient is not
        # tiny (Adam's g / (|g| + eps) amplifies rounding noise of near-zero gradients)
        for name in ("rpn_conv1", "rpn_out_class", "rpn_out_regress", "fc1", "fc2", "dense_class_7", "dense_regress_7"):
            for kk in ("kernel", "bias"):
                d_ref, d_gpu = P_first[name][kk] - P0[name][kk], w_after[name][kk] - P0[name][kk]
                g = (g_first[0] if name.startswith("rpn") else g_first[1])[name][kk]
                big = np.abs(g) > 1e-3 * np.abs(g).max()
                assert big.sum() > 0 and np.abs(d_gpu[big] - d_ref[big]).max() < 0.05 * 5e-5, (name, kk, float(np.abs(d_gpu[big] - d_ref[big]).max()))
                assert np.abs(d_gpu).max() <= 5e-5 * 1.0001
        # validation pass (train.py:478-561) on the twice-updated weights: Dropout off (Keras test phase), forward only
        ts.capture = []
        np.random.seed(7)
        rec = ts.validate([samples[1]])
        rng_v = int(np.random.randint(0, 2 ** 31 - 1))
        np.random.seed(7)
        vref = ostep.oracle_validate(ot, [samples[1]], override_R=[ts.capture[0]["R"]])
        assert rng_v == int(np.random.randint(0, 2 ** 31 - 1))
        assert rec["n"] == 1 and len(vref) == 1
        g_, r_ = rec["per_sample"][0], vref[0]
        assert abs(g_["rpn_cls"] - r_[0]) < 1e-3 * abs(r_[0]) and abs(g_["rpn_regr"] - r_[1]) < 1e-3 * abs(r_[1]) + 1e-6
        assert abs(g_["det_cls"] - r_[2]) < 3e-3 * abs(r_[2]) and abs(g_["det_regr"] - r_[3]) < 3e-3 * abs(r_[3]) + 1e-5
        assert abs(g_["det_acc"] - r_[4]) < 1e-6 and g_["n_pos"] == r_[5]
    finally:
        C.img_size = old_size
        eng.forced_masks = None
        eng.set_weights(copy.deepcopy(P0))
