"""BASELINE.json configurations at their FULL sizes against the oracle (the NumPy restatement runs a 1000x600 iteration in
a few seconds on the GPU box's host cores), stage by stage on identical tensors:

  cfg 2  one full train step at 1000x600 / 12 anchors / 20 RoIs: anchor targets, base + RPN activations, RPN losses, the
         DIRECT gradients of rpn_conv1 (Winograd weight gradient inside the engine) / rpn_out_class / rpn_out_regress,
         proposals bit-exact on the device's own tensors, RoI labels, selected samples, RNG consumption, detector losses,
         direct gradients of every stage-5 conv and both dense heads
  cfg 3  predict path on the 2048x2048 tile at img_size = 600 (38x38 map): device resize bit-exact vs oracle/resize.py,
         RPN activations, proposals bit-exact, the 300-RoI single head pass (M = 14 700) against the oracle's 15 chunks
  cfg 5  VGG16 base + RPN at 1000x600 (37x62 map, 9 anchors), proposals bit-exact on the device's tensors

Tolerances as tests/test_gpu_engine.py: activations 1e-3 * max|ref|, gradients 2e-3 * max|ref|, losses 1e-3 relative
(fp32 MFMA accumulation / Winograd re-association against the oracle's BLAS); integer and index outputs bit-exact.
The dense half of the oracle is parity-unpinned against TensorFlow itself (oracle/dense.py header); the glue half is pinned
by the reference's own outputs."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


from tolerances import check, measure  # noqa: E402  (max-norm + per-channel + RMS criteria, tests/tolerances.py)


def rel_err(a, b):
    b = np.asarray(b, dtype=np.float64)
    return np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-30)


def cfg2_sample():
    """BASELINE.md 3 / SURVEY.md 8d: panel (600,1000,3) seed 1; 8 boxes seed 2, sizes U[64,400] in a 2000x1200 frame."""
    from radnet_hip import synth
    meta = synth.synthetic_gt(2, n=8, src_w=2000, src_h=1200)
    return dict(img=synth.synthetic_panel(1, 600, 1000), bboxes=meta["bboxes"], width=2000, height=1200)


def test_cfg2_full_step_1000x600_vs_oracle():
    from faster_rcnn.config import Config
    from oracle import dense, glue, step as ostep
    from radnet_hip.engine import FasterRCNNEngine
    from radnet_hip.trainer import TrainStep
    C = Config()
    assert C.img_size == 600 and C.n_rois == 20 and len(C.anchor_box_scales) * len(C.anchor_box_ratios) == 12
    P = dense.init_params(seed=3)
    eng = FasterRCNNEngine(C)
    eng.set_weights(P)
    sample = cfg2_sample()
    fh, fw = 38, 63

    # ---------------- the product's step (what bench.py times), with per-stage capture
    np.random.seed(64)
    ts = TrainStep(eng)
    ts.capture = []
    ts.step([sample])
    got = ts.losses()
    rng_after_gpu = int(np.random.randint(0, 2 ** 31 - 1))
    cap = ts.capture[0]
    w_after = eng.get_weights()

    # proposals: bit-exact vs the oracle run on the device's own post-update RPN outputs
    pred = cap["pred"]
    assert pred.shape == (fh * fw, 64)
    Rref = glue.rpn_to_roi(pred[:, :12].reshape(1, fh, fw, 12), pred[:, 12:60].reshape(1, fh, fw, 48), C, True, 300, 0.7)
    assert cap["R"].shape == Rref.shape and np.array_equal(cap["R"], Rref)

    # ---------------- the oracle's iteration on the same sample, same RNG stream, labelling the device's proposals
    np.random.seed(64)
    ot = ostep.OracleTrainer(C, copy.deepcopy(P))
    detail = {}
    ref = ot.step(sample, detail, override_R=cap["R"])
    rng_after_ref = int(np.random.randint(0, 2 ** 31 - 1))
    assert rng_after_gpu == rng_after_ref
    assert ref[2] is not None and got["n_head"] == 1
    # the oracle's own proposals (from its own fp32 scores) agree except where ~1e-7 score noise reorders near-ties
    own = detail["R_own"]
    assert (own[:, None, :] == cap["R"][None, :, :]).all(-1).any(1).mean() > 0.9
    # post-update RPN outputs (Adam #1 applied on both sides)
    assert check(pred[:, :12], detail["p"].reshape(-1, 12), 1e-3) < 1e-3
    assert check(pred[:, 12:60], detail["r"].reshape(-1, 48), 1e-3) < 1e-3
    # RoI labelling and sampling on identical proposals: exact
    assert cap["keep"].sum() == detail["X2"].shape[1]
    assert np.array_equal(cap["cls"][cap["keep"]], detail["Y1"][0].argmax(-1))
    assert cap["sel_kept"] == detail["sel"]
    # losses
    assert abs(got["rpn_cls"] - ref[0]) < 1e-3 * abs(ref[0])
    assert abs(got["rpn_regr"] - ref[1]) < 1e-3 * abs(ref[1]) + 1e-6
    assert abs(got["det_cls"] - ref[2]) < 2e-3 * abs(ref[2])
    assert abs(got["det_regr"] - ref[3]) < 2e-3 * abs(ref[3]) + 1e-5
    assert abs(got["det_acc"] - ref[4]) < 1e-6
    # weights after one Adam step of each optimizer: |delta| <= lr, and equal to the oracle's where its gradient is not tiny
    for name in ("rpn_conv1", "rpn_out_class", "rpn_out_regress", "res5a_branch2a", "res5b_branch2b", "res5c_branch2c", "dense_class_7", "dense_regress_7"):
        for k in ("kernel", "bias"):
            d_ref, d_gpu = ot.P[name][k] - P[name][k], w_after[name][k] - P[name][k]
            g = (detail["g_rpn"] if name.startswith("rpn") else detail["g_head"])[name][k]
            big = np.abs(g) > 1e-3 * np.abs(g).max()
            assert big.sum() > 0 and np.abs(d_gpu[big] - d_ref[big]).max() < 0.05 * 5e-5, (name, k)
            assert np.abs(d_gpu).max() <= 5e-5 * 1.0001

    # ---------------- direct gradients: the same phases enqueued by hand, gradient arenas read BEFORE Adam consumes them
    eng.set_weights(P)
    for arena in (eng.rpn_arena, eng.head_arena):
        arena.g.zero_(); arena.m.zero_(); arena.v.zero_(); arena.t = 0
    gt = ts._gt(sample)
    np.random.seed(64)
    tp = eng.anchor_targets_launch(gt, sample["width"], sample["height"], 1000, 600, slot=90)
    bp = eng.upload_image(sample["img"], slot=90)
    Fg = eng.base_forward(bp)
    rp = eng.rpn_forward(bp)
    ycls, yregr, n_pos = eng.anchor_targets_finish(tp)
    # anchor targets at full size vs the oracle (labels exact; regression targets exact after the fp32 cast the net applies)
    np.random.seed(64)
    yc_ref, yr_ref = ot.targets(sample)
    assert np.array_equal(ycls.cpu().numpy().reshape(yc_ref.shape), yc_ref.astype(np.float32))
    assert np.array_equal(yregr.cpu().numpy().reshape(yr_ref.shape), yr_ref.astype(np.float32))
    # activations
    F_ref = detail["F"]
    assert tuple(Fg.shape) == F_ref.shape == (1, fh, fw, 1024)
    assert check(Fg.cpu().numpy(), F_ref, 1e-3) < 1e-3
    P0 = copy.deepcopy(P)
    p0, r0, _ = dense.rpn_forward(P0, F_ref)
    pred0 = rp["pred"].cpu().numpy()
    assert check(pred0[:, :12], p0.reshape(-1, 12), 1e-3) < 1e-3 and check(pred0[:, 12:60], r0.reshape(-1, 48), 1e-3) < 1e-3
    # RPN backward: gradients straight out of the arena
    eng.set_accumulate(rp["bwd"], False, prezeroed=True)
    eng.rpn_backward(rp, ycls, yregr)
    torch.cuda.synchronize()
    l_rpn = eng.rpn_losses.cpu().numpy()
    assert abs(l_rpn[0] - ref[0]) < 1e-3 * abs(ref[0]) and abs(l_rpn[1] - ref[1]) < 1e-3 * abs(ref[1]) + 1e-6
    g = detail["g_rpn"]
    c1, ch = eng.convs["rpn_conv1"], eng.convs["rpn_heads"]
    assert check(c1.dweight.cpu().numpy(), g["rpn_conv1"]["kernel"].reshape(-1, 512), 2e-3) < 2e-3          # Winograd-domain wgrad
    assert check(c1.dbias.cpu().numpy()[:512], g["rpn_conv1"]["bias"], 2e-3) < 2e-3
    dwh, dbh = ch.dweight.cpu().numpy(), ch.dbias.cpu().numpy()
    assert check(dwh[:, :12], g["rpn_out_class"]["kernel"].reshape(512, 12), 2e-3) < 2e-3
    assert check(dwh[:, 12:60], g["rpn_out_regress"]["kernel"].reshape(512, 48), 2e-3) < 2e-3
    assert check(dbh[:12], g["rpn_out_class"]["bias"], 2e-3) < 2e-3 and check(dbh[12:60], g["rpn_out_regress"]["bias"], 2e-3) < 2e-3
    assert np.all(dwh[:, 60:] == 0) and np.all(dbh[60:] == 0)
    # classifier phase on the oracle's sampled RoIs / targets (identical to the device's, asserted above): direct gradients
    sel = detail["sel"]
    rois = detail["X2"][0, sel].astype(np.float32)
    Y1, Y2 = detail["Y1"][0, sel].astype(np.float32), detail["Y2"][0, sel].astype(np.float32)
    hp = eng._plan_head(C.n_rois, fh, fw, bp["F"])
    hp["rois"].copy_(torch.from_numpy(rois)); hp["y1"].copy_(torch.from_numpy(Y1)); hp["y2"].copy_(torch.from_numpy(Y2))
    eng.head_forward(hp, training=True)
    eng.set_accumulate(hp["bwd"], False, prezeroed=True)
    eng.head_backward(hp, accumulate=False)
    torch.cuda.synchronize()
    gh = detail["g_head"]
    l_det = eng.det_losses.cpu().numpy()
    assert abs(l_det[0] - ref[2]) < 2e-3 * abs(ref[2]) and abs(l_det[1] - ref[3]) < 2e-3 * abs(ref[3]) + 1e-5
    # The stage-5 gradients cross up to eight ReLU masks; where an activation lands within rounding of zero the device's mask
    # and the oracle's differ in single elements, which moves isolated gradient entries by whole terms (~1e-4 of the largest
    # gradient) -- not rounding noise of one sum.  So the oracle's backward is evaluated ON THE DEVICE'S MASKS (its cached
    # post-ReLU outputs replaced by the device's, the way proposals are compared on the device's own tensors): what is left is
    # summation order, and the per-channel floor is the checker's default 1e-3 again (round 3 had raised it to 1e-2 here).
    pc_o, pr_o, cache_o = dense.head_forward(P, F_ref, rois, 7)
    for co, B in zip(cache_o["blocks"], hp["blocks"]):
        for part, buf in (("a", B["a"]), ("b", B["b"]), ("c", B["out"])):
            dev_y = buf.cpu().numpy()
            assert dev_y.shape == co[part]["y"].shape
            assert check(dev_y, co[part]["y"], 1e-3) < 1e-3
            co[part]["y"] = dev_y
    _, dpc_o = dense.class_loss_cls(Y1[None], pc_o)
    _, dpr_o = dense.smooth_l1_masked(Y2[None], pr_o, 4 * 6)
    gh_dev, _ = dense.head_backward(P, cache_o, dpc_o, dpr_o)
    for name in eng.head_conv_names:
        c = eng.convs[name]
        assert check(c.dweight.cpu().numpy(), gh_dev[name]["kernel"].reshape(-1, c.cout), 2e-3) < 2e-3, name
        assert check(c.dbias.cpu().numpy(), gh_dev[name]["bias"], 2e-3) < 2e-3, name
        # against the oracle's OWN masks: max-norm and RMS only -- the per-channel criterion is the one that single flipped mask
        # elements move (round 3 had loosened its floor to 1e-2 here; round 4's launch shapes put a channel at 3.4e-3 of a 2e-3
        # bound even so, exactly the fragility the device-mask comparison above removes)
        mx, _, rms = measure(c.dweight.cpu().numpy(), gh[name]["kernel"].reshape(-1, c.cout))
        assert mx < 2e-3 and rms < 0.1 * 2e-3, (name, mx, rms)
        assert check(c.dbias.cpu().numpy(), gh[name]["bias"], 2e-3) < 2e-3, name
    dk, db = eng.dense_dw.cpu().numpy(), eng.dense_db.cpu().numpy()
    assert check(dk[:, :7], gh["dense_class_7"]["kernel"], 2e-3) < 2e-3 and check(dk[:, 7:31], gh["dense_regress_7"]["kernel"], 2e-3) < 2e-3
    assert check(db[:7], gh["dense_class_7"]["bias"], 2e-3) < 2e-3 and check(db[7:31], gh["dense_regress_7"]["bias"], 2e-3) < 2e-3


def test_cfg3_predict_tile_2048_at_img_size_600_vs_oracle():
    from faster_rcnn import models as M
    from faster_rcnn import rpn
    from faster_rcnn.RADNet import RADNet, resize_cubic
    from faster_rcnn.base_models import resnet50
    from faster_rcnn.config import Config
    from oracle import dense, glue, resize as oresize, step as ostep
    C = Config()
    assert C.img_size == 600
    P = dense.init_params(seed=3)
    m_rpn, m_cls, m_all, m_rpn3, m_det = M.build_models(C, weights=copy.deepcopy(P))
    tile = np.random.RandomState(4).randint(0, 256, (2048, 2048, 3)).astype(np.uint8)
    net = RADNet(C, m_rpn3, m_det, resnet50.preprocess)
    # G9: device bicubic resize, bit-exact against the oracle's restatement of the 8-bit INTER_CUBIC definition
    small = resize_cubic(tile, 600, 600)
    assert np.array_equal(small, oresize.resize_bicubic_u8(tile, 600, 600))
    X, ratio = net.format_img(tile)
    assert X.shape == (1, 600, 600, 3) and abs(ratio - 600 / 2048) < 1e-12
    assert np.allclose(X, dense.preprocess_caffe_bgr(small), atol=1e-4)
    # RPN at 600x600 (38x38 map)
    Y1, Y2, F = m_rpn3.predict(X)
    p, r, F_ref = ostep.rpn_only_forward(P, small)
    assert F.shape == F_ref.shape == (1, 38, 38, 1024)
    assert check(F, F_ref, 1e-3) < 1e-3 and check(Y1, p, 1e-3) < 1e-3 and check(Y2, r, 1e-3) < 1e-3
    # proposals on the device's own tensors: bit-exact
    R = rpn.rpn_to_roi(Y1, Y2, C, overlap_thresh=0.7)
    assert np.array_equal(R, glue.rpn_to_roi(Y1, Y2, C, True, 300, 0.7))
    assert len(R) == 300
    Rx = R.copy()
    Rx[:, 2] -= Rx[:, 0]; Rx[:, 3] -= Rx[:, 1]
    # ALL 300 RoIs in one head pass (GEMM M = 300 * 49 = 14 700) vs the oracle walking 15 chunks of 20 on the same feature map
    pc, pr = m_det.predict([F, Rx[None]])
    assert pc.shape == (1, 300, 7) and pr.shape == (1, 300, 24)
    for k in range(0, 300, C.n_rois):
        rc, rr, _ = dense.head_forward(P, F, Rx[k:k + C.n_rois].astype(np.float32), 7)
        assert np.abs(pc[0, k:k + C.n_rois] - rc[0]).max() < 2e-3, k
        assert np.abs(pr[0, k:k + C.n_rois] - rr[0]).max() < 2e-3 * max(1.0, np.abs(rr).max()), k
    # decode: the facade's one-pass path == the oracle's chunked decode fed with the device's outputs
    bb, pp = net.apply_spatial_pyramid_pooling(Rx, F)
    bb_ref, pp_ref = glue.spp_decode(Rx, lambda rois: m_det.predict([F, rois]), C)
    assert sorted(bb) == sorted(bb_ref)
    for k in bb:
        assert np.array_equal(np.array(bb[k]), np.array(bb_ref[k]))
        assert np.allclose(np.array(pp[k]), np.array(pp_ref[k]), rtol=0, atol=1e-5)


def test_cfg5_vgg16_base_and_rpn_1000x600_vs_oracle():
    from faster_rcnn.config import Config
    from oracle import dense, glue, vgg
    from radnet_hip import make_engine, synth
    C = Config()
    C.network = "vgg16"
    C.anchor_box_scales = [128, 256, 512]
    P = vgg.init_params(seed=5, n_anchors=9)
    eng = make_engine(C)
    eng.set_weights(P)
    img = synth.synthetic_panel(1, 600, 1000)
    F = vgg.base_forward(P, dense.preprocess_caffe_bgr(img))
    p, r, _ = dense.rpn_forward(P, F)
    bp = eng.upload_image(img)
    Fg = eng.base_forward(bp)
    rp = eng.rpn_forward(bp)
    pred = rp["pred"].cpu().numpy()
    assert tuple(Fg.shape) == F.shape == (1, 37, 62, 512)
    assert check(Fg.cpu().numpy(), F, 1e-3) < 1e-3
    assert check(pred[:, :9], p.reshape(-1, 9), 1e-3) < 1e-3 and check(pred[:, 9:45], r.reshape(-1, 36), 1e-3) < 1e-3
    R, Rn = eng.proposals(rp, 0.7, 300)
    n = int(Rn.cpu()[0])
    Rref = glue.rpn_to_roi(pred[:, :9].reshape(1, 37, 62, 9), pred[:, 9:45].reshape(1, 37, 62, 36), C, True, 300, 0.7)
    assert n == len(Rref) and np.array_equal(R.cpu().numpy()[:n], Rref)
    # RPN backward at full size: rpn_conv1 (3x3, C = 512 -> 512) and the two 1x1 heads, gradients straight out of the arena
    from oracle import step as ostep
    from radnet_hip.trainer import TrainStep
    meta = synth.synthetic_gt(2, n=8, src_w=2000, src_h=1200)
    sample = dict(img=img, bboxes=meta["bboxes"], width=2000, height=1200)
    ot = ostep.OracleTrainerVGG(C, copy.deepcopy(P))
    np.random.seed(64)
    yc_ref, yr_ref = ot.targets(sample)
    ts = TrainStep(eng)
    np.random.seed(64)
    tp = eng.anchor_targets_launch(ts._gt(sample), sample["width"], sample["height"], 1000, 600, slot=91)
    ycls, yregr, n_pos = eng.anchor_targets_finish(tp)
    assert np.array_equal(ycls.cpu().numpy().reshape(yc_ref.shape), yc_ref.astype(np.float32))          # 9 anchors, 37x62 map
    assert np.array_equal(yregr.cpu().numpy().reshape(yr_ref.shape), yr_ref.astype(np.float32))
    l_ref, g = dense.rpn_losses_and_grads(P, F, yc_ref.astype(np.float32), yr_ref.astype(np.float32), 9, True)
    for arena in (eng.rpn_arena,):
        arena.g.zero_()
    eng.set_accumulate(rp["bwd"], False, prezeroed=True)
    eng.rpn_backward(rp, ycls, yregr)
    torch.cuda.synchronize()
    l_rpn = eng.rpn_losses.cpu().numpy()
    assert abs(l_rpn[0] - l_ref[1]) < 1e-3 * abs(l_ref[1]) and abs(l_rpn[1] - l_ref[2]) < 1e-3 * abs(l_ref[2]) + 1e-6
    c1, ch = eng.convs["rpn_conv1"], eng.convs["rpn_heads"]
    assert check(c1.dweight.cpu().numpy(), g["rpn_conv1"]["kernel"].reshape(-1, 512), 2e-3) < 2e-3
    assert check(c1.dbias.cpu().numpy()[:512], g["rpn_conv1"]["bias"], 2e-3) < 2e-3
    dwh, dbh = ch.dweight.cpu().numpy(), ch.dbias.cpu().numpy()
    assert check(dwh[:, :9], g["rpn_out_class"]["kernel"].reshape(512, 9), 2e-3) < 2e-3
    assert check(dwh[:, 9:45], g["rpn_out_regress"]["kernel"].reshape(512, 36), 2e-3) < 2e-3
    assert check(dbh[:9], g["rpn_out_class"]["bias"], 2e-3) < 2e-3 and check(dbh[9:45], g["rpn_out_regress"]["bias"], 2e-3) < 2e-3
    assert np.all(dwh[:, 45:] == 0) and np.all(dbh[45:] == 0)
    eng.rpn_arena.g.zero_()


def test_cfg4_per_gpu_batch_2_at_1000x600_one_program_vs_oracle():
    """BASELINE cfg 4 on one GPU at full size: two 1000x600 panels per step as ONE layer program (every base / RPN GEMM with
    M doubled, 40 RoIs through stage 5 in one pass) against the oracle's step_batch: per-image proposals bit-exact on the
    device's tensors, sample selection and RNG consumption exact, mean losses, first Adam step of both optimizers."""
    from faster_rcnn.config import Config
    from oracle import dense, glue, step as ostep
    from radnet_hip import synth
    from radnet_hip.engine import FasterRCNNEngine
    from radnet_hip.trainer import TrainStep
    C = Config()
    P = dense.init_params(seed=3)
    eng = FasterRCNNEngine(C)
    eng.set_weights(P)
    batch = []
    for i in range(2):                      # SURVEY.md 8d cfg 4: seeds 100 + rank*2 + i (rank 0)
        meta = synth.synthetic_gt(100 + i, n=8, src_w=2000, src_h=1200)
        batch.append(dict(img=synth.synthetic_panel(100 + i, 600, 1000), bboxes=meta["bboxes"], width=2000, height=1200))
    np.random.seed(64)
    ts = TrainStep(eng)
    assert ts.batched
    ts.capture = []
    ts.step(batch)
    got = ts.losses()
    rng_gpu = int(np.random.randint(0, 2 ** 31 - 1))
    w_after = eng.get_weights()
    assert len(ts.capture) == got["n_head"] and got["n_head"] >= 1
    for c in ts.capture:
        Rref = glue.rpn_to_roi(c["pred"][:, :12].reshape(1, 38, 63, 12), c["pred"][:, 12:60].reshape(1, 38, 63, 48), C, True, 300, 0.7)
        assert np.array_equal(c["R"], Rref)
    if got["n_head"] != 2:
        pytest.skip("one of the two synthetic panels kept no RoI: the per-image comparison below needs both")
    np.random.seed(64)
    ot = ostep.OracleTrainer(C, copy.deepcopy(P))
    det = []
    ref = ostep.step_batch(ot, batch, details=det, override_R=[c["R"] for c in ts.capture])
    assert int(np.random.randint(0, 2 ** 31 - 1)) == rng_gpu
    for c, d in zip(ts.capture, det):
        assert c["sel_kept"] == d["sel"]
    r = np.array(ref, dtype=np.float64)
    assert abs(got["rpn_cls"] - r[:, 0].mean()) < 1e-3 * abs(r[:, 0].mean())
    assert abs(got["rpn_regr"] - r[:, 1].mean()) < 1e-3 * abs(r[:, 1].mean()) + 1e-6
    assert abs(got["det_cls"] - r[:, 2].mean()) < 2e-3 * abs(r[:, 2].mean())
    assert abs(got["det_regr"] - r[:, 3].mean()) < 2e-3 * abs(r[:, 3].mean()) + 1e-5
    for name in ("rpn_conv1", "rpn_out_class", "rpn_out_regress", "res5a_branch2a", "res5b_branch2b", "res5c_branch2c", "dense_regress_7"):
        for k in ("kernel", "bias"):
            d_ref, d_gpu = ot.P[name][k] - P[name][k], w_after[name][k] - P[name][k]
            big = np.abs(d_ref) > 0.9 * 5e-5                               # first Adam step: |delta| ~ lr where the gradient is not tiny
            assert big.sum() > 0 and np.abs(d_gpu[big] - d_ref[big]).max() < 0.05 * 5e-5, (name, k)


def test_cfg3_predict_tile_2048_at_img_size_1000_rpn_and_proposals_vs_oracle():
    """BASELINE.md 3 also names img_size = 1000 for cfg 3 (1000x1000 network input, 63x63 map, 47 628 anchors): device resize
    bit-exact, RPN activations against the oracle, proposals bit-exact on the device's tensors, and one 20-RoI classifier
    chunk against the oracle (the 300-RoI single pass is covered at img_size = 600 above)."""
    from faster_rcnn import models as M
    from faster_rcnn import rpn
    from faster_rcnn.RADNet import RADNet, resize_cubic
    from faster_rcnn.base_models import resnet50
    from faster_rcnn.config import Config
    from oracle import dense, glue, resize as oresize, step as ostep
    C = Config()
    C.img_size = 1000
    P = dense.init_params(seed=3)
    m_rpn, m_cls, m_all, m_rpn3, m_det = M.build_models(C, weights=copy.deepcopy(P))
    tile = np.random.RandomState(4).randint(0, 256, (2048, 2048, 3)).astype(np.uint8)
    net = RADNet(C, m_rpn3, m_det, resnet50.preprocess)
    small = resize_cubic(tile, 1000, 1000)
    assert np.array_equal(small, oresize.resize_bicubic_u8(tile, 1000, 1000))
    X, ratio = net.format_img(tile)
    assert X.shape == (1, 1000, 1000, 3)
    Y1, Y2, F = m_rpn3.predict(X)
    p, r, F_ref = ostep.rpn_only_forward(P, small)
    assert F.shape == F_ref.shape == (1, 63, 63, 1024)
    assert check(F, F_ref, 1e-3) < 1e-3 and check(Y1, p, 1e-3) < 1e-3 and check(Y2, r, 1e-3) < 1e-3
    R = rpn.rpn_to_roi(Y1, Y2, C, overlap_thresh=0.7)
    assert np.array_equal(R, glue.rpn_to_roi(Y1, Y2, C, True, 300, 0.7))
    Rx = R[:20].copy()
    Rx[:, 2] -= Rx[:, 0]; Rx[:, 3] -= Rx[:, 1]
    pc, pr = m_det.predict([F, Rx[None]])
    rc, rr, _ = dense.head_forward(P, F, Rx.astype(np.float32), 7)
    assert np.abs(pc - rc).max() < 2e-3 and np.abs(pr - rr).max() < 2e-3 * max(1.0, np.abs(rr).max())
