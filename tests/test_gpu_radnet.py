"""Drop-in package (faster_rcnn.*) on the GPU: function-style API vs goldens from the reference, RADNet.predict with
fake models vs the reference's own output, Keras-like models vs the oracle."""
import copy

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_rpn_functions_match_reference_goldens():
    from faster_rcnn import rpn
    from faster_rcnn.config import Config
    g = load_golden("rpn_to_roi")
    for i in (0, 3, 4, 6):
        C = Config(); C.anchor_box_scales = [int(v) for v in g[f"c{i}_scales"]]
        R = rpn.rpn_to_roi(g[f"c{i}_cls"], g[f"c{i}_regr"], C, use_regr=True, max_boxes=int(g[f"c{i}_max"]), overlap_thresh=float(g[f"c{i}_thr"]))
        assert R.dtype == np.int64 and np.array_equal(R, g[f"c{i}_R"])
    g = load_golden("nms")
    for i in range(int(g["n_cases"])):
        b, p = rpn.non_max_suppression_fast(g[f"c{i}_boxes"], g[f"c{i}_probs"], overlap_thresh=float(g[f"c{i}_thr"]), max_boxes=int(g[f"c{i}_max"]))
        assert np.array_equal(b, g[f"c{i}_out_boxes"]) and np.array_equal(p, g[f"c{i}_out_probs"])
    assert rpn.non_max_suppression_fast(np.zeros((0, 4)), np.zeros(0)) == []
    with pytest.raises(AssertionError):
        rpn.non_max_suppression_fast(np.array([[1., 1., 1., 4.]]), np.array([0.3]))
    g = load_golden("calc_iou")
    C = Config()
    classes = {v: k for k, v in C.class_mapping.items()}
    for i in range(int(g["n_cases"])):
        W, H = (int(v) for v in g[f"c{i}_wh"])
        bboxes = [{"class": classes[int(c)], "x1": b[0], "y1": b[1], "x2": b[2], "y2": b[3]} for b, c in zip(g[f"c{i}_gt_boxes"], g[f"c{i}_gt_cls"])]
        X, Y1, Y2, ious = rpn.calc_iou(g[f"c{i}_R"], {"bboxes": bboxes, "width": W, "height": H}, C, C.class_mapping)
        assert np.array_equal(X, g[f"c{i}_X"]) and np.array_equal(Y1, g[f"c{i}_Y1"])
        assert np.array_equal(np.array(ious), g[f"c{i}_ious"])
        assert np.array_equal(Y2[..., :24], g[f"c{i}_Y2"][..., :24])
        assert np.allclose(Y2[..., 24:], g[f"c{i}_Y2"][..., 24:], rtol=5e-16, atol=0)        # device log vs NumPy log: 1 ulp
    bb = [{"class": "boat", "x1": 1900, "x2": 1990, "y1": 1100, "y2": 1190}]
    assert rpn.calc_iou(np.array([[0, 0, 2, 2]]), {"bboxes": bb, "width": 2000, "height": 1200}, C, C.class_mapping) == (None, None, None, None)


def test_calc_region_props_matches_reference_goldens():
    from faster_rcnn import utils
    from faster_rcnn.base_models import resnet50
    from faster_rcnn.config import Config
    g = load_golden("calc_region_props")
    for i in range(int(g["n_cases"])):
        W, H, rw, rh, isz, rseed = (int(v) for v in g[f"c{i}_wh"])
        C = Config(); C.img_size = isz
        bboxes = [{"class": "bg" if bg else "boat", "x1": b[0], "y1": b[1], "x2": b[2], "y2": b[3]} for b, bg in zip(g[f"c{i}_gt_boxes"], g[f"c{i}_gt_is_bg"])]
        np.random.seed(rseed)
        ycls, yregr, best, n_pos = utils.calc_region_props(C, {"bboxes": bboxes}, W, H, rw, rh, resnet50.get_img_output_length)
        assert n_pos == int(g[f"c{i}_n_pos"])
        assert np.random.randint(0, 2 ** 31 - 1) == int(g[f"c{i}_rng_after"])
        assert np.array_equal(ycls, g[f"c{i}_y_rpn_cls"])
        assert np.array_equal(best, g[f"c{i}_best_anchor"])
        ref = g[f"c{i}_y_rpn_regr"]
        assert np.array_equal(yregr[:, :48], ref[:, :48])
        assert np.array_equal(yregr.astype(np.float32), ref.astype(np.float32))          # what the network consumes
        assert np.all(np.abs(yregr - ref) <= 4.5e-16 * np.abs(ref))


class _FakeDet:
    def __init__(self, nc, seed, tie_free=False):
        from test_oracle_glue import fake_detector, fake_detector_tie_free
        self._f = (fake_detector_tie_free if tie_free else fake_detector)(nc, seed, [])
        self._tie_free = tie_free

    def predict(self, inputs):
        return self._f(inputs[1], inputs[0]) if self._tie_free else self._f(inputs[1])


class _FakeRPN:
    """The golden generator's closed-form RPN stand-in."""
    def __init__(self, A, seed):
        self.A, self.seed = A, seed

    def predict(self, X):
        from oracle import glue
        h, w = glue.resnet50_feat_len(X.shape[1]), glue.resnet50_feat_len(X.shape[2])
        rs = np.random.RandomState(self.seed + int(abs(float(X.sum()))) % 1000)
        n = h * w * self.A
        cls = (rs.permutation(n).astype(np.float32) / np.float32(n)).reshape(1, h, w, self.A)
        regr = (rs.standard_normal((1, h, w, 4 * self.A)) * 2.0).astype(np.float32)
        return [cls, regr, rs.standard_normal((1, h, w, 8)).astype(np.float32)]


def test_radnet_predict_matches_reference_output():
    """RADNet.predict (tiling, RPN->NMS->classifier decode->per-class NMS->box-averaging merge->cross-image NMS) with
    the same fake models the reference's RADNet was driven with: identical detections."""
    from faster_rcnn.config import Config
    from faster_rcnn.RADNet import RADNet
    g = load_golden("predict_fake")
    C = Config(); C.tile_size = 600; C.tile_overlap = 300; C.img_size = 600
    net = RADNet(C, _FakeRPN(12, 8), _FakeDet(7, 2), lambda x: x - np.float32(100.0))
    dets = net.predict([g["img"]])
    assert len(dets) == int(g["n"])
    assert [d["class"] for d in dets] == list(g["classes"])
    # The fake detector saturates (many probabilities are exactly equal), and among EQUAL scores the reference's
    # order is whatever np.argsort's unstable default produced (SURVEY.md A.4 rule 6); this build's rule is "stable
    # ascending, walk from the end".  Compare per class as multisets: same boxes, same probabilities.
    got = sorted((d["class"], int(d["x1"]), int(d["y1"]), int(d["x2"]), int(d["y2"]), float(d["prob"])) for d in dets)
    ref = sorted((str(c), int(b[0]), int(b[1]), int(b[2]), int(b[3]), float(p)) for c, b, p in zip(g["classes"], g["boxes"], g["probs"]))
    assert got == ref


def _same_detections(dets, g, prefix):
    assert len(dets) == int(g[prefix + "n"])
    got = sorted((d["class"], int(d["x1"]), int(d["y1"]), int(d["x2"]), int(d["y2"]), float(d["prob"])) for d in dets)
    ref = sorted((str(c), int(b[0]), int(b[1]), int(b[2]), int(b[3]), float(p)) for c, b, p in zip(g[prefix + "classes"], g[prefix + "boxes"], g[prefix + "probs"]))
    assert got == ref


def test_radnet_predict_full_image_pass_matches_reference_output():
    """C.include_full_img (RADNet.py:606-665): the whole panel at the network size as one more source of detections.
    (a) on top of the tiles, panel already at the network size; (b) full image ONLY (max_n_tiles_train = 0), two panels of
    different sizes scaled by the device bicubic kernel -- the reference ran with cv2.resize bound to oracle/resize.py, which
    that kernel equals bit for bit -- so format_img's ratio and get_real_coordinates are exercised off the identity."""
    from faster_rcnn.config import Config
    from faster_rcnn.RADNet import RADNet
    g = load_golden("predict_fake")
    C = Config(); C.tile_size = 600; C.tile_overlap = 300; C.img_size = 600; C.include_full_img = True
    net = RADNet(C, _FakeRPN(12, 8), _FakeDet(7, 2), lambda x: x - np.float32(100.0))
    _same_detections(net.predict([g["img"]]), g, "full_")
    assert int(g["full_n"]) != int(g["n"])                       # the pass changes the result
    C = Config(); C.img_size = 300; C.include_full_img = True; C.max_n_tiles_train = 0
    net = RADNet(C, _FakeRPN(12, 4), _FakeDet(7, 6, tie_free=True), lambda x: x - np.float32(100.0))
    dets = net.predict([g["only_img_a"], g["only_img_b"]])
    _same_detections(dets, g, "only_")
    assert [d["class"] for d in dets] == list(g["only_classes"])                 # tie-free scores: the ORDER is pinned too
    assert np.array_equal(np.array([[d["x1"], d["y1"], d["x2"], d["y2"]] for d in dets]), g["only_boxes"])


def test_tile_feed_full_image_pass_device_resize():
    """The *_full cases of tests/golden/tile_feed.json (the reference's generator with C.include_full_img) through TileFeed's
    DEFAULT resize, the device bicubic kernel: same samples, same pixel sums as the reference's resized panels."""
    import json
    import os
    from faster_rcnn import data_feed as F
    from faster_rcnn.config import Config
    from test_data_feed import AUG, CLASSES, dataset
    from oracle import glue
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tile_feed.json")))
    cases = [c for c in G["cases"] if c.get("full")]
    assert len(cases) == 3
    for case in cases:
        C = Config()
        C.img_size, C.tile_size, C.tile_overlap = 300, 300, 150
        C.max_n_tiles_train, C.max_n_tiles_val = 2, 3
        C.balanced_classes, C.include_full_img, C.use_img_type = case["balanced"], True, False
        for k in AUG:
            setattr(C, k, False)
        data, imgs = dataset(case["data_seed"], [tuple(s) for s in case["sizes"]])
        class_count = {c: sum(1 for d in data for b in d["bboxes"] if b["class"] == c) for c in CLASSES}
        np.random.seed(case["seed"])
        feed = iter(F.TileFeed(data, C, class_count, lambda d, t: imgs[d["filepath"]], train_mode=case["train"]))
        for ref in case["yields"]:
            s = next(feed)
            assert (s["filepath"], s["width"], s["height"]) == (ref["filepath"], ref["width"], ref["height"])
            assert int(s["img"].astype(np.int64).sum()) == ref["img_sum"]
            boxes = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in s["bboxes"]], dtype=np.float64)
            is_bg = np.array([b["class"] == "bg" for b in s["bboxes"]])
            rw, rh = glue.new_img_size(s["width"], s["height"], C.img_size)
            glue.anchor_targets(C, boxes, is_bg, s["width"], s["height"], rw, rh, lambda w, h: (glue.resnet50_feat_len(w), glue.resnet50_feat_len(h)))
        assert np.random.randint(0, 2 ** 31 - 1) == case["rng_after"]


@pytest.fixture(scope="module")
def models():
    from faster_rcnn import models as M
    from faster_rcnn.config import Config
    from oracle import dense
    C = Config(); C.img_size = 300
    P = dense.init_params(seed=3)
    return C, P, M.build_models(C, weights=copy.deepcopy(P))


def test_keras_like_models_vs_oracle(models):
    from oracle import dense, glue
    from faster_rcnn.base_models import resnet50
    C, P, (m_rpn, m_cls, m_all, m_rpn3, m_det) = models
    P = copy.deepcopy(P)
    rs = np.random.RandomState(21)
    img = rs.randint(0, 256, (300, 480, 3)).astype(np.uint8)
    X = resnet50.preprocess(img[:, :, (2, 1, 0)].astype(np.float32)[None])
    assert np.allclose(X, dense.preprocess_caffe_bgr(img), atol=1e-4)
    # inference flavour: 3 outputs
    Y1, Y2, F = m_rpn3.predict(X)
    p, r, cache = dense.rpn_forward(P, dense.base_forward(P, X))
    assert Y1.shape == p.shape and Y2.shape == r.shape and F.shape == cache["F"].shape
    assert np.abs(Y1 - p).max() < 1e-3 and np.abs(Y2 - r).max() < 1e-3 * np.abs(r).max()
    # detector on the returned feature map
    rois = np.stack([rs.randint(0, 20, 20), rs.randint(0, 10, 20), rs.randint(1, 9, 20), rs.randint(1, 8, 20)], 1)[None]
    pc, pr = m_det.predict([F, rois])
    rc, rr, _ = dense.head_forward(P, cache["F"], rois[0].astype(np.float32), 7)
    assert np.abs(pc - rc).max() < 1e-3 and np.abs(pr - rr).max() < 1e-3 * max(1.0, np.abs(rr).max())
    # a foreign feature-map array (not the cached one) takes the upload path and gives the same answer
    pc2, _ = m_det.predict([F.copy(), rois])
    assert np.array_equal(pc, pc2)
    # RPN train / test on batch: Keras order [total, cls, regr]
    fh, fw = F.shape[1:3]
    valid = (rs.uniform(size=(1, fh, fw, 12)) < 0.05).astype(np.float32)
    ov = ((rs.uniform(size=(1, fh, fw, 12)) < 0.3) * valid).astype(np.float32)
    Y = [np.concatenate([valid, ov], -1), np.concatenate([np.repeat(ov, 4, -1), rs.standard_normal((1, fh, fw, 48)).astype(np.float32)], -1)]
    lt = m_rpn.test_on_batch(X, Y)
    ref, grads = dense.rpn_losses_and_grads(P, cache["F"], Y[0], Y[1], 12, True)
    assert abs(lt[1] - ref[1]) < 1e-3 * abs(ref[1]) and abs(lt[2] - ref[2]) < 1e-3 * abs(ref[2]) + 1e-6
    l1 = m_rpn.train_on_batch(X, Y)
    assert abs(l1[0] - (l1[1] + l1[2])) < 1e-5 and abs(l1[1] - lt[1]) < 1e-6
    l2 = m_rpn.test_on_batch(X, Y)
    assert l2[0] < l1[0]                                     # one Adam step on the same batch lowers the loss
    # classifier train on batch: [total, cls, regr, acc]
    cls = rs.randint(0, 7, 20)
    T1 = np.eye(7, dtype=np.float32)[cls][None]
    lab = np.zeros((20, 24), np.float32)
    for i, c in enumerate(cls):
        if c != 6:
            lab[i, 4 * c:4 * c + 4] = 1
    T2 = np.concatenate([lab, rs.standard_normal((20, 24)).astype(np.float32) * lab], -1)[None]
    d0 = m_cls.test_on_batch([X, rois], [T1, T2])
    refd, _ = dense.head_losses_and_grads(P, cache["F"], rois[0].astype(np.float32), T1, T2, 7)
    assert abs(d0[1] - refd[1]) < 2e-3 * abs(refd[1]) and abs(d0[2] - refd[2]) < 2e-3 * abs(refd[2]) + 1e-6 and abs(d0[3] - refd[3]) < 1e-6
    d1 = m_cls.train_on_batch([X, rois], [T1, T2])
    d2 = m_cls.test_on_batch([X, rois], [T1, T2])
    assert len(d1) == 4 and d2[0] < d1[0]


def test_save_load_weights_roundtrip(models, tmp_path):
    C, P, (m_rpn, m_cls, m_all, m_rpn3, m_det) = models
    path = str(tmp_path / "weights.npz")
    m_all.save_weights(path)
    z = np.load(path)
    assert "rpn_conv1/kernel" in z.files and "bn_conv1/gamma" in z.files and z["res5a_branch2a/kernel"].shape == (1, 1, 1024, 512)
    X = np.random.RandomState(3).standard_normal((1, 300, 300, 3)).astype(np.float32) * 50
    a = m_rpn3.predict(X)
    m_all.load_weights(path, by_name=True)
    b = m_rpn3.predict(X)
    assert all(np.array_equal(u, v) for u, v in zip(a, b))
    # Keras HDF5 weights (train.py:574 save_weights / RADNet.py:754,769 load_weights(by_name=True)); parity vs h5py unpinned
    from faster_rcnn import keras_h5
    h5 = str(tmp_path / "weights.hdf5")
    m_all.save_weights(h5)
    layers, names = keras_h5.read_keras_weights(h5)
    assert names["conv1"] == ["conv1/kernel:0", "conv1/bias:0"] and layers["res5a_branch2a"][0].shape == (1, 1, 1024, 512)
    assert names["bn_conv1"][3] == "bn_conv1/bn_conv1_running_std:0" and len(layers["rpn_out_regress"]) == 2
    W0 = m_all._s.eng.get_weights()
    m_rpn.train_on_batch(X, [np.ones((1,) + a[0].shape[1:3] + (24,), np.float32), np.ones((1,) + a[0].shape[1:3] + (96,), np.float32)])
    Wt = m_all._s.eng.get_weights()
    assert not np.array_equal(Wt["rpn_conv1"]["kernel"], W0["rpn_conv1"]["kernel"])      # the step really moved the weights
    m_all.load_weights(h5, by_name=True)                         # restores the saved weights exactly
    W1 = m_all._s.eng.get_weights()
    assert all(np.array_equal(W0[n][k], W1[n][k]) for n in W0 for k in W0[n])
    c = m_rpn3.predict(X)
    assert all(np.array_equal(u, v) for u, v in zip(a, c))


def test_cfg3_predict_tile_2048(models):
    """BASELINE config 3: predict.py path on a 2048x2048 synthetic tile (seed 4), resized to short side C.img_size on
    the device.  Stage-wise against the oracle on the device's own tensors: proposals bit-exact, per-chunk classifier
    outputs within fp32 tolerance, and the decoded detections identical to the oracle's decode of those outputs."""
    from faster_rcnn.RADNet import RADNet, resize_cubic
    from faster_rcnn.base_models import resnet50
    from oracle import dense, glue
    C, P0, (m_rpn, m_cls, m_all, m_rpn3, m_det) = models
    tile = np.random.RandomState(4).randint(0, 256, (2048, 2048, 3)).astype(np.uint8)
    net = RADNet(C, m_rpn3, m_det, resnet50.preprocess)
    X, ratio = net.format_img(tile)
    assert X.shape == (1, 300, 300, 3) and abs(ratio - 300 / 2048) < 1e-12
    small = resize_cubic(tile, 300, 300)
    assert small.dtype == np.uint8 and 100 < small.mean() < 155 and small.std() < tile.std()      # low-pass of white noise
    Y1, Y2, F = m_rpn3.predict(X)
    from faster_rcnn import rpn
    R = rpn.rpn_to_roi(Y1, Y2, C, overlap_thresh=0.7)
    assert np.array_equal(R, glue.rpn_to_roi(Y1, Y2, C, True, 300, 0.7))
    R[:, 2] -= R[:, 0]; R[:, 3] -= R[:, 1]
    bb, pp = net.apply_spatial_pyramid_pooling(R, F)
    W = m_all._s.eng.get_weights()
    Pnow = copy.deepcopy(P0); Pnow.update(W)
    calls = []

    def oracle_det(rois):
        pc, pr, _ = dense.head_forward(Pnow, F, rois[0].astype(np.float32), 7)
        calls.append((pc, pr))
        return [pc, pr]
    # the oracle's classifier on the same feature map agrees chunk by chunk
    pc_gpu, pr_gpu = m_det.predict([F, R[:20][None]])
    pc_ref, pr_ref = oracle_det(R[:20][None])
    assert np.abs(pc_gpu - pc_ref).max() < 2e-3 and np.abs(pr_gpu - pr_ref).max() < 2e-3 * max(1.0, np.abs(pr_ref).max())
    # decode parity: the oracle's spp_decode fed with the DEVICE outputs reproduces the facade's boxes exactly
    bb_ref, pp_ref = glue.spp_decode(R, lambda rois: m_det.predict([F, rois]), C)
    # (the facade sends all 300 RoIs through ONE head pass, the oracle walk calls the detector per chunk of 20: same
    # arithmetic per RoI, different GEMM tiling -> probabilities agree to fp32 rounding, decoded boxes exactly)
    assert sorted(bb) == sorted(bb_ref)
    for k in bb:
        assert np.array_equal(np.array(bb[k]), np.array(bb_ref[k]))
        assert np.allclose(np.array(pp[k]), np.array(pp_ref[k]), rtol=0, atol=1e-5)
    # and the chunked path itself (detector without the any-count capability) is still exact
    m_det.__class__.accepts_any_roi_count = False
    try:
        bb_c, pp_c = net.apply_spatial_pyramid_pooling(R, F)
    finally:
        m_det.__class__.accepts_any_roi_count = True
    for k in bb_ref:
        assert np.array_equal(np.array(bb_c[k]), np.array(bb_ref[k])) and np.array_equal(np.array(pp_c[k]), np.array(pp_ref[k]))
    dets = net.predict([tile])
    assert isinstance(dets, list)
    for d in dets:
        assert set(d) == {"class", "prob", "x1", "y1", "x2", "y2"}


def test_device_resident_detect_equals_numpy_facing_path(models):
    """RADNet._detect keeps a tile on the device from the resize to the classifier outputs when the models are engine-backed;
    forcing the NumPy-facing calls the reference makes (predict -> rpn_to_roi -> detector.predict) must give the very same
    proposals, decoded boxes and scores: same kernels, only the PCIe round trips differ."""
    from faster_rcnn import rpn
    from faster_rcnn.RADNet import RADNet
    from faster_rcnn.base_models import resnet50
    C, P0, (m_rpn, m_cls, m_all, m_rpn3, m_det) = models
    net = RADNet(C, m_rpn3, m_det, resnet50.preprocess)
    net.bbox_threshold = 0.0                                   # synthetic weights: keep every non-background RoI in the comparison
    for shape in ((2048, 2048, 3), (300, 300, 3), (700, 1100, 3)):
        tile = np.random.RandomState(sum(shape)).randint(0, 256, shape).astype(np.uint8)
        # stage-wise: proposals
        img_dev, ratio_d = net.format_img_size(tile, keep_on_device=True)
        R_dev, F_dev = m_rpn3.propose_device(img_dev, overlap_thresh=0.7)
        X, ratio_h = net.format_img(tile)
        Y1, Y2, F = m_rpn3.predict(X)
        R_host = rpn.rpn_to_roi(Y1, Y2, C, overlap_thresh=0.7)
        assert ratio_d == ratio_h and np.array_equal(R_dev, R_host)
        # end to end
        net.device_resident = True
        a = net._detect(tile)
        net.device_resident = False
        b = net._detect(tile)
        assert a.keys() == b.keys() and len(a) > 0
        for k in a:
            assert a[k][0] == b[k][0] and np.array_equal(np.array(a[k][1]), np.array(b[k][1]))
    # two tiles in flight (side lane: upload .. proposals of tile j+1; main lane: classifier of tile j) == one at a time
    tiles = [np.random.RandomState(70 + i).randint(0, 256, sh).astype(np.uint8) for i, sh in enumerate(((640, 640, 3), (640, 640, 3), (300, 420, 3), (640, 640, 3), (2048, 2048, 3)))]
    net.device_resident = True
    piped = net._detect_all(tiles)
    one_by_one = [net._detect(t) for t in tiles]
    assert len(piped) == len(tiles)
    for a, b in zip(piped, one_by_one):
        assert a.keys() == b.keys()
        for k in a:
            assert a[k][0] == b[k][0] and np.array_equal(np.array(a[k][1]), np.array(b[k][1]))
