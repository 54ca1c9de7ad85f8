"""The chain kernel (radnet_chain_build / radnet_chain_run, include/radnet_hip.h): nn_base stages 2-4 (resnet50.py:150-228) as
ONE persistent launch whose workgroups draw output tiles / Winograd transform blocks from a dependency-ordered list and wait
on arrival counters -- against the launch-by-launch layer program (same kernels' code) and the oracle.

  * same feature map as the launch list to fp32 rounding (other K-split / tile shapes: another summation order) and within
    the suite's activation tolerance of the oracle;
  * reproducible: two replays give the same bits (K-split slices are added in slice order);
  * correct for ANY grid width -- 3 workgroups (items almost in list order), 64, the default, 1024 (more than fit the chip):
    the dependency-ordered list cannot deadlock;
  * replayable from a hipGraph; no item ever gave up waiting (radnet_chain_status)."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tolerances import check  # noqa: E402


def _engine(chain, wgs=0, img_size=300):
    from faster_rcnn.config import Config
    from oracle import dense
    from radnet_hip.engine import FasterRCNNEngine
    C = Config()
    C.img_size = img_size
    eng = FasterRCNNEngine(C)
    eng.use_chain, eng.chain_wgs = chain, wgs
    P = dense.init_params(seed=3)
    eng.set_weights(P)
    return C, P, eng


@pytest.mark.parametrize("shape", [(300, 500), (600, 1000)])
def test_chain_base_forward_equals_launch_list_and_oracle(shape):
    from oracle import dense
    from radnet_hip import synth
    H, W = shape
    img = synth.synthetic_panel(1, H, W)
    C, P, e0 = _engine(False, img_size=min(H, W))
    bp0 = e0.upload_image(img)
    F0 = e0.base_forward(bp0).cpu().numpy()
    C, P, e1 = _engine(True, img_size=min(H, W))
    bp1 = e1.upload_image(img)
    assert bp1.get("chain") is not None and len(bp1["ops"]) == 3           # conv1, max-pool, the chain
    F1 = e1.base_forward(bp1).cpu().numpy()
    err, runs, n_items, n_stages, fe, fa = e1.chain_status(bp1)
    assert err == 0 and runs == 1 and n_stages == 32 + 3 * 10              # 32 direct convs + 10 Winograd layers x 3 stages
    assert fa > fe > 0
    # same function, another fp32 summation order (tile shapes, K slices): 2e-5 of the largest activation overall, and every
    # channel within 5e-4 of its own scale (a weak channel of a 16-block-deep map carries its rounding noise relative to that)
    assert check(F1, F0, 5e-4, "chain vs launch list") < 2e-5
    if shape == (300, 500):
        F_ref = dense.base_forward(P, dense.preprocess_caffe_bgr(img))
        check(F1, F_ref, 1e-3, "chain vs oracle")
    # replays (the second and third through the recorded hipGraph): the same bits
    for _ in range(3):
        bp1["F"].zero_()
        Fr = e1.base_forward(bp1).cpu().numpy()
        assert np.array_equal(Fr, F1)
    err, runs, *_ = e1.chain_status(bp1)
    assert err == 0 and runs == 4


@pytest.mark.parametrize("wgs", [3, 64, 1024])
def test_chain_any_grid_width(wgs):
    from radnet_hip import synth
    img = synth.synthetic_panel(2, 300, 500)
    C, P, e0 = _engine(True, 0)
    F0 = e0.base_forward(e0.upload_image(img)).cpu().numpy()
    C, P, e1 = _engine(True, wgs)
    bp = e1.upload_image(img)
    F1 = e1.base_forward(bp).cpu().numpy()
    err, runs, *_ = e1.chain_status(bp)
    assert err == 0 and runs == 1
    assert np.array_equal(F1, F0)                                          # the grid width changes who runs an item, not its arithmetic


def test_chain_refuses_what_it_cannot_run():
    """An op list with a launch the chain has no item type for: RADNET_ERR_UNSUPPORTED with a message, nothing built."""
    import ctypes as C_
    C, P, eng = _engine(False)
    from radnet_hip import synth
    bp = eng.upload_image(synth.synthetic_panel(1, 300, 500))
    arr = eng._compile(bp["ops"])                                          # starts with conv1 (4 channels) and the max-pool
    h = C_.c_void_p()
    rc = eng.lib.radnet_chain_build(eng.ctx.h, C_.cast(arr, C_.c_void_p), len(bp["ops"]), 0, C_.byref(h))
    assert rc == -3 and not h.value and b"chain" in eng.lib.radnet_last_error(eng.ctx.h)


def test_train_step_with_chain_matches_launch_list():
    """A whole training iteration with the base forward as a chain: losses within rounding of the launch-list engine, same
    RNG consumption, proposals identical or reordered only among near-ties."""
    from radnet_hip import synth
    from radnet_hip.trainer import TrainStep
    meta = synth.synthetic_gt(40, n=6, src_w=1000, src_h=600, smin=60, smax=300)
    sample = dict(img=synth.synthetic_panel(30, 300, 500), bboxes=meta["bboxes"], width=1000, height=600)
    res = []
    for chain in (False, True):
        C, P, eng = _engine(chain)
        np.random.seed(64)
        ts = TrainStep(eng)
        ts.step([sample])
        res.append((ts.losses(), int(np.random.randint(0, 2 ** 31 - 1))))
    (l0, r0), (l1, r1) = res
    assert r0 == r1 and l0["n_head"] == l1["n_head"] == 1
    for k in ("rpn_cls", "rpn_regr"):
        assert abs(l0[k] - l1[k]) <= 1e-4 * abs(l0[k]), (k, l0[k], l1[k])
    for k in ("det_cls", "det_regr"):
        assert abs(l0[k] - l1[k]) <= 2e-3 * abs(l0[k]) + 1e-5, (k, l0[k], l1[k])
