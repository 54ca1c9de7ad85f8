"""Host-side logic of the product (no GPU): RNG-parity sampling helpers, synthetic generators, layer bookkeeping."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import dense, glue
from faster_rcnn.config import Config
from radnet_hip import engine as E
from radnet_hip import synth


@pytest.mark.parametrize("seed,n,size", [(0, 50, 10), (1, 20000, 19984), (2, 137, 9), (3, 5000, 4999), (4, 300, 300), (5, 7, 1)])
def test_choice_without_replacement_equals_numpy(seed, n, size):
    rs = np.random.RandomState(seed + 100)
    p = rs.uniform(0.5, 1.5, n)
    if seed % 2 == 1:
        p[:] = 1.0            # the uniform table the labeller produces
    p /= p.sum()
    np.random.seed(seed)
    ref = np.random.choice(n, size, replace=False, p=p)
    after_ref = np.random.randint(0, 2 ** 31 - 1)
    np.random.seed(seed)
    got = E.choice_without_replacement(n, size, p)
    after_got = np.random.randint(0, 2 ** 31 - 1)
    assert np.array_equal(got, ref)
    assert after_got == after_ref           # identical consumption of the global MT19937 stream
    with pytest.raises(ValueError):
        E.choice_without_replacement(n, n + 1, p)


def test_subsample_valid_matches_reference_goldens():
    """engine.subsample_valid (host half of calc_region_props) on the oracle's pre-subsampling maps reproduces the
    reference's final y_rpn_cls bit-for-bit with the same RNG consumption."""
    g = load_golden("calc_region_props")
    for i in range(int(g["n_cases"])):
        W, H, rw, rh, isz, rseed = (int(v) for v in g[f"c{i}_wh"])
        C = Config(); C.img_size = isz
        fw, fh = glue.resnet50_feat_len(rw), glue.resnet50_feat_len(rh)
        d = glue.anchor_targets_dense(C, g[f"c{i}_gt_boxes"], g[f"c{i}_gt_is_bg"], W, H, rw, rh, fw, fh)
        valid = np.ascontiguousarray(np.transpose(d["valid"], (2, 0, 1))).astype(np.uint8)
        overlap = np.ascontiguousarray(np.transpose(d["overlap"], (2, 0, 1))).astype(np.uint8)
        np.random.seed(rseed)
        n_pos = E.subsample_valid(valid, overlap)
        assert n_pos == int(g[f"c{i}_n_pos"])
        assert np.random.randint(0, 2 ** 31 - 1) == int(g[f"c{i}_rng_after"])
        A = 12
        assert np.array_equal(valid, g[f"c{i}_y_rpn_cls"][0, :A])
        assert np.array_equal(overlap, g[f"c{i}_y_rpn_cls"][0, A:])


def test_subsample_keyerror_like_reference():
    valid = np.zeros((2, 20, 20), np.uint8); overlap = np.zeros((2, 20, 20), np.uint8)
    valid[0, :, :10] = 1; overlap[0, :, :10] = 1          # 200 positives in channel 0, no negatives there
    valid[1, :, :] = 1                                     # negatives only in channel 1
    with pytest.raises(KeyError):
        E.subsample_valid(valid, overlap)


def test_select_samples_same_stream_as_oracle():
    rs = np.random.RandomState(3)
    for n_pos, n_neg in [(3, 60), (15, 40), (0, 25), (12, 3), (8, 0)]:
        cls = np.concatenate([rs.randint(0, 6, n_pos), np.full(n_neg, 6)])
        rs.shuffle(cls)
        Y1 = np.eye(7)[cls][None]
        np.random.seed(11)
        a = E.select_samples(cls, 6, 20)
        np.random.seed(11)
        b = glue.select_samples(Y1, 20)
        assert a == b


def test_synthetic_weights_equal_oracle_init():
    a, b = dense.init_params(seed=3), synth.synthetic_weights(seed=3)
    assert sorted(a) == sorted(b)
    for n in a:
        for k in a[n]:
            assert np.array_equal(a[n][k], b[n][k]), (n, k)


def test_feature_size_formula():
    for L in (600, 800, 1000, 2000, 333, 240):
        assert E.feat_len(L) == glue.resnet50_feat_len(L)


def test_plan_cache_is_bounded_per_kind():
    """engine.PlanCache: least-recently-used plans of a kind are evicted beyond the limit; other kinds are untouched."""
    from radnet_hip.engine import PlanCache
    gone = []
    pc = PlanCache(3, lambda k, v: gone.append(k))
    for i in range(3):
        pc[("head", i)] = {"i": i}
    pc[("base", 0)] = {}
    assert pc[("head", 0)]["i"] == 0                 # touch: ("head", 1) is now the oldest head plan
    pc[("head", 3)] = {}
    assert gone == [("head", 1)] and ("head", 0) in pc and ("base", 0) in pc and len(pc) == 4
    pc[("head", 4)] = {}
    assert gone == [("head", 1), ("head", 2)]
