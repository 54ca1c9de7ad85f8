"""RADNet facade (drop-in surface) driven with the closed-form fake models the golden generator used on the
reference's own RADNet class: host-side parts only (no GPU)."""
import numpy as np
import pytest

from conftest import load_golden
from faster_rcnn.config import Config
from faster_rcnn.RADNet import RADNet, _spans
from test_oracle_glue import fake_detector


class _Det:
    def __init__(self, nc, seed):
        self.calls = []
        self._f = fake_detector(nc, seed, self.calls)

    def predict(self, inputs):
        return self._f(inputs[1])


def test_apply_spatial_pyramid_pooling_golden():
    g = load_golden("spp")
    det = _Det(7, 5)
    net = RADNet(Config(), None, det, lambda x: x)
    bb, pp = net.apply_spatial_pyramid_pooling(g["R"], np.zeros((1, 38, 63, 8), np.float32))
    assert sorted(bb) == list(g["classes"])
    assert np.array_equal(np.stack(det.calls), g["detector_calls"])          # same chunks, same padding
    for k in bb:
        assert np.array_equal(np.array(bb[k], dtype=np.int64), g[f"boxes_{k}"])
        assert np.array_equal(np.array(pp[k], dtype=np.float64), g[f"probs_{k}"])
    det2 = _Det(7, 9)
    net2 = RADNet(Config(), None, det2, None)
    bb2, _ = net2.apply_spatial_pyramid_pooling(g["R"][:40], None)
    assert len(det2.calls) == int(g["n_calls_40"])
    for k in bb2:
        assert np.array_equal(np.array(bb2[k], dtype=np.int64), g[f"m40_boxes_{k}"])


def test_final_nms_and_real_coordinates_golden():
    g = load_golden("final_nms")
    net = RADNet(Config(), None, None, None)
    for i in range(int(g["n_cases"])):
        b, p = net.final_nms(g[f"c{i}_boxes"].copy(), g[f"c{i}_probs"].copy(), obj_avg_threshold=0.2, obj_confidence_threshold=0.8, n_obj_avg=5)
        assert np.array_equal(b, g[f"c{i}_out_boxes"]) and np.array_equal(p, g[f"c{i}_out_probs"])
    out = np.array([[net.get_real_coordinates(r, *[int(v) for v in c]) for c in g["grc_in"]] for r in g["grc_ratios"]])
    assert np.array_equal(out, g["grc_out"])
    assert net.final_nms(np.zeros((0, 4)), np.zeros(0)) == []


def test_tile_spans():
    assert _spans(900, 600, 300) == [(0, 600), (300, 900)]
    assert _spans(600, 600, 300) == [(0, 600)]
    assert _spans(500, 2000, 400) == [(0, 500)]                  # image smaller than a tile: one flush window
    assert _spans(2500, 2000, 400) == [(0, 2000), (400, 2400), (500, 2500)]


def test_surface_attributes():
    C = Config()
    net = RADNet(C, "rpn", "det", "prep")
    assert (net.is_object_threshold, net.bbox_threshold) == (0.5, 0.7)
    assert net.class_mapping == {v: k for k, v in C.class_mapping.items()}
    assert net.model_rpn == "rpn" and net.model_detector == "det" and net.preprocess_func == "prep"


def test_config_unpickler_admits_config_only(tmp_path):
    """load_radnet reads config.pickle (RADNet.py:724) through a restricted unpickler: a Config round-trips (including the
    attributes train.py adds dynamically), a pickle naming any other callable is refused before it can run."""
    import io
    import pickle

    from faster_rcnn.RADNet import _ConfigUnpickler
    from faster_rcnn.config import Config
    C = Config()
    C.weights_path = "models/x/weights.hdf5"            # train.py:174
    C.class_mapping = dict(C.class_mapping)
    back = _ConfigUnpickler(io.BytesIO(pickle.dumps(C))).load()
    assert isinstance(back, Config) and back.__dict__ == C.__dict__

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))
    with pytest.raises(pickle.UnpicklingError):
        _ConfigUnpickler(io.BytesIO(pickle.dumps(Evil()))).load()
    assert not (tmp_path / "pwned").exists()
