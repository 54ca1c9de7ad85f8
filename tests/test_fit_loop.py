"""radnet_hip.fit: the epoch driver around the training step (train.py:273-642 epochs / validation / save-on-best / record.csv;
cont_train.py:112-206 resume) over a recording stand-in for TrainStep -- host logic, no GPU."""
import csv
import os
import pickle

import numpy as np
import pytest

from radnet_hip import fit as F


class FakeStep:
    """Scripted TrainStep: sample k carries its own losses; 'skip' samples take no classifier step (n_pos 0 appended, not an
    iteration), 'drop' samples are dropped by the labeller (invisible to the loop)."""

    def __init__(self, val_totals):
        self.last_n_pos, self.last_took_head = [], []
        self.log, self.calls, self.flushes, self.validated = [], [], 0, 0
        self.val_totals = list(val_totals)
        self.announced = []

    def start_loss_log(self, capacity):
        self.capacity = capacity
        self.log = []

    def read_loss_log(self):
        return np.array(self.log, np.float32).reshape(-1, 5)

    def step(self, batch, upcoming=None):
        self.calls.append(batch[0]["k"])
        self.announced.append([b[0]["k"] for b in (upcoming or [])])
        self.last_n_pos, self.last_took_head = [], []
        for s in batch:
            if s["kind"] == "drop":
                self.last_n_pos.append(None); self.last_took_head.append(False)
            elif s["kind"] == "skip":
                self.last_n_pos.append(0); self.last_took_head.append(False)
            else:
                self.last_n_pos.append(s["n_pos"]); self.last_took_head.append(True)
                self.log.append(s["loss"])
                assert len(self.log) <= self.capacity

    def flush(self):
        self.flushes += 1

    def validate(self, samples):
        self.validated += 1
        t = self.val_totals.pop(0)
        if t is None:
            return {"n": 0, "skipped": 1, "dropped": 0}
        return {"n": 2, "rpn_cls": t / 4, "rpn_regr": t / 4, "det_cls": t / 4, "det_regr": t / 4, "det_acc": 0.5, "mean_overlapping_bboxes": 3.0, "total": t}


def samples(kinds):
    for k, kind in enumerate(kinds):
        yield dict(k=k, kind=kind, n_pos=k % 5, loss=[1.0 + k, 0.5, 0.25 * (k % 3), 0.125, 0.1 * (k % 10)])


def test_epochs_count_classifier_steps_and_means_follow_the_reference(tmp_path):
    kinds = ["ok", "skip", "ok", "drop", "ok", "ok", "skip", "ok", "ok", "ok", "ok"]
    ts = FakeStep([])
    saved = []
    rec = tmp_path / "record.csv"
    rows, best = F.fit(ts, samples(kinds), epochs=2, epoch_length=3, save_weights=saved.append, weights_path="w.hdf5", record_path=str(rec), lookahead=2)
    # epoch 1 = samples 0..4 (three classifier steps: 0, 2, 4; one skip; one drop), epoch 2 = samples 5..8 (5, 7, 8; skip at 6)
    assert ts.calls == [0, 1, 2, 3, 4, 5, 6, 7, 8] and ts.flushes == 2 and ts.validated == 0
    assert ts.announced[0] == [1, 2] and ts.announced[3] == [4, 5]                  # lookahead crosses the epoch boundary
    L1 = np.array([[1.0, 0.5, 0.0, 0.125, 0.0], [3.0, 0.5, 0.5, 0.125, 0.2], [5.0, 0.5, 0.25, 0.125, 0.4]])
    assert rows[0]["loss_rpn_cls"] == round(L1[:, 0].mean(), 3) and rows[0]["detector_acc"] == round(L1[:, 4].mean(), 3)
    assert rows[0]["total_loss"] == round(L1[:, :4].mean(0).sum(), 3)
    assert rows[0]["mean_overlapping_bboxes"] == round((0 + 0 + 2 + 4) / 4, 3)      # n_pos of 0, the skip's 0, n_pos of 2 and 4; the drop is invisible
    assert rows[0]["val_total_loss"] is None and rows[0]["model_improvement"] is None   # first save: from inf
    assert saved == ["w.hdf5"]                                                      # epoch 2's training total is larger: no second save
    assert rows[1]["model_improvement"] is None and best == pytest.approx(L1[:, :4].mean(0).sum())
    got = list(csv.reader(open(rec)))
    assert got[0] == F.RECORD_COLUMNS and len(got) == 3
    assert F.read_record(str(rec))[1]["total_loss"] == rows[1]["total_loss"]


def test_validation_total_decides_the_save(tmp_path):
    ts = FakeStep([4.0, 5.0, None, 3.0])
    saved = []
    rows, best = F.fit(ts, samples(["ok"] * 40), epochs=4, epoch_length=2, val_samples=lambda: ["v1", "v2"], save_weights=lambda p: saved.append(p),
                       weights_path="w", lookahead=3)
    assert ts.validated == 4 and saved == ["w", "w"] and best == 3.0
    assert [r["val_total_loss"] for r in rows] == [4.0, 5.0, None, 3.0]
    assert [r["model_improvement"] for r in rows] == [None, None, None, -1.0]       # inf -> 4 records None; 4 -> 3 records -1


def test_feed_exhaustion_ends_the_run_after_recording_the_partial_epoch():
    ts = FakeStep([])
    rows, best = F.fit(ts, samples(["ok"] * 5), epochs=3, epoch_length=3)
    assert len(rows) == 2 and ts.calls == [0, 1, 2, 3, 4]


def test_resume_reads_config_and_best_recorded_loss(tmp_path):
    from faster_rcnn.config import Config
    C = Config()
    C.weights_path = "models/x/weights.hdf5"
    with open(tmp_path / "config.pickle", "wb") as f:
        pickle.dump(C, f)
    rows = [dict.fromkeys(F.RECORD_COLUMNS), dict.fromkeys(F.RECORD_COLUMNS)]
    rows[0].update(total_loss=2.5, val_total_loss=3.5)
    rows[1].update(total_loss=2.0, val_total_loss=None)
    F.write_record(str(tmp_path / "record.csv"), rows)
    C2, got, best = F.resume_state(str(tmp_path), use_validation=True)
    assert C2.weights_path == C.weights_path and len(got) == 2 and best == 3.5
    assert F.resume_state(str(tmp_path), use_validation=False)[2] == 2.0
    os.remove(tmp_path / "record.csv")
    assert F.resume_state(str(tmp_path), use_validation=True)[2] == float("inf")
    # a continued run: an epoch whose validation total beats the recorded best saves, one that does not leaves the file alone
    ts = FakeStep([3.4, 3.6])
    saved = []
    rows2, best2 = F.fit(ts, samples(["ok"] * 20), epochs=2, epoch_length=2, val_samples=["v"], save_weights=saved.append, weights_path="w", rows=got, best_total_loss=best)
    assert saved == ["w"] and best2 == 3.4 and len(rows2) == 4 and rows2[2]["model_improvement"] == pytest.approx(-0.1)
