"""The epoch driver around the training step: train.py:273-642 (epochs of EPOCH_LENGTH iterations, per-epoch means, validation,
save-on-best, record.csv) and cont_train.py:112-206 (resume from config.pickle + weights + the best loss recorded so far).

    fit(ts, feed, epochs, epoch_length, val_samples=..., save_weights=model_all.save_weights, weights_path=C.weights_path,
        record_path=os.path.join(model_dir, "record.csv"))

`ts` is a radnet_hip.trainer.TrainStep (or anything with its step / flush / validate / start_loss_log / read_loss_log /
last_n_pos surface: the CPU tests drive this loop with a recording stand-in), `feed` an iterator of samples
(faster_rcnn.data_feed.TileFeed), `save_weights(path)` the reference's model_all.save_weights.

What is the reference's and is kept:
  * an ITERATION is a sample whose classifier step ran: a sample whose proposals overlap no box appends 0 to the epoch's
    "overlapping boxes" list and is not counted (train.py:378-380 `continue`s before iter_num += 1); a sample the labeller
    dropped never reaches the loop (utils.py:461-465);
  * epoch figures are plain means over the epoch's iterations (train.py:441-452); the total is the sum of the four loss means;
  * with validation the weights are saved when the VALIDATION total improves, without it when the training total does
    (train.py:565-599), `model_improvement` = new - old best (None otherwise);
  * record.csv: the reference's 16 columns in its order, values rounded to 3 decimals (train.py:214-233, 632-642);
  * resume: the best loss so far is the minimum of the recorded column (cont_train.py:203-206); optimizer state is not kept.
What is this build's: the five losses of an epoch's iterations are read back from the device ONCE per epoch
(TrainStep.start_loss_log) instead of per iteration, and `lookahead` batches are announced ahead (the pipelined step)."""
import csv
import math
import os
import time

import numpy as np

RECORD_COLUMNS = ["elapsed_time", "mean_overlapping_bboxes", "val_mean_overlapping_bboxes", "loss_rpn_cls", "val_loss_rpn_cls",
                  "loss_rpn_regr", "val_loss_rpn_regr", "loss_detector_cls", "val_loss_detector_cls", "loss_detector_regr",
                  "val_loss_detector_regr", "total_loss", "val_total_loss", "detector_acc", "val_detector_acc", "model_improvement"]


def read_record(path):
    """record.csv -> list of dict rows (floats; empty cells -> None)."""
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append({k: (float(v) if v not in ("", "None", "nan", None) else None) for k, v in r.items()})
    return rows


def write_record(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(RECORD_COLUMNS)
        for r in rows:
            w.writerow(["" if r.get(c) is None else r[c] for c in RECORD_COLUMNS])


def best_recorded_loss(rows, use_validation):
    """cont_train.py:203-206: df_record['val_total_loss' | 'total_loss'].min() (inf when nothing is recorded)."""
    col = "val_total_loss" if use_validation else "total_loss"
    vals = [r[col] for r in rows if r.get(col) is not None and not math.isnan(r[col])]
    return min(vals) if vals else float("inf")


def resume_state(model_dir, use_validation):
    """cont_train.py:114-119,203-206: (Config from config.pickle -- only the attribute bag is unpickled --, rows recorded so
    far, best total loss so far).  The caller builds its models for C and calls model.load_weights(C.weights_path, by_name=True)
    (cont_train.py:155,164) before fit(..., rows=rows, best_total_loss=best)."""
    from faster_rcnn.RADNet import _ConfigUnpickler
    with open(os.path.join(model_dir, "config.pickle"), "rb") as f:
        C = _ConfigUnpickler(f).load()
    rec = os.path.join(model_dir, "record.csv")
    rows = read_record(rec) if os.path.exists(rec) else []
    return C, rows, best_recorded_loss(rows, use_validation)


def _r3(v):
    return None if v is None else round(float(v), 3)


def fit(ts, feed, epochs, epoch_length, val_samples=None, save_weights=None, weights_path=None, record_path=None, rows=None,
        best_total_loss=None, lookahead=3, log=None, on_epoch=None):
    """Run `epochs` epochs of `epoch_length` iterations.  val_samples: None (no validation: the training total decides),
    a list of samples, or a callable returning a fresh iterable per epoch (the reference builds a new validation generator
    every epoch, train.py:482).  Returns (rows, best_total_loss); stops early (after writing what it has) when the feed ends."""
    log = log or (lambda *a: None)
    rows = list(rows or [])
    best = float("inf") if best_total_loss is None else float(best_total_loss)
    use_val = val_samples is not None
    it = iter(feed)
    window, exhausted = [], False
    start_time = time.time()

    def fill():
        nonlocal exhausted
        while not exhausted and len(window) < lookahead + 1:
            try:
                s = next(it)
            except (StopIteration, RuntimeError) as e:          # PEP 479 turns the reference generators' `raise StopIteration` into RuntimeError
                if isinstance(e, RuntimeError) and "StopIteration" not in str(e):
                    raise
                exhausted = True
                break
            window.append(s if isinstance(s, list) else [s])

    for epoch in range(epochs):
        per_batch = max((len(b) for b in window), default=1)
        ts.start_loss_log(epoch_length + 8 * per_batch)
        n_iter, overlapping = 0, []
        while n_iter < epoch_length:
            fill()
            if not window:
                break
            batch = window.pop(0)
            fill()
            ts.step(batch, upcoming=window[:lookahead] if lookahead else None)
            for n_pos in ts.last_n_pos:
                if n_pos is None:                                # dropped by the labeller: the reference's loop never saw it
                    continue
                overlapping.append(n_pos)                        # train.py:378 (0) / 385-386
            n_iter += sum(1 for took in _took_head(ts) if took)
        ts.flush()
        if n_iter == 0:
            break
        L = np.asarray(ts.read_loss_log(), dtype=np.float64)[:n_iter]
        m = L.mean(0)
        rec = {"loss_rpn_cls": m[0], "loss_rpn_regr": m[1], "loss_detector_cls": m[2], "loss_detector_regr": m[3], "detector_acc": m[4]}
        rec["mean_overlapping_bboxes"] = float(sum(overlapping)) / max(len(overlapping), 1)
        rec["total_loss"] = m[0] + m[1] + m[2] + m[3]
        rec["elapsed_time"] = (time.time() - start_time) / 60.0
        improved_from = None
        if use_val:
            v = ts.validate(val_samples() if callable(val_samples) else val_samples)
            if v.get("n", 0) > 0:
                rec.update(val_mean_overlapping_bboxes=v["mean_overlapping_bboxes"], val_detector_acc=v["det_acc"], val_loss_rpn_cls=v["rpn_cls"],
                           val_loss_rpn_regr=v["rpn_regr"], val_loss_detector_cls=v["det_cls"], val_loss_detector_regr=v["det_regr"],
                           val_total_loss=v["total"])
                decide = v["total"]
            else:                                                # the reference divides by len([]) here (train.py:541): nothing to compare
                decide = float("inf")
        else:
            decide = rec["total_loss"]
        if decide < best:
            improved_from = best
            rec["model_improvement"] = decide - best if math.isfinite(best) else None
            log("Total loss decreased from %s to %s, saving weights" % (best, decide))
            best = decide
            if save_weights is not None:
                save_weights(weights_path) if weights_path is not None else save_weights()
        row = {c: _r3(rec.get(c)) for c in RECORD_COLUMNS}
        rows.append(row)
        if record_path is not None:
            write_record(record_path, rows)
        if on_epoch is not None:
            on_epoch(epoch, row, improved_from is not None)
        if n_iter < epoch_length:                                # the feed ended inside this epoch
            break
    return rows, best


def _took_head(ts):
    """Per image of the last step: did its classifier step run?  (n_pos is the number of POSITIVE sampled RoIs and may be 0 for a
    step that ran on background RoIs only, so the trainer's own record decides.)"""
    took = getattr(ts, "last_took_head", None)
    if took is not None:
        return took
    return [n is not None and n > 0 for n in ts.last_n_pos]
