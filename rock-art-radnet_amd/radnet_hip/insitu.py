"""Launch shapes tuned IN SITU: against the throughput of the running step instead of each launch alone.

The engine's autotuner (csrc/conv_mfma.hip: run_igemm / run_wgrad) times every problem shape by itself on an idle chip.  In the
pipelined train step (trainer.TrainStep, four lanes) a launch shares the CUs with the other lanes' launches, and the shape that is
fastest alone is not always the one that packs best (+6 % at 1000x600).  `tune()` starts from the table of the shapes
the workload launches and walks it entry by entry: the neighbouring shapes the autotuner itself would consider (other tile, other
K-slice count / unit order, 4- or 8-wave form), layer programs re-recorded, a few hundred steps timed; a change is kept only if the
step got faster by more than the noise twice AND the unchanged table, measured again afterwards, still loses.  Every launch shape
computes the same convolution (sums associate differently, nothing else), so a training job can keep training while this runs; the
result is an ordinary tuning table (radnet_tune_save format) for engine.load_tuning / `bench.py --tune-cache` / radnet_hip/tuned/.

Command line: tools/insitu_tune.py."""
import time

BK = 32
FWD_SLICES = [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16]
WGRAD_SLICES = [1, 2, 3, 4, 6, 8, 12, 16]


def cdiv(a, b):
    return (a + b - 1) // b


def read_table(path):
    tab, header = {}, None
    for line in open(path):
        if line.startswith("#"):
            header = line
            continue
        f = line.split()
        if len(f) < 11:
            continue
        key = tuple(int(v) for v in f[:7])
        tab[key] = [int(f[7]), int(f[8]), int(f[9]), float(f[10]), int(f[11]) if len(f) > 11 else 4]
    return tab, header


def write_table(path, tab, header):
    with open(path, "w") as f:
        f.write(header or "# radnet tuned GEMM launch shapes v2: kind m n k c npos stride | tile_a tile_b slices ms waves\n")
        for key in sorted(tab):
            a, b, s, ms, w = tab[key]
            f.write("%s %d %d %d %.6f %d\n" % (" ".join(str(v) for v in key), a, b, s, ms, w))


WIDE = False      # --wide: every K-slice count (both unit orders) instead of the two or three next to the current one


def neighbours(key, cur):
    """Launch shapes next to `cur` inside the autotuner's own candidate space (conv_mfma.hip: run_igemm / run_wgrad)."""
    kind, m, n, k, c, npos, stride = key
    a, b, s, _, w = cur
    out = []
    if kind == 33:      # radnet_conv_bottleneck's decision: the fused launch with a rows per workgroup (slices 1) or the separate launches (slices 2)
        cands = [(64, 64, 1, 4), (32, 64, 1, 4), (a, b, 2, 4)]
        return [o for o in cands if o != (a, b, s, w) and not (s == 2 and o[2] == 2)]
    if kind == 32:      # radnet_conv_fwd_pair's decision for a pair of shapes: one launch on tile a x b (slices 1) or the two launches (slices 2)
        cands = [(64, 64, 1, 4), (32, 64, 1, 4), (32, 32, 1, 4), (a, b, 2, 4)]
        return [o for o in cands if o != (a, b, s, w) and not (s == 2 and o[2] == 2)]
    wgrad = (kind & 7) in (2, 3)
    batched = kind >= 8
    if wgrad:
        nmt = cdiv(m, BK)
        ok = lambda v: v == 1 or (nmt // v >= 2 and cdiv(nmt, cdiv(nmt, v)) == v)
        if not batched:
            i = WGRAD_SLICES.index(s) if s in WGRAD_SLICES else None
            if i is not None:
                for j in (range(len(WGRAD_SLICES)) if WIDE else (i - 1, i + 1, i + 2)):
                    if 0 <= j < len(WGRAD_SLICES) and ok(WGRAD_SLICES[j]):
                        out.append((a, b, WGRAD_SLICES[j], w))
        if batched and abs(s) == 1:
            out.append((a, b, -s, w))                      # plain / XCD-contiguous numbering of the batch's workgroups
        for ta in (64, 128):
            for tb in (64, 128):
                if (ta, tb) != (a, b) and c % ta == 0 and not (tb > 64 and n <= 64):
                    out.append((ta, tb, s, w))
        return [o for i, o in enumerate(out) if o != (a, b, s, w) and o not in out[:i]]
    nk = cdiv(k, BK)
    if not batched:
        mag = abs(s)
        i = FWD_SLICES.index(mag) if mag in FWD_SLICES else None
        if i is not None:
            for j in (range(len(FWD_SLICES)) if WIDE else (i - 1, i + 1, i + 2)):
                if 0 <= j < len(FWD_SLICES):
                    v = FWD_SLICES[j]
                    if v == 1 or nk // v >= 2:
                        out.append((a, b, v if s > 0 else -v, w))
                        if WIDE and cdiv(m, a) * cdiv(n, b) * v >= 16:
                            out.append((a, b, -v if s > 0 else v, w))
        tiles = cdiv(m, a) * cdiv(n, b)
        if tiles * abs(s) >= 16:
            out.append((a, b, -s, w))                      # the other workgroup order (plain / XCD-contiguous)
    else:
        out.append((a, b, -s, w))                          # batched launch: plain / XCD-contiguous numbering of its workgroups
        if w == 4 and (a, b) in ((64, 64), (32, 64), (64, 128), (32, 32)):      # persistent form: z consecutive problems per workgroup
            zs = [1, 2, 3, 4, 6, 9, 12]
            i = zs.index(abs(s)) if abs(s) in zs else 0
            for j in (range(len(zs)) if WIDE else (i - 1, i + 1)):
                if 0 <= j < len(zs) and zs[j] <= stride:
                    out.append((a, b, zs[j] if s > 0 else -zs[j], w))
    if a >= 64 and not (batched and abs(s) > 1):
        out.append((a, b, s, 12 - w))                      # 4 <-> 8 waves (the 32-row tiles and the persistent form exist with 4 waves only)
    small = npos * c == k and c % BK == 0 and cdiv(m, 64) * cdiv(n, 64) * (stride if batched else 1) <= 6 * 256
    for ta, tb in ((64, 64), (64, 128), (128, 64), (128, 128), (32, 64), (32, 32)):
        if (ta, tb) != (a, b) and not (tb > 64 and n <= 64) and not (ta > 64 and m <= 64) and (ta >= 64 or small):
            persist_ok = (ta, tb) in ((64, 64), (32, 64), (64, 128), (32, 32)) and (4 if ta < 64 else w) == 4
            sb = s if (abs(s) == 1 or persist_ok) else (1 if s > 0 else -1)
            out.append((ta, tb, sb if batched else (1 if abs(s) == 1 else s), 4 if ta < 64 else w))
    seen, uniq = set(), []
    for o in out:
        if o not in seen and o != (a, b, s, w):
            seen.add(o)
            uniq.append(o)
    return uniq



class NoComm:
    """Single process: nothing to agree on."""
    rank = 0

    def bcast(self, obj):
        return obj


class DistComm:
    """Ranks of a torch.distributed job walk the table TOGETHER: every step of the workload carries the gradient exchanges, so all
    ranks must run the same number of steps with the same control flow.  Rank 0's table, rank 0's clock and rank 0's verdict on
    every candidate are broadcast (a few bytes per decision); the other ranks apply the same tables -- any launch shape computes
    the same convolution, so the replicas stay consistent -- and end the walk holding rank 0's table."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group
        self.rank = dist.get_rank(group)

    def bcast(self, obj):
        box = [obj]
        self.dist.broadcast_object_list(box, src=0 if self.group is None else self.dist.get_global_rank(self.group, 0), group=self.group)
        return box[0]


def cache_path(network, workload, H, W, per_batch, device_name, cache_dir=None):
    """Where a job's in-situ table lives: one file per (workload, network, panel size, images per step, device)."""
    import os
    import re
    d = cache_dir or os.environ.get("RADNET_TUNE_CACHE_DIR") or os.path.join(os.path.expanduser("~"), ".cache", "radnet_hip", "tuned")
    dev = re.sub(r"[^A-Za-z0-9]+", "-", device_name or "gpu").strip("-")
    return os.path.join(d, "%s_%s_%dx%d_batch%d_%s.txt" % (workload, network, H, W, per_batch, dev))


def tune(eng, run, flush, out, passes=1, steps=200, budget_s=900.0, start=None, gain=0.0025, wide=False, n_prime=18, log=print,
         measure=None, comm=None, sync=None, only_new=False, kinds=None):
    """Walk `eng`'s launch-shape table against the throughput of the workload.
    measure(n) -> microseconds per step over n steps (default: run(20) untimed, then run(n) between two device synchronisations);
    comm: NoComm / DistComm (multi-rank jobs: rank 0 decides, see DistComm); sync(): device synchronisation (default torch's).
    only_new: walk only the shapes the engine measures during the first n_prime steps of THIS call (a job whose engine already
    holds shipped tables for other workloads: entries the job never launches are left alone).
    kinds: walk only entries of these kinds (first field of the key), e.g. (33,) after a new fused launch has been added.

    run(n): enqueue n steps of the workload (the caller's loop over TrainStep.step with its lookahead); flush(): drain it.
    The engine must come without tables of its own (RADNET_SHIPPED_TUNING=0) and must not have run the workload yet: the shapes
    it measures itself during the first n_prime steps are the shapes that are walked -- a change to a shape the workload never
    launches can only "win" by noise.  start: path of a table whose entries for those shapes replace the measured ones first.
    out: the table is written there after every kept change.  Returns (us_per_step_before, us_per_step_after, changes)."""
    global WIDE
    WIDE = bool(wide)
    comm = comm or NoComm()
    root = comm.rank == 0
    if sync is None:
        import torch
        sync = torch.cuda.synchronize
    t_begin = time.perf_counter()
    tmp = "%s.tmp%d" % (out, comm.rank)
    own_measure = measure

    def measure(n=steps):
        if own_measure is not None:
            return comm.bcast(own_measure(n))
        run(20)
        sync()
        t0 = time.perf_counter()
        run(n)
        sync()
        return comm.bcast((time.perf_counter() - t0) / n * 1e6)      # rank 0's clock decides for everybody

    def apply(tab, header):
        flush()
        sync()
        write_table(tmp, tab, header)
        eng.load_tuning(tmp)
        eng._graphs.clear()                  # programs run eagerly once (unit tables are built), then are recorded again
        run(n_prime)
        sync()

    if not root:
        log = lambda *a: None
    known = set()
    if only_new:
        eng.save_tuning(tmp)
        known = set(read_table(tmp)[0])
    run(n_prime)
    sync()
    eng.save_tuning(tmp)
    tab, header = read_table(tmp)
    walk = comm.bcast(sorted(k for k in tab if k not in known)) if only_new else None
    if start:
        used = set(tab)
        start_tab, _ = read_table(start)
        tab.update({k: v for k, v in start_tab.items() if k in used})
    tab, header = comm.bcast((tab, header))  # the walk starts from rank 0's measured shapes on every rank
    if start or not isinstance(comm, NoComm):   # (every rank of a job re-applies, also rank 0: apply() runs steps, and steps carry collectives)
        apply(tab, header)
    best = min(measure(), measure())
    log("start: %d entries, step %.1f us (%.1f steps/s)" % (len(tab), best, 1e6 / best))
    start_us = best
    changed = []
    n_tried = 0
    out_of_time = lambda: comm.bcast(time.perf_counter() - t_begin > budget_s)
    for p in range(passes):
        n_acc = 0
        # longest launches first: ms x (how often is unknown) -- the per-launch time is the proxy
        for key in sorted(tab, key=lambda kk: -tab[kk][3]):
            if (walk is not None and key not in walk) or (kinds is not None and key[0] not in kinds):
                continue
            cur = list(tab[key])
            for cand in neighbours(key, cur):
                if out_of_time():
                    break
                n_tried += 1
                if n_tried % 25 == 0:          # the box drifts (clocks, neighbours on a shared host): refresh the figure to beat
                    apply(tab, header)
                    ref = min(measure(), measure())
                    log("  reference re-measured: %.1f us (was %.1f)" % (ref, best))
                    best = ref
                trial = dict(tab)
                trial[key] = [cand[0], cand[1], cand[2], cur[3], cand[3]]
                try:
                    apply(trial, header)
                    t1 = measure()
                    ok = t1 < best * (1.0 - gain)
                    t2 = measure() if ok else t1
                    ok = ok and t2 < best * (1.0 - gain)
                    if ok:                     # A / B / A: the table without the change, measured again now, must still lose
                        apply(tab, header)
                        ref = min(measure(), measure())
                        ok = max(t1, t2) < ref * (1.0 - gain)
                        log("    reference now %.1f us" % ref)
                        if not ok:
                            best = ref
                except RuntimeError as e:      # the step's state is unknown after a failed launch: stop with what is kept so far
                    log("  %s -> %s: %s -- stopping" % (key, cand, str(e).splitlines()[0][:160]))
                    if root:
                        write_table(out, tab, header)
                    raise
                log("  %-44s %s -> %s : %.1f / %.1f us vs %.1f %s" % (key, tuple(cur[:3] + [cur[4]]), cand, t1, t2, best, "KEPT" if ok else ""))
                if ok:
                    if key[0] in (8, 18):      # the 36 / 16 GEMMs of a Winograd 3x3 layer: the north-star's per-layer table times them ALONE
                        log("    note: %s is a Winograd layer's batched GEMM; this change was accepted for the step's throughput and may cost "
                            "the layer's isolated figure -- re-read roofline.layers_3x3 of the next bench line" % (key,))
                    tab = trial
                    cur = list(tab[key])
                    best = max(t1, t2)
                    n_acc += 1
                    changed.append((key, cand))
                    if root:
                        write_table(out, tab, header)
            if out_of_time():
                log("budget reached")
                break
        log("pass %d: %d changes kept, step %.1f us" % (p + 1, n_acc, best))
        if n_acc == 0 or out_of_time():
            break
    apply(tab, header)
    end_us = min(measure(2 * steps), measure(2 * steps))
    if root:
        write_table(out, tab, header)
    import os
    os.remove(tmp)
    return start_us, end_us, changed


def tune_job(ts, next_batch, out, steps_done=None, lookahead=3, budget_s=120.0, steps=100, passes=1, comm=None, log=None, measure=None, sync=None):
    """In-situ tuning as part of a RUNNING training job (run_training(..., tune=True) / RADNET_INSITU=1): the job's own steps are the
    measurement.  next_batch() hands out the job's next batch (the samples train the model while the walk runs -- any launch shape
    computes the same step); steps_done(n) is told how many steps each call consumed.  Bounded by `budget_s` seconds; the table is
    written to `out` by rank 0 after every kept change and loaded by every rank when the walk ends.  Returns (us_before, us_after,
    number of changes)."""
    eng = ts.eng
    window = []

    def run(n):
        for _ in range(n):
            while len(window) < lookahead + 1:
                window.append(next_batch())
            batch = window.pop(0)
            ts.step(batch, upcoming=window[:lookahead] if lookahead else None)
        if steps_done is not None:
            steps_done(n)

    before, after, changed = tune(eng, run, ts.flush, out, passes=passes, steps=steps, budget_s=budget_s, n_prime=2 * getattr(ts, "NBUF", 1) + 2,
                                  log=log or (lambda *a: None), comm=comm, measure=measure, sync=sync, only_new=True)
    ts.flush()
    return before, after, len(changed), window
