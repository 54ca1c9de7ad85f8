"""Tune the GEMM launch shapes IN SITU: against the throughput of the pipelined train step instead of each launch alone.

The engine's autotuner times every problem shape by itself on an idle chip.  In the pipelined step (trainer.TrainStep, four
lanes) a launch shares the CUs with the other lanes' launches, and the shape that is fastest alone is not always the one that
packs best.  This tool starts from the autotuned table and walks it entry by entry: for every measured problem it tries the
neighbouring launch shapes the autotuner itself would consider (other tile, other K-slice count / unit order, 4- or 8-wave
form), re-records the layer programs, times a few hundred pipelined steps and keeps a change only if the step got faster by
more than the noise, twice.  The result is an ordinary tuning table (radnet_tune_save format) for `bench.py --tune-cache` /
`engine.load_tuning`.

usage: python tools/insitu_tune.py <out table> [--passes 1] [--steps 200] [--budget-s 900] [--start <table>] [--per-gpu-batch 1] [--trainable train|cont]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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
    elif abs(s) == 1:
        out.append((a, b, -s, w))                          # batched launch: plain / XCD-contiguous numbering of its workgroups
    out.append((a, b, s, 12 - w))                          # 4 <-> 8 waves
    for ta in (64, 128):
        for tb in (64, 128):
            if (ta, tb) != (a, b) and not (tb > 64 and n <= 64) and not (ta > 64 and m <= 64):
                out.append((ta, tb, s if batched else (1 if abs(s) == 1 else s), w))
    seen, uniq = set(), []
    for o in out:
        if o not in seen and o != (a, b, s, w):
            seen.add(o)
            uniq.append(o)
    return uniq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--passes", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--budget-s", type=float, default=900.0)
    ap.add_argument("--start", default=None, help="table to start from instead of this process's autotuning")
    ap.add_argument("--gain", type=float, default=0.0025, help="relative step-time gain a change must show, twice")
    ap.add_argument("--wide", action="store_true", help="try every K-slice count of a shape, not only the neighbouring ones")
    ap.add_argument("--dp-rehearsal", action="store_true", help="tune the data-parallel step's schedule (1-rank RCCL group, deferred head update)")
    ap.add_argument("--trainable", choices=("train", "cont"), default="train")
    ap.add_argument("--per-gpu-batch", type=int, default=1, help="images per step (2 = BASELINE cfg 4 on one GPU)")
    args = ap.parse_args()
    global WIDE
    WIDE = args.wide
    os.environ["RADNET_SHIPPED_TUNING"] = "0"      # start from what the engine measures itself (or --start), never from a shipped table
    sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]
    import torch
    import bench
    from faster_rcnn.config import Config
    from radnet_hip import make_engine, synth
    from radnet_hip.trainer import TrainStep
    t_begin = time.perf_counter()
    if args.trainable == "cont":           # cont_train.py mode: stages 3-4 train in both models, one lane (bench.py --trainable cont)
        from radnet_hip.engine_cont import ContEngine
        from radnet_hip.trainer_cont import ContTrainStep
        eng = ContEngine(Config())
        eng.set_weights(synth.synthetic_weights(seed=3))
        _ts = ContTrainStep(eng)

        class _Adapter:
            def step(self, batch, upcoming=None):
                return _ts.step(batch, upcoming=upcoming)

            def flush(self):
                pass

        ts = _Adapter()
    else:
        if args.dp_rehearsal:                 # the data-parallel step's schedule on ONE GPU (bench.py RADNET_BENCH_REHEARSAL=nccl1): a 1-rank
            import torch.distributed as dist  # RCCL group, both gradient exchanges issued from their lanes, head update deferred
            from radnet_hip import trainer as _tr
            _tr.FORCE_COLLECTIVES = True
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29535")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        eng = make_engine(Config())
        eng.set_weights(synth.synthetic_weights(seed=3))
        ts = TrainStep(eng, defer_head_update=True if args.dp_rehearsal else None)
    batch = bench.make_batch(0, args.per_gpu_batch, 600, 1000)
    look = getattr(ts, "LOOKAHEAD", 3)
    n_prime = 2 * getattr(ts, "NBUF", 6) + 6

    def run(n):
        for _ in range(n):
            ts.step(batch, upcoming=[batch] * look)

    def measure(n=args.steps):
        run(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    tmp = args.out + ".tmp"

    def apply(tab, header):
        ts.flush()
        torch.cuda.synchronize()
        write_table(tmp, tab, header)
        eng.load_tuning(tmp)
        eng._graphs.clear()                  # programs run eagerly once (unit tables are built), then are recorded again
        run(n_prime)
        torch.cuda.synchronize()

    # the shapes THIS workload launches: what the engine measures on its own when it starts from an empty table (shipped tables
    # may hold other workloads' shapes; a change to a shape that is never launched can only "win" by noise)
    run(n_prime)
    torch.cuda.synchronize()
    eng.save_tuning(tmp)
    tab, header = read_table(tmp)
    if args.start:
        used = set(tab)
        start_tab, _ = read_table(args.start)
        tab.update({k: v for k, v in start_tab.items() if k in used})
        apply(tab, header)
    best = min(measure(), measure())
    print("start: %d entries, step %.1f us (%.1f images/s)" % (len(tab), best, 1e6 / best), flush=True)
    start_us = best
    changed = []
    n_tried = 0
    for p in range(args.passes):
        n_acc = 0
        # longest launches first: ms x (how often is unknown) -- the per-launch time is the proxy
        for key in sorted(tab, key=lambda kk: -tab[kk][3]):
            cur = list(tab[key])
            for cand in neighbours(key, cur):
                if time.perf_counter() - t_begin > args.budget_s:
                    break
                n_tried += 1
                if n_tried % 25 == 0:          # the box drifts (clocks, neighbours on a shared host): refresh the figure to beat
                    apply(tab, header)
                    ref = min(measure(), measure())
                    print("  reference re-measured: %.1f us (was %.1f)" % (ref, best), flush=True)
                    best = ref
                trial = dict(tab)
                trial[key] = [cand[0], cand[1], cand[2], cur[3], cand[3]]
                try:
                    apply(trial, header)
                    t1 = measure()
                    ok = t1 < best * (1.0 - args.gain)
                    t2 = measure() if ok else t1
                    ok = ok and t2 < best * (1.0 - args.gain)
                    if ok:                     # A / B / A: the table without the change, measured again now, must still lose
                        apply(tab, header)
                        ref = min(measure(), measure())
                        ok = max(t1, t2) < ref * (1.0 - args.gain)
                        print("    reference now %.1f us" % ref, flush=True)
                        if not ok:
                            best = ref
                except RuntimeError as e:      # the step's state is unknown after a failed launch: stop with what is kept so far
                    print("  %s -> %s: %s -- stopping" % (key, cand, str(e).splitlines()[0][:160]), flush=True)
                    write_table(args.out, tab, header)
                    raise
                print("  %-44s %s -> %s : %.1f / %.1f us vs %.1f %s" % (key, tuple(cur[:3] + [cur[4]]), cand, t1, t2, best, "KEPT" if ok else ""),
                      flush=True)
                if ok:
                    tab = trial
                    cur = list(tab[key])
                    best = max(t1, t2)
                    n_acc += 1
                    changed.append((key, cand))
                    write_table(args.out, tab, header)
            if time.perf_counter() - t_begin > args.budget_s:
                print("budget reached", flush=True)
                break
        print("pass %d: %d changes kept, step %.1f us" % (p + 1, n_acc, best), flush=True)
        if n_acc == 0:
            break
    apply(tab, header)
    end_us = min(measure(400), measure(400))
    write_table(args.out, tab, header)
    print("in situ: step %.1f -> %.1f us (%.1f -> %.1f images/s), %d entries changed" % (start_us, end_us, 1e6 / start_us, 1e6 / end_us, len(changed)))
    for key, cand in changed:
        print("  changed %s -> tile %dx%d slices %d waves %d" % (key, cand[0], cand[1], cand[2], cand[3]))
    ts.flush()
    os.remove(tmp)


if __name__ == "__main__":
    main()
