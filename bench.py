#!/usr/bin/env python3
"""bench.py -- train images/sec, ResNet50 Faster R-CNN on synthetic 1000x600 panels (BASELINE.json metric).

One step = one reference training iteration (train.py:278-402) per image: anchor targets, base forward,
RPN train (fwd+loss+bwd+Adam#1), re-predict, proposals (decode+NMS), RoI labelling + sampling, classifier-head
train (fwd+loss+bwd+Adam#2) -- nothing skipped (a step whose head phase is skipped aborts the run).
Inputs (uint8 panels, GT boxes) are resident before the timed region; the panel is uploaded once per step
from pinned host memory exactly as the reference feeds its model (the PCIe copy is inside the step).

  python bench.py --gpus 1 --steps 300 --warmup 30
  python bench.py --gpus N ...          (no launcher: starts its N ranks itself, one per GPU -- radnet_hip/launch.py)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "rock-art-radnet_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# The step keeps 4 HIP streams busy (main, two prefetch lanes, head lane) and RCCL adds one per communicator; HIP multiplexes
# streams over 4 hardware queues by default, and two busy streams that land on one queue serialise (measured: the
# pipelined step 385 instead of 441 images/s once RCCL's streams exist).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
ALGO_GFLOP_PER_IMAGE = 226.4       # minimal train step, base frozen (train.py semantics)
ALGO_GFLOP_CONT = 537.0            # cont_train.py variant (stages 3-4 trainable in both models)


def make_batch(rank, per_gpu, H, W):
    from radnet_hip import synth
    # Every rank runs the workload BASELINE.md 3 names (panel seed 1, GT seed 2 for its first image): weak scaling on the
    # quoted configuration.  Ranks differ in their NumPy RNG seed (anchor / RoI sampling), not in the image: with per-rank
    # random boxes some ranks' batches sit at the edge of the reference's "no RoI overlaps a box -> skip the classifier step"
    # path (tools/skip_probe.py), and a step with work skipped is not a measurement of the named workload.
    batch = []
    for i in range(per_gpu):
        off = i
        meta = synth.synthetic_gt(2 + off, n=8, src_w=2 * W, src_h=2 * H)
        batch.append(dict(img=synth.synthetic_panel(1 + off, H, W), bboxes=meta["bboxes"], width=2 * W, height=2 * H))
    return batch


def cpu_baseline(budget_s=25.0, workload="train"):
    """The oracle (CPU restatement, kind 'port') timed on this box's host cores on a bounded sample of the same
    workload: whole 1000x600 training iterations until ~budget_s seconds are spent (at least 1)."""
    import copy
    from faster_rcnn.config import Config
    from oracle import step as ostep
    from radnet_hip import synth
    C = Config()
    if workload == "vgg16":
        C.network = "vgg16"
        C.anchor_box_scales = [128, 256, 512]
        ot = ostep.OracleTrainerVGG(C, copy.deepcopy(synth.synthetic_weights_vgg16(seed=3, n_anchors=9, n_classes=len(C.class_mapping))))
    elif workload == "cont":
        ot = ostep.OracleTrainerCont(C, copy.deepcopy(synth.synthetic_weights(seed=3)))
    else:
        ot = ostep.OracleTrainer(C, copy.deepcopy(synth.synthetic_weights(seed=3)))
    batch = make_batch(0, 1, 600, 1000)
    np.random.seed(64)
    n, t0 = 0, time.perf_counter()
    while True:
        ot.step(batch[0])
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 8:
            break
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    return {"value": n / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d full train iterations (%s, 1000x600, 8 GT, NumPy/BLAS oracle, fp32) in %.1f s" % (n, type(ot).__name__, dt)}


class _Ptr:
    """int device address with the tensor method the engine's descriptor builder calls."""

    def __init__(self, p):
        self.p = p

    def data_ptr(self):
        return self.p


def _time_us(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def _time_graph_us(eng, ops, reps=8, n=20):
    """Per-pass time of a launch list replayed the way the step launches it: `reps` passes recorded into ONE hipGraph (engine._run:
    eager first, captured on the second call), replayed n times between two events on the launch stream -- kernels back to back with
    no host launch latency between them (a ctypes call per kernel costs 5-10 us of host time, more than some of these kernels last)."""
    prog = list(ops) * reps
    eng._run(prog)
    eng._run(prog)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        eng._run(prog)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


def committed_pmc_traffic():
    """HBM bytes per launch of the dominant GEMM kernel from the committed PMC passes of this same command
    (profiles/r04_pmc_summary.txt, else the earlier rounds': FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc runs by
    tools/collect_profiles_r04.sh, FETCH_SIZE doubled as the gfx950 correction of MI355X_MICROARCH.md prescribes).  A profiler
    cannot run inside this process: the figure is the last committed measurement, or None when the file is not there."""
    import re
    # the dominant GEMM kernel by time per step: round 4 conv_igemm_kernel<32, 64, 0, false, 4> (17 launches, 485 us of 2 545 on one lane,
    # profiles/r04_trace_summary_one_lane.txt); earlier rounds' summaries know the 64x64 class only
    for name, kern in (("r04_pmc_summary.txt", r"conv_igemm_kernel<32, 64, 0, false, 4>"), ("r03_pmc_summary_insitu_tables.txt", r"conv_igemm_kernel<64, 64, 0, false, 4>"),
                       ("r03_pmc_summary.txt", r"conv_igemm_kernel<64, 64, 0, false, 4>"), ("r02_pmc_summary.txt", r"conv_igemm_kernel<64, 64, 0, false, 4>")):
        path = os.path.join(ROOT, "profiles", name)
        try:
            for line in open(path):
                m = re.search(r"^\s*(" + re.escape(kern) + r")\s+per launch: FETCH_SIZE\s+([0-9.]+) MB \(x2 =\s+([0-9.]+) MB\)\s+WRITE_SIZE\s+([0-9.]+) MB", line)
                if m:
                    return {"kernel": m.group(1), "bytes": (float(m.group(3)) + float(m.group(4))) * 1e6,
                            "fetch_bytes_corrected": float(m.group(3)) * 1e6, "write_bytes": float(m.group(4)) * 1e6,
                            "source": "profiles/%s (separate --pmc passes of this command; not measured in this run)" % name}
        except OSError:
            pass
    return None


def winograd_executed_share(eng):
    """Executed / algorithmic flops over the Winograd ops of every layer program the engine holds (forward, re-forward, weight
    gradient): 2*positions*tiles*C*N against 2*M*N*9C."""
    ex = al = 0.0
    import collections
    for key in list(eng._plans.keys()):
        plan = collections.OrderedDict.__getitem__(eng._plans, key)       # (the cache's own lookup reorders it)
        for name in ("ops", "fwd", "refwd", "bwd"):
            for kind, p in (plan.get(name) or []) if isinstance(plan, dict) else []:
                if kind in ("wino", "wino_reuse"):
                    nb, hh, ww, c, n, T, form = p[1], p[2], p[3], p[4], p[5], p[9], p[15]
                elif kind == "wino_wgrad":
                    nb, hh, ww, c, n, T, form = p[1], p[2], p[3], p[4], p[5], p[10], p[13]
                else:
                    continue
                ex += 2.0 * (form + 2) ** 2 * T * c * n
                al += 2.0 * nb * hh * ww * n * 9 * c
    return ex / al if al else 1.0


def layers_3x3_table(eng, bp, rp):
    """north_star target "MFMA utilisation on ResNet50 stage-3/4 3x3 convs" (+ rpn_conv1), per layer class, each form alone on
    the chip, back-to-back launches between two HIP events on the launch stream: the Winograd form the step runs (F(4x4,3x3) or
    F(2x2,3x3): input transform + 36 / 16 batched GEMMs + output transform with BN/ReLU) and the direct implicit GEMM.
    `executed` = flops the MFMAs really perform (Winograd: 2*positions*tiles*C*N), `algorithmic` = 2*M*N*9C; fractions are of 157.3 TFLOP/s."""
    import ctypes as C
    lib = eng.lib
    rows = []
    wanted = {"res3b_branch2b": "stage-3 3x3 128->128 (res3a-d_branch2b)", "res4b_branch2b": "stage-4 3x3 256->256 (res4a-f_branch2b)",
              "rpn_conv1": "rpn_conv1 3x3 1024->512"}
    found = {}
    for kind, p in list(bp["ops"]) + list(rp["fwd"]):
        if kind != "wino":
            continue
        for name in wanted:
            c = eng.convs[name]
            if c.wino_u is not None and p[7] == c.wino_u.data_ptr():
                found[name] = p
    for name, label in wanted.items():
        if name not in found:
            continue
        c = eng.convs[name]
        x, nb, hh, ww, cin, n, V, U, M, T, scale, shift, act, y, ldy, form = found[name]
        h = eng.ctx.h
        P = (form + 2) ** 2
        w_in, w_out = (lib.radnet_winograd4_input, lib.radnet_winograd4_output) if form == 4 else (lib.radnet_winograd_input, lib.radnet_winograd_output)
        t_in = _time_us(lambda: w_in(h, x, nb, hh, ww, cin, V))
        gemm = lib.radnet_gemm_batched
        t_g = _time_us(lambda: gemm(h, V, U, M, P, T, n, cin))
        t_out = _time_us(lambda: w_out(h, M, nb, hh, ww, n, scale, shift, act, y, ldy))

        def layer():
            w_in(h, x, nb, hh, ww, cin, V)
            gemm(h, V, U, M, P, T, n, cin)
            w_out(h, M, nb, hh, ww, n, scale, shift, act, y, ldy)
        t_layer_stream = _time_us(layer)
        d, _, _ = eng._desc(c, _Ptr(x), nb, hh, ww, _Ptr(y), relu=bool(act))
        t_dir_stream = _time_us(lambda: lib.radnet_conv_fwd(h, C.byref(d)))
        # the figures the fractions use: the layer as the step launches it (hipGraph replay, kernels back to back)
        try:
            t_layer = _time_graph_us(eng, [("wino", found[name])])
            t_dir = _time_graph_us(eng, [("conv", d)])
        except Exception:                                  # graphs switched off / capture failed: the stream-launch figures
            t_layer, t_dir = t_layer_stream, t_dir_stream
        # ... and as it runs INSIDE the step's layer program: the frozen base forward (or the RPN forward) replayed as the step
        # replays it, with and without the ops of this layer class (every op of the class removed: results of the stripped program
        # are garbage, its timing is not) -- the difference per removed layer is what the layer costs where it runs, between its
        # real neighbours, on buffers written once and read once.  The isolated loop above re-runs ONE layer on ONE set of
        # buffers eight times in a row; for the stage-3 layers (32 MB of operands and results, the size of the L2s) it reads
        # 5 us more than the step's own trace shows for the same three kernels (profiles/r04_kernel_sequence_one_lane.txt).
        t_prog = None
        try:
            prog = list(bp["ops"]) if name != "rpn_conv1" else list(rp["fwd"])
            cls = {"res3b_branch2b": ["res3%s_branch2b" % b for b in "abcd"], "res4b_branch2b": ["res4%s_branch2b" % b for b in "abcdef"],
                   "rpn_conv1": ["rpn_conv1"]}[name]
            us = {eng.convs[k].wino_u.data_ptr() for k in cls if eng.convs[k].wino_u is not None}
            same = lambda op: op[0] == "wino" and op[1][7] in us and (op[1][1], op[1][2], op[1][3], op[1][4], op[1][5]) == (nb, hh, ww, cin, n)
            stripped = [op for op in prog if not same(op)]
            n_removed = len(prog) - len(stripped)
            if n_removed and not any(k == "chain" for k, _ in prog):
                t_full = min(_time_graph_us(eng, prog, reps=1, n=30), _time_graph_us(eng, prog, reps=1, n=30))
                t_strip = min(_time_graph_us(eng, stripped, reps=1, n=30), _time_graph_us(eng, stripped, reps=1, n=30))
                t_prog = (t_full - t_strip) / n_removed
        except Exception:
            t_prog = None
        algo = 2.0 * nb * hh * ww * n * 9 * cin
        execd = 2.0 * P * T * cin * n
        t_frac = t_prog if (t_prog is not None and t_prog > 0) else t_layer
        rows.append({"layer": label, "M": nb * hh * ww, "N": n, "K": 9 * cin, "tiles": T, "winograd_form": "F(%dx%d,3x3)" % (form, form),
                     "winograd_us": {"input_transform": t_in, "gemm_batched": t_g, "gemms": P, "output_transform": t_out, "layer_back_to_back": t_layer,
                                     "layer_stream_launches": t_layer_stream, "layer_in_program": t_prog,
                                     "how": "layer_in_program: (the layer program that holds the layer, replayed as a hipGraph) minus (the same program "
                                            "without this layer class), per removed layer -- the layer's cost where the step runs it; "
                                            "layer_back_to_back / direct_us: ONE layer 8 times in one hipGraph on one set of buffers; "
                                            "the three parts and layer_stream_launches: one host launch per kernel, 30 in a row"},
                     "direct_us": t_dir, "direct_stream_launches_us": t_dir_stream, "algorithmic_gflop": algo / 1e9, "winograd_executed_gflop": execd / 1e9,
                     "winograd_gemm_executed_tflops": execd / t_g / 1e6, "winograd_gemm_executed_frac": execd / t_g / 1e6 / PEAK_FP32_MFMA_TFLOPS,
                     "winograd_layer_algorithmic_tflops": algo / t_frac / 1e6, "winograd_layer_algorithmic_frac": algo / t_frac / 1e6 / PEAK_FP32_MFMA_TFLOPS,
                     "winograd_layer_algorithmic_frac_basis": ("layer_in_program" if t_frac is t_prog else "layer_back_to_back"),
                     "winograd_layer_algorithmic_frac_isolated_loop": algo / t_layer / 1e6 / PEAK_FP32_MFMA_TFLOPS,
                     "direct_tflops": algo / t_dir / 1e6, "direct_frac": algo / t_dir / 1e6 / PEAK_FP32_MFMA_TFLOPS})
    return rows


def collectives_leg(eng, ts, dist, world, n=20):
    """Data-parallel exchanges alone on the chip (after the timed region): bytes and time of the two gradient all-reduces of
    a step, as the step issues them (AR#1 whole RPN arena on its communicator; AR#2 = three kernel-gradient buckets + tail
    on the head communicator)."""
    if dist is None:
        return None
    if getattr(ts, "native_comm", False):
        # round 4: the library's own RCCL binding (radnet_allreduce_grads): AR#1 in line on the main lane, AR#2 one call on the head
        # communicator's stream
        import contextlib
        from radnet_hip import native as N
        out = {"per_step": "AR#1 rpn arena (1 ncclAllReduce on the main lane) + AR#2 head arena (1 ncclAllReduce on its own stream, deferred update)",
               "path": "radnet_allreduce_grads (RCCL bound by the library; torch.distributed only for the rendezvous)", "world": world}
        for name, arena, ctx, stream in (("ar1_rpn", eng.rpn_arena, ts._main_ctx, None), ("ar2_head", eng.head_arena, ts._comm_ctx, ts._comm_stream)):
            buf = torch.zeros_like(arena.g)
            with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                for _ in range(3):
                    N.allreduce(eng, buf, ctx=ctx)
                torch.cuda.synchronize()
                dist.barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    N.allreduce(eng, buf, ctx=ctx)
                e1.record()
                e1.synchronize()
            out[name] = {"bytes": int(arena.n * 4), "ms_alone": e0.elapsed_time(e1) / n}
        return out
    out = {"per_step": "AR#1 rpn arena (1 call) + AR#2 head arena (3 block buckets + tail)", "path": "torch.distributed", "world": world}
    for name, arena, group in (("ar1_rpn", eng.rpn_arena, ts.group), ("ar2_head", eng.head_arena, ts.group_head)):
        buf = torch.zeros_like(arena.g)
        for _ in range(3):
            dist.all_reduce(buf, group=group)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            dist.all_reduce(buf, group=group)
        e1.record()
        e1.synchronize()
        out[name] = {"bytes": int(arena.n * 4), "ms_alone": e0.elapsed_time(e1) / n}
    return out


def _host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count()


def bench_predict(args):
    """BASELINE cfg 3 as a bench line: RADNet's tile path (predict.py:56-122 -> RADNet.py:502-718) over synthetic 2048x2048 tiles.
    A step = one tile: upload -> device bicubic resize to short side img_size -> preprocess -> base -> RPN -> decode / sort / NMS
    (<= 300 proposals) -> RoI crop-resize -> classifier on ALL proposals in one pass -> decode + per-class NMS on the host.  The
    tiles are uint8 arrays in host memory, as the reference's tiler hands them over (the upload is inside the step)."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    from faster_rcnn import models as M
    from faster_rcnn.RADNet import RADNet
    from faster_rcnn.base_models import resnet50
    from faster_rcnn.config import Config
    from radnet_hip import synth
    C = Config()
    C.img_size = args.img_size
    W = synth.synthetic_weights(seed=3)
    m_rpn, m_cls, m_all, m_rpn3, m_det = M.build_models(C, weights=W, workload="predict")
    net = RADNet(C, m_rpn3, m_det, resnet50.preprocess)
    eng = m_rpn3._s.eng
    if args.tune_cache is not None and os.path.exists(args.tune_cache):
        eng.load_tuning(args.tune_cache)
    tiles = [np.random.RandomState(4 + i).randint(0, 256, (2048, 2048, 3)).astype(np.uint8) for i in range(8)]      # tile seed 4
    seq = lambda n: [tiles[i % len(tiles)] for i in range(n)]
    net.device_resident = True
    net._detect_all(seq(4))                    # plans, launch shapes, graphs ("compile")
    net._detect_all(seq(4))
    torch.cuda.synchronize()
    import gc
    gc.collect()
    gc.freeze()
    if args.tune_cache is not None and not os.path.exists(args.tune_cache):
        eng.save_tuning(args.tune_cache)
    net._detect_all(seq(max(args.warmup, 2)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dets = net._detect_all(seq(args.steps))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_det = sum(len(v[0]) for d in dets[:len(tiles)] for v in d.values())
    # ---- roofline leg: the same tiles one at a time on ONE lane, GEMM launches timed from their own dispatch
    roof = None
    if args.roofline_steps > 0:
        for t in tiles[:2]:
            net._detect(t)
        torch.cuda.synchronize()
        eng.ctx.timing(True)
        eng.ctx.timing_reset()
        for k in range(args.roofline_steps):
            net._detect(tiles[k % len(tiles)])
        torch.cuda.synchronize()
        per, tot_ms, tot_fl = {}, 0.0, 0.0
        for cls, name in ((0, "conv_igemm_fwd"), (3, "conv3x3_winograd_layers")):
            ms, n, fl = eng.ctx.timing_read(cls)
            if n:
                per[name] = {"launches_per_step": n / args.roofline_steps, "avg_us": 1e3 * ms / n, "tflops": fl / max(ms, 1e-9) / 1e9}
                tot_ms += ms
                tot_fl += fl
        wms, wn, wfl = eng.ctx.timing_read(3)
        eng.ctx.timing(False)
        exec_fl = tot_fl - (wfl * (1.0 - winograd_executed_share(eng)) if wn else 0.0)
        ach = tot_fl / max(tot_ms, 1e-9) / 1e9
        roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                "kernel": "conv_igemm_kernel (fp32 v_mfma_f32_32x32x2_f32; stride-1 3x3 layers of stages 3-5 and rpn_conv1 via Winograd F(4x4,3x3))",
                "schedule": "one lane, one tile at a time, launches isolated (hipExtLaunchKernelGGL start / stop events; Winograd layers: two marker "
                            "events around their three kernels); `value` is measured with two tiles in flight",
                "executed_tflops": exec_fl / max(tot_ms, 1e-9) / 1e9, "executed_frac": exec_fl / max(tot_ms, 1e-9) / 1e9 / PEAK_FP32_MFMA_TFLOPS,
                "gemm_ms_per_tile": tot_ms / args.roofline_steps, "gemm_gflop_per_tile": tot_fl / args.roofline_steps / 1e9, "by_kernel": per}
    fsz = "%dx%d" % (args.img_size, args.img_size)
    algo = {600: 58.95 + 15 * 29.29, 1000: 162.89 + 15 * 29.29}.get(args.img_size)      # RPN pass + 300 RoIs through the head
    value = args.steps / elapsed
    out = {"metric": "predict tiles/sec, ResNet50 Faster R-CNN, 2048x2048 tiles", "value": value, "unit": "tiles/sec", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "predict.py tile path (BASELINE cfg 3): 2048x2048 uint8 tile -> %s -> RPN -> NMS (300 proposals) -> RoI crop-resize -> "
                                  "classifier on all proposals in one pass (GEMM M = 14 700) -> decode + per-class NMS; two tiles in flight" % fsz,
                      "bench_workload": "predict", "img_size": args.img_size, "anchors": eng.A, "proposals_per_tile": 300, "detections_on_the_8_tiles": n_det,
                      "launch_shapes": "measured per shape on first use" if not getattr(eng, "shipped_tuning", None) else ", ".join(eng.shipped_tuning),
                      "algorithmic_gflop_per_tile": algo, "step_tflops_algorithmic": (value * algo / 1e3) if algo else None}}
    if roof is not None:
        out["roofline"] = roof
    if not args.no_cpu_baseline:
        # the oracle's tile path on one tile (bounded sample: ~500 GFLOP of NumPy / BLAS convolutions)
        from oracle import dense, glue, resize as oresize, step as ostep
        t0 = time.perf_counter()
        small = oresize.resize_bicubic_u8(tiles[0], args.img_size, args.img_size)
        p, r, F = ostep.rpn_only_forward(W, small)
        R = glue.rpn_to_roi(p, r, C, True, 300, 0.7)
        R[:, 2] -= R[:, 0]
        R[:, 3] -= R[:, 1]
        glue.spp_decode(R, lambda rois: dense.head_forward(W, F, rois[0].astype(np.float32), len(C.class_mapping))[:2], C)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "tiles/sec", "cores": _host_cores(), "kind": "port",
                               "sample": "1 tile (2048x2048 -> %s, RPN + 300 proposals through the classifier in 15 chunks of 20, NumPy/BLAS oracle) in %.1f s" % (fsz, dt)}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~0.4 s of timed work -- a 20-step (70 ms) window is short enough for one host hiccup or a clock ramp
    # after the sync-heavy tuning phase to move the figure by several per cent (tools/variance_probe.py)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--per-gpu-batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=3)
    ap.add_argument("--trainable", choices=("train", "cont"), default="train",
                    help="train: train.py trainability, whole base frozen (the BASELINE metric); cont: cont_train.py, stages 3-4 "
                         "unfrozen in both models (secondary measurement, single GPU)")
    ap.add_argument("--workload", choices=("train", "cont", "vgg16", "predict"), default="train",
                    help="train: BASELINE cfg 2, the headline (default).  cont: the same step with cont_train.py trainability (= --trainable "
                         "cont).  vgg16: BASELINE cfg 5, the train step on the VGG16 base model (9 anchors).  predict: BASELINE cfg 3, "
                         "RADNet's tile path (RPN -> NMS -> RoI crop-resize -> classifier on 300 RoIs) over 2048x2048 tiles; a step = a tile")
    ap.add_argument("--img-size", type=int, default=600, help="predict workload: Config.img_size (short side the tile is resized to)")
    ap.add_argument("--tune-cache", default=None,
                    help="file with measured GEMM launch choices: loaded when present (no trial launches), written after warm-up otherwise")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the ranks, form the process group over gloo (no GPU), all-reduce the ranks, print one JSON line: "
                         "exercises the self-launch path on a machine without GPUs (tests/test_bench_launch.py)")
    args = ap.parse_args()

    if args.workload == "cont":
        args.trainable = "cont"
    elif args.trainable == "cont":
        args.workload = "cont"
    if args.workload in ("predict", "vgg16", "cont") and args.gpus != 1:
        raise SystemExit("bench.py --workload %s is a single-GPU line (the multi-GPU metric is the default workload)" % args.workload)
    if args.workload == "predict":
        return bench_predict(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process has made no GPU call (torch is imported, nothing
        # initialised) -- it starts N fresh rank processes, one per GPU, relays rank 0's JSON line and exits with their code
        from radnet_hip.launch import spawn_ranks
        sys.exit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (args.gpus, world))
    if args.launch_check:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": float(t.item()),
                              "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))}))
        if world > 1:
            dist.destroy_process_group()
        return
    dist = None
    nccl1 = os.environ.get("RADNET_BENCH_REHEARSAL") == "nccl1" and world == 1
    if nccl1:
        # rehearsal of the RCCL code path on ONE GPU: a 1-rank process group, both gradient exchanges issued (identity
        # reductions) from the lanes they are issued from on N GPUs, head update deferred as on N GPUs
        import torch.distributed as dist
        from radnet_hip import trainer as _tr
        _tr.FORCE_COLLECTIVES = True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    elif world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("RADNET_BENCH_REHEARSAL") == "1":
            # rehearsal of the multi-rank code path on a ONE-GPU box: every rank on cuda:0, gradients exchanged by gloo
            # (through host memory).  Exercises lanes + deferred head update + collectives; its number means nothing.
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")

    from faster_rcnn.config import Config
    from radnet_hip import synth
    from radnet_hip.engine import FasterRCNNEngine
    from radnet_hip.trainer import TrainStep

    C = Config()
    cont = args.trainable == "cont"
    vgg = args.workload == "vgg16"
    if cont:
        if world != 1:
            raise SystemExit("bench.py --trainable cont is single-GPU")
        from radnet_hip.engine_cont import ContEngine
        from radnet_hip.trainer_cont import ContTrainStep
        eng = ContEngine(C, device_index=local_rank)
        eng.set_weights(synth.synthetic_weights(seed=3))
        _ts = ContTrainStep(eng)

        class _Adapter:              # same calling convention as TrainStep for the loops below
            skipped_head_steps = property(lambda self: _ts.skipped_head_steps)

            def step(self, batch, next_batch=None, after_next=None, upcoming=None):
                return _ts.step(batch, upcoming=upcoming)

            def flush(self):
                pass

            def losses(self):
                return _ts.losses()

        ts = _Adapter()
    elif vgg:
        from radnet_hip import make_engine
        C.network = "vgg16"
        C.anchor_box_scales = [128, 256, 512]        # config.py:46: the 3-scale alternative BASELINE cfg 5 names (A = 9)
        eng = make_engine(C, device_index=local_rank)
        eng.set_weights(synth.synthetic_weights_vgg16(seed=3, n_anchors=eng.A, n_classes=eng.nc))
        ts = TrainStep(eng)
    else:
        eng = FasterRCNNEngine(C, device_index=local_rank)
        eng.set_weights(synth.synthetic_weights(seed=3))
        ts = TrainStep(eng, world_size=world, defer_head_update=True if nccl1 else None)
    have_cache = args.tune_cache is not None and os.path.exists(args.tune_cache)
    if have_cache:
        eng.load_tuning(args.tune_cache)
    batch = make_batch(rank, args.per_gpu_batch, args.height, args.width)
    np.random.seed(64 + rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Priming (the "compile" of this framework, outside W and K): every GEMM shape is measured once, every layer program of
    # every buffer set runs once eagerly and is then recorded into its hipGraph -- 2 uses per buffer set.  Without it a short
    # warm-up (W < 13) would leave graph captures, each with a device synchronisation, inside the timed region.
    LOOK = getattr(ts, "LOOKAHEAD", 3)        # announced batches the step uses (a prefetching loader knows them)
    n_prime = 2 * getattr(ts, "NBUF", 1) + 2
    for k in range(n_prime):
        ts.step(batch, upcoming=[batch] * min(LOOK, n_prime - 1 - k))
    ts.flush()
    barrier()
    # Everything long-lived exists now (plans, descriptors, graphs): collect once and move it to the permanent generation,
    # so CPython's full collection -- measured at 41 ms here, i.e. twelve steps' worth of GPU idle -- does not fire at a
    # random step of the run (tools/variance_probe.py).
    import gc
    gc.collect()
    gc.freeze()
    if args.tune_cache is not None and not have_cache and rank == 0:
        eng.save_tuning(args.tune_cache)
    # That collection (and the table write) leaves the GPU idle for ~50 ms, after which the first ~8 steps run 5-25 % slow
    # (clock ramp: tools/bench_sequence_probe2.py -- the first 20-step region after the pause took 41.2 ms, every repetition
    # of it 39.2).  The pause is this script's, not the workload's: it sits HERE, in the priming, followed by a few more
    # priming steps, so that the W warm-up steps and the timed region meet the GPU in the state a running training loop keeps it in.
    n_settle = 12
    for k in range(n_settle):
        ts.step(batch, upcoming=[batch] * min(LOOK, n_settle - 1 - k))
    ts.flush()
    barrier()
    for k in range(args.warmup):      # the pipeline drains at the end of the warm-up: nothing of the timed steps is enqueued early
        ts.step(batch, upcoming=[batch] * min(LOOK, args.warmup - 1 - k))
    ts.flush()
    barrier()
    skipped0 = ts.skipped_head_steps
    t0 = time.perf_counter()
    for k in range(args.steps):
        # the input pipeline knows the next batch: its label / base-forward phases are enqueued across the host sync
        ts.step(batch, upcoming=[batch] * min(LOOK, args.steps - 1 - k))
    ts.flush()                       # multi-GPU: the last step's deferred head update belongs to the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    skipped = ts.skipped_head_steps - skipped0
    if dist is not None:              # decided together: a rank that left alone would strand the others in the next collective
        t = torch.tensor([skipped], dtype=torch.int64, device=eng.dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        skipped = int(t.item())
    if skipped:
        if dist is not None:
            dist.destroy_process_group()
        raise SystemExit("invalid run: %d classifier-head steps were skipped inside the timed region" % skipped)
    losses = ts.losses()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=eng.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline leg: same step, GEMM launches bracketed by HIP events on the launch stream (not inside `value`)
    roof = None
    if args.roofline_steps > 0:
        # The leg runs the step on ONE lane (launches alone on the chip).  The pipelined run above kept the head phase on
        # its own context: give the main context its work-unit / row tables for those shapes first (built on first use with
        # a host-synchronous upload that would otherwise sit between a launch's two events).
        for _ in range(6):          # every buffer set once
            ts.step(batch)
        ts.flush()
        torch.cuda.synchronize()
        eng.ctx.timing(True)
        eng.ctx.timing_reset()
        for _ in range(args.roofline_steps):
            ts.step(batch)
        ts.flush()                    # data-parallel: the last deferred head update must not be left pending
        torch.cuda.synchronize()
        per = {}
        tot_ms = tot_fl = 0.0
        tot_n = 0
        for cls, name in ((0, "conv_igemm_fwd"), (1, "conv_igemm_dgrad"), (2, "conv_wgrad"), (4, "conv_bwd_pair (dgrad + wgrad of a layer, one launch)")):
            ms, n, fl = eng.ctx.timing_read(cls)
            if cls == 4 and not n:
                continue
            per[name] = {"launches_per_step": n / args.roofline_steps / args.per_gpu_batch, "avg_us": 1e3 * ms / max(n, 1),
                         "tflops": fl / max(ms, 1e-9) / 1e9}
            tot_ms += ms; tot_fl += fl; tot_n += n
        eng.ctx.timing(False)
        # 3x3 layers run as Winograd F(2x2,3x3) (input transform + 16 batched GEMMs on the same kernel + output transform):
        # timed per LAYER, credited the layer's algorithmic 2*M*N*9C flops, not the 2.25x fewer it executes
        wms, wn, wfl = eng.ctx.timing_read(3)          # Winograd layers of the programs: timed per layer inside radnet_program_run
        if wn:
            per["conv3x3_winograd_layers"] = {"launches_per_step": wn / args.roofline_steps / args.per_gpu_batch, "avg_us": 1e3 * wms / wn,
                                              "tflops": wfl / max(wms, 1e-9) / 1e9}
            tot_ms += wms; tot_fl += wfl
        ach = tot_fl / max(tot_ms, 1e-9) / 1e9
        # executed flops: what the MFMAs really performed (a Winograd layer executes 2*positions*tiles*C*N: 1/4 of its credit as
        # F(4x4,3x3) on whole tiles, 1/2.25 as F(2x2,3x3), a little more on maps that are not multiples of the tile)
        exec_fl = tot_fl - (wfl * (1.0 - winograd_executed_share(eng)) if wn else 0.0)
        roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                "traffic": (committed_pmc_traffic() or {}).get("bytes"), "traffic_detail": committed_pmc_traffic(), "kernel": "conv_igemm_kernel + conv_wgrad_kernel (fp32 v_mfma_f32_32x32x2_f32; 3x3 layers via Winograd transforms)",
                "schedule": "one lane, launches isolated (each GEMM launch alone on the chip, timed from its own dispatch: hipExtLaunchKernelGGL "
                            "start / stop events = the kernel's begin and end, as rocprofv3 --kernel-trace reports them; Winograd layers: two marker "
                            "events around their three kernels); `value` is measured on the pipelined schedule, where lanes overlap -- "
                            "gemm_ms_per_image here may exceed ms_per_step",
                "winograd_credit": "Winograd layers are timed per layer (3 kernels) and credited the algorithmic 2*M*N*9C flops, not the 4x (F(4x4,3x3)) / 2.25x (F(2x2,3x3)) fewer they execute",
                "executed_tflops": exec_fl / max(tot_ms, 1e-9) / 1e9, "executed_frac": exec_fl / max(tot_ms, 1e-9) / 1e9 / PEAK_FP32_MFMA_TFLOPS,
                "gemm_ms_per_image": tot_ms / args.roofline_steps / args.per_gpu_batch,
                "gemm_gflop_per_image": tot_fl / args.roofline_steps / args.per_gpu_batch / 1e9, "by_kernel": per}
        if rank == 0 and not cont and not vgg and hasattr(eng, "_plan_rpn"):
            try:
                nb0 = args.per_gpu_batch if getattr(ts, "batched", False) else 1       # the mini-batch runs as one program
                bp0 = eng._plan_base(nb0, args.height, args.width, 0)
                roof["layers_3x3"] = layers_3x3_table(eng, bp0, eng._plan_rpn(bp0["fh"], bp0["fw"], bp0["F"], nb0))
            except Exception as e:                       # a secondary table must never cost the bench line
                roof["layers_3x3_error"] = repr(e)
    coll = collectives_leg(eng, ts, dist, world) if (dist is not None and not cont) else None
    if dist is not None:
        dist.barrier()

    if rank == 0:
        n_img = world * args.per_gpu_batch * args.steps
        value = n_img / elapsed
        algo_gf = ALGO_GFLOP_CONT if cont else ALGO_GFLOP_PER_IMAGE
        if vgg:      # no survey figure for the VGG16 train step: the GEMM flops the step's launches are credited (2MNK, direct convs)
            algo_gf = roof["gemm_gflop_per_image"] if roof is not None else float("nan")
        out = {
            "metric": "train images/sec, %s Faster R-CNN 1000x600" % ("VGG16" if vgg else "ResNet50"),
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "full train step (RPN + RoiPoolingConv + classifier head + losses + 2x Adam), %s, %dx%d, "
                                   "batch=%d per GPU, %s; identical panel and boxes on every rank and step (per-rank RNG seed)"
                                   % ("VGG16 base model (BASELINE cfg 5: 3 anchor scales x 3 ratios, fc head on 20 sampled RoIs per step)" if vgg else "ResNet50",
                                      args.width, args.height, args.per_gpu_batch,
                                      "stages 3-4 trainable (cont_train.py)" if cont else "base frozen (train.py)"),
                       "bench_workload": args.workload,
                       "per_gpu_batch": args.per_gpu_batch, "global_batch": world * args.per_gpu_batch, "anchors": eng.A, "n_rois": C.n_rois,
                       "mini_batch_form": ("one layer program, images stacked along the GEMM M dimension" if getattr(ts, "batched", False) and args.per_gpu_batch > 1
                                           else "image by image"),
                       "parallelism": "dp%d" % world,
                       "schedule": ("one lane" if cont or not getattr(ts, "side_prefetch", False) else
                                    "pipelined over HIP streams: %d prefetch lanes (frozen base forward, %d batches ahead%s), RPN phase, head phase"
                                    % (getattr(eng, "n_side_lanes", 1), LOOK,
                                       ", two consecutive batches' base forwards as one nb=2 program" if getattr(ts, "stack_base", False) and args.per_gpu_batch == 1 else "")),
                       # the contexts' real flag (every lane's), not the environment variable
                       "reductions": ("ordered (radnet_set_deterministic: no floating-point atomics in the step; bit-identical across runs and "
                                      "schedules for a given launch-shape table -- another K-split count re-associates the sums)"
                                      if all(int(eng.lib.radnet_get_deterministic(c.h)) == 1 for c in eng.contexts()) else "fp32 atomics (radnet_set_deterministic off)"),
                       "launch_shapes": (("table %s" % os.path.basename(args.tune_cache)) if have_cache else
                                         ("tuned in situ, shipped (%s) + measured on first use" % ", ".join(eng.shipped_tuning)) if getattr(eng, "shipped_tuning", None)
                                         else "measured per shape on first use"),
                       "base_forward": ("chain kernel, %d workgroups (RADNET_CHAIN=1)" % (getattr(eng, "chain_wgs", 0) or 512)) if getattr(eng, "use_chain", False) else "launch list (hipGraph)",
                       "algorithmic_gflop_per_image": algo_gf,
                       "step_tflops_algorithmic": value / world * algo_gf / 1e3},
            "losses": losses,
        }
        if roof is not None:
            out["roofline"] = roof
        if coll is not None:
            out["config"]["collectives"] = coll
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(workload=args.workload)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
