"""Command line of radnet_hip/insitu.py: tune the GEMM launch shapes IN SITU -- against the throughput of the pipelined train step
instead of each launch alone -- for one of bench.py's workloads, and write the table (radnet_tune_save format: `bench.py
--tune-cache`, engine.load_tuning, radnet_hip/tuned/).  2-4 minutes per pass on one MI355X.

usage: python tools/insitu_tune.py <out table> [--passes 1] [--steps 200] [--budget-s 900] [--start <table>] [--wide]
                                   [--per-gpu-batch 1] [--trainable train|cont] [--dp-rehearsal] [--workload train|predict] [--network resnet50|vgg16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    ap.add_argument("--network", choices=("resnet50", "vgg16"), default="resnet50")
    ap.add_argument("--workload", choices=("train", "predict"), default="train", help="predict = RADNet.predict's tile loop (cfg 3)")
    ap.add_argument("--per-gpu-batch", type=int, default=1, help="images per step (2 = BASELINE cfg 4 on one GPU)")
    ap.add_argument("--kinds", default=None, help="comma-separated entry kinds to walk (default: all), e.g. 32,33 = the paired / fused launch decisions")
    args = ap.parse_args()
    kinds = tuple(int(v) for v in args.kinds.split(",")) if args.kinds else None
    os.environ["RADNET_SHIPPED_TUNING"] = "0"      # start from what the engine measures itself (or --start), never from a shipped table
    sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]
    import torch
    import bench
    from faster_rcnn.config import Config
    from radnet_hip import insitu, make_engine, synth
    from radnet_hip.trainer import TrainStep
    if args.workload == "predict":         # RADNet.predict's tile loop (BASELINE cfg 3): 2048x2048 tiles, two in flight (RADNet._detect_all)
        import numpy as np
        from faster_rcnn import models as M
        from faster_rcnn.RADNet import RADNet
        from faster_rcnn.base_models import resnet50
        C = Config()
        m_rpn, m_cls, m_all, m_rpn3, m_det = M.build_models(C, weights=synth.synthetic_weights(seed=3))
        net = RADNet(C, m_rpn3, m_det, resnet50.preprocess)
        net.device_resident = True
        eng = m_rpn3._s.eng
        tiles = [np.random.RandomState(40 + i).randint(0, 256, (2048, 2048, 3)).astype(np.uint8) for i in range(8)]

        def run(n):
            while n > 0:
                net._detect_all(tiles[:max(2, min(n, len(tiles)))])
                n -= len(tiles)

        before, after, changed = insitu.tune(eng, run, lambda: None, args.out, passes=args.passes, steps=args.steps, budget_s=args.budget_s,
                                             start=args.start, gain=args.gain, wide=args.wide, n_prime=8, log=lambda m: print(m, flush=True), kinds=kinds)
        print("in situ: tile %.1f -> %.1f us (%.1f -> %.1f tiles/s), %d entries changed" % (before, after, 1e6 / before, 1e6 / after, len(changed)))
        for key, cand in changed:
            print("  changed %s -> tile %dx%d slices %d waves %d" % (key, cand[0], cand[1], cand[2], cand[3]))
        return
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
        C = Config()
        if args.network == "vgg16":           # BASELINE cfg 5: VGG16 base, 3 scales x 3 ratios, fc head
            C.network, C.anchor_box_scales = "vgg16", [128, 256, 512]
        eng = make_engine(C)
        eng.set_weights(synth.synthetic_weights_vgg16(seed=3, n_anchors=eng.A, n_classes=eng.nc) if args.network == "vgg16"
                        else synth.synthetic_weights(seed=3))
        ts = TrainStep(eng, defer_head_update=True if args.dp_rehearsal else None)
    batch = bench.make_batch(0, args.per_gpu_batch, 600, 1000)
    look = getattr(ts, "LOOKAHEAD", 3)

    def run(n):
        for _ in range(n):
            ts.step(batch, upcoming=[batch] * look)

    before, after, changed = insitu.tune(eng, run, ts.flush, args.out, passes=args.passes, steps=args.steps, budget_s=args.budget_s,
                                         start=args.start, gain=args.gain, wide=args.wide, n_prime=2 * getattr(ts, "NBUF", 6) + 6,
                                         log=lambda m: print(m, flush=True), kinds=kinds)
    per = args.per_gpu_batch
    print("in situ: step %.1f -> %.1f us (%.1f -> %.1f images/s), %d entries changed" % (before, after, per * 1e6 / before, per * 1e6 / after, len(changed)))
    for key, cand in changed:
        print("  changed %s -> tile %dx%d slices %d waves %d" % (key, cand[0], cand[1], cand[2], cand[3]))
    ts.flush()


if __name__ == "__main__":
    main()
