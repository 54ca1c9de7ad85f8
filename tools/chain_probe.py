#!/usr/bin/env python3
"""Diagnosis of the chain kernel on the GPU box: runs prefixes of the base layer program (stages 2-4 at a given size) as chains
of growing length, watching each launch from the host through radnet_chain_peek; a launch that does not finish within a few
seconds is reported with the item its queue head points at and the counters that item waits for."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rock-art-radnet_amd")):
    sys.path.insert(0, p)
from faster_rcnn.config import Config  # noqa: E402
from radnet_hip import synth  # noqa: E402
from radnet_hip.engine import FasterRCNNEngine  # noqa: E402


def say(*a):
    print(*a, flush=True)


def main():
    H, W = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (300, 500)))
    wgs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    C_ = Config()
    C_.img_size = min(H, W)
    eng = FasterRCNNEngine(C_)
    eng.set_weights(synth.synthetic_weights(seed=3))
    bp = eng.upload_image(synth.synthetic_panel(1, H, W))
    Fref = eng.base_forward(bp).cpu().numpy()
    ops = bp["ops"]
    say("program:", len(ops), "ops")
    # reference outputs of every op: rerun the launch list, read y of each conv / wino
    out = (C.c_uint32 * 160)()
    for n in [1, 2, 3, 4, 5, 8, 12, len(ops) - 2]:
        sub = ops[2:2 + n]
        arr = eng._compile(sub)
        h = C.c_void_p()
        rc = eng.lib.radnet_chain_build(eng.ctx.h, C.cast(arr, C.c_void_p), len(sub), wgs, C.byref(h))
        if rc != 0:
            say("prefix", n, "build failed:", eng.lib.radnet_last_error(eng.ctx.h).decode())
            continue
        torch.cuda.synchronize()
        t0 = time.time()
        eng.ctx.check(eng.lib.radnet_chain_run(eng.ctx.h, h), "chain_run")
        done = False
        while time.time() - t0 < 6.0:
            eng.lib.radnet_chain_peek(h, -1, out, 160)
            if out[4] >= 1:                     # runs
                done = True
                break
            time.sleep(0.05)
        if done:
            torch.cuda.synchronize()
            say("prefix %3d ops: finished in < %.2f s, first error %d" % (n, time.time() - t0, out[3]))
        else:
            say("prefix %3d ops: NOT finished after 6 s: next item %d, workgroups gone %d, error %d" % (n, out[0], out[1], out[2]))
            eng.lib.radnet_chain_peek(h, -2, out, 160)
            say("   waves by phase (0 never ran, 1 drew an item, 2 inputs complete, 3 item computed, 4 stores drained, 5 left):", [out[20 + j] for j in range(8)],
                "a stuck wave: phase %d item %d" % (out[28] & 255, out[28] >> 8))
            # the items around the queue head: which one waits for what
            lo = max(0, int(out[0]) - 1200)
            stuck = 0
            for item in range(lo, int(out[0])):
                k = eng.lib.radnet_chain_peek(h, item, out, 160)
                unmet = [(out[20 + 2 * j], out[21 + 2 * j]) for j in range(max(k, 0)) if out[20 + 2 * j] < out[21 + 2 * j]]
                if unmet and stuck < 12:
                    stuck += 1
                    say("   item %d stage %d (bx %d by %d bz %d) deps [%d+%d, %d+%d] waits for %s" % (item, out[8], out[9], out[10], out[11], out[12], out[13], out[14], out[15], unmet[:6]))
            say("   giving the launch 3 more seconds (its own 1 s watchdog should end it)")
            time.sleep(3.0)
            eng.lib.radnet_chain_peek(h, -1, out, 160)
            say("   now: next %d gone %d error %d first error %d runs %d" % (out[0], out[1], out[2], out[3], out[4]))
            if out[4] < 1:
                say("   still running: leaving")
                os._exit(3)
        eng.lib.radnet_chain_destroy(h)
    say("done")


if __name__ == "__main__":
    main()
