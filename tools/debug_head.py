"""Debug aid: head forward/backward vs the oracle after poisoning the allocator's free memory.

Reads of uninitialised device memory show up as errors here that a fresh process (zeroed pages) hides.
usage: python tools/debug_head.py [poison_value]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]

from faster_rcnn.config import Config  # noqa: E402
from oracle import dense  # noqa: E402
from radnet_hip.engine import FasterRCNNEngine  # noqa: E402


def main():
    poison = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    junk = [torch.full((256 << 20,), poison, device="cuda") for _ in range(6)]   # 6 GiB of poison
    torch.cuda.synchronize()
    del junk
    C = Config()
    P = dense.init_params(seed=3)
    eng = FasterRCNNEngine(C)
    eng.set_weights(P)
    rs = np.random.RandomState(7)
    F = np.maximum(rs.standard_normal((1, 20, 31, 1024)), 0).astype(np.float32) * 3
    R = C.n_rois
    rois = np.stack([rs.randint(0, 25, R), rs.randint(0, 14, R), rs.randint(1, 12, R), rs.randint(1, 10, R)], 1).astype(np.float32)
    cls = rs.randint(0, 7, R)
    Y1 = np.eye(7, dtype=np.float32)[cls][None]
    lab = np.zeros((R, 24), np.float32)
    for i, c in enumerate(cls):
        if c != 6:
            lab[i, 4 * c:4 * c + 4] = 1
    Y2 = np.concatenate([lab, rs.standard_normal((R, 24)).astype(np.float32) * lab], -1)[None]
    losses, grads = dense.head_losses_and_grads(P, F, rois, Y1, Y2, 7)
    Fd = torch.from_numpy(F).cuda()
    hp = eng._plan_head(R, 20, 31, Fd)
    hp["rois"].copy_(torch.from_numpy(rois)); hp["y1"].copy_(torch.from_numpy(Y1[0])); hp["y2"].copy_(torch.from_numpy(Y2[0]))
    _, _, cache = dense.head_forward(P, F, rois, 7)
    for rep in range(2):
        eng.head_forward(hp)
        if rep == 0:
            for bi, B in enumerate(hp["blocks"]):
                for which, key in (("a", "a"), ("b", "b")):
                    g = B[key].cpu().numpy().reshape(-1, B[key].shape[-1])
                    r = cache["blocks"][bi][which]["y"].reshape(g.shape)
                    flips = np.nonzero((g > 0) != (r > 0))
                    print("block %d conv %s: act err %.2e, ReLU sign flips %d %s" % (
                        bi, which, np.abs(g - r).max() / np.abs(r).max(), len(flips[0]),
                        [(int(m), int(n), float(g[m, n]), float(r[m, n])) for m, n in list(zip(*flips))[:4]]))
        eng.set_accumulate(hp["bwd"], False)
        eng.head_backward(hp, accumulate=False)
        torch.cuda.synchronize()
        for name in eng.head_conv_names:
            c = eng.convs[name]
            got = c.dweight.cpu().numpy().astype(np.float64)
            ref = grads[name]["kernel"].reshape(-1, c.cout).astype(np.float64)
            err = np.abs(got - ref) / np.abs(ref).max()
            tag = "" if err.max() < 2e-3 else "   <-- BAD"
            print("rep %d %-16s dW err %.2e  nan %d%s" % (rep, name, err.max(), int(np.isnan(got).sum()), tag))
            if tag:
                bad = err > 2e-3
                rows, cols = np.nonzero(bad)
                print("   bad elements %d; rows %d..%d (%d distinct) cols %d..%d (%d distinct)" % (
                    bad.sum(), rows.min(), rows.max(), len(set(rows)), cols.min(), cols.max(), len(set(cols))))
                print("   sample got/ref:", [(int(r), int(cc), float(got[r, cc]), float(ref[r, cc])) for r, cc in list(zip(rows, cols))[:6]])


if __name__ == "__main__":
    main()
