"""rpn_conv1's 36 batched Winograd GEMMs ([160 x 1024] x [1024 x 512] each) alone, one forced launch shape per process: the workload
of tools/rpn_gemm_traffic.sh, which runs it under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) to see what each shape
fetches for its 111 MB of operands and results.  usage: rpn_gemm_traffic.py <tile_a> <tile_b> <slices> [pad]
pad != 0: V / U / M rows padded by `pad` floats (a pitch that is not a power of two)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]
from radnet_hip import lib as L  # noqa: E402


def main():
    a, b, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    ctx = L.Context(0)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.radnet_set_workspace(ctx.h, ws.data_ptr(), ws.numel()), "ws")
    P, T, c, n = 36, 160, 1024, 512
    V = torch.randn(P, T, c, device="cuda")
    U = torch.randn(P, c, n, device="cuda")
    M = torch.empty(P, T, n, device="cuda")
    ctx.check(ctx.lib.radnet_force_config(ctx.h, a, b, s), "force")
    ctx.check(ctx.lib.radnet_force_waves(ctx.h, 4), "waves")
    junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(12):
        junk.random_(0, 255)              # 512 MB through the caches between launches: every launch starts cold, as in the step
        e0.record()
        ctx.check(ctx.lib.radnet_gemm_batched(ctx.h, V.data_ptr(), U.data_ptr(), M.data_ptr(), P, T, n, c), "gemm")
        e1.record()
        e1.synchronize()
        tot += e0.elapsed_time(e1)
    print("tile %dx%d slices %d: %.1f us per launch (cold caches)" % (a, b, s, tot / 12 * 1e3))


if __name__ == "__main__":
    main()
