"""Latency of a 1-rank RCCL all-reduce (identity) on the gradient arenas' sizes, as seen by the issuing stream.
usage: python tools/rccl_1rank_probe.py"""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29535")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for mb in (0.25, 19, 24, 60):
    t = torch.zeros(int(mb * 1e6 / 4), device="cuda")
    for _ in range(5):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        dist.all_reduce(t)
    e1.record()
    torch.cuda.synchronize()
    print("%6.2f MB: %.1f us per all_reduce (stream time)" % (mb, e0.elapsed_time(e1) * 1e3 / n), flush=True)
    import time
    t0 = time.perf_counter()
    for _ in range(n):
        dist.all_reduce(t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("           host time per call %.1f us" % ((t1 - t0) * 1e6 / n), flush=True)
dist.destroy_process_group()
