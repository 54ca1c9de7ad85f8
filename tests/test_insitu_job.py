"""In-situ tuning as part of a running job (run_training(tune=True) / RADNET_INSITU=1; radnet_hip/insitu.py tune_job): control flow
on CPU with a scripted engine -- the table cache, the walk over the job's own steps, and the multi-rank rule (world 2 over gloo):
rank 0's table, clock and verdicts are broadcast, every rank runs the same number of steps (each step carries a collective, so a
rank that stepped once more or less would hang the job) and ends with rank 0's table."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HEADER = "# radnet tuned GEMM launch shapes v2: kind m n k c npos stride | tile_a tile_b slices ms waves\n"
KEY_A = (0, 980, 512, 2048, 2048, 1, 1)        # the scripted clock rewards 32x32 here ...
KEY_B = (0, 2394, 1024, 256, 256, 1, 1)        # ... and nothing here
SHIPPED = (0, 37101, 64, 64, 64, 1, 1)         # an entry the engine held before the job: not walked


def _paths():
    import sys
    for p in (ROOT, os.path.join(ROOT, "rock-art-radnet_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


class FakeEngine:
    NETWORK, workload, shipped_tuning, dev = "resnet50", "train", [], "cpu"

    def __init__(self, rank):
        from radnet_hip import insitu
        self.I = insitu
        self.rank = rank
        self.table = {SHIPPED: [64, 64, 1, 0.009, 4]}
        self._graphs = {}
        self.measured = False

    def first_use(self):                      # what the autotuner would measure when the job's shapes first launch: rank-dependent
        if not self.measured:
            self.measured = True
            self.table.setdefault(KEY_A, [64, 64, 2 if self.rank == 0 else 3, 0.0275, 8])      # a loaded entry is not measured again
            self.table.setdefault(KEY_B, [64, 64, 1, 0.018, 4])

    def save_tuning(self, path):
        self.I.write_table(path, self.table, HEADER)

    def load_tuning(self, path):
        tab, _ = self.I.read_table(path)
        self.table.update(tab)


class FakeStep:
    NBUF = 2

    def __init__(self, rank, dist=None):
        self.eng = FakeEngine(rank)
        self.dist = dist
        self.steps = 0
        self.insitu_steps = 4
        self.insitu_sync = lambda: None
        self.insitu_measure = self._clock

    def _clock(self, n):
        # rank 0's clock: 100 us per step, 96 with a 32-row tile on KEY_A; rank 1's clock disagrees (always 100): it must not matter
        for _ in range(n):
            self.step([None])
        a = self.eng.table[KEY_A]
        return 96.0 if (self.eng.rank == 0 and a[0] == 32 and a[1] == 32) else 100.0

    def step(self, batch, upcoming=None):
        self.eng.first_use()
        self.steps += 1
        if self.dist is not None:
            t = torch.ones(1)
            self.dist.all_reduce(t)           # the gradient exchange every step carries: asymmetric step counts would hang here

    def flush(self):
        pass


def _feed():
    k = 0
    while True:
        yield dict(img=torch.zeros(60, 100, 3).numpy(), k=k)
        k += 1


def test_single_process_walk_writes_the_cache_and_the_next_job_loads_it(tmp_path):
    _paths()
    from faster_rcnn import data_feed
    from radnet_hip import insitu
    ts = FakeStep(0)
    logs = []
    done = data_feed.run_training(ts, _feed(), 400, lookahead=2, tune=True, tune_budget_s=30.0, tune_cache_dir=str(tmp_path), log=logs.append)
    assert done == 400 and ts.steps >= 400                      # the walk's steps are the job's steps (measure() ran some more inside the fake clock)
    path = insitu.cache_path("resnet50", "train", 60, 100, 1, "cpu", str(tmp_path))
    tab, _ = insitu.read_table(path)
    assert tab[KEY_A][:2] == [32, 32] and tab[KEY_B][:2] == [64, 64]
    assert tab[SHIPPED] == [64, 64, 1, 0.009, 4] and ts.eng.table[KEY_A][:2] == [32, 32]
    assert any("tuned in situ" in m for m in logs)
    # a second job of the same (network, panel, batch, device): no walk, the table is loaded before the first step
    ts2 = FakeStep(0)
    logs2 = []
    data_feed.run_training(ts2, _feed(), 10, lookahead=2, tune=True, tune_cache_dir=str(tmp_path), log=logs2.append)
    assert ts2.steps == 10 and ts2.eng.table[KEY_A][:2] == [32, 32] and any("loaded" in m for m in logs2)
    # a workload with a shipped table for this very panel size is left alone
    ts3 = FakeStep(0)
    ts3.eng.shipped_tuning = ["train_resnet50_60x100_batch1.txt"]
    data_feed.run_training(ts3, _feed(), 10, lookahead=2, tune=True, tune_cache_dir=str(tmp_path / "other"), log=lambda m: None)
    assert ts3.steps == 10 and not os.path.exists(str(tmp_path / "other"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cache, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    _paths()
    import torch.distributed as dist
    from faster_rcnn import data_feed
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ts = FakeStep(rank, dist)
    done = data_feed.run_training(ts, _feed(), 300, lookahead=2, tune=True, tune_budget_s=30.0, tune_cache_dir=cache, log=lambda m: None)
    out[rank] = (done, ts.steps, {k: list(v) for k, v in ts.eng.table.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_walk_together_and_end_with_rank_0s_table(tmp_path):
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, str(tmp_path), out), nprocs=world, join=True)
    (d0, s0, t0), (d1, s1, t1) = out[0], out[1]
    assert d0 == d1 == 300 and s0 == s1                       # same number of steps on both ranks (or the all-reduces would have hung)
    assert t0 == t1                                           # rank 1 adopted rank 0's starting shapes (slices 2, not its own 3) and verdicts
    assert t0[KEY_A][:3] == [32, 32, 1] or t0[KEY_A][:2] == [32, 32]
    files = [f for f in os.listdir(str(tmp_path)) if f.endswith(".txt")]
    assert len(files) == 1                                    # written by rank 0 only; no stray temporaries
