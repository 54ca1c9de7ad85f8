"""Data-parallel path on CPU: world_size 2 over gloo.  Each rank computes the RPN and head gradients of ITS image with
the oracle, flattens them like the engine's arenas, and runs the product's exchange (trainer.allreduce_grad_arena);
the Adam update from the reduced arena must equal a single process accumulating both images (the reference's step
under gradient accumulation, SURVEY.md 8d cfg 4)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny_problem(img_idx):
    """Small RPN-head + dense-head gradient problem on a seeded feature map (no backbone: keeps the CPU suite fast)."""
    from oracle import dense
    rs = np.random.RandomState(100 + img_idx)
    A, nc = 12, 7
    P = {n: {k: v.astype(np.float64) for k, v in d.items()} for n, d in dense.init_params(seed=3).items() if n.startswith(("rpn", "dense"))}
    F = np.maximum(rs.standard_normal((1, 5, 6, 1024)), 0)
    valid = (rs.uniform(size=(1, 5, 6, A)) < 0.3).astype(np.float64)
    ov = (rs.uniform(size=(1, 5, 6, A)) < 0.4) * valid
    y_cls = np.concatenate([valid, ov], -1)
    y_regr = np.concatenate([np.repeat(ov, 4, -1), rs.standard_normal((1, 5, 6, 4 * A))], -1)
    _, g = dense.rpn_losses_and_grads(P, F, y_cls, y_regr, A, True)
    names = list(dense.RPN_TRAINABLE)
    flat = np.concatenate([g[n][k].ravel() for n in names for k in ("kernel", "bias")])
    return P, names, flat


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "rock-art-radnet_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from radnet_hip.trainer import allreduce_grad_arena
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, _, flat = _tiny_problem(rank)
    t = torch.from_numpy(flat.copy())
    scale = allreduce_grad_arena(t, world)
    out[rank] = (t.numpy().copy(), scale)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_mean_equals_accumulation():
    from oracle import dense
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    (g0, s0), (g1, s1) = out[0], out[1]
    assert s0 == s1 == 0.5
    assert np.array_equal(g0, g1)                           # every rank holds the same reduced arena
    P, names, f0 = _tiny_problem(0)
    _, _, f1 = _tiny_problem(1)
    assert np.allclose(g0, f0 + f1, rtol=0, atol=1e-12)
    # Adam from the reduced arena (grad_scale = 1/world) == Adam from the accumulated mean gradient
    mean = (f0 + f1) / 2
    p_a = np.zeros_like(mean); m_a = np.zeros_like(mean); v_a = np.zeros_like(mean)
    p_b = np.zeros_like(mean); m_b = np.zeros_like(mean); v_b = np.zeros_like(mean)
    dense.adam_step(p_a, g0 * s0, m_a, v_a, 1, 5e-5)
    dense.adam_step(p_b, mean, m_b, v_b, 1, 5e-5)
    assert np.allclose(p_a, p_b, rtol=0, atol=1e-15)


def test_single_rank_is_identity():
    from radnet_hip.trainer import allreduce_grad_arena
    t = torch.arange(8, dtype=torch.float32)
    assert allreduce_grad_arena(t, 1) == 1.0 and torch.equal(t, torch.arange(8, dtype=torch.float32))
