"""One process per GPU without an external launcher: `python bench.py --gpus N` (no torchrun) starts its N ranks itself.

The parent must not have touched the GPU (importing torch is fine, any HIP call is not: a forked/exec'ed child of a
process that initialised the device takes the machine down on this pool, and the children need the devices to themselves).
Children get the environment torch.distributed.run would give them (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE,
MASTER_ADDR = 127.0.0.1, MASTER_PORT = a free port) and run the same command line; rank 0's stdout is the parent's stdout
(the bench's JSON line), the other ranks' stdout goes to stderr.  If a rank dies, exactly the processes started here are
terminated (by PID) and its exit code is returned."""
import os
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this driver (RCCL needs it)
    return env


def spawn_ranks(world, argv, poll_s=0.05, timeout_s=None):
    """Run `argv` (a full command line, e.g. [sys.executable, 'bench.py', ...]) as `world` ranks; returns the exit code
    (0 when every rank exited 0, else the first failing rank's code)."""
    port = free_port()
    procs = []
    for r in range(world):
        out = None if r == 0 else sys.stderr
        procs.append(subprocess.Popen(argv, env=rank_env(r, world, port), stdout=out, stderr=None))
    t0 = time.monotonic()
    rc = 0
    live = set(range(world))
    try:
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (timeout_s is not None and time.monotonic() - t0 > timeout_s):
                if rc == 0:
                    rc = 124
                break
            if live:
                time.sleep(poll_s)
    finally:
        for r in sorted(live):                 # only what was started here, by PID
            procs[r].terminate()
        for r in sorted(live):
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    return rc
