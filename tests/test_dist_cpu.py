"""CPU test of the N>1 bench path (replicas: no data-path collective, only barrier + MAX of the
per-rank elapsed time), with gloo and world_size 2 on 127.0.0.1."""
import os
import socket

import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from conftest import load_pkg
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_pkg()
    dist.barrier()
    got = pkg.max_over_ranks(1.0 + rank, "cpu")      # rank r "took" 1+r seconds
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_max_over_ranks_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: 2.0, 1: 2.0}


def test_max_over_ranks_single_process(pkg):
    assert pkg.max_over_ranks(3.5) == 3.5


def test_bench_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` itself starts two ranks (gloo here): n_gpus comes from real ranks, and a
    --gpus that disagrees with WORLD_SIZE is refused."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"launch_check": True, "n_gpus": 2}
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--launch-check"],
                         env=dict(env, RANK="0", WORLD_SIZE="2"), capture_output=True, text=True, timeout=120)
    assert bad.returncode == 2 and "WORLD_SIZE" in bad.stderr
