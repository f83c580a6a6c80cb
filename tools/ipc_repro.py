#!/usr/bin/env python3
"""Reproducer for the round-3 observation "hipIpcOpenMemHandle of a peer's 3.2 GB segment did not return when two 85 GB
processes shared one GPU" (DESIGN.md §9), through the library's own IPC exchange (csrc/ipc_comm.cpp) -- which now maps in
turn, gives every open a deadline and degrades to host staging instead of hanging.

    python tools/ipc_repro.py --world 2 --filler-gb 85 --segment-gb 3.2 [--exchange-mb 256]

Starts `world` processes on GPU 0. Each allocates `filler-gb` of device memory (touched), then ONE allocation of
`segment-gb` whose tail holds its send buffer, and runs three all-gathers of `exchange-mb` through ovm_hip_ipc_*; prints per
rank the time of each exchange and the communicator's status (mapped / degraded + reason). Always wrap it in `timeout`."""
import argparse
import json
import os
import subprocess
import sys
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(args):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import torch
    from conftest import load_pkg
    pkg = load_pkg()
    pkg.Device(0)
    t0 = time.time()
    filler = []
    left = int(args.filler_gb * (1 << 30))
    while left > 0:
        n = min(left, 8 << 30)
        filler.append(torch.ones(n, dtype=torch.uint8, device="cuda:0"))
        left -= n
    seg = torch.empty(int(args.segment_gb * (1 << 30)), dtype=torch.uint8, device="cuda:0")
    nb = args.exchange_mb << 20
    send = seg[seg.numel() - nb:]
    send.fill_(args.rank + 1)
    recv = torch.empty(nb * args.world, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    print(f"[rank {args.rank}] allocated in {time.time() - t0:.1f} s; device memory in use {(total - free) / 2**30:.1f} of {total / 2**30:.0f} GiB", flush=True)
    shard = pkg.Shard(args.rank, args.world, args.shm, bulk="ipc")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    times = []
    for i in range(3):
        t = time.time()
        rc = pkg._lib().ovm_hip_ipc_allgather_device(shard.ipc, send.data_ptr(), recv.data_ptr(), nb, stream)
        times.append(round(time.time() - t, 3))
        ok = bool((recv.view(args.world, nb)[:, 0].cpu() == torch.arange(1, args.world + 1, dtype=torch.uint8)).all())
        print(f"[rank {args.rank}] exchange {i}: rc {rc}, {times[-1]} s, data {'ok' if ok else 'WRONG'}", flush=True)
    print(f"[rank {args.rank}] " + json.dumps({"exchange_s": times, "ipc": shard.ipc_status()}), flush=True)
    shard.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--filler-gb", type=float, default=85.0)
    ap.add_argument("--segment-gb", type=float, default=3.2)
    ap.add_argument("--exchange-mb", type=int, default=256)
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--shm", default="")
    args = ap.parse_args()
    if args.rank >= 0:
        return worker(args)
    shm = f"/ovm-ipc-repro-{uuid.uuid4().hex[:10]}"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), "--shm", shm] + sys.argv[1:], env=env)
             for r in range(args.world)]
    sys.exit(max(p.wait() for p in procs))


if __name__ == "__main__":
    main()
