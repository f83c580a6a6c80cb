#!/usr/bin/env python3
"""How the memory pool (csrc/block_pool.h) behaves on the benchmark shape: per proof, the wall time, the hipMalloc calls
it made, what the pool holds from the driver and how much of that is free afterwards.

    python tools/pool_report.py [--log-rows 22] [--proofs 6]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-rows", type=int, default=22)
    ap.add_argument("--cols", type=int, default=100)
    ap.add_argument("--proofs", type=int, default=6)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    import airs
    pkg = ge._load_pkg()
    dev = pkg.Device(0)
    h = 1 << args.log_rows
    params = airs.app_params(args.log_rows + 2)
    pk = pkg.ProvingKey(airs.pk_blob(params, [airs.uniform_runner_air(args.cols)]))
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(1)
    trace = torch.randint(0, 2, (args.cols * h,), dtype=torch.int32, device="cuda:0", generator=gen)
    trace *= int(pkg.to_mont(np.array([1], dtype=np.uint32))[0])
    rows, last = [], pkg.memory_stats()["driver_allocs"]
    for k in range(args.proofs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.prove(pk, [(0, trace, h, [])])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        m = pkg.memory_stats()
        rows.append({"proof": k, "ms": round(ms, 2), "driver_allocs": m["driver_allocs"] - last,
                     "segments_GiB": round(m["segments"] / 2**30, 2), "free_GiB": round(m["cached"] / 2**30, 2),
                     "peak_live_GiB": round(m["peak"] / 2**30, 2)})
        last = m["driver_allocs"]
        print(rows[-1], file=sys.stderr)
    print(json.dumps({"workload": f"2^{args.log_rows} x {args.cols} BenchmarkAir, app params", "proofs": rows}))


if __name__ == "__main__":
    main()
