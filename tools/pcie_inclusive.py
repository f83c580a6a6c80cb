#!/usr/bin/env python3
"""The headline configuration INCLUDING the host -> device hand-over of the trace: the
reference's timed region (`stark_prove_excluding_trace`, prover/mod.rs:98-103) and bench.py's `value` start with
the trace resident in HBM; a caller that holds the trace in host memory pays
`transport_matrix_to_device` (hal.rs:152) first. Times upload (pageable and pinned source) + prove.

    python tools/pcie_inclusive.py [--log-rows 22] [--cols 100]
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
    ap.add_argument("--log-stacked-height", type=int, default=24)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    import airs
    pkg = ge._load_pkg()
    dev = pkg.Device(0)
    h, cols = 1 << args.log_rows, args.cols
    pk = pkg.ProvingKey(airs.pk_blob(airs.app_params(args.log_stacked_height), [airs.uniform_runner_air(cols)]))
    one = int(pkg.to_mont(np.array([1], dtype=np.uint32))[0])
    host = (torch.randint(0, 2, (cols * h,), dtype=torch.int32, generator=torch.Generator().manual_seed(1)) * one)
    pinned = host.pin_memory()
    out = {}
    for name, src in (("pageable", host), ("pinned", pinned)):
        d = src.to("cuda:0")
        dev.prove(pk, [(0, d, h, [])])  # warm-up
        up, total = [], []
        for _ in range(args.steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d = src.to("cuda:0", non_blocking=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dev.prove(pk, [(0, d, h, [])])
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            up.append(t1 - t0)
            total.append(t2 - t0)
        u, t = sorted(up)[len(up) // 2], sorted(total)[len(total) // 2]
        out[name] = {"upload_ms": round(u * 1e3, 2), "upload_GBps": round(4 * cols * h / u / 1e9, 1),
                     "upload_plus_prove_ms": round(t * 1e3, 2), "cells_per_sec_inclusive": round(cols * h / t, 1)}
    print(json.dumps({"workload": f"2^{args.log_rows} x {cols} BenchmarkAir, app params", "trace_bytes": 4 * cols * h, **out}))


if __name__ == "__main__":
    main()
