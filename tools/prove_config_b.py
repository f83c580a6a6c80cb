#!/usr/bin/env python3
"""Proves the BASELINE config B shape (one Keccak-shaped synthetic AIR, 2634 columns x 2^20 rows, app
parameters) `--reps` times on cuda:0 and prints the wall time of each. For profiling:

    rocprofv3 --kernel-trace -d gpurun_out/cfgb -- python3 tools/prove_config_b.py --reps 2
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--log-height", type=int, default=20)
    args = ap.parse_args()
    import importlib
    import torch
    import airs
    pkg = importlib.import_module("stark-backend_amd")
    segs = airs.load_reth_shapes()
    rec = next(r for s in segs for r in s["airs"] if r[1] == 2634)
    air, width = airs.synthetic_air(rec)
    blob = airs.pk_blob(airs.app_params(24), [air])
    pk = pkg.ProvingKey(blob)
    dev = pkg.Device(0)
    h = 1 << args.log_height
    trace = torch.zeros(width * h, dtype=torch.int32, device="cuda:0")
    times = []
    for _ in range(args.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.prove(pk, [(0, trace, h, [])])
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"workload": f"config B shape {width} x 2^{args.log_height}", "prove_ms": times,
                      "cells_per_s": width * h / (min(times) / 1e3)}))


if __name__ == "__main__":
    main()
