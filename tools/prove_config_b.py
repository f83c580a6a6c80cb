#!/usr/bin/env python3
"""Proves BASELINE config B `--reps` times on cuda:0 and prints the wall time of each: by default the REAL Keccak-f AIR
(p3-keccak-air's KeccakAir restated in tests/airs.py: 2 633 columns, 3 182 degree-3 constraints, a trace of 43 690 seeded
Keccak-f[1600] permutations + padding, 2^20 rows, app parameters); `--shape-only` proves the Keccak-SHAPED synthetic AIR of
the reference's synthetic benchmark instead (2 634 columns, all-zero trace: rounds 1-2's stand-in). For profiling:

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
    ap.add_argument("--shape-only", action="store_true")
    ap.add_argument("--profile", action="store_true", help="print the per-stage HIP-event times of the last proof")
    args = ap.parse_args()
    import importlib
    import numpy as np
    import torch
    import airs
    pkg = importlib.import_module("stark-backend_amd")
    dev = pkg.Device(0)
    h = 1 << args.log_height
    if args.shape_only:
        segs = airs.load_reth_shapes()
        rec = next(r for s in segs for r in s["airs"] if r[1] == 2634)
        air, width = airs.synthetic_air(rec)
        trace = torch.zeros(width * h, dtype=torch.int32, device="cuda:0")
        what = f"config B shape (synthetic Keccak-shaped AIR) {width} x 2^{args.log_height}"
    else:
        air, width = airs.keccak_air(), airs.KeccakCols.WIDTH
        one = int(pkg.to_mont(np.array([1], dtype=np.uint32))[0])
        trace, _ = airs.keccak_trace(airs.keccak_inputs(h // 24), args.log_height, device="cuda:0", mont_one=one)
        what = f"config B (Keccak-f AIR, {h // 24} permutations) {width} x 2^{args.log_height}"
    blob = airs.pk_blob(airs.app_params(24), [air])
    pk = pkg.ProvingKey(blob)
    times = []
    for i in range(args.reps):
        if args.profile and i == args.reps - 1:
            pkg.profile_enable(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.prove(pk, [(0, trace, h, [])])
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    out = {"workload": what, "prove_ms": [round(t, 2) for t in times], "cells_per_s": width * h / (min(times) / 1e3)}
    if args.profile:
        out["stage_ms"] = {k: round(v[1], 3) for k, v in sorted(pkg.profile_report().items(), key=lambda kv: -kv[1][1])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
