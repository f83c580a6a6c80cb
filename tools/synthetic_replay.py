#!/usr/bin/env python3
"""Replay of the reference's synthetic workload (benchmarks/synthetic/src/bin/synthetic_runner.rs):
every sampled segment of the captured reth-block profile is rebuilt as synthetic AIRs with all-zero
traces resident on the GPU, and `prove` alone is timed (keygen, trace allocation excluded), exactly
the reference's `total_prove_ms`.

    python tools/synthetic_replay.py [--every 10] [--max-log-height 22] [--out scorecard.json]

--every 1 is the reference's "rigorous full sweep" (209/209 segments; README: ~43.6 s of GPU prove
on an unnamed NVIDIA GPU with the CUDA backend). The reference's 0.1-fraction sample is a seeded
Rust StdRng shuffle that cannot be reproduced here, so --every 10 (21 segments: 0, 10, ..., 200) is
a same-size sample, not the same segments.
Parameters: app_params_with_100_bits_security(max_log_height) with the captured global constraint
degree 4 (the profile predates APP_MAX_CONSTRAINT_DEGREE = 3).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(argv=None, verify=None):
    """verify(blob, proof): optional checker called on every proof (tests/replay_verified.py passes the oracle's
    verifier restatement; this tool itself never touches oracle/)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--max-log-height", type=int, default=22)
    ap.add_argument("--out", default=None)
    ap.add_argument("--segment", type=int, default=None, help="replay only this segment index, with the stage profile")
    args = ap.parse_args(argv)

    import torch
    import __graft_entry__ as ge
    import airs

    pkg = ge._load_pkg()
    dev = pkg.Device(0)
    segs = airs.load_reth_shapes()
    params = airs.app_params(args.max_log_height)
    params["max_constraint_degree"] = 4
    picked = list(range(0, len(segs), args.every)) if args.segment is None else [args.segment]
    results, total_ms, total_cells = [], 0.0, 0
    for n, si in enumerate(picked):
        air_list, shapes = airs.synthetic_segment(segs[si], args.max_log_height)
        blob = airs.pk_blob(params, air_list)
        pk = pkg.ProvingKey(blob)
        traces = [(i, torch.zeros(w << lh, dtype=torch.int32, device="cuda:0"), 1 << lh, []) for i, lh, w in shapes]
        cells = sum(w << lh for _, lh, w in shapes)
        if n == 0:
            dev.prove(pk, traces)  # warm-up: allocator, twiddles
        torch.cuda.synchronize()
        if args.segment is not None:
            pkg.profile_enable(True)
        t0 = time.perf_counter()
        proof, _ = dev.prove(pk, traces)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        if args.segment is not None:
            prof = pkg.profile_report()
            pkg.profile_enable(False)
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:25]:
                print(f"   {k:32s} calls {v[0]:5d}  {v[1]:9.3f} ms", file=sys.stderr)
            print(f"   sum of profiled device time: {sum(v[1] for v in prof.values()):.1f} ms", file=sys.stderr)
        if verify is not None:
            verify(blob, proof)
        total_ms += ms
        total_cells += cells
        results.append({"segment_idx": segs[si]["idx"], "num_airs": len(air_list), "main_cells": cells,
                        "prove_ms": round(ms, 2), "proof_bytes": len(proof)})
        print(f"[{n + 1}/{len(picked)}] segment {segs[si]['idx']}: {len(air_list)} AIRs, {cells / 1e6:.1f} M cells, "
              f"{ms:.1f} ms", file=sys.stderr)
        del traces, pk
    mem = pkg.memory_stats()
    free_b, total_b = torch.cuda.mem_get_info(0)
    card = {"workload": "reth-block-23992138 synthetic replay", "every": args.every,
            # the library's allocator over the whole sweep (no trim between segments): peak of what it handed out, what its
            # free lists hold at the end, and what the driver reports in use on the device (incl. torch's own pool)
            "allocator_peak_live_GiB": round(mem["peak"] / 2**30, 2), "allocator_cached_end_GiB": round(mem["cached"] / 2**30, 2),
            "device_in_use_end_GiB": round((total_b - free_b) / 2**30, 2),
            "max_log_height": args.max_log_height, "sampled_segments": len(picked), "total_segments": len(segs),
            "total_prove_ms": round(total_ms, 1), "total_main_cells": total_cells,
            "cells_per_sec": round(total_cells / (total_ms / 1e3), 1), "verified": verify is not None,
            "device": torch.cuda.get_device_name(0), "segments": results}
    text = json.dumps(card)
    if args.out:
        open(args.out, "w").write(text)
    print(json.dumps({k: v for k, v in card.items() if k != "segments"}))


if __name__ == "__main__":
    main()
