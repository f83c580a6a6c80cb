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


def replay_engines(args, verify):
    """K engines over the sweep: the main thread builds each segment's key and (all-zero) traces and hands it to the first free
    engine; an engine times nothing but its `prove` calls, as the sequential sweep does."""
    import queue
    import threading
    import torch
    import __graft_entry__ as ge
    import airs

    pkg = ge._load_pkg()
    pkg.Device(0)
    segs = airs.load_reth_shapes()
    params = airs.app_params(args.max_log_height)
    params["max_constraint_degree"] = 4
    picked = list(range(0, len(segs), args.every))
    work = queue.Queue(maxsize=args.engines + 1)
    results, errors, busy = [], [], [0.0] * args.engines
    lock = threading.Lock()

    def engine(e):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=0)):
                d = pkg.Device(0)
                first = True
                while True:
                    item = work.get()
                    if item is None:
                        return
                    si, blob, pk, traces, cells, n_airs = item
                    if first:
                        d.prove(pk, traces)  # this engine's allocator and twiddles, as the sequential sweep's warm-up
                        first = False
                    torch.cuda.current_stream().synchronize()
                    t0 = time.perf_counter()
                    proof, _ = d.prove(pk, traces)
                    torch.cuda.current_stream().synchronize()
                    ms = (time.perf_counter() - t0) * 1e3
                    if verify is not None:
                        verify(blob, proof)
                    with lock:
                        busy[e] += ms
                        results.append({"segment_idx": segs[si]["idx"], "engine": e, "num_airs": n_airs, "main_cells": cells,
                                        "prove_ms": round(ms, 2), "proof_bytes": len(proof)})
                    del traces, pk, item
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex))
            while work.get() is not None:
                pass

    threads = [threading.Thread(target=engine, args=(e,)) for e in range(args.engines)]
    for t in threads:
        t.start()
    w0 = time.perf_counter()
    for si in picked:
        air_list, shapes = airs.synthetic_segment(segs[si], args.max_log_height)
        blob = airs.pk_blob(params, air_list)
        pk = pkg.ProvingKey(blob)
        traces = [(i, torch.zeros(w << lh, dtype=torch.int32, device="cuda:0"), 1 << lh, []) for i, lh, w in shapes]
        torch.cuda.current_stream().synchronize()  # (this thread's fills only: a device-wide wait would stall behind the engines)
        work.put((si, blob, pk, traces, sum(w << lh for _, lh, w in shapes), len(air_list)))
    for _ in threads:
        work.put(None)
    for t in threads:
        t.join()
    wall = (time.perf_counter() - w0) * 1e3
    if errors:
        raise SystemExit("engine failed: " + "; ".join(errors))
    total_cells = sum(r["main_cells"] for r in results)
    card = {"workload": "reth-block-23992138 synthetic replay, throughput mode", "engines": args.engines, "every": args.every,
            "sampled_segments": len(picked), "total_segments": len(segs), "sum_of_prove_ms": round(sum(r["prove_ms"] for r in results), 1),
            "engine_busy_ms": [round(b, 1) for b in busy], "makespan_of_prove_ms": round(max(busy), 1),
            "wall_ms_incl_keygen_and_trace_allocation": round(wall, 1), "total_main_cells": total_cells,
            "cells_per_sec_at_makespan": round(total_cells / (max(busy) / 1e3), 1), "verified": verify is not None,
            "device": torch.cuda.get_device_name(0), "segments": sorted(results, key=lambda r: r["segment_idx"])}
    if args.out:
        open(args.out, "w").write(json.dumps(card))
    print(json.dumps({k: v for k, v in card.items() if k != "segments"}))


def main(argv=None, verify=None):
    """verify(blob, proof): optional checker called on every proof (tests/replay_verified.py passes the oracle's
    verifier restatement; this tool itself never touches oracle/)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--max-log-height", type=int, default=22)
    ap.add_argument("--out", default=None)
    ap.add_argument("--segment", type=int, default=None, help="replay only this segment index, with the stage profile")
    ap.add_argument("--profile", action="store_true",
                    help="after each timed proof, prove the segment once more with the stage profile on and add the stage times up "
                         "over the sweep (`stage_ms` in the card: where a whole block's proving time goes)")
    ap.add_argument("--engines", type=int, default=1,
                    help="K > 1: the reference's throughput mode (crates/cuda-backend/examples/keccakf.rs:55-131): K host threads, each "
                         "with its own stream and engine, prove the segments of the sweep concurrently (segments are independent "
                         "proofs); reports per-engine busy time and the makespan next to the sum of the prove times")
    args = ap.parse_args(argv)
    if args.engines > 1 and args.segment is None:
        return replay_engines(args, verify)

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
    stage_ms = {}
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
        if args.profile and args.segment is None:
            pkg.profile_enable(True)
            dev.prove(pk, traces)
            torch.cuda.synchronize()
            for k, v in pkg.profile_report().items():
                stage_ms[k] = stage_ms.get(k, 0.0) + v[1]
            pkg.profile_enable(False)
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
    if stage_ms:
        card["stage_ms"] = {k: round(v, 2) for k, v in sorted(stage_ms.items(), key=lambda kv: -kv[1])}
    text = json.dumps(card)
    if args.out:
        open(args.out, "w").write(text)
    print(json.dumps({k: v for k, v in card.items() if k != "segments"}))


if __name__ == "__main__":
    main()
