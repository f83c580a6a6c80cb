#!/usr/bin/env python3
"""Parity fuzzing of the SHARDED prover on one GPU (test infrastructure, not collected by pytest): random multi-AIR
contexts, a random number of logical ranks (2-8 processes sharing the GPU), random assignment of the traces to ranks
(a rank may own none), random memory-bounded modes and with / without the all-to-all form of the mu-batch sum; the joint
proof must be byte-identical to the oracle's single proof.

    python tests/fuzz_shard.py --cases 40 --seed 1      (~2 s per case: the worker processes start up)
"""
import argparse
import os
import pathlib
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    os.environ["ORC_SKIP_DEBUG_ASSERTS"] = "1"
    os.environ["OVM_TUNING_LIVE"] = "1"  # (the knobs drawn per case are re-read at every prover entry)
    import airs
    import oracle_lib
    import shard_cases
    from test_gpu_shard import run_sharded
    orc = oracle_lib.load()
    rng = np.random.default_rng(args.seed)
    bad = 0
    with_absent = 0
    for case in range(args.cases):
        seed, world = int(rng.integers(0, 1 << 30)), int(rng.integers(2, 9))
        save_memory, alltoall = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        bulk = str(rng.choice(["ipc", "ipc", "shm"]))
        params, air_list, traces, owners = shard_cases.make(f"fuzz:{seed}", world)
        blob = airs.pk_blob(params, air_list)
        try:
            want = orc.prove(blob, [t[:4] for t in traces])["proof"]
        except RuntimeError as e:
            print(f"case {case}: skipped (the protocol rejects the context: {e})", flush=True)
            continue
        with tempfile.TemporaryDirectory() as tmp:
            got = run_sharded(f"fuzz:{seed}", world, pathlib.Path(tmp), save_memory=save_memory, alltoall=alltoall, bulk=bulk)
        ok = got == want
        bad += not ok
        n_absent = len(air_list) - len(traces)
        idle = world - len(set(owners))
        with_absent += n_absent > 0
        print(f"case {case}: {'ok' if ok else 'MISMATCH'} seed={seed} world={world} airs={len(air_list)} absent={n_absent} "
              f"ranks_without_a_trace={idle} owners={owners} "
              f"save_memory={save_memory} alltoall={alltoall} bulk={bulk}", flush=True)
    print(f"{bad} mismatches in {args.cases} cases ({with_absent} proved contexts had absent AIRs)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
