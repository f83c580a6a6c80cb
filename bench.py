#!/usr/bin/env python3
"""Headline benchmark: prove a synthetic 2^22-row x 100-column BabyBear AIR (uniform_runner shape,
app parameters, blowup 2) on N MI355X GPUs, one process per GPU.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full `prove` (commit + LogUp-GKR + batch sumcheck + stacked reduction + WHIR) of
one trace that is already resident in HBM. For N > 1 every rank proves its own AIR instance
(independent proofs, weak scaling, no data-path collective -- DESIGN.md "Multi-GPU"); rank 0
prints ONE JSON line. `roofline` is for the dominant kernel (live HIP-event timing on the launch
stream); `cpu_baseline` times the CPU oracle on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-rows", type=int, default=22)
    ap.add_argument("--cols", type=int, default=100)
    ap.add_argument("--log-stacked-height", type=int, default=24)
    ap.add_argument("--cpu-log-rows", type=int, default=18, help="rows of the CPU-baseline sample")
    ap.add_argument("--log-blowup", type=int, default=1,
                    help="1 = app parameters (the headline config); 2 = leaf-like parameters of BASELINE config E")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--zero-trace", action="store_true",
                    help="all-zero trace (what the reference's uniform_runner proves) instead of seeded random bits")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, rendezvous (gloo when there is no GPU) and print n_gpus: the launcher's CPU test")
    args = ap.parse_args()

    # `python bench.py --gpus N` starts the N ranks itself: one child `torch.distributed.run` with one rank
    # per GPU, started BEFORE this process imports torch or touches a GPU (children are fresh processes,
    # nothing that initialised the GPU is ever re-executed); this process only waits and passes the exit
    # code on. Under the driver's own torchrun launch RANK / WORLD_SIZE are set and must agree with --gpus.
    if "RANK" not in os.environ and args.gpus > 1:
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.run(cmd, env=env).returncode)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}", file=sys.stderr)
        sys.exit(2)
    if args.launch_check:
        import torch
        import torch.distributed as dist
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        if world > 1:
            dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
            t = torch.ones(1, device="cuda" if torch.cuda.is_available() else "cpu")
            dist.all_reduce(t)
            assert int(t.item()) == world
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world}), flush=True)
        return

    import torch
    import __graft_entry__ as ge
    import airs

    pkg = ge._load_pkg()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    dev = pkg.Device(local_rank)

    # ---- workload: uniform_runner BenchmarkAir (benchmarks/synthetic/src/bin/uniform_runner.rs) ----
    log_rows, cols = args.log_rows, args.cols
    h = 1 << log_rows
    params = (airs.app_params(args.log_stacked_height) if args.log_blowup == 1
              else airs.stress_params(args.log_stacked_height, args.log_blowup))
    air = airs.uniform_runner_air(cols)
    pk = pkg.ProvingKey(airs.pk_blob(params, [air]))
    gen = torch.Generator(device=f"cuda:{local_rank}")
    gen.manual_seed(0xD15EA5E + rank)
    bits = torch.randint(0, 2, (cols * h,), dtype=torch.int32, device=f"cuda:{local_rank}", generator=gen)
    one_mont = int(pkg.to_mont(np.array([1], dtype=np.uint32))[0])
    trace = bits * one_mont  # boolean cells in Montgomery form, column-major
    if args.zero_trace:
        trace.zero_()
    del bits
    traces = [(0, trace, h, [])]
    cells = cols * h

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    proof = b""
    for _ in range(args.warmup):
        proof, _ = dev.prove(pk, traces)
    pkg.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof, _ = dev.prove(pk, traces)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = pkg.profile_report()
    pkg.profile_enable(False)
    elapsed = pkg.max_over_ranks(elapsed, f"cuda:{local_rank}")

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * cells * args.steps / elapsed
        # dominant kernel = the label with the largest device time
        stages = {k: v[1] / args.steps for k, v in prof.items()}
        dom = max(stages, key=lambda k: stages[k]) if stages else None
        roof = None
        if "merkle.hash_rows" in prof:
            calls, total_ms = prof["merkle.hash_rows"]
            rows = 1 << (args.log_stacked_height + params["log_blowup"])
            width = (cols * h) >> args.log_stacked_height
            # algorithmic bytes per launch: read the codeword once + write one digest per row
            algo_bytes = 4 * rows * width + 32 * rows
            avg_s = total_ms / calls * 1e-3
            achieved = algo_bytes / avg_s / 1e9
            # measured HBM bytes per launch: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
            # command (profiles/r01/pmc_hbm_traffic_*.json, gfx950 x2 fetch correction applied); only valid
            # for the default workload, null otherwise
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01", "pmc_hbm_traffic_v5.json")
            if (log_rows, cols, args.log_stacked_height) == (22, 100, 24) and os.path.exists(pmc):
                k = json.load(open(pmc))["kernels"].get("hash_rows_kernel<1>")
                if k:
                    traffic = (k["fetch_bytes_corrected"] + k["write_bytes"]) // k["launches_per_prove"]
            roof = {"kernel": "hash_rows_kernel<1> (Poseidon2 leaf hash; integer-ALU bound, see DESIGN.md)",
                    "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": "profiles/r01/pmc_hbm_traffic_v5.json (PMC pass, not live)" if traffic else None,
                    "avg_launch_ms": round(total_ms / calls, 3), "algorithmic_bytes": algo_bytes}
            # The kernel is VALU-issue bound, not HBM bound: report that view too. 7233 VALU instructions per
            # permutation (static count, tools/isa_count.py on merkle.hip); nominal issue peak = CUs x 4 SIMDs x
            # 16 lanes x clock = one wave64 instruction per 4 cycles per SIMD.
            perms = rows * ((width + 7) // 8)
            props = torch.cuda.get_device_properties(local_rank)
            valu_peak = props.multi_processor_count * 4 * 16 * 2.4e9
            roof["issue_bound"] = {"valu_instr_per_permutation": 7233, "permutations": perms,
                                   "achieved_Tlaneops": round(7233 * perms / avg_s / 1e12, 2),
                                   "nominal_peak_Tlaneops": round(valu_peak / 1e12, 2),
                                   "frac": round(7233 * perms / avg_s / valu_peak, 3)}
        cpu = None
        if not args.no_cpu_baseline:
            import oracle_lib
            orc = oracle_lib.load()
            lr = args.cpu_log_rows
            small = args.log_stacked_height - (log_rows - lr)
            p2 = airs.app_params(small) if args.log_blowup == 1 else airs.stress_params(small, args.log_blowup)
            blob = airs.pk_blob(p2, [air])
            tr = airs.splitmix64_bits(0xD15EA5E, cols << lr)
            c0 = time.perf_counter()
            orc.prove(blob, [(0, tr, 1 << lr, [])])
            cdt = time.perf_counter() - c0
            cpu = {"value": round((cols << lr) / cdt, 1), "unit": "cells/s", "cores": os.cpu_count(), "kind": "port",
                   "sample": f"C++ oracle (OpenMP), one prove of 2^{lr} x {cols} BenchmarkAir, app params with "
                             f"log_stacked_height {args.log_stacked_height - (log_rows - lr)}, {cdt:.1f} s"}
        # ~60 algorithmic HBM bytes per trace cell end to end for this configuration
        # (6 trace passes + codeword write/read at blowup 2 + the 2^27-leaf fraction tree)
        e2e_gbs = 60.0 * cells / (elapsed / args.steps) / 1e9
        line = {
            "metric": "trace_cells_per_sec", "value": round(value, 1), "unit": "cells/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "prove_ms": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (BabyBear Montgomery; F_p^4 challenges)", "data": "synthetic",
            "config": {"workload": f"uniform_runner BenchmarkAir 2^{log_rows} rows x {cols} cols per GPU, "
                                   f"200 boolean constraints, 13 send/receive pairs, app params "
                                   f"(l_skip 4, n_stack {args.log_stacked_height - 4}, log_blowup {args.log_blowup}, k_whir 4, "
                                   f"PoW {params['logup_pow_bits']}/{params['mu_pow_bits']}/"
                                   f"{params['folding_pow_bits']}/{params['query_phase_pow_bits']} bits), " + ("all-zero trace" if args.zero_trace else "seeded random boolean trace"),
                       "proof_bytes": len(proof), "parallelism": f"replicas x{world}"},
            "roofline": roof, "cpu_baseline": cpu,
            "end_to_end": {"algorithmic_bytes_per_cell": 60, "achieved_GBps": round(e2e_gbs, 1),
                           "frac_of_hbm_peak": round(e2e_gbs / HBM_PEAK_GBS, 4)},
            "stage_ms": {k: round(v, 3) for k, v in sorted(stages.items(), key=lambda kv: -kv[1])},
            "dominant_stage": dom,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
