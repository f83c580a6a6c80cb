#!/usr/bin/env python3
"""Headline benchmark: prove synthetic 2^22-row x 100-column BabyBear AIRs (uniform_runner shape, app
parameters, blowup 2) on N MI355X GPUs, one process per GPU.

    python bench.py --gpus N --steps K --warmup W

N = 1: a "step" is one full `prove` (commit + LogUp-GKR + batch sumcheck + stacked reduction + WHIR) of one
BenchmarkAir trace that is already resident in HBM -- BASELINE config D as the reference's own bench proves
it (one common-main commitment).
N > 1: ONE JOINT PROOF per step over N AIRs, one per GPU (BASELINE config D in its multi-GPU form, SURVEY
§8(e)): each AIR's 100 columns are its own cached-main commitment (committed INSIDE the timed region, on its
own GPU), the per-round sumcheck values of all ranks are combined through shared memory, and the only bulk
collective is the RCCL all-gather of the mu-batched WHIR vector (plus 16 MB per rank of common-main
columns at commit time). `value` = cells of all N AIRs / time of the joint proof (weak scaling).
`python bench.py --gpus N` starts the N ranks itself; under the driver's torchrun launch it is one of them.
Rank 0 prints ONE JSON line. `roofline` is per kernel class with the dominant kernel first (live HIP-event
timing on the launch stream); `cpu_baseline` times the CPU oracle on a bounded sample of the same workload.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

# (multi-process GPU work on this pool: the host driver only supports dmabuf IPC; harmless when already set)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# (the shared-memory exchange waits up to OVM_SHARD_TIMEOUT_S = 300 s by default for a SILENT peer -- long enough for a
# first step on 8 GPUs, where one rank may still be creating its RCCL communicators or loading code objects; a peer that
# fails is announced through the abort word and noticed within milliseconds)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def launch_ranks(args):
    """`python bench.py --gpus N`: one child `torch.distributed.run` with one rank per GPU, started BEFORE this
    process imports torch or touches a GPU (children are fresh processes; nothing that initialised the GPU is
    ever re-executed). This process only waits and passes the exit code on."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.run(cmd, env=env).returncode)


# The ISSUE roof of the integer kernels: a wave64 integer VALU instruction occupies its SIMD for ~4.05-4.1 shader
# cycles at 2.4 GHz (measured on the Poseidon2 instruction stream, tools/ubench_poseidon2.hip; per instruction,
# tools/ubench_int64.hip: multiplies, 64-bit multiply-adds and adds, min ~4.3-4.7 in a dependent-chain harness, v_add_u32 /
# v_sub_u32 / 32-bit shifts ~2.5-2.9; the guide's 2-cycle figure is for v_fma_f32), so the chip issues about
# CUs x 4 SIMDs x 2.4e9 / 4.05 wave-level VALU instructions per second -- a little more for a stream rich in plain adds.
ISSUE_CYCLES_PER_WAVE_VALU = 4.05
SHADER_CLOCK_HZ = 2.4e9
# which kernels make up a class (prefixes of the demangled names in profiles/*/sq_valu*.json)
CLASS_KERNELS = {
    "ntt (eval_to_coeff + DIF passes)": ("eval_to_coeff", "dif16_pass_kernel<bb::Fp", "dif_pass_kernel<bb::Fp", "bitrev"),
    "gkr (leaves + tree + layer sumchecks)": ("gkr_", "eq_heap", "eq_double", "eq_split", "sum_partials"),
    "folds (fold_ple + MLE rounds)": ("fold_ple", "mle_round", "fold_mle"),
    "whir mu-batch": ("whir_mu_batch", "zeta_pass"),
    "poseidon2 (row hash + compress + grind)": ("hash_rows", "compress_layer", "grind"),
}


def checked_digest_verdict(args, world, digest):
    """Does the proof this run timed equal the one the GPU suite compares byte for byte with the oracle
    (tests/test_gpu_parity_at_scale.py::test_config_d_full_byte_identical pins its sha256 in
    tests/golden/config_d_proof_sha256.txt)? true / false for the headline workload in its plain form with the seeded
    trace; null for any other workload (no checked digest exists for it)."""
    headline = (world == 1 and not args.partitioned and not args.zero_trace and args.num_airs == 1 and
                (args.log_rows, args.cols, args.log_stacked_height, args.log_blowup) == (22, 100, 24, 1))
    path = os.path.join(ROOT, "tests", "golden", "config_d_proof_sha256.txt")
    if not headline or not os.path.exists(path):
        return None
    return open(path).read().strip() == digest


def newest_profile(pattern, rounds=(6, 5, 4, 3, 2)):
    """The committed counter file of the newest round that has one: profiles/rNN/<pattern with {r} = rNN>, or ""."""
    for n in rounds:
        q = os.path.join(ROOT, "profiles", f"r{n:02d}", pattern.format(r=f"r{n:02d}"))
        if os.path.exists(q):
            return q
    return ""


def load_sq_valu():
    q = newest_profile("sq_valu_{r}.json")
    if q:
        return json.load(open(q))["kernels"], os.path.relpath(q, ROOT)
    return None, None


def roofline_classes(prof, steps, rows, width, cells, log_stacked, gkr_leaves, n_cus=256, headline=False):
    """Per kernel class, BOTH roofs: `frac` = algorithmic bytes per proof (SURVEY §8(d) formulas) /
    device time of the class's launches (HIP events on the launch stream) against the 8 TB/s HBM peak, and `issue_frac` =
    the wave-level VALU instructions the class's kernels execute per proof (SQ_INSTS_VALU, a property of the workload: read
    from the committed counter pass like `traffic`) / the same device time against the measured issue ceiling. `bound` names
    the roof the class sits closer to."""
    def ms(label):
        return prof[label][1] / steps if label in prof else None
    sq, sq_src = load_sq_valu() if headline else (None, None)
    issue_peak = n_cus * 4 * SHADER_CLOCK_HZ / ISSUE_CYCLES_PER_WAVE_VALU
    E_code, E_msg = rows * width, (rows // 2) * width
    out = {}
    cls = {
        # chunk-iNTT+zeta 8 B/elem over the stacked matrix; big NTT reported against 2 passes = 16 B/elem of the codeword
        "ntt (eval_to_coeff + DIF passes)": (8 * E_msg + 16 * E_code, ["ntt.eval_to_coeff", "ntt.dif_passes"]),
        # leaf gen 4 B/cell read + 32 B/leaf written; tree build one read+write sweep 2*32 B/leaf;
        # layer sumchecks: read all, write half per fold = 1.5 * 32 B per fraction over sum_j 2^j ~ 2 * leaves
        "gkr (leaves + tree + layer sumchecks)": (4 * cells + 32 * gkr_leaves + 64 * gkr_leaves + 48 * 2 * gkr_leaves,
                                                  ["gkr.leaves", "gkr.build_tree", "gkr.layer_sumchecks"]),
        # fold_ple: 4 B/cell + 16 B per 2^l_skip cells; MLE folds 1.5 * 16 B per element over a halving series
        "folds (fold_ple + MLE rounds)": (4 * cells + cells + 48 * (cells // 16), ["constraints.fold_ple", "constraints.mle_rounds"]),
        "whir mu-batch": (4 * E_msg + 16 * (1 << log_stacked), ["whir.mu_batch"]),
        # not an HBM class (integer-ALU bound): listed so that its issue fraction stands beside the others'
        "poseidon2 (row hash + compress + grind)": (4 * rows * width + 32 * rows // 16 + 96 * 2 * (rows // 16),
                                                    ["merkle.hash_rows", "merkle.hash_rows_ext", "merkle.compress_layers", "grind"]),
    }
    for name, (algo, labels) in cls.items():
        t = [ms(l) for l in labels]
        if any(x is None for x in t):
            continue
        tot = sum(t)
        ent = {"algorithmic_bytes": int(algo), "ms": round(tot, 3), "achieved_GBps": round(algo / (tot * 1e-3) / 1e9, 1),
               "frac": round(algo / (tot * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if sq:
            insts = sum(v["valu_wave_insts"] for k, v in sq.items() if k.startswith(CLASS_KERNELS[name]))
            if insts:
                ent["valu_wave_insts"] = int(insts)
                ent["issue_frac"] = round(insts / (tot * 1e-3) / issue_peak, 4)
                top = max(ent["issue_frac"], ent["frac"])
                ent["bound"] = ("valu issue" if ent["issue_frac"] > ent["frac"] else "hbm") if top >= 0.5 else \
                    "latency (neither roof above one half: sequential small launches and host round trips)"
        out[name] = ent
    if sq:
        out["_issue_roof"] = {"peak_wave_valu_per_s": round(issue_peak, 1), "cycles_per_wave_valu": ISSUE_CYCLES_PER_WAVE_VALU,
                              "clock_hz": SHADER_CLOCK_HZ, "valu_counts_source": sq_src}
    return out


def dry_run(args, pkg, dist, shard, rank, world, devname, bulk_ok):
    """`--dry-run`: what the first multi-GPU run needs to be boring. After the rendezvous and the
    pre-flight the caller already did: the two bulk steps of the mu-batch sum at the FULL vector size through the very
    callbacks the prover calls (all-to-all of slices + all-gather of the summed slices; and the whole-vector all-gather
    fallback), with the carrier the pre-flight chose and with the library's IPC pulls, and -- on the nccl backend -- one
    timed transfer per peer (one xGMI link each). Rank 0 prints one JSON line."""
    import ctypes
    import torch
    nbytes = 16 << args.log_stacked_height          # one rank's mu-batch vector: 2^(l_skip + n_stack) extension elements
    slice_b = -(-(nbytes // 16) // world) * 16        # ceil(2^m / N) elements per pair, as the prover cuts it
    dev = torch.device(devname)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    send = torch.empty(slice_b * world, dtype=torch.uint8, device=dev).fill_(rank + 1)
    recv = torch.empty(slice_b * world, dtype=torch.uint8, device=dev)
    whole = torch.empty(nbytes * world, dtype=torch.uint8, device=dev) if nbytes * world <= (32 << 30) else None

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        return pkg.max_over_ranks(dt, "cpu" if args.one_gpu else devname) * 1e3

    res = {}
    carriers = [shard.bulk] + (["ipc"] if shard.bulk != "ipc" else [])
    for carrier in carriers:
        shard.bulk = carrier
        shard._set_abi()
        a2a = timed(lambda: shard._alltoall_device(None, send.data_ptr(), recv.data_ptr(), slice_b, stream))
        ag = timed(lambda: shard._allgather_device(None, send.data_ptr(), recv.data_ptr(), slice_b, stream))
        ent = {"alltoall_ms": round(a2a, 3), "allgather_of_slices_ms": round(ag, 3),
               "bytes_in_per_gpu_per_phase": slice_b * (world - 1),
               "per_link_GBps_alltoall": round(slice_b / (a2a * 1e-3) / 1e9, 1) if a2a > 0 else None}
        ok = bool((recv.view(world, slice_b)[:, 0].cpu() == torch.arange(1, world + 1, dtype=torch.uint8)).all())
        ent["data_ok"] = ok
        if whole is not None:
            mine = whole[:nbytes]
            big = timed(lambda: shard._allgather_device(None, send.data_ptr(), whole.data_ptr(), min(nbytes, send.numel()), stream), reps=2)
            ent["whole_vector_allgather_ms"] = round(big, 3)
        res["torch/" + dist.get_backend() if carrier == "torch" else carrier] = ent
    links = None
    if dist is not None and dist.get_backend() == "nccl":
        links = {}
        buf_s = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        buf_r = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        for d in range(1, world):
            to, frm = (rank + d) % world, (rank - d) % world

            def xfer():
                ops = [dist.P2POp(dist.isend, buf_s, to), dist.P2POp(dist.irecv, buf_r, frm)]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            ms = timed(xfer)
            links[f"+{d}"] = {"ms_64MiB": round(ms, 3), "GBps_each_way": round((64 << 20) / (ms * 1e-3) / 1e9, 1)}
    every = [None] * world
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    dist.all_gather_object(every, {"rank": rank, "device_index": torch.cuda.current_device(), "device_name": props.name,
                                   "device_uuid": str(getattr(props, "uuid", ""))})
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                          "bulk_preflight_ok": bool(bulk_ok), "vector_bytes": nbytes, "slice_bytes": slice_b,
                          "exchanges": res, "p2p_links": links, "ranks": every}), flush=True)
    shard.close()
    dist.destroy_process_group()


def engines_run(args, pkg, torch, pk, traces, cells, barrier, local_rank):
    """The reference's throughput mode on one GPU (crates/cuda-backend/examples/keccakf.rs:55-131: N OS threads x N engines
    x N streams, one shared proving key): K host threads, each with its own stream and `Device` (the library keeps its
    mailboxes, pools and side stream per thread), prove the SAME resident trace `steps` times each; K = 1 is measured the
    same way for the comparison. Returns the record for the JSON line."""
    import threading
    out = {"reference": "crates/cuda-backend/examples/keccakf.rs:55-131", "runs": []}
    steps = max(2, min(args.steps, 8))
    for k in sorted({1, args.engines}):
        per_engine, errors, proofs = [0.0] * k, [], [None] * k
        start = threading.Barrier(k + 1)

        def worker(i):
            try:
                # --engines-priority first-high: engine 0's stream gets the high dispatch priority, the others the default one
                prio = -1 if (args.engines_priority == "first-high" and i == 0) else 0
                with torch.cuda.stream(torch.cuda.Stream(device=local_rank, priority=prio)):
                    d = pkg.Device(local_rank)
                    d.prove(pk, traces)            # this thread's pools, mailboxes and twiddles: outside the timing
                    torch.cuda.current_stream().synchronize()
                    start.wait()
                    if args.engines_stagger_ms > 0 and i > 0:
                        time.sleep(i * args.engines_stagger_ms * 1e-3)
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        proofs[i], _ = d.prove(pk, traces)
                    torch.cuda.current_stream().synchronize()
                    per_engine[i] = (time.perf_counter() - t0) / steps * 1e3
                    pkg._lib().ovm_hip_thread_shutdown()
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))
                try:
                    start.abort()
                except Exception:  # noqa: BLE001
                    pass
        ts = [threading.Thread(target=worker, args=(i,)) for i in range(k)]
        for t in ts:
            t.start()
        try:
            start.wait()
        except threading.BrokenBarrierError:
            pass
        t0 = time.perf_counter()
        for t in ts:
            t.join()
        wall = time.perf_counter() - t0
        barrier()
        rec = {"engines": k, "steps_per_engine": steps, "stagger_ms": args.engines_stagger_ms, "stream_priority": args.engines_priority,
               "wall_ms": round(wall * 1e3, 2),
               "cells_per_s": round(k * steps * cells / wall, 1), "per_engine_prove_ms": [round(x, 3) for x in per_engine],
               # every engine's own rate added up (what a long-running pipeline sees; `cells_per_s` also pays the start-up stagger)
               "cells_per_s_steady": round(sum(cells / (x * 1e-3) for x in per_engine if x > 0), 1),
               "identical_proofs": len({hashlib.sha256(p or b"").hexdigest() for p in proofs}) == 1}
        if errors:
            rec["errors"] = errors
        out["runs"].append(rec)
    base = out["runs"][0]["cells_per_s"]
    out["gain_over_one_engine"] = round(out["runs"][-1]["cells_per_s"] / base, 4) if base else None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)  # (20 x ~75 ms: the first few timed steps of a cold process read 2-3 % high)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-rows", type=int, default=22)
    ap.add_argument("--cols", type=int, default=100)
    ap.add_argument("--num-airs", type=int, default=1,
                    help="N = 1 only: this many BenchmarkAirs of the same shape in ONE proof (uniform_runner's --num-airs; "
                         "BASELINE config C is --num-airs 4 --log-rows 20 --log-stacked-height 24)")
    ap.add_argument("--log-stacked-height", type=int, default=24)
    ap.add_argument("--cpu-log-rows", type=int, default=18, help="rows of the CPU-baseline sample")
    ap.add_argument("--log-blowup", type=int, default=1,
                    help="1 = app parameters (the headline config); 2 = leaf-like parameters of BASELINE config E")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--zero-trace", action="store_true",
                    help="all-zero trace (what the reference's uniform_runner proves) instead of seeded random bits")
    ap.add_argument("--bulk", default="rccl", choices=["rccl", "ipc", "shm"],
                    help="N > 1: the bulk device exchange -- torch.distributed (RCCL; default), the library's own IPC pulls, "
                         "or host staging through shared memory (the pre-flight falls back along this chain by itself)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, rendezvous (gloo when there is no GPU) and print n_gpus: the launcher's CPU test")
    ap.add_argument("--profile-steps", type=int, default=-1,
                    help="fully profiled steps AFTER the timed region that stage_ms / roofline.classes come from (default: "
                         "min(steps, 3); 0: none -- what tools/profile_round.sh uses so that the process makes exactly "
                         "warmup + steps proofs)")
    ap.add_argument("--no-pin", action="store_true",
                    help="do not pin this rank's process to two physical host cores (default: two distinct physical cores on the "
                         "NUMA node of the rank's GPU, stark-backend_amd/placement.py: the prover's host thread spins on device "
                         "flags between ~900 launches per proof, and staying on one core pair is worth ~1 ms of 76 at config D)")
    ap.add_argument("--partitioned", action="store_true",
                    help="N = 1: make the PARTITIONED form the headline -- the AIR's 100 columns as a cached-main commitment "
                         "(committed inside the timed region) + a 1-column common main, i.e. exactly what each rank of an N > 1 "
                         "run proves. Without the flag the N = 1 line still reports it as `n1_partitioned_form`, so that the "
                         "N = 1 point of a scaling curve can be read in the same units as N >= 2")
    ap.add_argument("--engines-stagger-ms", type=float, default=0.0,
                    help="--engines K: engine i starts its timed proofs i x this many milliseconds after engine 0 (two provers in "
                         "lock step hash at the same time; half a proof apart, one's VALU-bound hashing runs beside the other's "
                         "memory- and latency-bound stages)")
    ap.add_argument("--engines-priority", default="equal", choices=["equal", "first-high"],
                    help="--engines K: 'first-high' gives engine 0's stream the high dispatch priority (one proof at nearly full "
                         "speed, the others in what it leaves free) instead of K equal streams")
    ap.add_argument("--no-other-form", action="store_true",
                    help="N = 1: skip the few extra steps that measure the OTHER form (partitioned beside plain, or plain beside "
                         "--partitioned): tools/profile_round.sh uses it so that the process makes exactly warmup + steps proofs")
    ap.add_argument("--engines", type=int, default=0,
                    help="N = 1: after the timed region, K host threads x K streams prove the same resident trace concurrently "
                         "(the reference's throughput mode, crates/cuda-backend/examples/keccakf.rs:55-131); reported as the "
                         "extra key `throughput`, the headline is unchanged")
    ap.add_argument("--dry-run", action="store_true",
                    help="N > 1: rendezvous, the pre-flight of the bulk exchange, then the two collectives of the mu-batch sum and "
                         "one transfer per peer at the FULL vector size (2^log_stacked_height x 16 B = 268 MB at config D), timed; "
                         "prints one JSON line with per-link GB/s and proves nothing")
    ap.add_argument("--one-gpu", action="store_true",
                    help="N > 1 ranks share GPU 0 with the gloo backend (host-staged bulk exchange): exercises the sharded "
                         "bench path on a 1-GPU box at a reduced --log-rows; not a benchmark")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        launch_ranks(args)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}", file=sys.stderr)
        sys.exit(2)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        import torch
        import torch.distributed as dist
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
    # Host placement: the prover thread (it spins on host-mapped flags between launches) and its helper thread stay on two
    # PHYSICAL cores -- measured on the 2 x 64-core box (profiles/r03/host_pinning.txt): unpinned 76.0-76.6 ms, cores 8,9
    # 74.9-75.4 ms, two SMT siblings of one core 78 ms. Done after the imports (the OpenMP runtime the CPU baseline uses
    # sizes itself from the mask it sees when it loads) and undone before the CPU baseline. Stated in the JSON line.
    # The pair: two distinct PHYSICAL cores on the NUMA node of this rank's GPU (placement.py reads sysfs; VERDICT r3: the
    # fixed pair 8 + 2r put all eight ranks on socket 0 and assumed that CPUs c, c + 1 are different cores).
    host_affinity, all_cpus, placement_rec = None, None, {"cpus": None, "reason": "--no-pin"}
    if hasattr(os, "sched_getaffinity"):
        all_cpus = sorted(os.sched_getaffinity(0))
    if not args.no_pin:
        from stark_backend_amd import placement
        placement_rec = placement.pin_this_process(local_rank, slot_rank=int(os.environ.get("LOCAL_RANK", "0")) if args.one_gpu else None)
        host_affinity = placement_rec["cpus"]
    dist = None
    devname = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if args.one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(devname))
    dev = pkg.Device(local_rank)

    # ---- workload: uniform_runner BenchmarkAir (benchmarks/synthetic/src/bin/uniform_runner.rs) ----
    log_rows, cols = args.log_rows, args.cols
    h = 1 << log_rows
    params = (airs.app_params(args.log_stacked_height) if args.log_blowup == 1
              else airs.stress_params(args.log_stacked_height, args.log_blowup))
    one_mont = int(pkg.to_mont(np.array([1], dtype=np.uint32))[0])
    gen = torch.Generator(device=devname)
    gen.manual_seed(0xD15EA5E + rank)

    def bits(n):
        t = torch.randint(0, 2, (n,), dtype=torch.int32, device=devname, generator=gen) * one_mont
        if args.zero_trace:
            t.zero_()
        return t

    cells = cols * h  # per AIR (N > 1: the 100 cached-main columns; the extra 1-column common main is not counted)
    shard = None
    step_partitioned = None
    if world == 1:
        air = airs.uniform_runner_air(cols)
        pk = pkg.ProvingKey(airs.pk_blob(params, [air] * args.num_airs))
        traces = [(a, bits(cols * h), h, []) for a in range(args.num_airs)]
        if args.num_airs == 1:
            # the same AIR as ONE rank of an N > 1 run proves it: 100 columns as a cached-main commitment
            # made inside the step + a 1-column common main; the cells counted are the 100 columns', as for N > 1
            pk_p = pkg.ProvingKey(airs.pk_blob(params, [airs.uniform_runner_air_partitioned(cols)]))
            cached_p = traces[0][1] if not args.partitioned else bits(cols * h)
            common_p = bits(h)

            def step_partitioned():
                handle = dev.commit_matrix(pk_p, cached_p, cols, h)
                out = dev.prove(pk_p, [(0, common_p, h, [], [handle])])
                handle.free()
                return out
        cells *= args.num_airs

        def step_plain():
            return dev.prove(pk, traces)
        step = step_partitioned if args.partitioned and step_partitioned else step_plain
    else:
        # one partitioned BenchmarkAir per rank: 100 columns as a cached-main commitment + a 1-column common main
        air = airs.uniform_runner_air_partitioned(cols)
        pk = pkg.ProvingKey(airs.pk_blob(params, [air] * world))
        name = [f"/ovm-bench-{os.getpid()}-{int(time.time() * 1e3) & 0xFFFFFF}"]
        dist.broadcast_object_list(name, src=0)
        shard = pkg.Shard(rank, world, name[0], bulk={"rccl": "torch"}.get(args.bulk, args.bulk))
        # pre-flight of the RCCL exchanges on rank-dependent data; if any rank sees wrong bytes or an error, every rank
        # switches to host staging through the shared-memory segment (and the JSON line says so)
        bulk_ok = shard.check_bulk()
        if not bulk_ok and rank == 0:
            print("bench.py: the device exchange pre-flight FAILED; staging the bulk exchange through shared memory", file=sys.stderr)
        cached, common = bits(cols * h), bits(h)
        owners = list(range(world))

        def step():
            handle = dev.commit_matrix(pk, cached, cols, h)  # this AIR's own commitment: inside the timed region
            traces_ = [(a, common, h, [], [handle]) if a == rank else (a, None, h, [], None) for a in range(world)]
            out = dev.prove_sharded(pk, shard, traces_, owners)
            handle.free()
            return out

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.dry_run:
        dry_run(args, pkg, dist, shard, rank, world, devname, bulk_ok if world > 1 else True)
        return

    proof = b""
    for _ in range(args.warmup):
        proof, _ = step()
    # The timed region carries ONE HIP-event pair per proof: the dominant kernel's (roofline.achieved must be measured
    # over the timed region, on the launch stream). Everything else in `stage_ms` / `roofline.classes` comes from
    # separate, fully profiled steps AFTER the timed region (the headline no longer includes the
    # event overhead of ~900 launches).
    pkg.profile_enable(2)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof, _ = step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof_dom = pkg.profile_report()
    prof_steps = max(1, min(args.steps, 3)) if args.profile_steps < 0 else args.profile_steps
    prof, prof_detail, prof_step_ms = {}, {}, None
    if prof_steps:
        pkg.profile_enable(1)
        barrier()
        p0 = time.perf_counter()
        for _ in range(prof_steps):
            step()
        barrier()
        prof_step_ms = (time.perf_counter() - p0) / prof_steps * 1e3
        prof = pkg.profile_report()
        prof_detail = pkg.profile_report_detail()
    pkg.profile_enable(0)
    other_form = None
    if world == 1 and step_partitioned is not None and not args.no_other_form:
        # the N = 1 point in the units of N >= 2 (or, with --partitioned, the plain form beside it): a few more steps
        other = step_plain if args.partitioned else step_partitioned
        other()
        barrier()
        o0 = time.perf_counter()
        n_other = max(1, min(args.steps, 5))
        for _ in range(n_other):
            other()
        barrier()
        o_ms = (time.perf_counter() - o0) / n_other * 1e3
        other_form = {"form": "plain: one common-main commitment (the reference bench's form)" if args.partitioned else
                              "partitioned: 100 cached-main columns committed inside the step + 1-column common main "
                              "(what each rank of an N > 1 run proves)",
                      "steps": n_other, "ms_per_step": round(o_ms, 3), "value": round(cells / (o_ms * 1e-3), 1), "unit": "cells/s"}
    throughput = None
    if world == 1 and args.engines > 0:
        throughput = engines_run(args, pkg, torch, pk, traces, cells, barrier, local_rank)
    elapsed = pkg.max_over_ranks(elapsed, "cpu" if args.one_gpu else devname)
    stages_mine = {k: v[1] / max(prof_steps, 1) for k, v in prof.items()}
    dist_info, stages_ranks = None, None
    if world > 1:
        # positive evidence of what ran: the process group's backend and size as torch reports
        # them, each rank's device, and the carrier of the bulk exchange the pre-flight settled on
        props = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "local_rank": local_rank, "device_index": torch.cuda.current_device(), "device_name": props.name,
                "device_uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
                "pid": os.getpid(), "bulk_carrier": shard.bulk, "alltoall_calls": shard.alltoall_calls,
                "bulk_bytes": shard.bulk_bytes, "stage_ms": stages_mine, "host_placement": placement_rec}
        every = [None] * world
        dist.all_gather_object(every, mine)
        stages_ranks = [e["stage_ms"] for e in every]
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "nccl_is_rccl": bool(getattr(torch.version, "hip", None)),
                     "distinct_devices": len({(e["device_index"], e["device_uuid"]) for e in every}),
                     "bulk_preflight_ok": bool(bulk_ok), "bulk_carrier": sorted({e["bulk_carrier"] for e in every}),
                     "ranks": [{k: e[k] for k in ("rank", "local_rank", "device_index", "device_name", "device_uuid",
                                                  "pci_bus_id", "pid", "alltoall_calls", "bulk_bytes", "host_placement")} for e in every]}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * cells * args.steps / elapsed
        # N > 1: a stage's time is the MAX over ranks (the ranks run in lock step through the replicated transcript)
        stages = dict(stages_mine)
        if stages_ranks:
            for sr in stages_ranks:
                for k, v in sr.items():
                    stages[k] = max(stages.get(k, 0.0), v)
        dom = max(stages, key=lambda k: stages[k]) if stages else None
        rows = 1 << (args.log_stacked_height + params["log_blowup"])
        width = (cols * h * (args.num_airs if world == 1 else 1)) >> args.log_stacked_height
        roof = None
        if "merkle.hash_rows" in prof_dom and width > 0:
            calls, total_ms = prof_dom["merkle.hash_rows"]   # HIP events over the TIMED region
            rows_per_query = 1 << params["k_whir"]
            # The dominant kernel is the Poseidon2 row hash. Algorithmic bytes per launch: the codeword
            # read once + one digest per QUERY (4*E + 32*rows/rows_per_query); the kernel writes one digest per row
            # (the k_whir in-query compress layers run as separate launches), which shows in `traffic`.
            n_main = max(1, calls // args.steps) if world == 1 else None
            avg_ms = total_ms / calls
            algo_bytes = 4 * rows * width + 32 * rows // rows_per_query
            achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
            traffic, src = None, None
            pmc = newest_profile("pmc_hbm_traffic_{r}.json")
            if world == 1 and (log_rows, cols, args.log_stacked_height, args.log_blowup) == (22, 100, 24, 1) and os.path.exists(pmc):
                k = json.load(open(pmc))["kernels"].get("hash_rows_kernel<1>")
                if k:
                    traffic = int((k["fetch_bytes_corrected"] + k["write_bytes"]) // k["launches"])
                    src = os.path.relpath(pmc, ROOT) + " (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
            roof = {"kernel": "hash_rows_kernel<1> (Poseidon2 leaf hash of the codeword; VALU-bound, see issue_bound)",
                    "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
                    "avg_launch_ms": round(avg_ms, 3), "algorithmic_bytes": algo_bytes, "launches_timed": calls}
            # VALU view of the same kernel. VALU instructions per permutation: the row hash runs ONE permutation body (round 6:
            # p2::permute_lazy_t, 3 676 VALU by static count, tools/isa_count.py; 7 184 in rounds 3-5) for every permutation
            # of a row; with a committed SQ counter pass the number is the kernel's own SQ_INSTS_VALU per permutation.
            # Peaks: (a) MI355X_MICROARCH.md's SIMD-32 figure, one wave64 VALU instruction per 2 cycles per SIMD
            # (256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz); (b) the MEASURED ceiling of this instruction mix: the same
            # permutation code on register-resident states, no memory (tools/ubench_poseidon2.hip ->
            # profiles/r06/poseidon2_mix_ceiling.json), which is what bounds the kernel in practice.
            ppr = (width + 7) // 8
            valu_pp, valu_src = 3676, "static count of p2::permute (tools/isa_count.py)"
            sqk, sq_file = load_sq_valu()
            if sqk and "hash_rows_kernel<1>" in sqk and (log_rows, cols, args.log_stacked_height, args.log_blowup, args.num_airs) == (22, 100, 24, 1, 1):
                k1 = sqk["hash_rows_kernel<1>"]
                valu_pp = round(k1["valu_wave_insts"] * 64 / max(1, k1.get("launches", 1)) / (rows * ppr))
                valu_src = f"{sq_file}: SQ_INSTS_VALU of the kernel / permutations (loads, stores and sponge bookkeeping included)"
            perms = rows * ppr
            props = torch.cuda.get_device_properties(local_rank)
            guide_peak = props.multi_processor_count * 4 * 32 * 2.4e9
            ib = {"valu_instr_per_permutation": valu_pp, "valu_instr_source": valu_src, "permutations_per_launch": perms,
                  "achieved_Gperm_s": round(perms / (avg_ms * 1e-3) / 1e9, 3),
                  "achieved_Tlaneops": round(valu_pp * perms / (avg_ms * 1e-3) / 1e12, 2),
                  "guide_simd32_peak_Tlaneops": round(guide_peak / 1e12, 2),
                  "frac_of_guide_peak": round(valu_pp * perms / (avg_ms * 1e-3) / guide_peak, 3)}
            ceil_path = newest_profile("poseidon2_mix_ceiling.json")
            if os.path.exists(ceil_path):
                c = json.load(open(ceil_path))
                ib["measured_mix_ceiling_Gperm_s"] = c["gperm_per_s"]
                ib["frac_of_measured_ceiling"] = round(ib["achieved_Gperm_s"] / c["gperm_per_s"], 3)
                ib["ceiling_source"] = os.path.relpath(ceil_path, ROOT)
            roof["issue_bound"] = ib
            if world == 1:
                gkr_leaves = 26 * h * args.num_airs  # 13 send/receive pairs -> 26 interactions per row (padding leaves are never materialised)
                roof["classes"] = roofline_classes(prof, prof_steps, rows, width, cells, args.log_stacked_height, gkr_leaves,
                                                   n_cus=props.multi_processor_count,
                                                   headline=(log_rows, cols, args.log_stacked_height, args.log_blowup, args.num_airs) == (22, 100, 24, 1, 1))
                roof["classes_source"] = f"{prof_steps} fully profiled steps after the timed region"
        cpu = None
        if not args.no_cpu_baseline:
            if host_affinity is not None:
                os.sched_setaffinity(0, set(all_cpus))  # the baseline gets every host core again
            import oracle_lib
            orc = oracle_lib.load()
            orc.lib.orc_set_num_threads(len(all_cpus) if all_cpus else (os.cpu_count() or 1))
            lr = args.cpu_log_rows
            small = args.log_stacked_height - (log_rows - lr)
            p2 = airs.app_params(small) if args.log_blowup == 1 else airs.stress_params(small, args.log_blowup)
            blob = airs.pk_blob(p2, [airs.uniform_runner_air(cols)])
            tr = airs.splitmix64_bits(0xD15EA5E, cols << lr)
            c0 = time.perf_counter()
            orc.prove(blob, [(0, tr, 1 << lr, [])])
            cdt = time.perf_counter() - c0
            cpu = {"value": round((cols << lr) / cdt, 1), "unit": "cells/s", "cores": os.cpu_count(), "kind": "port",
                   "sample": f"C++ oracle (OpenMP restatement, not the reference's Rust backend: no cargo here), one prove of "
                             f"2^{lr} x {cols} BenchmarkAir, app params with log_stacked_height {small}, {cdt:.1f} s"}
            # The bounded sample understates even the port (fixed costs: the oracle proves the FULL 2^22 x 100 workload at
            # about twice this rate). The full-size figure is measured once per round by the always-on byte-parity test of
            # config D (tests/test_gpu_parity_at_scale.py writes gpurun_out/cpu_baseline_full_size.json) and committed.
            # A box with cargo replaces both by the reference itself: tools/run_on_cargo_box.sh times
            # BabyBearPoseidon2CpuEngine on this workload (kind "reference").
            ref = os.path.join(ROOT, "profiles", "cargo_box", "cpu_baseline_reference.json")
            if os.path.exists(ref):  # (tools/run_on_cargo_box.sh step 4: the reference's CPU engine on this very workload)
                cpu["reference"] = dict(json.load(open(ref)), source="profiles/cargo_box/cpu_baseline_reference.json")
            for rel in (("gpurun_out", "cpu_baseline_full_size.json"), ("profiles", "r06", "cpu_baseline_full_size.json"),
                        ("profiles", "r05", "cpu_baseline_full_size.json")):
                q = os.path.join(ROOT, *rel)
                if os.path.exists(q):
                    full = json.load(open(q))
                    cpu["full_size"] = {"value": full["cells_per_s"], "unit": "cells/s", "seconds": full["seconds"],
                                        "cores": full.get("cores"), "workload": full["workload"],
                                        "source": os.path.join(*rel) + " (the oracle's prove inside test_config_d_full_byte_identical)"}
                    break
        # ~60 algorithmic HBM bytes per trace cell end to end for this configuration
        # (6 trace passes + codeword write/read at blowup 2 + the 2^27-leaf fraction tree)
        e2e_gbs = 60.0 * world * cells / (elapsed / args.steps) / 1e9
        par = "1 GPU" if world == 1 else (f"sharded x{world}: ONE joint proof, one AIR (cached-main commitment) per GPU, "
                                          f"per-round values over shared memory, the mu-batched WHIR vectors summed by "
                                          + ("RCCL all-to-all + all-gather" if shard.bulk == "torch" and not args.one_gpu else
                                             "the library's IPC exchange (peer buffers mapped, device-to-device pulls)"
                                             + ("" if args.bulk == "ipc" else " -- the RCCL pre-flight FAILED") if shard.bulk == "ipc" else
                                             "an exchange STAGED THROUGH HOST shared memory"
                                             + ("" if args.bulk == "shm" else " -- the pre-flights of RCCL and IPC FAILED") if shard.bulk == "shm"
                                             else "gloo, host-staged")
                                          + (" [--one-gpu: ranks share GPU 0]" if args.one_gpu else ""))
        line = {
            "metric": "trace_cells_per_sec", "value": round(value, 1), "unit": "cells/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "prove_ms": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (BabyBear Montgomery; F_p^4 challenges)", "data": "synthetic",
            "config": {"workload": ("" if args.num_airs == 1 else f"{args.num_airs} x ") + f"uniform_runner BenchmarkAir 2^{log_rows} rows x {cols} cols per GPU, "
                                   f"200 boolean constraints, 13 send/receive pairs, app params "
                                   f"(l_skip 4, n_stack {args.log_stacked_height - 4}, log_blowup {args.log_blowup}, k_whir 4, "
                                   f"PoW {params['logup_pow_bits']}/{params['mu_pow_bits']}/"
                                   f"{params['folding_pow_bits']}/{params['query_phase_pow_bits']} bits), "
                                   + ("all-zero trace" if args.zero_trace else "seeded random boolean trace")
                                   + ("" if world == 1 else "; per AIR: 100 columns as a cached-main commitment (committed inside the "
                                      "timed region) + a 1-column common main (not counted in cells)"),
                       "proof_bytes": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(),
                       "proof_matches_checked_digest": checked_digest_verdict(args, world, hashlib.sha256(proof).hexdigest()),
                       "parallelism": par,
                       "host_affinity": host_affinity, "host_placement": placement_rec,
                       "form": ("partitioned (cached-main commitment inside the step + 1-column common main)"
                                if (world > 1 or (args.partitioned and step_partitioned is not None)) else "plain (one common-main commitment)")},
            "roofline": roof, "cpu_baseline": cpu,
            "end_to_end": {"algorithmic_bytes_per_cell": 60, "achieved_GBps": round(e2e_gbs, 1),
                           "frac_of_hbm_peak": round(e2e_gbs / (HBM_PEAK_GBS * world), 4)},
            "stage_ms": {k: round(v, 3) for k, v in sorted(stages.items(), key=lambda kv: -kv[1])},
            "dominant_stage": dom,
        }
        if prof_detail and world == 1:
            # The ADDITIVE table (the labels of `stage_ms` nest -- whir.rs_commit contains Merkle and NTT labels --
            # and one runs on the side stream, so they summed to more than the step): per label only the time measured at
            # top level on the caller's stream; nested and side-stream time are listed apart; the remainder of a profiled
            # step is host round trips and unlabelled launches.
            add = {k: (v["ms"] - v["nested_ms"] - v["side_ms"]) / prof_steps for k, v in prof_detail.items()}
            add = {k: round(v, 3) for k, v in sorted(add.items(), key=lambda kv: -kv[1]) if v > 0.0005}
            line["stage_ms_additive"] = {
                "top_level_on_the_callers_stream": add, "sum": round(sum(add.values()), 3),
                "profiled_step_ms": round(prof_step_ms, 3),
                "unlabelled_or_idle": round(prof_step_ms - sum(add.values()), 3),
                "nested_in_another_label": {k: round(v["nested_ms"] / prof_steps, 3) for k, v in prof_detail.items() if v["nested_ms"] > 0},
                "on_the_side_stream": {k: round(v["side_ms"] / prof_steps, 3) for k, v in prof_detail.items() if v["side_ms"] > 0}}
        if other_form is not None:
            line["n1_plain_form" if args.partitioned else "n1_partitioned_form"] = other_form
        if throughput is not None:
            line["throughput"] = throughput
        if shard is not None:
            line["config"]["bulk_exchange_bytes_per_proof"] = shard.bulk_bytes // max(args.steps + args.warmup + prof_steps, 1)
            line["dist"] = dist_info
            line["stage_ms_per_rank"] = [{k: round(v, 3) for k, v in sr.items()} for sr in stages_ranks]
        print(json.dumps(line), flush=True)
    if shard is not None:
        shard.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
