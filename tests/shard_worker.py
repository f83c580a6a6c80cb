"""One rank of a sharded proof (tests/test_gpu_shard.py starts `world` of these on ONE GPU: logical shards
with their own streams and caches, host staging through the shared-memory exchange standing in for RCCL;
bench.py runs the same code path with one GPU per rank and the nccl backend). Rank 0 writes the joint proof
to --out."""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--case", required=True)
    ap.add_argument("--shm", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--save-memory", action="store_true",
                    help="prove in the memory-bounded modes (codeword dropped, round 0 in slabs of 8 rows, leaf layer dropped)")
    ap.add_argument("--bulk", default="ipc", choices=["ipc", "shm"],
                    help="the bulk device exchange: the library's IPC pulls, or host staging through the shared-memory segment")
    ap.add_argument("--fail-mode", default="none", choices=["none", "invalid", "midproof"],
                    help="before the real proof: 'invalid' = --fail-rank passes a public-values list of the wrong length (every "
                         "rank must refuse, the communicator stays usable); 'midproof' = --fail-rank throws after its commitment "
                         "(the peers must return OVM_ESHARD_PEER_ABORTED quickly, the communicator is poisoned and is recreated)")
    ap.add_argument("--fail-rank", type=int, default=1)
    ap.add_argument("--expect-ipc", default="any", choices=["any", "mapped", "degraded"],
                    help="after the proof: the library's IPC exchange must have stayed on mapped peer buffers / must have degraded "
                         "to host staging (OVM_IPC_FORCE_OPEN_FAIL / OVM_IPC_FORCE_OPEN_HANG in the environment)")
    ap.add_argument("--no-alltoall", action="store_true",
                    help="hand the library a null alltoall_device: the mu-batch vectors are all-gathered whole")
    args = ap.parse_args()

    import time
    t_start = time.time()
    import torch
    from conftest import load_pkg
    import airs
    import shard_cases

    pkg = load_pkg()
    # (no process group: the ranks share one GPU; the bulk exchange is the library's own -- IPC-mapped peer buffers and
    # device-to-device pulls, or staged through its shared-memory exchange; torch.distributed's part is covered by
    # tests/test_shard_cpu.py (gloo) and the nccl test)
    t_pg = time.time()
    dev = pkg.Device(0)
    if args.save_memory:
        pkg.set_prover_config(cache_rs_code_matrix=False, zerocheck_save_memory=True, round0_slab_log_rows=3, gkr_drop_leaves=True)
    shard = pkg.Shard(args.rank, args.world, args.shm, bulk=args.bulk, alltoall=not args.no_alltoall)
    if args.case.startswith("gen:"):
        # "gen:<log_rows>:<cols>:<log_stacked_height>[:<log_blowup>]": one partitioned BenchmarkAir per rank under app
        # parameters (log_blowup 2: the leaf-like parameters of BASELINE config E), the seeded boolean trace generated on
        # the device (BASELINE config D's / E's multi-GPU form at any size)
        fields = [int(x) for x in args.case.split(":")[1:]]
        log_rows, cols, log_stacked = fields[:3]
        log_blowup = fields[3] if len(fields) > 3 else 1
        h = 1 << log_rows
        air = airs.uniform_runner_air_partitioned(cols)
        params = airs.app_params(log_stacked) if log_blowup == 1 else airs.stress_params(log_stacked, log_blowup)
        pk = pkg.ProvingKey(airs.pk_blob(params, [air] * args.world))
        gen = torch.Generator(device="cuda:0")
        gen.manual_seed(0xD15EA5E + args.rank)
        one = int(pkg.to_mont(np.array([1], dtype=np.uint32))[0])
        cached = torch.randint(0, 2, (cols * h,), dtype=torch.int32, device="cuda:0", generator=gen) * one
        common = torch.randint(0, 2, (h,), dtype=torch.int32, device="cuda:0", generator=gen) * one
        handle = dev.commit_matrix(pk, cached, cols, h)
        owners = list(range(args.world))
        dtr = [(a, common, h, [], [handle]) if a == args.rank else (a, None, h, [], None) for a in range(args.world)]
        traces = []
    else:
        params, air_list, traces, owners = shard_cases.make(args.case, args.world)
        pk = pkg.ProvingKey(airs.pk_blob(params, air_list))
        dtr = []
    up = lambda a: dev.upload(pkg.to_mont(np.asarray(a, dtype=np.uint32)))
    for (a, common, h, pvs, cached), o in zip(traces, owners):
        if o != args.rank:
            dtr.append((a, None, h, [], None))
            continue
        handles = []
        if cached is not None:
            off = 0
            for w in air_list[a]["cached_main_widths"]:
                handles.append(dev.commit_matrix(pk, up(cached[off:off + w * h]), w, h))
                off += w * h
        dtr.append((a, up(common), h, pvs, handles))
    if args.fail_mode != "none":
        t0 = time.time()
        if args.fail_mode == "invalid":
            bad = [(a, c, h, list(pv) + [1], k) if o == args.rank == args.fail_rank else (a, c, h, pv, k)
                   for (a, c, h, pv, k), o in zip(dtr, owners)]
            try:
                dev.prove_sharded(pk, shard, bad, owners)
                raise SystemExit("the invalid input was accepted")
            except pkg.OvmError as e:
                mine = "public values length" in str(e)
                assert mine == (args.rank == args.fail_rank) and (mine or f"rank {args.fail_rank} rejected" in str(e)), str(e)
            assert not shard.poisoned                      # nobody had started working: the same communicator proves next
        else:
            os.environ["OVM_TEST_FAIL_RANK"] = str(args.fail_rank)
            try:
                dev.prove_sharded(pk, shard, dtr, owners)
                raise SystemExit("the failing proof succeeded")
            except pkg.OvmError as e:
                assert e.code == (2 if args.rank == args.fail_rank else 10007), (e.code, str(e))
            finally:
                del os.environ["OVM_TEST_FAIL_RANK"]
            assert shard.poisoned
            try:
                dev.prove_sharded(pk, shard, dtr, owners)   # a poisoned communicator refuses at once
                raise SystemExit("a poisoned communicator was used")
            except pkg.OvmError as e:
                assert e.code == 10008, (e.code, str(e))
            shard.close()
            shard = pkg.Shard(args.rank, args.world, args.shm + "-again", bulk=args.bulk, alltoall=not args.no_alltoall)
        dt = time.time() - t0
        assert dt < 20, f"failure handling took {dt:.1f} s"
        print(f"[shard_worker rank {args.rank}] {args.fail_mode}: refused / aborted in {dt:.2f} s", file=sys.stderr, flush=True)
    proof = b""
    t_ready = time.time()
    for _ in range(args.repeat):
        proof, secs = dev.prove_sharded(pk, shard, dtr, owners)
    t_done = time.time()
    print(f"[shard_worker rank {args.rank}] import+rendezvous {t_pg - t_start:.1f} s, device+key+shm+upload {t_ready - t_pg:.1f} s, "
          f"prove x{args.repeat} {t_done - t_ready:.2f} s", file=sys.stderr, flush=True)
    if args.bulk == "ipc" and args.world > 1:
        st = shard.ipc_status()
        print(f"[shard_worker rank {args.rank}] ipc: {st}", file=sys.stderr, flush=True)
        if args.expect_ipc == "degraded":
            assert st["degraded"] and st["staged_exchanges"] > 0 and st["reason"], st
        elif args.expect_ipc == "mapped":
            assert not st["degraded"] and st["staged_exchanges"] == 0 and st["mappings"] >= 1, st
    if args.rank == 0:
        with open(args.out, "wb") as f:
            f.write(proof)
    shard.close()


if __name__ == "__main__":
    main()
