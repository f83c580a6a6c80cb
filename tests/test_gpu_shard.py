"""The sharded prover on real hardware: `world` LOGICAL SHARDS ON ONE GPU (one process each: own stream,
own caches; the shared-memory exchange between them, host staging through it standing in for RCCL) produce
ONE joint proof that must equal, byte for byte, the oracle's proof of the same multi-AIR,
per-AIR-commitment context -- and the single-process HIP proof. (torch.distributed's part of the bulk exchange:
tests/test_shard_cpu.py with gloo, test_bulk_exchange_callback_on_nccl below, bench.py --one-gpu.)"""
import os
import socket
import subprocess
import sys
import uuid

import numpy as np
import pytest

import airs
import shard_cases

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def hooks_env():
    """The workers of a fault-injection test load the TEST TWIN of the library (tools/build_hooks.sh): the product reads no hook."""
    import conftest
    conftest.load_pkg(hooks=True)  # (builds the twin if it is missing)
    return {"OVM_HIP_LIB": conftest.HOOKS_LIB, "OVM_TUNING_LIVE": "1"}


def run_sharded(case, world, tmp_path, repeat=1, save_memory=False, alltoall=True, bulk="ipc", fail_mode="none", fail_rank=1,
                extra_env=None, expect_ipc="any"):
    port, shm, out = _free_port(), f"/ovm-test-{uuid.uuid4().hex[:12]}", str(tmp_path / "proof.bin")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OVM_HIP_LIB")}
    env["OVM_SHARD_TIMEOUT_S"] = "60"  # a rank that died must not hold the others for minutes
    if fail_mode == "midproof":
        env.update(hooks_env())  # (the worker arms OVM_TEST_FAIL_RANK)
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "shard_worker.py"), "--rank", str(r), "--world", str(world),
                               "--port", str(port), "--case", case, "--shm", shm, "--out", out, "--repeat", str(repeat)]
                              + (["--save-memory"] if save_memory else []) + ([] if alltoall else ["--no-alltoall"])
                              + ["--bulk", bulk, "--fail-mode", fail_mode, "--fail-rank", str(fail_rank), "--expect-ipc", expect_ipc],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    for l in logs:  # (where a run's time goes: pytest -s shows it)
        print("\n".join(x for x in l.splitlines() if x.startswith("[shard_worker")))
    return open(out, "rb").read()


def oracle_proof(orc, case, world):
    params, air_list, traces, owners = shard_cases.make(case, world)
    blob = airs.pk_blob(params, air_list)
    want = orc.prove(blob, [t if t[4] is not None else t[:4] for t in traces])["proof"]
    return blob, want


# (every context with 2 shards; with 4 shards the three that differ in structure: one AIR per rank, mixed
# heights with two AIRs per rank, and the mixture of plain / cached / public-value AIRs)
RUNS = [(c, 2) for c in shard_cases.CASES] + [(c, 4) for c in ("bench8", "bench_mixed_heights", "mixture")] + \
       [("bench8", 3), ("three_airs", 4), ("absent_airs", 4), ("optional_air:2", 3)] + \
       [("bench8", 8), ("bench_mixed_heights", 8), ("three_airs", 8)]
# (absent_airs / optional_air: keys with more AIRs than the context has traces -- with 4 (3) ranks over 3 (2) traces a rank
# owns nothing; 3: a world that is not a power of two; three_airs with 4 or 8 ranks: ranks that own no trace; 8 = the north star's
# world size -- one AIR per rank, 16 AIRs of mixed heights (two per rank), and 3 AIRs over 8 ranks)


@pytest.mark.parametrize("case,world", RUNS)
def test_logical_shards_match_oracle(orc, tmp_path, case, world):
    import time
    t0 = time.time()
    blob, want = oracle_proof(orc, case, world)
    t1 = time.time()
    got = run_sharded(case, world, tmp_path)
    print(f"[{case}-{world}] oracle {t1 - t0:.1f} s, {world} worker processes {time.time() - t1:.1f} s")
    assert got == want
    orc.verify(blob, got)


@pytest.mark.parametrize("case,world", [("bench_app", 2), ("no_interactions_on_one_rank", 2), ("bench_mixed_heights", 4)])
def test_sharded_in_the_memory_bounded_modes(orc, tmp_path, case, world):
    """Codeword dropped on every rank (the remote ranks re-encode when rank 0's queries arrive), round 0 in
    slabs, the fraction tree's leaf layer dropped and generated again (also on a rank that has no leaves):
    the joint proof does not change."""
    blob, want = oracle_proof(orc, case, world)
    assert run_sharded(case, world, tmp_path, save_memory=True) == want


@pytest.mark.parametrize("case,world", [("bench_app", 2), ("bench8", 4), ("mixture", 3)])
def test_bulk_exchange_staged_through_shared_memory(orc, tmp_path, case, world):
    """The last fall-back of the bulk exchange (device -> host -> shared-memory segment -> device), which every other test
    here replaces by the library's IPC exchange: the same proof."""
    blob, want = oracle_proof(orc, case, world)
    assert run_sharded(case, world, tmp_path, bulk="shm") == want


@pytest.mark.parametrize("hook,value,world", [("OVM_IPC_FORCE_OPEN_FAIL", "1", 2), ("OVM_IPC_FORCE_OPEN_FAIL", "1", 4),
                                              ("OVM_IPC_FORCE_OPEN_FAIL", "2", 3), ("OVM_IPC_FORCE_OPEN_HANG", "1", 3)])
def test_ipc_exchange_degrades_to_host_staging_instead_of_hanging(orc, tmp_path, hook, value, world):
    """`hipIpcOpenMemHandle` once did not return (two 85 GB processes on one GPU). Every open now has
    a deadline and the ranks map in turn; a failed (FAIL hook) or overdue (HANG hook: the helper thread never returns,
    deadline 1 s) open is agreed by all ranks and the communicator stages this and every later bulk exchange through
    host memory: the proof in flight completes, byte-identical, and a second proof over the same communicator too.
    FAIL = 2 fails the open with hipErrorInvalidValue -- what a stale handle or a buffer that is not a hipMalloc allocation
    really produces, and once mistaken for "the ranks disagree on the size" (ADVICE r4): it degrades like any other."""
    import time
    blob, want = oracle_proof(orc, "bench_app", world)
    t0 = time.time()
    got = run_sharded("bench_app", world, tmp_path, repeat=2, extra_env=dict(hooks_env(), **{hook: value, "OVM_IPC_OPEN_TIMEOUT_S": "1"}),
                      expect_ipc="degraded")
    assert got == want
    assert time.time() - t0 < 120


def test_ipc_exchange_stays_on_mapped_buffers_when_nothing_fails(orc, tmp_path):
    blob, want = oracle_proof(orc, "bench_app", 4)
    assert run_sharded("bench_app", 4, tmp_path, repeat=3, expect_ipc="mapped") == want


def _ipc_repro(segment_gb, extra_env, timeout=240):
    """tools/ipc_repro.py: two processes on GPU 0, each exchanging 256 MB out of ONE allocation of `segment_gb` GiB three times
    through the library's IPC exchange; returns every rank's final status line."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, **extra_env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "ipc_repro.py"), "--world", "2", "--filler-gb", "0",
                          "--segment-gb", str(segment_gb)], env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "WRONG" not in out.stdout, out.stdout
    recs = [json.loads(line.split("] ", 1)[1]) for line in out.stdout.splitlines() if line.startswith("[rank") and "] {" in line]
    assert len(recs) == 2, out.stdout
    return recs


def test_ipc_exchange_sends_a_copy_out_of_allocations_the_runtime_cannot_map():
    """Root cause of round 3's hang, found with tools/ipc_repro.py (profiles/r04/ipc_repro_sizes_r04.log): on this runtime
    `hipIpcOpenMemHandle` never returns for an allocation whose SIZE has bit 31 set (2.0-3.99, 6.0-7.99, 11 GiB hang; 1.9, 4.0,
    5.99, 12 GiB map in 2 ms; nothing else matters). The owner of such a send buffer exports a device-side copy in a buffer
    of a safe size instead: mapped, fast, no deadline to wait out. A 2.5 GiB allocation is the smallest clean instance."""
    for st in _ipc_repro(2.5, {"OVM_IPC_OPEN_TIMEOUT_S": "15"}):
        assert not st["ipc"]["degraded"] and st["ipc"]["staged_exchanges"] == 0, st
        assert st["ipc"]["bounced_exchanges"] == 3 and st["ipc"]["mappings"] == 1, st
        assert max(st["exchange_s"]) < 5, st
    for st in _ipc_repro(1.5, {"OVM_IPC_OPEN_TIMEOUT_S": "15"}):  # a safe size is exported as it is
        assert not st["ipc"]["degraded"] and st["ipc"]["bounced_exchanges"] == 0 and st["ipc"]["mappings"] == 1, st


def test_ipc_open_of_a_2_5_gib_allocation_still_hangs_without_the_rule_and_the_deadline_catches_it():
    """The rule switched off (OVM_IPC_NO_BOUNCE=1): the open of the peer's 2.5 GiB allocation does not return, the 3 s deadline
    fires on both ranks in turn and the exchanges are staged through host memory -- data intact. If a later runtime fixes the
    open, this test skips and says so."""
    if os.environ.get("OVM_IPC_RAW_HANG_TEST", "0") != "1":
        pytest.skip("opt-in (OVM_IPC_RAW_HANG_TEST=1): provokes the runtime's hang on purpose; its last run is "
                    "profiles/r04/gputest_ipc_r04n.log and the reproducer logs beside it")
    recs = _ipc_repro(2.5, dict(hooks_env(), **{"OVM_IPC_OPEN_TIMEOUT_S": "3", "OVM_IPC_NO_BOUNCE": "1"}))
    if not any(st["ipc"]["degraded"] for st in recs):
        pytest.skip("this runtime maps a 2.5 GiB allocation: the size rule of csrc/ipc_comm.cpp is no longer needed here")
    for st in recs:
        assert st["ipc"]["degraded"] and st["ipc"]["staged_exchanges"] == 3 and st["ipc"]["bounced_exchanges"] == 0, st
        assert "did not return" in st["ipc"]["reason"], st


@pytest.mark.parametrize("case,world", [("bench_app", 2), ("bench8", 3)])
def test_sum_of_the_mu_batch_vectors_by_allgather_only(orc, tmp_path, case, world):
    """A communicator without alltoall_device (the callback is optional): the ranks' mu-batch vectors are all-gathered
    whole and added up -- the same proof as with the all-to-all + add + all-gather every other test here runs
    (3 ranks: slices of ceil(2^m / 3) elements, a padded tail)."""
    blob, want = oracle_proof(orc, case, world)
    assert run_sharded(case, world, tmp_path, alltoall=False) == want


@pytest.mark.parametrize("mode,world,fail_rank", [("invalid", 2, 1), ("invalid", 4, 0), ("midproof", 2, 1), ("midproof", 4, 2)])
def test_a_failing_rank_stops_every_rank_and_the_next_proof_is_right(orc, tmp_path, mode, world, fail_rank):
    """ADVICE r2 (medium). 'invalid': one rank's inputs are invalid -> EVERY rank refuses before any work, with that
    rank's code, and the SAME communicator then makes the right proof. 'midproof': one rank throws after its commitment
    -> it raises the abort flag, its peers return OVM_ESHARD_PEER_ABORTED within seconds instead of the time-out, the
    communicator is poisoned on every rank (a second call returns OVM_ESHARD_POISONED at once: no stale slot is ever
    read), and a recreated communicator makes the right proof."""
    blob, want = oracle_proof(orc, "bench_app", world)
    assert run_sharded("bench_app", world, tmp_path, fail_mode=mode, fail_rank=fail_rank) == want


def test_single_process_equals_sharded(dev, pkg, orc, tmp_path):
    """The same context through ovm_hip_prove in this process (one device, one rank)."""
    params, air_list, traces, owners = shard_cases.make("bench_app", 2)
    blob = airs.pk_blob(params, air_list)
    pk = pkg.ProvingKey(blob)
    up = lambda a: dev.upload(pkg.to_mont(np.asarray(a, dtype=np.uint32)))
    dtr = []
    for a, common, h, pvs, cached in traces:
        w = air_list[a]["cached_main_widths"][0]
        dtr.append((a, up(common), h, pvs, [dev.commit_matrix(pk, up(cached), w, h)]))
    single, _ = dev.prove(pk, dtr)
    assert single == run_sharded("bench_app", 2, tmp_path, repeat=2)


def test_sharded_2_16_x_100_per_rank_byte_identical(orc, tmp_path):
    """Config D's sharded form under app parameters at 2^16 rows x 100 columns per rank, 2 ranks: a 2^22-leaf fraction
    tree over two leaf runs with base-field numerators, the distributed Merkle trees, the all-to-all form of the mu-batch
    sum on 2^18-element vectors -- byte-identical to the oracle's proof of the same two AIRs."""
    blob, want = oracle_proof(orc, "bench_scale:16", 2)
    got = run_sharded("bench_scale:16", 2, tmp_path)
    assert got == want


@pytest.mark.skipif(os.environ.get("OVM_FULL_PARITY") == "0", reason="OVM_FULL_PARITY=0: three minutes of host time skipped")
def test_config_d_full_size_two_shards_byte_identical(orc, tmp_path):
    """The same at the FULL size of BASELINE config D (2^22 x 100 per rank, 2 ranks on this one GPU), byte for byte: all
    1 267 685 bytes of the joint proof. ALWAYS ON since round 4: 184 s of a 617 s suite
    (profiles/r04/gputest_all_with_two_rank_full_size_r04i.log) after the oracle's large transforms were spread over the
    host's cores and the seeded traces vectorised; OVM_FULL_PARITY=0 skips it."""
    import time
    t0 = time.time()
    blob, want = oracle_proof(orc, "bench_scale:22", 2)
    t1 = time.time()
    got = run_sharded("bench_scale:22", 2, tmp_path)
    print(f"oracle prove of 2 x (2^22 x 100): {t1 - t0:.1f} s; sharded prove incl. upload {time.time() - t1:.1f} s; "
          f"proof {len(got)} bytes; identical: {got == want}")
    assert got == want


def test_config_d_full_size_two_shards_on_one_gpu_verifies(orc, tmp_path):
    """BASELINE config D in its sharded form AT FULL SIZE -- two AIRs of 2^22 rows x 100 columns (each its own
    cached-main commitment) + 1-column common mains, app parameters, 2^28-leaf fraction tree split over two
    ranks -- as two logical shards on this one GPU: the joint proof is accepted by the verifier restatement."""
    got = run_sharded("gen:22:100:24", 2, tmp_path)
    blob = airs.pk_blob(airs.app_params(24), [airs.uniform_runner_air_partitioned(100)] * 2)
    orc.verify(blob, got)
    bad = bytearray(got)
    bad[len(bad) // 3] ^= 1
    with pytest.raises(Exception):
        orc.verify(blob, bytes(bad))


def test_config_e_two_shards_on_one_gpu_verifies(orc, tmp_path):
    """BASELINE config E in its sharded form -- per rank a 2^24-row x 100-column AIR as its own commitment under the
    leaf-like parameters (blowup 4, constraint degree 4): 26.8 GB of codeword per rank, a 2^30-leaf fraction tree over two
    leaf runs -- as two logical shards on this one GPU: the joint proof is accepted by the verifier restatement."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 200 * (1 << 30):
        pytest.skip("needs ~200 GB of free HBM")
    # (round 3 ran this with the bulk exchange staged through shared memory: hipIpcOpenMemHandle of the peer's 3.2 GB segment
    # did not return. Root cause, round 4: bit 31 of that allocation's size -- csrc/ipc_comm.cpp exports a safe-sized copy in
    # that case, so the test is back on the IPC carrier every other logical-shard test runs; should an open stall for another
    # reason, the exchange degrades to host staging after its deadline and the proof still has to verify. pytest -s shows
    # which carrier ran: "[shard_worker rank r] ipc: {...}")
    got = run_sharded("gen:24:100:24:2", 2, tmp_path, bulk="ipc")
    blob = airs.pk_blob(airs.stress_params(24, 2), [airs.uniform_runner_air_partitioned(100)] * 2)
    orc.verify(blob, got)
    bad = bytearray(got)
    bad[len(bad) // 2] ^= 1
    with pytest.raises(Exception):
        orc.verify(blob, bytes(bad))


def test_bulk_exchange_callback_on_nccl(tmp_path):
    """The RCCL branch of the Python host's bulk exchange (the one line that differs from the gloo branch the
    logical-shard tests use): zero-copy torch views of raw device ranges + all_gather_into_tensor /
    all_to_all_single on the nccl backend. One GPU is all this box has, so the group has one rank; run in a child process (its own
    process group)."""
    code = """
import os, sys, ctypes
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from conftest import load_pkg
pkg = load_pkg()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = pkg.Device(0)
shard = pkg.Shard(0, 1, "/ovm-test-nccl-%%d" %% os.getpid())
src = torch.arange(1 << 20, dtype=torch.int32, device="cuda:0")
dst = torch.zeros(1 << 20, dtype=torch.int32, device="cuda:0")
rc = shard._allgather_device(None, src.data_ptr(), dst.data_ptr(), src.numel() * 4, dev.stream.value or 0)
torch.cuda.synchronize()
assert rc == 0 and torch.equal(src, dst), rc
assert shard.bulk_bytes == 4 << 20
dst.zero_()
rc = shard._alltoall_device(None, src.data_ptr(), dst.data_ptr(), src.numel() * 4, dev.stream.value or 0)
torch.cuda.synchronize()
assert rc == 0 and torch.equal(src, dst) and shard.alltoall_calls == 1, rc
shard.close()
dist.destroy_process_group()
print("nccl bulk exchange ok")
""" % (HERE, str(_free_port()))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "nccl bulk exchange ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_bulk_preflight_falls_back_to_the_builtin_exchange(tmp_path):
    """Shard.check_bulk: two ranks (gloo, one GPU) run the pre-flight of the device exchanges; with the all-gather
    callback of rank 1 sabotaged (it delivers wrong bytes) BOTH ranks must report failure and switch to the next
    exchange in line, the library's IPC pulls -- and that exchange must then deliver the right data."""
    code = """
import os, sys
sys.path.insert(0, %r)
rank, world, port, shm, sabotage = int(sys.argv[1]), 2, sys.argv[2], sys.argv[3], sys.argv[4] == "1"
import torch, torch.distributed as dist
from conftest import load_pkg
pkg = load_pkg()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE="2")
dist.init_process_group("gloo", rank=rank, world_size=2)
dev = pkg.Device(0)
shard = pkg.Shard(rank, world, shm)
if sabotage and rank == 1:
    real = shard._allgather_device
    def broken(ctx, d_mine, d_all, nbytes, stream):
        rc = real(ctx, d_mine, d_all, nbytes, stream)
        if shard.bulk == "torch":  # (only the process-group exchange is broken: the library's own must then take over)
            pkg.check(pkg._lib().ovm_hip_memset(d_all, 0xFF, 16, stream), "memset")   # (the pattern bytes are all < 251)
            pkg.check(pkg._lib().ovm_hip_stream_synchronize(stream), "sync")
        return rc
    shard._allgather_device = broken
ok = shard.check_bulk()
assert ok == (not sabotage), (rank, ok)
assert shard.bulk == ("ipc" if sabotage else "torch")
# the exchange the prover would now use delivers the right bytes
src = (torch.arange(4096, dtype=torch.int32, device="cuda:0") + 1000 * rank)
dst = torch.zeros(2 * 4096, dtype=torch.int32, device="cuda:0")
cb = type(shard)._allgather_device.__get__(shard)
assert cb(None, src.data_ptr(), dst.data_ptr(), 4096 * 4, dev.stream.value or 0) == 0
torch.cuda.synchronize()
want = torch.cat([torch.arange(4096, dtype=torch.int32) + 1000 * r for r in range(2)])
assert torch.equal(dst.cpu(), want)
dist.barrier()
dist.destroy_process_group()
print("preflight ok", rank)
""" % (HERE,)
    script = tmp_path / "preflight.py"
    script.write_text(code)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OVM_SHARD_TIMEOUT_S"] = "60"
    for sabotage in ("0", "1"):
        port, shm = str(_free_port()), f"/ovm-test-{uuid.uuid4().hex[:12]}"
        procs = [subprocess.Popen([sys.executable, str(script), str(r), port, shm, sabotage], env=env, stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = [p.communicate(timeout=300)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)


def test_bench_with_two_ranks_on_one_gpu(tmp_path):
    """`bench.py --gpus 2` end to end on this one GPU (--one-gpu: gloo, host-staged) at a reduced size: the launcher
    starts its own ranks, the partitioned BenchmarkAir per rank is committed and the joint proof made; one JSON line
    with n_gpus 2 comes back, with each of the three bulk exchanges, and the proof is the same."""
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OVM_SHARD_TIMEOUT_S"] = "60"
    hashes = []
    for bulk, says in (("rccl", "host-staged"), ("ipc", "IPC exchange"), ("shm", "shared memory")):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--one-gpu", "--log-rows", "12",
                              "--log-stacked-height", "16", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--bulk", bulk],
                             env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        import json
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["proof_bytes"] > 0
        assert line["value"] > 0 and says in line["config"]["parallelism"] and "FAILED" not in line["config"]["parallelism"]
        hashes.append(line["config"]["proof_sha256"])
    assert len(set(hashes)) == 1   # the same joint proof whichever way the bulk data travelled


def test_bench_with_eight_ranks_on_one_gpu(tmp_path):
    """The north star's world size through `bench.py --gpus 8` on this one GPU (--one-gpu, reduced size): eight ranks,
    one partitioned BenchmarkAir each, one joint proof; the JSON line must carry positive evidence of what ran -- the
    process group's backend and size as torch reports them, one entry per rank with its device, the bulk carrier the
    pre-flight settled on, per-rank stage times -- and the proof must not depend on the carrier."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OVM_SHARD_TIMEOUT_S"] = "60"
    hashes = []
    for bulk in ("ipc", "rccl"):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--one-gpu", "--log-rows", "12",
                              "--log-stacked-height", "16", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--bulk", bulk],
                             env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0
        d = line["dist"]
        assert d["world_size"] == 8 and d["backend"] == "gloo" and len(d["ranks"]) == 8
        assert sorted(r["rank"] for r in d["ranks"]) == list(range(8)) and len({r["pid"] for r in d["ranks"]}) == 8
        assert d["bulk_carrier"] == (["ipc"] if bulk == "ipc" else ["torch"]) and d["bulk_preflight_ok"]
        if bulk == "rccl":  # (the IPC carrier has no Python in its data path, so nothing counts its calls)
            assert all(r["alltoall_calls"] >= 2 for r in d["ranks"])   # the all-to-all form of the mu-batch sum, every rank
        assert len(line["stage_ms_per_rank"]) == 8 and all(any(k.startswith("merkle.hash_rows") for k in s) for s in line["stage_ms_per_rank"])
        hashes.append(line["config"]["proof_sha256"])
    assert len(set(hashes)) == 1


def test_bench_dry_run_two_ranks(tmp_path):
    """`bench.py --gpus 2 --one-gpu --dry-run`: rendezvous + pre-flight + the bulk steps at the full vector size of the
    given stacked height, no proof; one JSON line with the timings and a data check per carrier."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OVM_SHARD_TIMEOUT_S"] = "60"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--one-gpu", "--dry-run",
                          "--log-stacked-height", "20", "--bulk", "ipc"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["dry_run"] and line["n_gpus"] == 2 and line["vector_bytes"] == 16 << 20
    assert all(e["data_ok"] for e in line["exchanges"].values())
