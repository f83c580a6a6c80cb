"""CPU tests of the sharded prover's host side (no GPU):
  * the LogUp-GKR driver with 1..8 simulated ranks against the unsharded run (csrc/selftest.cpp);
  * the exchange protocol between two REAL processes: the library's shared-memory all-gather for the
    per-round values, and the bulk all-gather callback supplied by the Python host over torch.distributed
    with the gloo backend (world size 2, 127.0.0.1) followed by the modular add of the gathered blocks."""
import os
import socket
import uuid

import numpy as np
import pytest
import torch.multiprocessing as mp


def test_gkr_driver_one_rank_with_and_without_padding(pkg):
    pkg.gkr_shard_selftest(1, 6, [64])
    pkg.gkr_shard_selftest(1, 6, [52])       # config D's shape in small: 26 of 32 slices used
    pkg.gkr_shard_selftest(1, 6, [52], three_points=True)
    pkg.gkr_shard_selftest(1, 1, [2])
    pkg.gkr_shard_selftest(1, 3, [1])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gkr_driver_config_d_layout(pkg, world):
    """world AIRs of 26 interaction slices each (13 send/receive pairs), slices of 2^3 leaves."""
    slices = 26 * world
    TR = (slices * 8 - 1).bit_length()
    pkg.gkr_shard_selftest(world, TR, [26 * 8 * (r + 1) for r in range(world)], seed=world)


def test_gkr_driver_random_ranges(pkg):
    """Arbitrary (also odd, also empty) leaf runs per rank, random padding, both point counts."""
    rng = np.random.default_rng(11)
    for it in range(60):
        TR = int(rng.integers(2, 11))
        world = int(rng.integers(1, 9))
        used = int(rng.integers(1, (1 << TR) + 1))
        ends = sorted(int(x) for x in rng.integers(0, used + 1, size=world - 1)) + [used]
        pkg.gkr_shard_selftest(world, TR, ends, seed=it, three_points=it % 4 == 0)


def test_gkr_driver_rejects_bad_input(pkg):
    with pytest.raises(pkg.OvmError):
        pkg.gkr_shard_selftest(2, 4, [9, 5])  # ends not monotone


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shm, q, alltoall=True):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from conftest import load_pkg
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_pkg()
    try:
        shard = pkg.Shard(rank, world, shm, slot_bytes=1 << 16, host_buffers=True, alltoall=alltoall)
        pkg.shard_exchange_selftest(shard, n=5001, rounds=200, seed=5)
        q.put((rank, "ok", shard.bulk_bytes, shard.alltoall_calls))
        shard.close()
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e), 0, 0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("alltoall", [True, False])
def test_exchange_protocol_world2_gloo(alltoall):
    """Two real processes: the small exchanges through shared memory, the bulk ones through the Python callbacks on
    gloo -- the all-gather of whole vectors and (alltoall=True) the all-to-all + add + all-gather form of the same sum,
    on a length that does not divide by the world size."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, shm = _free_port(), f"/ovm-test-{uuid.uuid4().hex[:12]}"
    ps = [ctx.Process(target=_worker, args=(r, 2, port, shm, q, alltoall)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == ["ok", "ok"], res
    n, sl = 5001, 2501                                       # slice = ceil(n / 2)
    whole = 2 * n * 16                                        # the all-gather of whole vectors
    assert all(r[2] == (whole + 2 * (2 * sl * 16) if alltoall else whole) for r in res)
    assert all(r[3] == (1 if alltoall else 0) for r in res)
    assert not os.path.exists("/dev/shm" + shm)              # the segment's name is gone after the handshake


def _abort_worker(rank, world, shm, q, mode):
    """Rank 1 makes 3 exchanges and then fails ('abort': raises the abort flag; 'die': just stops contributing); the
    others keep exchanging and report what their next exchange returned and how long that took."""
    import ctypes
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import load_pkg
    pkg = load_pkg()
    lib = pkg._lib()
    h = ctypes.c_void_p(0)
    pkg.check(lib.ovm_hip_shm_comm_create(shm.encode(), rank, world, 4096, ctypes.byref(h)), "create")
    mine = np.array([rank], dtype=np.uint32)
    every = np.zeros(world, dtype=np.uint32)
    rcs = []
    for i in range(3):
        rcs.append(lib.ovm_hip_shm_allgather(h, mine.ctypes.data, every.ctypes.data, 4))
    t0 = time.time()
    why = ""
    if rank == 1:
        if mode == "abort":
            lib.ovm_hip_shm_comm_abort(h, 2)
        rcs.append(lib.ovm_hip_shm_allgather(h, mine.ctypes.data, every.ctypes.data, 4) if mode == "abort" else -1)
    else:
        rcs.append(lib.ovm_hip_shm_allgather(h, mine.ctypes.data, every.ctypes.data, 4))   # rank 1 never joins this one
        why = lib.ovm_hip_shard_last_error().decode()
        rcs.append(lib.ovm_hip_shm_allgather(h, mine.ctypes.data, every.ctypes.data, 4))   # afterwards: poisoned, at once
    q.put((rank, rcs, time.time() - t0, int(lib.ovm_hip_shm_comm_poisoned(h)), why))


@pytest.mark.parametrize("mode", ["abort", "die"])
def test_a_failed_rank_does_not_hold_its_peers(mode):
    """a rank that fails in the middle of a joint call. With the abort flag its peers return
    OVM_ESHARD_PEER_ABORTED within milliseconds, naming the rank; without it (the process just died) they time out after
    OVM_SHARD_TIMEOUT_S naming the rank they waited for. Either way the communicator is POISONED: the next exchange
    fails at once instead of pairing a fresh slot with a stale one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    shm = f"/ovm-test-{uuid.uuid4().hex[:12]}"
    os.environ["OVM_SHARD_TIMEOUT_S"] = "2"
    try:
        ps = [ctx.Process(target=_abort_worker, args=(r, 3, shm, q, mode)) for r in range(3)]
        for p in ps:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(3))
        for p in ps:
            p.join(timeout=60)
    finally:
        del os.environ["OVM_SHARD_TIMEOUT_S"]
    TIMEOUT, ABORTED, POISONED = 10006, 10007, 10008
    for rank, rcs, dt, poisoned, msg in res:
        assert rcs[:3] == [0, 0, 0]
        if rank == 1:
            assert rcs[3] == (POISONED if mode == "abort" else -1)
            continue
        assert rcs[3] == (ABORTED if mode == "abort" else TIMEOUT) and rcs[4] == POISONED and poisoned == 1, (rank, rcs, msg)
        assert (dt < 1.0) if mode == "abort" else (1.5 < dt < 10.0), dt
    assert any("rank 1" in r[4] for r in res if r[0] != 1)
