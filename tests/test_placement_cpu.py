"""Host placement rules (stark-backend_amd/placement.py) over a synthetic sysfs tree shaped like the MI355X host
(2 sockets x 64 cores x 2 threads, SMT sibling of CPU k is k + 128, GPUs 0-3 on node 0 and 4-7 on node 1) and over
degenerate trees: no hardware needed."""
import os

import pytest

from conftest import load_pkg

load_pkg()
from stark_backend_amd import placement  # noqa: E402


def make_sysfs(root, sockets=2, cores=64, smt=2, adjacent_siblings=False, gpu_nodes=(0, 0, 0, 0, 1, 1, 1, 1)):
    ncpu = sockets * cores * smt
    def sibs(c):
        if adjacent_siblings:
            base = c - c % smt
            return list(range(base, base + smt))
        phys = c % (sockets * cores)
        return [phys + k * sockets * cores for k in range(smt)]
    def node_of(c):
        phys = (c // smt) if adjacent_siblings else c % (sockets * cores)
        return phys // cores
    for c in range(ncpu):
        d = root / "devices" / "system" / "cpu" / f"cpu{c}" / "topology"
        d.mkdir(parents=True)
        (d / "thread_siblings_list").write_text(",".join(str(s) for s in sibs(c)) + "\n")
    for n in range(sockets):
        d = root / "devices" / "system" / "node" / f"node{n}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in range(ncpu) if node_of(c) == n) + "\n")
    bdfs = []
    for i, n in enumerate(gpu_nodes):
        bdf = f"0000:{0x05 + 0x10 * i:02x}:00.0"
        d = root / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{n}\n")
        bdfs.append(bdf)
    return bdfs, list(range(ncpu))


def test_parse_cpulist():
    assert placement.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert placement.parse_cpulist("") == []


def test_each_rank_gets_two_physical_cores_on_its_gpus_node(tmp_path):
    bdfs, cpus = make_sysfs(tmp_path)
    used = set()
    for r in range(8):
        pair, facts = placement.pick_cores(r, bdfs, cpus, sysfs=str(tmp_path))
        node = 0 if r < 4 else 1
        assert facts["numa_node"] == node and facts["slot_on_node"] == r % 4 and facts["physical_cores_on_node"] == 64
        assert all(64 * node <= c < 64 * (node + 1) for c in pair)       # first hardware thread of cores on that socket
        assert pair[0] != pair[1] and abs(pair[0] - pair[1]) != 128      # not SMT siblings
        assert not used & set(pair)
        used |= set(pair)
    assert placement.pick_cores(0, bdfs, cpus, sysfs=str(tmp_path))[0] == [8, 9]      # the measured pair of round 3
    assert placement.pick_cores(4, bdfs, cpus, sysfs=str(tmp_path))[0] == [72, 73]    # rank 4: the other socket


def test_adjacent_smt_numbering_never_pairs_siblings(tmp_path):
    """Hosts that number the two threads of a core 2k, 2k + 1: CPUs c and c + 1 would be ONE core."""
    bdfs, cpus = make_sysfs(tmp_path, adjacent_siblings=True)
    pair, _ = placement.pick_cores(1, bdfs, cpus, sysfs=str(tmp_path))
    assert pair == [20, 22] and pair[0] // 2 != pair[1] // 2


def test_restricted_affinity_and_missing_information(tmp_path):
    bdfs, cpus = make_sysfs(tmp_path)
    # a container that may only use 8 CPUs of node 0: not enough physical cores after the skipped ones -> no pinning
    pair, why = placement.pick_cores(0, bdfs, list(range(8)), sysfs=str(tmp_path))
    assert pair is None and "not enough" in why
    # GPU without NUMA information (numa_node = -1) -> pairs from all allowed CPUs, by local rank
    (tmp_path / "bus" / "pci" / "devices" / bdfs[2] / "numa_node").write_text("-1\n")
    pair, facts = placement.pick_cores(2, bdfs, cpus, sysfs=str(tmp_path))
    assert facts["numa_node"] is None and pair == [12, 13]
    # unknown PCI address, unreadable sibling lists -> None, with the reason
    assert placement.pick_cores(9, bdfs, cpus, sysfs=str(tmp_path))[0] is None
    assert placement.pick_cores(0, bdfs, cpus, sysfs=str(tmp_path / "nowhere"))[0] in (None, [8, 9])
    pair, why = placement.pick_cores(0, [None], cpus, sysfs=str(tmp_path / "nowhere"))
    assert pair is None and "unreadable" in why


def test_pin_this_process_states_what_it_did(tmp_path):
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity")
    before = os.sched_getaffinity(0)
    try:
        rec = placement.pin_this_process(0, gpu_bdfs=[None], sysfs="/sys")
        if rec["cpus"] is None:   # this container: 8 CPUs -> "not enough", and nothing was changed
            assert os.sched_getaffinity(0) == before and rec["reason"]
        else:
            assert os.sched_getaffinity(0) == set(rec["cpus"]) and len(set(rec["cpus"])) == 2
    finally:
        os.sched_setaffinity(0, before)


def _fake_kfd(tmp_path, gpus, render_nodes):
    """A sysfs + /dev tree with one CPU node and `gpus` GPU nodes (location ids), `render_nodes` accessible render nodes."""
    base = tmp_path / "sys" / "class" / "kfd" / "kfd" / "topology" / "nodes"
    (base / "0").mkdir(parents=True)
    (base / "0" / "properties").write_text("simd_count 0\nlocation_id 0\ndomain 0\n")
    for i, loc in enumerate(gpus):
        (base / str(i + 1)).mkdir()
        (base / str(i + 1) / "properties").write_text(f"simd_count 1024\nlocation_id {loc}\ndomain 0\n")
    dri = tmp_path / "dev" / "dri"
    dri.mkdir(parents=True)
    for i in range(render_nodes):
        (dri / f"renderD{128 + i}").write_text("")
    return str(tmp_path / "sys"), str(tmp_path / "dev")


def test_kfd_topology_visibility_lists_and_restricted_containers(tmp_path):
    """HIP_ / CUDA_VISIBLE_DEVICES are ONE variable (HIP first, never applied twice); a topology that lists more
    GPUs than the process can open (a cgroup-restricted container) is not trusted; every refusal says why."""
    locs = [(0x05 << 8), (0x15 << 8), (0x65 << 8), (0x75 << 8)]
    sysfs, dev = _fake_kfd(tmp_path, locs, 4)
    all4 = ["0000:05:00.0", "0000:15:00.0", "0000:65:00.0", "0000:75:00.0"]
    assert placement.kfd_gpu_bdfs_why(sysfs, dev, {}) == (all4, None)
    # aliases: both set to the same list selects ONCE (applied twice, "2,3" of a 2-list would be out of range)
    got, why = placement.kfd_gpu_bdfs_why(sysfs, dev, {"HIP_VISIBLE_DEVICES": "2,3", "CUDA_VISIBLE_DEVICES": "2,3"})
    assert got == all4[2:] and why is None
    # HIP wins over its alias; ROCR is applied first, HIP indexes into what ROCR left
    assert placement.kfd_gpu_bdfs(sysfs, dev, {"HIP_VISIBLE_DEVICES": "1", "CUDA_VISIBLE_DEVICES": "3"}) == [all4[1]]
    assert placement.kfd_gpu_bdfs(sysfs, dev, {"CUDA_VISIBLE_DEVICES": "3"}) == [all4[3]]
    assert placement.kfd_gpu_bdfs(sysfs, dev, {"ROCR_VISIBLE_DEVICES": "2,3", "HIP_VISIBLE_DEVICES": "1"}) == [all4[3]]
    got, why = placement.kfd_gpu_bdfs_why(sysfs, dev, {"HIP_VISIBLE_DEVICES": "GPU-abcdef"})
    assert got is None and "HIP_VISIBLE_DEVICES" in why
    # the host has 4 GPUs, the container sees one render node: index 0 of the topology may be another GPU's NUMA node
    sysfs1, dev1 = _fake_kfd(tmp_path / "restricted", locs, 1)
    got, why = placement.kfd_gpu_bdfs_why(sysfs1, dev1, {})
    assert got is None and "render nodes" in why
    got, why = placement.kfd_gpu_bdfs_why(str(tmp_path / "nothing"), dev, {})
    assert got is None and "not listable" in why
    # the pool's boxes (BENCH r06: "kfd node 2 properties unreadable (PermissionError)"): the nodes of GPUs the container may
    # not use cannot be read -- they are skipped, and the one readable GPU node matches the one accessible render node
    sysfs2, dev2 = _fake_kfd(tmp_path / "pool", locs, 1)
    base = os.path.join(sysfs2, "class", "kfd", "kfd", "topology", "nodes")
    denied = {os.path.join(base, n, "properties") for n in ("1", "2", "4")}
    real_open = open

    def guarded_open(path, *a, **k):  # (root reads everything: permission bits on the fake tree would mean nothing to it)
        if path in denied:
            raise PermissionError(13, "Permission denied", path)
        return real_open(path, *a, **k)

    placement.open = guarded_open
    try:
        assert placement.kfd_gpu_bdfs_why(sysfs2, dev2, {}) == ([all4[2]], None)
    finally:
        del placement.open
