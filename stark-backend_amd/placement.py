"""Host placement of a prover process: which two host cores the thread that calls the prover (it spins on host-mapped
flags between ~900 kernel launches per proof) and its helper thread should run on.

Rules (measured on the 2 x 64-core MI355X host, profiles/r03/host_pinning.txt: unpinned 76.0-76.6 ms per config D proof,
two physical cores 74.9-75.4 ms, two SMT siblings of ONE core 78 ms):
  * two DISTINCT PHYSICAL cores -- never a core and its SMT sibling (`topology/thread_siblings_list`);
  * on the NUMA node of the GPU the rank drives (`/sys/bus/pci/devices/<bdf>/numa_node`): with 8 GPUs on two sockets half of
    the ranks would otherwise spin on flags their GPU writes across the socket link;
  * the ranks that share a node take consecutive core pairs, after a few cores left to the OS and the runtime's threads;
  * anything missing or ambiguous in sysfs -> no pinning (None): placement is an optimisation, never a requirement.

Pure functions over a sysfs root so that the rules are testable without the hardware (tests/test_placement_cpu.py).
The reference binds a device per engine/thread (crates/cuda-backend/src/device.rs:58-79) and leaves host placement to the
embedding application; bench.py does it here and states the choice in its JSON line (INTEGRATION.md "Host placement").
"""
import os

SKIP_CORES = 8  # physical cores per node left alone (interrupts and the runtime's own threads land on the first ones; 8,9 is the measured pair)


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def gpu_numa_node(pci_bdf, sysfs="/sys"):
    """NUMA node of a PCI device ('0000:05:00.0'), or None when sysfs does not say (-1 = no affinity)."""
    if not pci_bdf:
        return None
    txt = _read(os.path.join(sysfs, "bus", "pci", "devices", pci_bdf.lower(), "numa_node"))
    try:
        node = int(txt)
    except (TypeError, ValueError):
        return None
    return node if node >= 0 else None


def physical_cores(cpus, sysfs="/sys"):
    """The allowed CPUs grouped into physical cores: [[cpu, sibling, ...], ...] sorted by each core's lowest CPU; only
    the allowed members of a core are listed. None when a CPU's sibling list cannot be read."""
    allowed, seen, cores = set(cpus), set(), []
    for c in sorted(allowed):
        if c in seen:
            continue
        txt = _read(os.path.join(sysfs, "devices", "system", "cpu", f"cpu{c}", "topology", "thread_siblings_list"))
        if txt is None:
            return None
        sibs = [s for s in parse_cpulist(txt) if s in allowed] or [c]
        seen.update(sibs)
        cores.append(sorted(sibs))
    return cores


def pick_cores(local_rank, gpu_bdfs, allowed_cpus, sysfs="/sys", skip=SKIP_CORES):
    """Two CPUs (distinct physical cores, on the node of GPU `local_rank`) for this rank, with the facts the choice rests
    on, or (None, reason). `gpu_bdfs`: the PCI address of every local rank's GPU, by local rank (ranks that share a node
    take consecutive pairs in local-rank order)."""
    if local_rank >= len(gpu_bdfs):
        return None, "no PCI address for this rank's GPU"
    nodes = [gpu_numa_node(b, sysfs) for b in gpu_bdfs]
    node = nodes[local_rank]
    if node is None:
        cpus = sorted(allowed_cpus)  # no NUMA information: every allowed CPU is as good as another
        slot = local_rank
        why = "numa_node unknown"
    else:
        txt = _read(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist"))
        if txt is None:
            return None, f"node{node}/cpulist unreadable"
        cpus = sorted(set(parse_cpulist(txt)) & set(allowed_cpus))
        slot = sum(1 for r in range(local_rank) if nodes[r] == node)
        why = f"numa node {node}"
    cores = physical_cores(cpus, sysfs)
    if cores is None:
        return None, "thread_siblings_list unreadable"
    first = skip + 2 * slot
    if first + 1 >= len(cores):
        return None, f"{len(cores)} physical cores on {why}: not enough for slot {slot}"
    pair = [cores[first][0], cores[first + 1][0]]
    return pair, {"numa_node": node, "slot_on_node": slot, "physical_cores_on_node": len(cores), "basis": why}


def kfd_gpu_bdfs_why(sysfs="/sys", dev="/dev", environ=None):
    """(PCI addresses of the GPUs in HIP's device order, None) WITHOUT touching the runtime, or (None, reason). The KFD
    topology lists the GPU nodes (simd_count > 0) in the order the runtime enumerates them, with `domain` and
    `location_id` (bus << 8 | device << 3 | function). ROCR_VISIBLE_DEVICES, then HIP_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES
    is its alias: read only when HIP_ is unset, never applied on top of it) are honoured as plain index lists; anything
    else (UUIDs) gives up. In a container whose cgroup exposes fewer GPUs than the host has, the topology still lists
    every GPU of the host while the runtime enumerates only the accessible ones: the number of render nodes the process
    can open is compared with the topology's count, and a difference gives up too (the caller asks torch then)."""
    environ = os.environ if environ is None else environ
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError as e:
        return None, f"kfd topology not listable ({e.__class__.__name__})"
    gpus, skipped = [], 0
    for n in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(base, str(n), "properties")) if len(line.split()) >= 2)
        except PermissionError:
            # a GPU of the host this container may not use (the pool's boxes: every node but the leased GPU's reads
            # "permission denied"): the runtime does not enumerate it either, so it is skipped, not a reason to give up;
            # the count of render nodes below still has to agree with what is left
            skipped += 1
            continue
        except OSError as e:
            return None, f"kfd node {n} properties unreadable ({e.__class__.__name__})"
        if int(props.get("simd_count", "0")) == 0:
            continue
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        gpus.append(f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}")
    if not gpus:
        return None, "kfd topology lists no GPU node" + (f" this process may read ({skipped} unreadable)" if skipped else "")
    try:
        render = [n for n in os.listdir(os.path.join(dev, "dri")) if n.startswith("renderD")]
        usable = [n for n in render if os.access(os.path.join(dev, "dri", n), os.R_OK | os.W_OK)]
    except OSError:
        usable = None
    if usable is not None and len(usable) != len(gpus):
        return None, f"kfd topology lists {len(gpus)} GPUs, {len(usable)} render nodes are accessible (restricted container)"
    hip = environ.get("HIP_VISIBLE_DEVICES")
    if hip is None or hip == "":
        hip = environ.get("CUDA_VISIBLE_DEVICES")
    for name, v in (("ROCR_VISIBLE_DEVICES", environ.get("ROCR_VISIBLE_DEVICES")), ("HIP_VISIBLE_DEVICES", hip)):
        if v is None or v == "":
            continue
        try:
            gpus = [gpus[int(i)] for i in v.split(",")]
        except (ValueError, IndexError):
            return None, f"{name}={v!r} is not a list of indices into {len(gpus)} GPUs"
    return (gpus, None) if gpus else (None, "no GPU left after the visibility lists")


def kfd_gpu_bdfs(sysfs="/sys", dev="/dev", environ=None):
    return kfd_gpu_bdfs_why(sysfs, dev, environ)[0]


def torch_gpu_bdfs():
    """PCI addresses of the visible GPUs by device index, from torch's device properties. This INITIALISES the HIP / HSA
    runtime: its helper threads start with the mask the process has at that moment (pin_this_process re-pins every thread
    of the process afterwards)."""
    import torch
    out = []
    for i in range(torch.cuda.device_count()):
        p = torch.cuda.get_device_properties(i)
        dom, bus, dev = getattr(p, "pci_domain_id", None), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None)
        out.append(None if bus is None or dev is None else f"{(dom or 0):04x}:{bus:02x}:{dev:02x}.0")
    return out


def pin_this_process(local_rank, gpu_bdfs=None, sysfs="/sys", slot_rank=None):
    """Pins EVERY thread of the calling process (threads created later inherit the mask) and returns the JSON-able record of
    what was done: {"cpus": [a, b], ...facts} or {"cpus": None, "reason": ...}. The GPUs' PCI addresses come from the KFD
    topology in sysfs -- no runtime call, so the HIP / HSA helper threads are created AFTER the pinning and inherit the pair
     -- and from torch only if sysfs does not have them; either way the threads that already exist (the runtime's,
    an OpenMP pool's) are pinned one by one through /proc/self/task. `slot_rank`: ranks that share one GPU (bench.py
    --one-gpu) pass their own rank so that they do not share cores."""
    if not hasattr(os, "sched_setaffinity"):
        return {"cpus": None, "reason": "no sched_setaffinity on this platform"}
    try:
        allowed = sorted(os.sched_getaffinity(0))
        source = "caller"
        kfd_reason = None
        if gpu_bdfs is None:
            (gpu_bdfs, kfd_reason), source = kfd_gpu_bdfs_why(sysfs), "kfd topology (sysfs)"
        if gpu_bdfs is None:
            gpu_bdfs = torch_gpu_bdfs()
            source = f"torch device properties (the runtime was initialised before the pinning; sysfs: {kfd_reason})"
        if slot_rank is not None:  # several ranks on GPU `local_rank`: one pair each, as if each had its own GPU there
            gpu_bdfs = [gpu_bdfs[local_rank] if local_rank < len(gpu_bdfs) else None] * (slot_rank + 1)
            local_rank = slot_rank
        pair, facts = pick_cores(local_rank, gpu_bdfs, allowed, sysfs)
        if pair is None:
            return {"cpus": None, "reason": facts}
        os.sched_setaffinity(0, set(pair))
        repinned = 0
        try:  # the threads that already exist keep their own masks: pin them too
            for tid in os.listdir("/proc/self/task"):
                try:
                    os.sched_setaffinity(int(tid), set(pair))
                    repinned += 1
                except (OSError, ValueError):
                    pass
        except OSError:
            pass
        return dict({"cpus": pair, "all_cpus": len(allowed), "gpu_addresses_from": source, "threads_pinned": repinned}, **facts)
    except (OSError, ValueError) as e:
        return {"cpus": None, "reason": repr(e)}
