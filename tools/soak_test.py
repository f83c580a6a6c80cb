"""Soak test (GPU): 300 identical proofs must be byte-identical and must not grow device memory; 50 keys with a
preprocessed trace are created and freed. Run: python tools/soak_test.py"""
import sys, os, numpy as np, torch, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as ge, airs
pkg = ge._load_pkg(); dev = pkg.Device(0)
params = airs.test_params(l_skip=4, n_stack=10, k_whir=4)
pk = pkg.ProvingKey(airs.pk_blob(params, [airs.uniform_runner_air(20)]))
h = 1 << 12
tr = dev.upload(pkg.to_mont(airs.splitmix64_bits(1, 20 * h)))
first = None
for i in range(300):
    proof, _ = dev.prove(pk, [(0, tr, h, [])])
    if first is None: first = proof
    assert proof == first
    if i % 100 == 0:
        free, total = torch.cuda.mem_get_info(); print(i, "used GB", round((total - free) / 2**30, 3), flush=True)
free, total = torch.cuda.mem_get_info(); print("end used GB", round((total - free) / 2**30, 3))
# many keys in a row (key lifetime: preprocessed commit + free)
sels=[i%2==0 for i in range(16)]
for i in range(50):
    pre={0:(np.array([int(s) for s in sels],np.uint32),16)}
    k = pkg.ProvingKey(airs.pk_blob(airs.test_params(), [airs.fibonacci_selector_air()], preprocessed=pre)); del k
free, total = torch.cuda.mem_get_info(); print("after 50 keys used GB", round((total - free) / 2**30, 3))
# many-AIR proofs (batched MLE rounds, staged VM, per-job reduction counters): the captured segment with the
# most AIRs at heights clamped to 2^13, 60 times; identical bytes every time, no memory growth
segs = airs.load_reth_shapes()
si = max(range(len(segs)), key=lambda i: len(segs[i]["airs"]))
air_list, shapes = airs.synthetic_segment(segs[si], 13)
params2 = airs.app_params(22)
params2["max_constraint_degree"] = 4
pk2 = pkg.ProvingKey(airs.pk_blob(params2, air_list))
traces = [(i, torch.zeros(w << lh, dtype=torch.int32, device="cuda:0"), 1 << lh, []) for i, lh, w in shapes]
first = None
t0 = time.time()
for i in range(60):
    proof, _ = dev.prove(pk2, traces)
    if first is None: first = proof
    assert proof == first, f"many-AIR proof {i} differs"
free, total = torch.cuda.mem_get_info(); print(f"{len(air_list)} AIRs x 60 proofs identical, {time.time() - t0:.1f} s, used GB", round((total - free) / 2**30, 3))
# DIFFERENT proofs interleaved (a repeated identical proof would hide a buffer that still holds the previous proof's
# bytes: stale uploads, a pool block handed out twice): three contexts of different shapes in rotation, every proof
# compared with the first one of its context
rng = np.random.default_rng(7)
ctxs = []
for k, (n_airs, lh) in enumerate([(1, 12), (4, 9), (7, 6)]):
    al = [airs.random_air(rng, int(rng.integers(3, 16)), int(rng.integers(5, 40)), int(rng.integers(0, 3))) for _ in range(n_airs)]
    p3 = airs.test_params(l_skip=int(rng.integers(2, 5)), n_stack=10, k_whir=3)
    key = pkg.ProvingKey(airs.pk_blob(p3, al))
    trs = []
    for i, a in enumerate(al):
        hh = 1 << int(rng.integers(max(lh - 3, 0), lh + 1))
        w = a["common_main_width"]
        trs.append((i, dev.upload(pkg.to_mont(rng.integers(0, airs.P, size=w * hh, dtype=np.uint64).astype(np.uint32))), hh, []))
    ctxs.append((key, trs, None))
t0 = time.time()
for it in range(300):
    k = it % len(ctxs)
    key, trs, first = ctxs[k]
    proof, _ = dev.prove(key, trs)
    if first is None:
        ctxs[k] = (key, trs, proof)
    else:
        assert proof == first, f"interleaved proof {it} (context {k}) differs"
m = pkg.memory_stats()
print(f"300 interleaved proofs of 3 contexts identical to their first, {time.time() - t0:.1f} s; pool holds {m['segments'] / 2**20:.0f} MiB, "
      f"{m['driver_allocs']} hipMalloc calls in the process so far")
