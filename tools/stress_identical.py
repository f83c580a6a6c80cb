"""Stress (GPU): many proofs of one mid-size context must all be byte-identical -- a rare ordering race between a kernel's
mailbox / record stores and the flag or counter that announces them (device_common.h: release_to_host / release_to_device,
write-through stores + wait instead of fences) would show up as a differing or rejected proof.

    python tools/stress_identical.py [--proofs 400] [--log-rows 20] [--cols 100]
"""
import argparse, hashlib, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proofs", type=int, default=400)
    ap.add_argument("--log-rows", type=int, default=20)
    ap.add_argument("--cols", type=int, default=100)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge, airs
    pkg = ge._load_pkg(); dev = pkg.Device(0)
    params = airs.app_params(args.log_rows + 2)
    pk = pkg.ProvingKey(airs.pk_blob(params, [airs.uniform_runner_air(args.cols)]))
    h = 1 << args.log_rows
    tr = dev.upload(pkg.to_mont(airs.splitmix64_bits(7, args.cols * h)))
    first, t0 = None, time.time()
    for i in range(args.proofs):
        proof, _ = dev.prove(pk, [(0, tr, h, [])])
        d = hashlib.sha256(bytes(proof)).hexdigest()
        if first is None:
            first = d
        assert d == first, f"proof {i} differs: {d} vs {first}"
    print(f"{args.proofs} proofs of 2^{args.log_rows} x {args.cols} identical ({first[:16]}...), {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()
