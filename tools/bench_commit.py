"""Micro-benchmark of the commit path (stack -> RS encode -> Merkle) at BASELINE config D shape."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g._load_pkg()
import torch

def main():
    log_h = int(os.environ.get("LOG_H", 24)); width = int(os.environ.get("WIDTH", 25))
    l_skip, lb, k = 4, 1, 4
    dev = pkg.Device(0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    st = torch.randint(0, 2, (width << log_h,), dtype=torch.int32, device="cuda", generator=gen)
    scratch = dev.empty(width << log_h)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for it in range(3):
        ev[0].record()
        code = dev.rs_encode(st, width, log_h, l_skip, lb, d_scratch=scratch)
        ev[1].record()
        layers = dev.merkle_tree(code, width, log_h + lb, k)
        ev[2].record()
        torch.cuda.synchronize()
        print(f"iter {it}: rs_encode {ev[0].elapsed_time(ev[1]):.2f} ms, merkle {ev[1].elapsed_time(ev[2]):.2f} ms")
        del code, layers
    # poseidon2 throughput
    n = 1 << 24
    s = torch.randint(0, 2**31 - 1, (16 * n,), dtype=torch.int32, device="cuda") % pkg.P
    dev.poseidon2_permute(s, n); torch.cuda.synchronize()
    t0 = time.time(); dev.poseidon2_permute(s, n); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"poseidon2: {n/dt/1e9:.2f} Gperm/s")
main()
