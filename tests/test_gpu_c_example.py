"""The C ABI from a plain-C program (examples/prove_from_c.c): no PyTorch, no HIP headers in the
caller. Its proof must equal the one obtained through the Python binding and be accepted by the
verifier restatement."""
import os
import subprocess
import sys

import numpy as np
import pytest

import airs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_caller(dev, pkg, orc, tmp_path):
    exe = tmp_path / "prove_from_c"
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "prove_from_c.c"),
                           "-o", str(exe), "-L", os.path.dirname(pkg.LIB_PATH), "-lopenvm_hip",
                           f"-Wl,-rpath,{os.path.dirname(pkg.LIB_PATH)}"])
    params = airs.test_params(l_skip=4, n_stack=8, k_whir=4)
    air = airs.uniform_runner_air(20)
    blob = airs.pk_blob(params, [air])
    h = 1 << 8
    tr = airs.splitmix64_bits(0xD15EA5E, 20 * h).astype(np.uint32)
    blob.astype("<u4").tofile(tmp_path / "key.u32")
    tr.astype("<u4").tofile(tmp_path / "trace.u32")
    env = dict(os.environ)
    out = subprocess.run([str(exe), str(tmp_path / "key.u32"), str(tmp_path / "trace.u32"), "0", str(h),
                          str(tmp_path / "proof.bin")], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    got = (tmp_path / "proof.bin").read_bytes()
    pk = pkg.ProvingKey(blob)
    want, _ = dev.prove(pk, [(0, dev.upload(pkg.to_mont(tr)), h, [])])
    assert got == want
    orc.verify(blob, got)
