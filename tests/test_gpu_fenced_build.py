"""The fenced twin of the library (tools/build_fenced.sh: -DOVM_RELEASE_FENCES=1) must make the same proofs as the
fence-free default (the default relies on every announced word being a write-through store that
`s_waitcnt vmcnt(0)` covers -- a gfx9-family property; the fenced build is the cross-check that stays in the suite).
A child process loads the twin through OVM_HIP_LIB, proves a mix of contexts (single AIR with the unbatched rounds, many
AIRs with absent ones, a conformance mixture) several times each, and prints a digest of all proof bytes; this process
does the same with the default library."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import airs
import conformance

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FENCED = os.path.join(ROOT, "stark-backend_amd", "build_fenced", "libopenvm_hip_fenced.so")


def contexts():
    out = []
    h = 1 << 14
    out.append((airs.test_params(l_skip=4, n_stack=12, k_whir=4), [airs.uniform_runner_air(24)],
                [(0, airs.splitmix64_bits(0xFE0CE, 24 * h), h, [])], None))
    for make in (lambda: conformance.mixture_traces_stark(10), lambda: conformance.optional_air(2),
                 lambda: conformance.multi_interaction_traces_stark(3)):
        c = make()
        out.append((c.params, c.airs, c.traces, c.preprocessed))
    return out


def digest(pkg, dev, repeat=5):
    import parity_util
    hsh = hashlib.sha256()
    for params, air_list, traces, pre in contexts():
        pk = pkg.ProvingKey(airs.pk_blob(params, air_list, preprocessed=pre))
        dtr = parity_util.upload_traces(dev, pkg, pk, air_list, traces)
        first = None
        for _ in range(repeat):
            proof, _ = dev.prove(pk, dtr)
            assert first is None or proof == first, "prove is not deterministic"
            first = proof
        hsh.update(first)
    return hsh.hexdigest()


def test_fenced_build_makes_the_same_proofs(dev, pkg):
    if not os.path.exists(FENCED):
        pytest.fail(f"{FENCED} is missing: __graft_entry__.build() / tools/build_fenced.sh builds it")
    want = digest(pkg, dev)
    env = dict(os.environ, OVM_HIP_LIB=FENCED)
    out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = dict(l.split("=", 1) for l in out.stdout.splitlines() if "=" in l)
    assert lines["lib"] == FENCED, lines
    assert lines["digest"] == want, "the fenced build's proofs differ from the default build's"


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    from conftest import load_pkg
    pkg = load_pkg()
    print(f"lib={pkg.LIB_PATH}")
    print(f"digest={digest(pkg, pkg.Device(0))}")
