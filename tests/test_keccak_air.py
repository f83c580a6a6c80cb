"""BASELINE config B's AIR: p3-keccak-air's `KeccakAir`, restated (tests/airs.py: keccak_air, keccak_trace).
CPU part: the trace generator against an independent Keccak-f[1600] (itself pinned by hashlib's SHA3-256), the shape
facts the reference tree does hold (the captured KeccakfPermAir shape = this AIR + 1 column + 1 constraint), the oracle
proving and verifying a small instance, and a flipped trace bit being rejected.
GPU part: byte parity of the HIP proof against the oracle, the reference bench's own instance (2^10 permutations,
`app_params_with_100_bits_security(21)`: crates/cpu-backend/benches/keccakf.rs:26,45-52), and BASELINE's 2^20 rows."""
import hashlib
import time

import numpy as np
import pytest

import airs

K = airs.KeccakCols


def _sha3_256(msg):
    rate = 136
    m = bytearray(msg) + b"\x06"
    m += b"\x00" * ((-len(m)) % rate)
    m[-1] |= 0x80
    st = [0] * 25
    for off in range(0, len(m), rate):
        for i in range(rate // 8):
            st[i] ^= int.from_bytes(m[off + 8 * i:off + 8 * i + 8], "little")
        st = airs.keccak_f1600(st)
    return b"".join(s.to_bytes(8, "little") for s in st)[:32]


def test_keccak_f1600_is_sha3s_permutation():
    for msg in (b"", b"abc", bytes(range(200))):
        assert _sha3_256(msg) == hashlib.sha3_256(msg).digest()


def test_trace_generator_computes_keccak_f():
    """Row 23 of every permutation block holds the permutation's output (A''' of the last round) and row 0 its input."""
    inp = airs.keccak_inputs(5, seed=7)
    tr, final = airs.keccak_trace(inp, 7)          # 128 rows: 5 real permutations + a cut-off zero one
    h = 128
    tr = tr.numpy().reshape(K.WIDTH, h)
    for p in range(5):
        want = airs.keccak_f1600([int(v) for v in inp[p]])
        assert final[p] == want
        for i in (0, 7, 24):
            y, x = i // 5, i % 5
            limbs_in = [int(tr[K.A + (5 * y + x) * 4 + l, 24 * p]) for l in range(4)]
            assert sum(v << (16 * l) for l, v in enumerate(limbs_in)) == int(inp[p][x + 5 * y])
            col = (lambda l: K.A_PPP_00_LIMBS + l) if i == 0 else (lambda l: K.A_PRIME_PRIME + (5 * y + x) * 4 + l)
            limbs_out = [int(tr[col(l), 24 * p + 23]) for l in range(4)]
            assert sum(v << (16 * l) for l, v in enumerate(limbs_out)) == want[x + 5 * y]
    assert tr[K.EXPORT].sum() == 0 and tr[K.STEP_FLAGS:K.STEP_FLAGS + 24].sum() == h


def test_shape_matches_what_the_reference_tree_records():
    """2 633 columns / 3 182 constraints / degree 3; OpenVM's KeccakfPermAir (this AIR + 1 column + 1 constraint) was
    captured at 2 634 / 3 183 in the reference's reth-block profile."""
    air = airs.keccak_air()
    assert air["common_main_width"] == K.WIDTH == 2633
    assert len(air["constraint_idx"]) == 3182 and air["max_constraint_degree"] == 3 and air["need_rot"]
    assert not air["interactions"]
    assert any(r[1] == 2634 and r[2] == 3183 for s in airs.load_reth_shapes() for r in s["airs"])


def test_oracle_proves_and_a_flipped_bit_is_rejected(orc):
    air = airs.keccak_air()
    blob = airs.pk_blob(airs.app_params(12), [air])
    tr, _ = airs.keccak_trace(airs.keccak_inputs(1), 5)
    tr = tr.numpy().astype(np.uint32)
    proof = orc.prove(blob, [(0, tr, 32, [])])["proof"]
    orc.verify(blob, proof)
    for col, row in ((K.A_PRIME + 700, 3), (K.C + 5, 17), (K.STEP_FLAGS + 2, 2), (K.A_PP_00_BITS + 63, 23)):
        bad = tr.copy()
        bad[col * 32 + row] ^= 1
        with pytest.raises(Exception):
            orc.verify(blob, orc.prove(blob, [(0, bad, 32, [])])["proof"])


def test_the_compiled_program_fits_the_executors(pkg):
    """The zerocheck program of the real AIR: emitted in DAG order it needed 645 simultaneously live values (more than
    any executor holds: ovm_hip_prove failed with hipErrorNotSupported); demand-driven scheduling with recomputation of
    leaf-leaf operations needs a handful. The staged form, the plain interpreter and a direct evaluation of the DAG
    agree on a random row (host self-test)."""
    blob = airs.pk_blob(airs.app_params(16), [airs.keccak_air()])
    st = pkg.program_stats(blob)
    assert st["slots"] <= 16 and st["instrs"] < 50000 and st["remat_limit"] >= 1, st
    assert pkg.staged_selftest(blob, 8, 8) > 400


def _hip_and_oracle(dev, pkg, orc, n_perms, log_rows, log_stacked):
    import torch
    air = airs.keccak_air()
    blob = airs.pk_blob(airs.app_params(log_stacked), [air])
    one = int(pkg.to_mont(np.array([1], dtype=np.uint32))[0])
    inp = airs.keccak_inputs(n_perms)
    canon, _ = airs.keccak_trace(inp, log_rows, device="cuda:0")
    mont = ((canon.to(torch.int64) * one) % airs.P).to(torch.int32)
    pk = pkg.ProvingKey(blob)
    got, _ = dev.prove(pk, [(0, mont, 1 << log_rows, [])])
    t0 = time.time()
    want = orc.prove(blob, [(0, canon.cpu().numpy().astype(np.uint32), 1 << log_rows, [])])["proof"]
    print(f"oracle prove of KeccakAir 2^{log_rows} rows: {time.time() - t0:.1f} s; proof {len(got)} bytes")
    orc.verify(blob, got)
    return got, want


@pytest.mark.gpu
@pytest.mark.parametrize("n_perms,log_rows,log_stacked", [(1, 5, 12), (42, 10, 16)])
def test_gpu_keccak_air_byte_identical(dev, pkg, orc, n_perms, log_rows, log_stacked):
    got, want = _hip_and_oracle(dev, pkg, orc, n_perms, log_rows, log_stacked)
    assert got == want


@pytest.mark.gpu
def test_gpu_keccak_reference_bench_instance_byte_identical(dev, pkg, orc):
    """The instance the reference's own keccakf bench proves: 2^10 permutations (24 576 rows -> 2^15), app parameters
    with log_stacked_height 21. Byte for byte against the oracle."""
    got, want = _hip_and_oracle(dev, pkg, orc, 1 << 10, 15, 21)
    assert got == want


@pytest.mark.gpu
def test_gpu_keccak_flipped_bit_is_rejected(dev, pkg, orc):
    """A proof made by the HIP prover from a trace with ONE flipped bit must not verify."""
    import torch
    air = airs.keccak_air()
    blob = airs.pk_blob(airs.app_params(16), [air])
    one = int(pkg.to_mont(np.array([1], dtype=np.uint32))[0])
    canon, _ = airs.keccak_trace(airs.keccak_inputs(42), 10, device="cuda:0")
    canon[(K.A_PRIME + 11) * 1024 + 100] ^= 1
    mont = ((canon.to(torch.int64) * one) % airs.P).to(torch.int32)
    proof, _ = dev.prove(pkg.ProvingKey(blob), [(0, mont, 1024, [])])
    with pytest.raises(Exception):
        orc.verify(blob, proof)


@pytest.mark.gpu
def test_gpu_config_b_keccak_2_20_rows_verifies(dev, pkg, orc):
    """BASELINE config B: KeccakAir, 2^20 rows (43 690 seeded permutations + padding), blowup 2, app parameters --
    2.76e9 cells, codeword 22 GB. The verifier restatement accepts; proving twice gives the same bytes; a flipped
    proof byte is rejected."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * (1 << 30):
        pytest.skip("needs ~150 GB of free HBM")
    air = airs.keccak_air()
    blob = airs.pk_blob(airs.app_params(24), [air])
    one = int(pkg.to_mont(np.array([1], dtype=np.uint32))[0])
    h = 1 << 20
    t0 = time.time()
    mont, final = airs.keccak_trace(airs.keccak_inputs(43690), 20, device="cuda:0", mont_one=one)
    torch.cuda.synchronize()
    print(f"trace generation on the GPU: {time.time() - t0:.1f} s")
    pk = pkg.ProvingKey(blob)
    proof, _ = dev.prove(pk, [(0, mont, h, [])])
    orc.verify(blob, proof)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    proof2, _ = dev.prove(pk, [(0, mont, h, [])])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(f"config B (real KeccakAir): {K.WIDTH} x 2^20 = {K.WIDTH * h / 1e9:.2f}e9 cells proved in {ms:.1f} ms "
          f"({K.WIDTH * h / ms / 1e6:.2f}e9 cells/s)")
    assert proof2 == proof
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    with pytest.raises(Exception):
        orc.verify(blob, bytes(bad))
    del mont
    pkg._lib().ovm_hip_trim()
