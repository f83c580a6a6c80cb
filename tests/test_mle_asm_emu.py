"""The generated assembly of the multilinear rounds' word executor (tools/gen_mle_asm.py -> csrc/mle_vm_asm.inc), checked on
the CPU: the committed file is what the generator writes, and the statement -- run on a one-lane emulator of the
instruction subset it uses (tools/mle_asm_emu.py) -- computes what a direct evaluation of the same words computes, for
random programs that put every operation x operand kind into every word position of both landing sets, ends, and reads
neither a slot nor a buffer it should not. (The GPU parity tests compare whole proofs; this one localises a wrong register.)"""
import os
import random
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_mle_asm as gen  # noqa: E402
import mle_asm_emu as emu  # noqa: E402


def test_committed_file_is_current():
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_mle_asm.py"), "--check"]).returncode == 0


def rand_ef(rng):
    return [rng.randrange(emu.P) for _ in range(4)]


def random_program(rng, n, n_cols, n_slots, n_weights):
    words, written = [], []
    for i in range(n):
        op = rng.choice([emu.W_LOAD, emu.W_ADD, emu.W_SUB, emu.W_RSUB, emu.W_MUL, emu.W_ACC, emu.W_ACC, emu.W_NOP])
        kinds = [emu.WK_COL, emu.WK_COL, emu.WK_CONST, emu.WK_SELF] + ([emu.WK_SLOT, emu.WK_SLOT] if written else [])
        kind = rng.choice(kinds)
        if op == emu.W_NOP:
            words.append(emu.word(emu.W_NOP, emu.WK_CONST))
            continue
        if op == emu.W_LOAD and kind == emu.WK_SELF:
            kind = emu.WK_COL
        payload = {emu.WK_COL: rng.randrange(n_cols), emu.WK_CONST: rng.choice([0, 1, rng.randrange(emu.P)]),
                   emu.WK_SELF: 0, emu.WK_SLOT: rng.choice(written) if written else 0}[kind]
        if op == emu.W_ACC:
            words.append(emu.word(op, kind, payload, k=rng.randrange(2), widx=rng.randrange(n_weights)))
            continue
        store = op != emu.W_LOAD and rng.random() < 0.4
        dst = rng.randrange(n_slots)
        words.append(emu.word(op, kind, payload, store=store, dst=dst))
        if store and dst not in written:
            written.append(dst)
    return words


@pytest.mark.parametrize("D", [1, 2, 3, 4])
def test_statement_matches_reference(D):
    rng = random.Random(1000 + D)
    lines = gen.Gen(D).build()
    for case in range(12):
        n = rng.choice([0, 1, 3, 8, 9, 24, 40, 77])
        n_cols, n_weights = 11, 5
        body, padded = emu.pad_words(random_program(rng, n, n_cols, 5, n_weights))
        cols = {c: (rand_ef(rng), rand_ef(rng)) for c in range(n_cols)}
        weights = [rand_ef(rng) for _ in range(n_weights)]
        out_init = [[rand_ef(rng), rand_ef(rng)] for _ in range(D)]
        log_h = rng.choice([1, 5, 12])
        yo = 32 * rng.randrange(1 << (log_h - 1))
        want = emu.reference(body, D, lambda c: cols[c], weights, out_init)
        got, steps = emu.Emu(lines, D, padded, weights, lambda c: cols[c], log_h, yo, out_init).run()
        assert got == want, (D, case, n)
        assert steps < 400 * (len(body) + 16)


def test_every_operation_in_every_position():
    """one word of interest at position p (0..7) of a pair of groups, for each operation x operand kind"""
    D = 3
    rng = random.Random(7)
    lines = gen.Gen(D).build()
    cols = {c: (rand_ef(rng), rand_ef(rng)) for c in range(4)}
    weights = [rand_ef(rng) for _ in range(2)]
    out_init = [[rand_ef(rng), rand_ef(rng)] for _ in range(D)]
    for p in range(8):
        for op in (emu.W_LOAD, emu.W_ADD, emu.W_SUB, emu.W_RSUB, emu.W_MUL, emu.W_ACC):
            for kind in (emu.WK_SLOT, emu.WK_COL, emu.WK_CONST, emu.WK_SELF):
                if op == emu.W_LOAD and kind == emu.WK_SELF:
                    continue
                # a prefix that leaves a value in the result registers and one in slot 2, NOPs up to position p (mod 8)
                prefix = [emu.word(emu.W_LOAD, emu.WK_COL, 1), emu.word(emu.W_MUL, emu.WK_COL, 2, store=True, dst=2),
                          emu.word(emu.W_ADD, emu.WK_CONST, 12345)]
                while len(prefix) % 8 != p:
                    prefix.append(emu.word(emu.W_NOP, emu.WK_CONST))
                payload = {emu.WK_SLOT: 2, emu.WK_COL: 3, emu.WK_CONST: 777, emu.WK_SELF: 0}[kind]
                w = emu.word(op, kind, payload, k=1, widx=1) if op == emu.W_ACC else emu.word(op, kind, payload, store=True, dst=1)
                tail = [emu.word(emu.W_ACC, emu.WK_SELF, 0, k=0, widx=0), emu.word(emu.W_ACC, emu.WK_SLOT, 2, k=1, widx=1)]
                if op != emu.W_ACC and op != emu.W_LOAD:
                    tail.append(emu.word(emu.W_ACC, emu.WK_SLOT, 1, k=0, widx=1))
                body, padded = emu.pad_words(prefix + [w] + tail)
                want = emu.reference(body, D, lambda c: cols[c], weights, out_init)
                got, _ = emu.Emu(lines, D, padded, weights, lambda c: cols[c], 6, 64, out_init).run()
                assert got == want, (p, op, kind)
