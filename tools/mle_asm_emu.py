#!/usr/bin/env python3
"""One-lane emulator of the instruction subset tools/gen_mle_asm.py emits, and a reference evaluation of MleWord programs
(csrc/program.h) in Python: the executor's LOGIC (which register, which branch, does the loop end) is checked on the CPU
(tests/test_mle_asm_emu.py) before a GPU ever runs it. Control flow in the statement is wave-uniform and all data is
lane-private, so one lane is the whole machine. Not a gfx950 model: no timing, no hazards, loads complete at issue."""
import re

P = 0x78000001
M32 = 0xFFFFFFFF
MONTY_M = 0x88000001
R = (1 << 32) % P
RINV = pow(R, P - 2, P)
W_LOAD, W_ADD, W_SUB, W_RSUB, W_MUL, W_ACC, W_NOP = 0, 1, 2, 3, 4, 5, 7
WK_SLOT, WK_COL, WK_CONST, WK_SELF = 0, 1, 2, 3


def to_mont(x):
    return x % P * R % P


def mont_mul(a, b):
    return a * b * RINV % P


ELEVEN = to_mont(11)


# ---- extension field on Montgomery words (the reference) ----
def ef_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def ef_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def ef_mul(a, b):
    c = [0] * 4
    for i in range(4):
        for j in range(4):
            t = mont_mul(a[i], b[j])
            if i + j >= 4:
                t = mont_mul(t, ELEVEN)
            c[(i + j) % 4] = (c[(i + j) % 4] + t) % P
    return c


def word(op, kind, payload=0, store=False, dst=0, k=0, widx=0):
    return (op | kind << 3 | (32 if store else 0) | (k & 1) << 6 | dst << 16, payload, widx, 0)


def pad_words(words):
    """a group's executable words (a multiple of 8, at least 8) followed by the two groups of NOPs the fetch runs into"""
    words = list(words)
    nop = word(W_NOP, WK_CONST)
    while not words or len(words) % 8:
        words.append(nop)
    return words, words + [nop] * 8


def reference(words, D, col_pair, weights, out_init):
    """words: (x, y, z, w) tuples; col_pair(c) -> (row 2y, row 2y+1) as EF lists; weights: EF lists; out_init[x][a]"""
    acc = [[0] * 4 for _ in range(D)]
    slots = {}
    out = [[list(out_init[x][a]) for a in (0, 1)] for x in range(D)]
    for wx, wy, wz, _ in words:
        op, kind = wx & 7, (wx >> 3) & 3
        if op == W_NOP:
            continue
        if kind == WK_SLOT:
            v = [list(slots[(wy, x)]) for x in range(D)]
        elif kind == WK_COL:
            lo, hi = col_pair(wy)
            d = ef_sub(hi, lo)
            v = [list(hi)]
            for x in range(1, D):
                v.append(ef_add(v[-1], d))
        elif kind == WK_CONST:
            v = [[wy, 0, 0, 0] for _ in range(D)]
        else:
            v = [list(a) for a in acc]
        if op == W_ACC:
            for x in range(D):
                a = (wx >> 6) & 1
                out[x][a] = ef_add(out[x][a], ef_mul(v[x], weights[wz]))
            continue
        for x in range(D):
            if op == W_LOAD:
                acc[x] = v[x]
            elif op == W_ADD:
                acc[x] = ef_add(acc[x], v[x])
            elif op == W_SUB:
                acc[x] = ef_sub(acc[x], v[x])
            elif op == W_RSUB:
                acc[x] = ef_sub(v[x], acc[x])
            else:
                acc[x] = ef_mul(acc[x], v[x])
        if wx & 32:
            for x in range(D):
                slots[(wx >> 16, x)] = list(acc[x])
    return out


class Emu:
    """lines: Gen(D).build(); operands: name -> value (scalars) or ('v', index) for vector operands bound to registers"""

    WORDS_BASE, WEIGHTS_BASE, MATS_BASE = 0x1000_0000, 0x2000_0000, 0x1_0000_0000

    def __init__(self, lines, D, words_padded, weights, col_pair, log_h, yo, out_init, lds_rows=8):
        self.D, self.log_h, self.yo, self.col_pair = D, log_h, yo, col_pair
        self.s = [0] * 104
        self.v = [0xDEAD0000 + i for i in range(256)]  # poison: a register read before it is written shows up
        self.vcc = self.scc = 0
        self.lds = {}
        self.words_mem = [c for w in words_padded for c in w]
        self.weights_mem = []
        for e in weights:
            self.weights_mem += list(e) + [mont_mul(e[1], ELEVEN), mont_mul(e[2], ELEVEN), mont_mul(e[3], ELEVEN), 0]
        # operands: scalars live in pseudo registers, the out registers in v0..v(8D-1), yo in v40, lds in v41
        self.ops = {"M": MONTY_M, "P": P, "E": ELEVEN, "sh": log_h + 4, "ng": (len(words_padded) - 8) // 4,
                    "wp": self.WORDS_BASE, "wts": self.WEIGHTS_BASE, "mats": self.MATS_BASE}
        self.vops = {"yo": 40, "lds": 41}
        self.v[40], self.v[41] = yo, 0x40  # (the lane's LDS byte address)
        for x in range(D):
            for a in (0, 1):
                for k in range(4):
                    idx = 8 * x + 4 * a + k
                    self.vops[f"o{x}{a}{k}"] = idx
                    self.v[idx] = out_init[x][a][k]
        self.prog, self.labels = [], {}
        for ln in lines:
            if ln.endswith(":"):
                self.labels[ln[:-1]] = len(self.prog)
            else:
                self.prog.append(ln)
        self.loads_in_flight_checked = 0

    # ---- operand access ----
    def rd(self, tok):
        tok = tok.strip()
        m = re.fullmatch(r"%\[(\w+)\]", tok)
        if m:
            n = m.group(1)
            return self.v[self.vops[n]] if n in self.vops else self.ops[n]
        m = re.fullmatch(r"s(\d+)", tok)
        if m:
            return self.s[int(m.group(1))]
        m = re.fullmatch(r"v(\d+)", tok)
        if m:
            return self.v[int(m.group(1))]
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
        if m:
            a, b = int(m.group(1)), int(m.group(2))
            return sum(self.s[a + i] << (32 * i) for i in range(b - a + 1))
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            a, b = int(m.group(1)), int(m.group(2))
            return sum(self.v[a + i] << (32 * i) for i in range(b - a + 1))
        return int(tok, 0)

    def wr(self, tok, val, width=1):
        tok = tok.strip()
        m = re.fullmatch(r"%\[(\w+)\]", tok)
        if m:
            self.v[self.vops[m.group(1)]] = val & M32
            return
        m = re.fullmatch(r"([sv])(\d+)", tok)
        if m:
            (self.s if m.group(1) == "s" else self.v)[int(m.group(2))] = val & M32
            return
        m = re.fullmatch(r"([sv])\[(\d+):(\d+)\]", tok)
        a, b = int(m.group(2)), int(m.group(3))
        f = self.s if m.group(1) == "s" else self.v
        for i in range(b - a + 1):
            f[a + i] = (val >> (32 * i)) & M32

    def mem_dwords(self, addr, n):
        if self.WORDS_BASE <= addr < self.WEIGHTS_BASE:
            i = (addr - self.WORDS_BASE) // 4
            return self.words_mem[i:i + n]
        if self.WEIGHTS_BASE <= addr < self.WEIGHTS_BASE + 4 * len(self.weights_mem):
            i = (addr - self.WEIGHTS_BASE) // 4
            return self.weights_mem[i:i + n]
        off = addr - self.MATS_BASE
        col, within = off >> (self.log_h + 4), off & ((1 << (self.log_h + 4)) - 1)
        assert within in (self.yo, self.yo + 16), (hex(addr), col, within)
        return list(self.col_pair(col)[0 if within == self.yo else 1])

    def run(self, max_steps=2_000_000):
        pc, steps = 0, 0
        while pc < len(self.prog):
            steps += 1
            if steps > max_steps:
                raise RuntimeError("the statement does not end")
            ln = self.prog[pc]
            pc += 1
            op, _, rest = ln.partition(" ")
            a = [t.strip() for t in rest.split(",")] if rest else []
            if op in ("s_waitcnt",):
                continue
            if op == "s_branch":
                pc = self.labels[a[0]]
            elif op == "s_cbranch_scc0":
                if not self.scc:
                    pc = self.labels[a[0]]
            elif op == "s_cbranch_scc1":
                if self.scc:
                    pc = self.labels[a[0]]
            elif op in ("s_mov_b32", "s_mov_b64", "v_mov_b32"):
                self.wr(a[0], self.rd(a[1]))
            elif op == "s_movk_i32":
                self.wr(a[0], int(a[1], 0))
            elif op in ("s_load_dwordx16", "s_load_dwordx8"):
                n = 16 if op.endswith("16") else 8
                d = self.mem_dwords(self.rd(a[1]) + self.rd(a[2]), n)
                assert len(d) == n, "scalar load beyond the buffer"
                self.wr(a[0], sum(x << (32 * i) for i, x in enumerate(d)))
            elif op == "s_bfe_u32":
                c = self.rd(a[2])
                r = (self.rd(a[1]) >> (c & 31)) & ((1 << ((c >> 16) & 127)) - 1)
                self.wr(a[0], r)
                self.scc = int(r != 0)
            elif op == "s_cmp_eq_u32":
                self.scc = int(self.rd(a[0]) == self.rd(a[1]))
            elif op == "s_cmp_lg_u32":
                self.scc = int(self.rd(a[0]) != self.rd(a[1]))
            elif op == "s_bitcmp1_b32":
                self.scc = (self.rd(a[0]) >> (self.rd(a[1]) & 31)) & 1
            elif op == "s_lshl_b64":
                self.wr(a[0], (self.rd(a[1]) << (self.rd(a[2]) & 63)) & ((1 << 64) - 1))
            elif op == "s_lshl_b32":
                self.wr(a[0], self.rd(a[1]) << (self.rd(a[2]) & 31))
            elif op == "s_lshr_b32":
                self.wr(a[0], self.rd(a[1]) >> (self.rd(a[2]) & 31))
            elif op == "s_add_u32":
                r = self.rd(a[1]) + self.rd(a[2])
                self.wr(a[0], r)
                self.scc = r >> 32
            elif op == "s_addc_u32":
                r = self.rd(a[1]) + self.rd(a[2]) + self.scc
                self.wr(a[0], r)
                self.scc = r >> 32
            elif op == "s_sub_u32":
                x, y = self.rd(a[1]), self.rd(a[2])
                self.wr(a[0], x - y)
                self.scc = int(y > x)
            elif op == "s_and_b32":
                r = self.rd(a[1]) & self.rd(a[2])
                self.wr(a[0], r)
                self.scc = int(r != 0)
            elif op == "s_mul_i32":
                self.wr(a[0], self.rd(a[1]) * self.rd(a[2]))
            elif op == "v_add_u32":
                self.wr(a[0], self.rd(a[1]) + self.rd(a[2]))
            elif op == "v_sub_u32":
                self.wr(a[0], self.rd(a[1]) - self.rd(a[2]))
            elif op == "v_min_u32":
                self.wr(a[0], min(self.rd(a[1]), self.rd(a[2])))
            elif op == "v_mul_lo_u32":
                self.wr(a[0], self.rd(a[1]) * self.rd(a[2]))
            elif op == "v_mul_hi_u32":
                self.wr(a[0], (self.rd(a[1]) * self.rd(a[2])) >> 32)
            elif op == "v_mad_u64_u32":
                r = self.rd(a[2]) * self.rd(a[3]) + self.rd(a[4])
                assert r < (1 << 64), "a sum of products overflows 64 bits"
                self.wr(a[0], r)
            elif op == "ds_read_b128":
                adr, off = a[1].split(" offset:")
                key = self.rd(adr) + int(off)
                assert key in self.lds, f"LDS read of a slot never written ({key:#x})"
                self.wr(a[0], self.lds[key])
            elif op == "ds_write_b128":
                data, off = a[1].split(" offset:")
                self.lds[self.rd(a[0]) + int(off)] = self.rd(data)
            elif op == "global_load_dwordx4":
                base, _, off = a[2].partition(" offset:")
                d = self.mem_dwords(self.rd(base) + self.rd(a[1]) + (int(off) if off else 0), 4)
                self.wr(a[0], sum(x << (32 * i) for i, x in enumerate(d)))
            else:
                raise NotImplementedError(ln)
        return [[[self.v[8 * x + 4 * a_ + k] for k in range(4)] for a_ in (0, 1)] for x in range(self.D)], steps
