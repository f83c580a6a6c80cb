#!/usr/bin/env python3
"""Generates stark-backend_amd/csrc/mle_vm_asm.inc: the gfx950 assembly of the multilinear rounds' word executor
(csrc/mle_vm.cuh, program.h: MleWord), one statement per number of points D = 1..4.

    python tools/gen_mle_asm.py            # rewrites the file
    python tools/gen_mle_asm.py --check    # exit 1 if the committed file differs (tests/test_tools_cpu.py)

Why a generator: the executor is ONE inline-assembly statement per D with its whole state in registers it names itself
(the result registers, the operand registers, two sets of landing registers for the column loads in flight, the
multiplier's 64-bit sums), because every form that left register allocation to the compiler paid for it: the state of the
interpreter changed registers between the code of one word position and the next (24-44 v_mov per word), and with the
three-waves-per-SIMD register budget it was spilled. Register numbers that depend on D cannot be pasted together by the
preprocessor; a few Python loops can.

Register map (D points; B = 64):
    acc[x][k]  v[B + 4x + k]                 the result registers (x: point, k: limb)
    opv[x][k]  v[B + 4D + 4x + k]            the operand at the D points
    land[s][j] v[LB + 8(4s + j) ..+7]        rows 2y, 2y+1 of word j's column, set s (LB = B + 8D)
    sums       v[T .. T+7]  (T = LB + 64)    four 64-bit sums of products;  T+8..T+12: 11*b / temporaries;  T+13: LDS address
    s[36:51] the group being executed (shifted down a word at a time: the current word is s[36:39]),
    s[52:67] the next group (its columns are requested while this one runs), s[68:83] the group after (being fetched),
    s[84:85] column address (s84 also the weight record's offset, s85 the operand kind and scratch), s86 op,
    s87 byte offset of the next group fetch, s[88:95] weight record, s96 groups left, s97 position 0..7 (set * 4 + word),
    s[98:99] base of the matrices. (s32 and s100 up are reserved: stack pointer, vcc / flat_scratch / xnack_mask.)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "stark-backend_amd", "csrc", "mle_vm_asm.inc")
B = 64
P, NEG_P = "0x78000001", "0x87ffffff"
W_LOAD, W_ADD, W_SUB, W_RSUB, W_MUL, W_ACC, W_NOP = 0, 1, 2, 3, 4, 5, 7
WK_SLOT, WK_COL, WK_CONST, WK_SELF = 0, 1, 2, 3


class Gen:
    def __init__(self, D):
        self.D = D
        self.LB = B + 8 * D
        self.T = self.LB + 64
        self.lines = []

    # ---- registers ----
    def acc(self, x, k):
        return f"v{B + 4 * x + k}"

    def opv(self, x, k):
        return f"v{B + 4 * self.D + 4 * x + k}"

    def quad(self, first):
        return f"v[{first}:{first + 3}]"

    def land(self, s, j, hi, k=None):
        base = self.LB + 8 * (4 * s + j) + (4 if hi else 0)
        return self.quad(base) if k is None else f"v{base + k}"

    def sum64(self, i):
        return f"v[{self.T + 2 * i}:{self.T + 2 * i + 1}]"

    def slo(self, i):
        return f"v{self.T + 2 * i}"

    def shi(self, i):
        return f"v{self.T + 2 * i + 1}"

    def tmp(self, i):  # T+8 .. T+12
        return f"v{self.T + 8 + i}"

    @property
    def adr(self):
        return f"v{self.T + 13}"

    @property
    def last_vgpr(self):
        return self.T + 13

    # ---- emit ----
    def e(self, s):
        self.lines.append(s)

    def label(self, name):
        self.lines.append(f".Lmle%=_{name}:")

    def L(self, name):
        return f".Lmle%=_{name}"

    # a = a + b / a - b / b - a  (mod P), t a temporary; b may be an SGPR or 0
    def fadd(self, a, b, t):
        self.e(f"v_add_u32 {t}, {b}, {a}")
        self.e(f"v_add_u32 {a}, {NEG_P}, {t}")
        self.e(f"v_min_u32 {a}, {a}, {t}")

    def fsub(self, a, b, t):
        self.e(f"v_sub_u32 {t}, {a}, {b}")
        self.e(f"v_add_u32 {a}, {P}, {t}")
        self.e(f"v_min_u32 {a}, {a}, {t}")

    def frsub(self, a, b, t):
        self.e(f"v_sub_u32 {t}, {b}, {a}")
        self.e(f"v_add_u32 {a}, {P}, {t}")
        self.e(f"v_min_u32 {a}, {a}, {t}")

    # dst = a * b (Montgomery), b an SGPR or VGPR; t0, t1 temporaries (dst may be a)
    def fmul(self, dst, a, b, t0, t1):
        self.e(f"v_mul_lo_u32 {t0}, {a}, {b}")
        self.e(f"v_mul_hi_u32 {t1}, {a}, {b}")
        self.e(f"v_mul_lo_u32 {t0}, {t0}, %[M]")
        self.e(f"v_mul_hi_u32 {t0}, {t0}, %[P]")
        self.e(f"v_sub_u32 {t1}, {t1}, {t0}")
        self.e(f"v_add_u32 {t0}, {P}, {t1}")
        self.e(f"v_min_u32 {dst}, {t1}, {t0}")

    # the sixteen products of a[0..3] x (b[0..3]; w[1..3] = 11 b[1..3]) into the four sums
    def schoolbook(self, a, b, w):
        rows = [
            (b[0], b[1], b[2], b[3]),
            (w[3], b[0], b[1], b[2]),
            (w[2], w[3], b[0], b[1]),
            (w[1], w[2], w[3], b[0]),
        ]
        for i, row in enumerate(rows):
            for c in range(4):
                src2 = "0" if i == 0 else self.sum64(c)
                self.e(f"v_mad_u64_u32 {self.sum64(c)}, vcc, {a[i]}, {row[c]}, {src2}")

    # Montgomery reduction of the four sums (bb31.h mont_reduce_wide) into dst[0..3]; temporaries T+8..T+11
    def reduce4(self, dst):
        for i in range(4):
            self.e(f"v_add_u32 {self.tmp(i)}, {NEG_P}, {self.shi(i)}")
            self.e(f"v_mul_lo_u32 {self.slo(i)}, {self.slo(i)}, %[M]")
        for i in range(4):
            self.e(f"v_min_u32 {self.shi(i)}, {self.shi(i)}, {self.tmp(i)}")
            self.e(f"v_mul_hi_u32 {self.slo(i)}, {self.slo(i)}, %[P]")
        for i in range(4):
            self.e(f"v_sub_u32 {self.shi(i)}, {self.shi(i)}, {self.slo(i)}")
            self.e(f"v_add_u32 {self.tmp(i)}, {P}, {self.shi(i)}")
            self.e(f"v_min_u32 {dst[i]}, {self.shi(i)}, {self.tmp(i)}")

    # the D points of a column operand from landing registers (set s, word j) into reg(x, k): X = 1 is row 2y + 1, every
    # further point adds the difference (kept in the registers of row 2y)
    def column_points(self, s, j, reg):
        t = self.tmp(0)
        for k in range(4):
            lo, hi = self.land(s, j, False, k), self.land(s, j, True, k)
            if self.D > 1:
                self.e(f"v_sub_u32 {t}, {hi}, {lo}")
                self.e(f"v_add_u32 {lo}, {P}, {t}")
                self.e(f"v_min_u32 {lo}, {lo}, {t}")
            self.e(f"v_mov_b32 {reg(0, k)}, {hi}")
            for x in range(1, self.D):
                self.e(f"v_add_u32 {t}, {reg(x - 1, k)}, {lo}")
                self.e(f"v_add_u32 {reg(x, k)}, {NEG_P}, {t}")
                self.e(f"v_min_u32 {reg(x, k)}, {reg(x, k)}, {t}")

    def requests(self, s):
        """the column operands of the group in s[52:67] into landing set s"""
        for j in range(4):
            wx, wy = 52 + 4 * j, 53 + 4 * j
            skip = f"req{s}{j}"
            self.e(f"s_bfe_u32 s86, s{wx}, 0x20003")
            self.e(f"s_cmp_eq_u32 s86, {WK_COL}")
            self.e(f"s_cbranch_scc0 {self.L(skip)}")
            self.e(f"s_mov_b32 s84, s{wy}")
            self.e("s_mov_b32 s85, 0")
            self.e("s_lshl_b64 s[84:85], s[84:85], %[sh]")
            self.e("s_add_u32 s84, s84, s98")
            self.e("s_addc_u32 s85, s85, s99")
            self.e(f"global_load_dwordx4 {self.land(s, j, False)}, %[yo], s[84:85]")
            self.e(f"global_load_dwordx4 {self.land(s, j, True)}, %[yo], s[84:85] offset:16")
            self.label(skip)

    def slot_address(self, sreg):
        """adr = this lane's 16 bytes of point 0 of slot `sreg`"""
        self.e(f"s_mul_i32 s85, {sreg}, {self.D * 4096}")
        self.e(f"v_add_u32 {self.adr}, s85, %[lds]")

    def build(self):
        D = self.D
        e, L, label = self.e, self.L, self.label
        accs = lambda x: [self.acc(x, k) for k in range(4)]
        opvs = lambda x: [self.opv(x, k) for k in range(4)]
        outs = lambda x, a: [f"%[o{x}{a}{k}]" for k in range(4)]
        # ---- prologue ----
        e("s_mov_b64 s[98:99], %[mats]")
        e("s_load_dwordx16 s[36:51], %[wp], 0x0")
        e("s_load_dwordx16 s[52:67], %[wp], 0x40")
        e("s_movk_i32 s87, 0x80")
        e("s_mov_b32 s96, %[ng]")
        e("s_mov_b32 s97, 0")
        for x in range(D):
            for k in range(4):
                e(f"v_mov_b32 {self.acc(x, k)}, 0")
        e("s_waitcnt lgkmcnt(0)")
        # the first group's columns: it is "the next group" once, before the loop (set 0 from s[36:51])
        for j in range(4):
            wx, wy = 36 + 4 * j, 37 + 4 * j
            skip = f"req_first{j}"
            e(f"s_bfe_u32 s86, s{wx}, 0x20003")
            e(f"s_cmp_eq_u32 s86, {WK_COL}")
            e(f"s_cbranch_scc0 {L(skip)}")
            e(f"s_mov_b32 s84, s{wy}")
            e("s_mov_b32 s85, 0")
            e("s_lshl_b64 s[84:85], s[84:85], %[sh]")
            e("s_add_u32 s84, s84, s98")
            e("s_addc_u32 s85, s85, s99")
            e(f"global_load_dwordx4 {self.land(0, j, False)}, %[yo], s[84:85]")
            e(f"global_load_dwordx4 {self.land(0, j, True)}, %[yo], s[84:85] offset:16")
            label(skip)
        # ---- group top ----
        label("group")
        e("s_load_dwordx16 s[68:83], %[wp], s87")
        e("s_add_u32 s87, s87, 0x40")
        e("s_waitcnt vmcnt(0)")  # the set this group reads has landed (nothing else is in flight)
        e("s_bitcmp1_b32 s97, 2")
        e(f"s_cbranch_scc1 {L('req_to_0')}")
        self.requests(1)
        e(f"s_branch {L('word')}")
        label("req_to_0")
        self.requests(0)
        # ---- one word ----
        label("word")
        e("s_and_b32 s86, s36, 7")
        e(f"s_cmp_eq_u32 s86, {W_NOP}")
        e(f"s_cbranch_scc1 {L('next')}")
        e("s_bfe_u32 s85, s36, 0x20003")
        e(f"s_cmp_lg_u32 s86, {W_ACC}")
        e(f"s_cbranch_scc1 {L('no_weight')}")
        e("s_lshl_b32 s84, s38, 5")  # (s84: nothing writes it before the wait in `mac`)
        e("s_load_dwordx8 s[88:95], %[wts], s84")
        label("no_weight")
        e(f"s_cmp_eq_u32 s85, {WK_COL}")
        e(f"s_cbranch_scc1 {L('col')}")
        e(f"s_cmp_eq_u32 s85, {WK_SLOT}")
        e(f"s_cbranch_scc1 {L('slot')}")
        e(f"s_cmp_eq_u32 s85, {WK_CONST}")
        e(f"s_cbranch_scc1 {L('const')}")
        # WK_SELF
        for x in range(D):
            for k in range(4):
                e(f"v_mov_b32 {self.opv(x, k)}, {self.acc(x, k)}")
        e(f"s_branch {L('have')}")
        # ---- slot operand ----
        label("slot")
        self.slot_address("s37")
        e(f"s_cmp_eq_u32 s86, {W_LOAD}")
        e(f"s_cbranch_scc1 {L('slot_load')}")
        for x in range(D):
            e(f"ds_read_b128 {self.quad(B + 4 * D + 4 * x)}, {self.adr} offset:{x * 4096}")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_branch {L('have')}")
        label("slot_load")
        for x in range(D):
            e(f"ds_read_b128 {self.quad(B + 4 * x)}, {self.adr} offset:{x * 4096}")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_branch {L('next')}")
        # ---- constant operand: limb 0 only ----
        label("const")
        t = self.tmp(0)
        e(f"s_cmp_eq_u32 s86, {W_ADD}")
        e(f"s_cbranch_scc1 {L('addc')}")
        e(f"s_cmp_eq_u32 s86, {W_SUB}")
        e(f"s_cbranch_scc1 {L('subc')}")
        e(f"s_cmp_eq_u32 s86, {W_RSUB}")
        e(f"s_cbranch_scc1 {L('rsubc')}")
        e(f"s_cmp_eq_u32 s86, {W_LOAD}")
        e(f"s_cbranch_scc1 {L('loadc')}")
        # multiply by / accumulate a constant: through the general forms
        for x in range(D):
            e(f"v_mov_b32 {self.opv(x, 0)}, s37")
            for k in range(1, 4):
                e(f"v_mov_b32 {self.opv(x, k)}, 0")
        e(f"s_branch {L('have')}")
        label("addc")
        for x in range(D):
            self.fadd(self.acc(x, 0), "s37", t)
        e(f"s_branch {L('store')}")
        label("subc")
        for x in range(D):
            self.fsub(self.acc(x, 0), "s37", t)
        e(f"s_branch {L('store')}")
        label("rsubc")
        for x in range(D):
            self.frsub(self.acc(x, 0), "s37", t)
            for k in range(1, 4):
                self.frsub(self.acc(x, k), "0", t)
        e(f"s_branch {L('store')}")
        label("loadc")
        for x in range(D):
            e(f"v_mov_b32 {self.acc(x, 0)}, s37")
            for k in range(1, 4):
                e(f"v_mov_b32 {self.acc(x, k)}, 0")
        e(f"s_branch {L('next')}")
        # ---- column operand: from the landing registers of this position ----
        label("col")
        for p in range(8):
            if p < 7:
                e(f"s_cmp_eq_u32 s97, {p}")
                e(f"s_cbranch_scc1 {L(f'col{p}')}")
            else:
                e(f"s_branch {L(f'col{p}')}")
        for p in range(8):
            s, j = p >> 2, p & 3
            label(f"col{p}")
            e(f"s_cmp_eq_u32 s86, {W_LOAD}")
            e(f"s_cbranch_scc1 {L(f'col{p}_load')}")
            self.column_points(s, j, self.opv)
            e(f"s_branch {L('have')}")
            label(f"col{p}_load")
            self.column_points(s, j, self.acc)
            e(f"s_branch {L('next')}")
        # ---- the operation on (acc, opv) ----
        label("have")
        e(f"s_cmp_eq_u32 s86, {W_ADD}")
        e(f"s_cbranch_scc1 {L('add')}")
        e(f"s_cmp_eq_u32 s86, {W_SUB}")
        e(f"s_cbranch_scc1 {L('sub')}")
        e(f"s_cmp_eq_u32 s86, {W_MUL}")
        e(f"s_cbranch_scc1 {L('mul')}")
        e(f"s_cmp_eq_u32 s86, {W_RSUB}")
        e(f"s_cbranch_scc1 {L('rsub')}")
        e(f"s_cmp_eq_u32 s86, {W_ACC}")
        e(f"s_cbranch_scc1 {L('mac')}")
        # W_LOAD of the result register itself: nothing to do
        e(f"s_branch {L('next')}")
        for name, f in (("add", self.fadd), ("sub", self.fsub), ("rsub", self.frsub)):
            label(name)
            for x in range(D):
                for k in range(4):
                    f(self.acc(x, k), self.opv(x, k), t)
            e(f"s_branch {L('store')}")
        label("mul")
        for x in range(D):
            a, b = accs(x), opvs(x)
            w = [None, self.tmp(0), self.tmp(1), self.tmp(2)]
            for k in range(1, 4):
                self.fmul(w[k], b[k], "%[E]", self.tmp(3), self.tmp(4))
            self.schoolbook(a, b, w)
            self.reduce4(a)
        e(f"s_branch {L('store')}")
        label("mac")
        e("s_waitcnt lgkmcnt(0)")
        e("s_bitcmp1_b32 s36, 6")
        e(f"s_cbranch_scc1 {L('mac1')}")
        for a_idx in (0, 1):
            if a_idx == 1:
                label("mac1")
            for x in range(D):
                b = ["s88", "s89", "s90", "s91"]
                w = [None, "s92", "s93", "s94"]
                self.schoolbook(opvs(x), b, w)
                red = [self.slo(i) for i in range(4)]
                self.reduce4(red)
                o = outs(x, a_idx)
                for k in range(4):
                    self.fadd(o[k], red[k], self.tmp(k))
            e(f"s_branch {L('next')}")
        # ---- the result to its slot ----
        label("store")
        e("s_bitcmp1_b32 s36, 5")
        e(f"s_cbranch_scc0 {L('next')}")
        e("s_lshr_b32 s85, s36, 16")
        self.slot_address("s85")
        for x in range(D):
            e(f"ds_write_b128 {self.adr}, {self.quad(B + 4 * x)} offset:{x * 4096}")
        # ---- next word / next group ----
        label("next")
        e("s_mov_b64 s[36:37], s[40:41]")
        e("s_mov_b64 s[38:39], s[42:43]")
        e("s_mov_b64 s[40:41], s[44:45]")
        e("s_mov_b64 s[42:43], s[46:47]")
        e("s_mov_b64 s[44:45], s[48:49]")
        e("s_mov_b64 s[46:47], s[50:51]")
        e("s_add_u32 s97, s97, 1")
        e("s_and_b32 s85, s97, 3")
        e("s_cmp_lg_u32 s85, 0")
        e(f"s_cbranch_scc1 {L('word')}")
        e("s_and_b32 s97, s97, 7")
        e("s_waitcnt lgkmcnt(0)")  # the group after the next has arrived
        for i in range(0, 16, 2):
            e(f"s_mov_b64 s[{36 + i}:{37 + i}], s[{52 + i}:{53 + i}]")
        for i in range(0, 16, 2):
            e(f"s_mov_b64 s[{52 + i}:{53 + i}], s[{68 + i}:{69 + i}]")
        e("s_sub_u32 s96, s96, 1")
        e("s_cmp_lg_u32 s96, 0")
        e(f"s_cbranch_scc1 {L('group')}")
        e("s_waitcnt vmcnt(0)")  # (the requests of the group of NOPs behind the last one: none, but nothing stays in flight)
        return self.lines


def render():
    out = [
        "// GENERATED by tools/gen_mle_asm.py -- do not edit; regenerate with `python tools/gen_mle_asm.py`.",
        "// The word executor of the multilinear rounds (csrc/mle_vm.cuh) for D = 1..4 points: the statement's text, its",
        "// operand lists and the registers it names (clobbers).",
        "#pragma once",
        "",
    ]
    for D in (1, 2, 3, 4):
        g = Gen(D)
        lines = g.build()
        out.append(f"#define OVM_MLE_ASM_D{D} \\")
        for ln in lines:
            out.append(f'    "{ln}\\n\\t" \\')
        out.append('    ""')
        ops = []
        for x in range(D):
            for a in (0, 1):
                for k in range(4):
                    ops.append(f'[o{x}{a}{k}] "+v"(out[{x}][{a}].c[{k}].v)')
        out.append(f"#define OVM_MLE_OUT_D{D} " + ", ".join(ops))
        vregs = ", ".join(f'"v{i}"' for i in range(B, g.last_vgpr + 1))
        out.append(f"#define OVM_MLE_VGPRS_D{D} {vregs}")
        out.append(f"#define OVM_MLE_LAST_VGPR_D{D} {g.last_vgpr}")
        out.append("")
    sregs = ", ".join(f'"s{i}"' for i in range(36, 100))
    out.append(f"#define OVM_MLE_SGPRS {sregs}")
    out.append("")
    return "\n".join(out)


def main():
    text = render()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("mle_vm_asm.inc is stale: run python tools/gen_mle_asm.py")
            sys.exit(1)
        return
    with open(OUT, "w") as f:
        f.write(text)
    print(f"wrote {OUT}: " + ", ".join(f"D={D}: {len(Gen(D).build())} instructions" for D in (1, 2, 3, 4)))


if __name__ == "__main__":
    main()
