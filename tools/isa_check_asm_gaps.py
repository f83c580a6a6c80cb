#!/usr/bin/env python3
"""Build-time guard for the hand-written interpreters: loads issued inside one asm statement and waited for in
a later one (the s_load_dwordx16 of the word groups, the global_load_dword into the landing registers of cached_vm.cuh)
are invisible to the compiler. If it ever placed a spill (v_writelane / v_readlane), a copy or a re-materialisation that
touches those registers BETWEEN two statements, the interpreter would read stale data silently.

For every kernel whose name matches, this compiles the file to gfx950 assembly (no GPU needed), takes the registers that
s_load / global_load / ds_read instructions INSIDE asm statements write, and fails if a compiler-placed instruction
between two asm statements of that kernel names one of them, or is a lane-spill instruction at all.

    python tools/isa_check_asm_gaps.py stark-backend_amd/csrc/stages.hip round0_cached gkr_leaves_cached
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\b([sv])(\d+)\b|\b([sv])\[(\d+):(\d+)\]")
SIGNATURE = "s_bitcmp1_b32"  # the word decode of the interpreters (cached_vm.cuh: OVM_WORD_ASM): marks a word loop
SPILL = re.compile(r"^(v_writelane_b32|v_readlane_b32|scratch_store|scratch_load|buffer_store_dword.*offen|buffer_load_dword.*offen)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def compile_to_asm(src, out):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(os.path.abspath(src)), "-o", out, src],
                          stderr=subprocess.DEVNULL)


def check(asm_path, patterns):
    """-> (per-kernel stats, list of findings). The loads at risk are the ones that stay in flight from one trip of an
    interpreter's word loop to the next. So: every INNERMOST loop of the kernel (a label and a backward branch to it with no
    other loop inside) that contains asm statements with loads; the registers those loads write; and every instruction the
    COMPILER placed inside that loop must not name one of them, nor be a spill. Only word loops are looked at (their asm
    statements decode program words: SIGNATURE); epilogue loops that merely contain a self-contained asm statement are not."""
    kernels, cur, in_asm = {}, None, False
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1) if any(p in m.group(1) for p in patterns) else None
            if cur:
                kernels[cur] = []  # (kind, text): kind in 'asm', 'gap', 'label'
            in_asm = False
            continue
        if not cur:
            continue
        if ".Lfunc_end" in line or line.startswith("\t.section"):
            cur = None
            continue
        if ";;#ASMSTART" in line:
            in_asm = True
            kernels[cur].append(("asm_start", ""))
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m and not in_asm:
            kernels[cur].append(("label", m.group(1)))
            continue
        ins = line.strip()
        if re.match(r"^(v_|s_|global_|ds_|buffer_|flat_|scratch_)", ins):
            kernels[cur].append(("asm" if in_asm else "gap", ins))
    findings, stats = [], {}
    for name, items in kernels.items():
        labels = {t: i for i, (k, t) in enumerate(items) if k == "label"}
        # loops = (label position, position of a backward branch to it); innermost = no other loop strictly inside
        loops = []
        for i, (kind, ins) in enumerate(items):
            if kind != "gap":
                continue
            m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", ins)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
        n_checked = 0
        for lo, hi in inner:
            body = items[lo:hi + 1]
            # registers whose load is still outstanding when its asm statement ENDS (a statement that waits for its own
            # loads leaves nothing in flight for the compiler to trip over)
            inflight, lgkm, vm = set(), set(), set()
            for kind, ins in body + [("asm_start", "")]:
                if kind == "asm_start":
                    inflight |= lgkm | vm
                    lgkm, vm = set(), set()
                elif kind == "asm":
                    if ins.startswith("s_waitcnt"):
                        if "lgkmcnt(0)" in ins:
                            lgkm = set()
                        if "vmcnt(0)" in ins:
                            vm = set()
                    elif re.match(r"^(s_load_|ds_read)", ins):
                        lgkm |= regs_of(ins.split(",")[0])
                    elif re.match(r"^global_load_", ins):
                        vm |= regs_of(ins.split(",")[0])
            if not inflight or not any(kind == "asm" and SIGNATURE in ins for kind, ins in body):
                continue  # a loop without asm-issued loads in flight, or not an interpreter's word loop
            for kind, ins in body:
                if kind != "gap":
                    continue
                n_checked += 1
                if SPILL.match(ins):
                    findings.append(f"{name}: spill inside a loop of asm statements: {ins}")
                hit = regs_of(ins) & inflight
                if hit:
                    findings.append(f"{name}: `{ins}` touches {sorted(hit)[:4]}, written by loads inside the loop's asm statements")
        stats[name] = (sum(1 for k, _ in items if k == "asm"), len(inner), n_checked)
    return stats, findings


def main():
    src, patterns = sys.argv[1], sys.argv[2:] or [""]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        compile_to_asm(src, out)
        kernels, findings = check(out, patterns)
    for name, (n_asm, n_loops, n_checked) in kernels.items():
        print(f"{name[:110]}: {n_asm} instructions inside asm statements, {n_loops} innermost loops, {n_checked} compiler-placed instructions checked")
    for f in findings:
        print("FINDING:", f)
    if not kernels:
        print("no kernel matched", patterns)
        return 2
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
