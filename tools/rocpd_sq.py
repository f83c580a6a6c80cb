#!/usr/bin/env python3
"""Per-kernel SQ counter ratios from two rocprofv3 --pmc passes (rocpd sqlite).
    python tools/rocpd_sq.py pass_a.db pass_b.db
pass A: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS
pass B: SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
Only the later half of each kernel's dispatches is used (the timed prove).
    python tools/rocpd_sq.py pass_a.db pass_b.db --json out.json
also writes, per kernel of the timed prove, the wave-level VALU instructions it executed (SQ_INSTS_VALU: a property of the
workload, not of the clock), its launches and its time under the profiler: what bench.py's `roofline.classes[*].issue_frac`
is computed from."""
import collections
import sqlite3
import sys


def load(path):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection "
                      "order by dispatch_id").fetchall()
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for n, disp, c, v, dur in rows:
        n = n.replace("void ", "").replace("ovm::", "").split("(")[0]
        d[n][c].append((v, dur))
    return d


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])

    def agg(d, k, c):
        v = d[k].get(c, [])
        v = v[len(v) // 2:]
        return sum(x[0] for x in v)

    if "--json" in sys.argv:
        import json
        out = {}
        for k in a:
            v = a[k].get("SQ_INSTS_VALU", [])
            v = v[len(v) // 2:]
            if v:
                out[k] = {"valu_wave_insts": int(sum(x[0] for x in v)), "launches": len(v), "ms": round(sum(x[1] for x in v) / 1e6, 4)}
        json.dump({"what": "wave-level VALU instructions per kernel over ONE prove (rocprofv3 --pmc SQ_INSTS_VALU ..., later half of "
                           "each kernel's dispatches of `bench.py --steps 1 --warmup 1`)", "kernels": out},
                  open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    ks = sorted(a, key=lambda k: -sum(x[1] for x in a[k].get("SQ_WAVE_CYCLES", [])))[:14]
    print("kernel                           ms   wait_any wait_inst act_inst valu_act lds_wait | per wave-cycle; "
          "valu lds vmem_rd vmem_wr salu insts (1e6) | lds bank-conflict frac")
    for k in ks:
        wc = agg(a, k, "SQ_WAVE_CYCLES") or 1
        v = a[k]["SQ_WAVE_CYCLES"]
        dur = sum(x[1] for x in v[len(v) // 2:]) / 1e6
        print(f"{k[:30]:30s} {dur:6.2f}  {agg(a, k, 'SQ_WAIT_ANY') / wc:6.2f} {agg(a, k, 'SQ_WAIT_INST_ANY') / wc:8.2f} "
              f"{agg(a, k, 'SQ_ACTIVE_INST_ANY') / wc:8.2f} {agg(a, k, 'SQ_ACTIVE_INST_VALU') / wc:8.2f} "
              f"{agg(a, k, 'SQ_WAIT_INST_LDS') / wc:8.2f} | {agg(a, k, 'SQ_INSTS_VALU') / 1e6:9.1f} "
              f"{agg(b, k, 'SQ_INSTS_LDS') / 1e6:8.1f} {agg(b, k, 'SQ_INSTS_VMEM_RD') / 1e6:8.1f} "
              f"{agg(b, k, 'SQ_INSTS_VMEM_WR') / 1e6:8.1f} {agg(b, k, 'SQ_INSTS_SALU') / 1e6:8.1f} | "
              f"{agg(b, k, 'SQ_LDS_BANK_CONFLICT') / max(agg(b, k, 'SQ_LDS_IDX_ACTIVE'), 1):5.2f}")


if __name__ == "__main__":
    main()
