#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (rocpd sqlite databases).

    python tools/rocpd_pmc.py fetch.db write.db out.json

FETCH_SIZE / WRITE_SIZE are in KiB. gfx950 reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so fetch_bytes_corrected = FETCH_SIZE * 1024 * 2; the
correction is uncalibrated for strided sub-line accesses. Only the later half of each kernel's
dispatches is kept (bench.py --steps 1 --warmup 1: the second prove is the timed one).
"""
import collections
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, dispatch_id, value, duration from counters_collection "
                      "where counter_name = ? order by dispatch_id", (counter,)).fetchall()
    d = collections.defaultdict(list)
    for n, _, v, dur in rows:
        n = n.replace("void ", "").replace("ovm::", "").split("(")[0]
        d[n].append((v, dur))
    out = {}
    for k, v in d.items():
        v = v[len(v) // 2:]
        out[k] = {"launches": len(v), "sum": sum(x[0] for x in v), "ms": sum(x[1] for x in v) / 1e6}
    return out


def main():
    fe, wr = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"_note": __doc__.strip().split("\n\n", 1)[1], "kernels": {}}
    tot_f = tot_w = 0
    for k in sorted(fe, key=lambda k: -fe[k]["ms"]):
        if k.startswith("at::") or "elementwise_kernel" in k or k.startswith("__amd_rocclr"):
            continue  # bench.py's own set-up kernels (torch) and runtime copies are not the prover's
        w = wr.get(k, {"sum": 0})
        tot_f += int(fe[k]["sum"] * 2048)
        tot_w += int(w["sum"] * 1024)
        if fe[k]["ms"] < 0.05:
            continue
        out["kernels"][k] = {"launches": fe[k]["launches"], "ms": round(fe[k]["ms"], 3),
                             "fetch_bytes_corrected": int(fe[k]["sum"] * 2048), "write_bytes": int(w["sum"] * 1024)}
    out["total_per_prove"] = {"fetch_bytes_corrected": tot_f, "write_bytes": tot_w}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(f"whole prove: fetch {tot_f / 1e9:.2f} GB  write {tot_w / 1e9:.2f} GB")
    for k, v in list(out["kernels"].items())[:20]:
        print(f"{k[:44]:44s} {v['launches']:4d} {v['ms']:8.3f} ms  fetch {v['fetch_bytes_corrected'] / 1e9:7.2f} GB  write {v['write_bytes'] / 1e9:7.2f} GB")


if __name__ == "__main__":
    main()
