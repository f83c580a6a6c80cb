#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels in one .hip file (no GPU needed).

    python tools/isa_count.py stark-backend_amd/csrc/merkle.hip [name-substring]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S",
                               "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-o", out, src],
                              stderr=subprocess.DEVNULL)
        cur, cnt, meta, entries = None, collections.defaultdict(collections.Counter), {}, []
        for line in open(out):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = m.group(1)
                continue
            if cur and re.match(r"^\s+(v_|s_|global_|ds_|buffer_|flat_|scratch_)", line):
                cnt[cur][line.split()[0]] += 1
            # amdhsa.kernels metadata: one YAML list item per kernel, keys in alphabetical order
            if re.match(r"^\s+- \.", line):
                entry = {}
                entries.append(entry)
            m = re.match(r"^\s+(?:- )?\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)", line)
            if m and entries:
                entries[-1][m.group(1)] = m.group(2)
        for e in entries:
            if "name" in e:
                meta[e["name"]] = {k: int(v) for k, v in e.items() if k != "name"}
        for k, v in cnt.items():
            if pat not in k:
                continue
            tot = sum(v.values())
            valu = sum(n for o, n in v.items() if o.startswith("v_"))
            print(f"{k[:90]}\n   total {tot}  valu {valu}  {meta.get(k, {})}\n   {v.most_common(10)}")


if __name__ == "__main__":
    main()
