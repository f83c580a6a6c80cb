"""The profile post-processing the round's evidence goes through (tools/rocpd_gaps.py): idle time must be the UNION over
streams (round 3's file summed per-stream gaps and contradicted itself) and the long gaps must be listed with the
kernels on either side. Synthetic rocpd database, no GPU."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _db(path, rows):
    db = sqlite3.connect(path)
    db.execute("create table kernels (name text, start integer, end integer, stream_id integer)")
    db.executemany("insert into kernels values (?, ?, ?, ?)", rows)
    db.commit()
    db.close()


def test_idle_is_the_union_over_streams_and_long_gaps_are_listed(tmp_path):
    us = 1000
    rows = [
        ("void ovm::a_kernel(int)", 0, 100 * us, 0),
        # the side stream's kernel covers the 40 us between a and b: that stretch is NOT idle
        ("void ovm::side_kernel(int)", 90 * us, 150 * us, 1),
        ("void ovm::b_kernel(int)", 140 * us, 200 * us, 0),
        # a real gap of 60 us after b, then a short one of 5 us after c
        ("void ovm::c_kernel(int)", 260 * us, 300 * us, 0),
        ("void ovm::d_kernel(int)", 305 * us, 400 * us, 0),
    ]
    path = str(tmp_path / "t.db")
    _db(path, rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_gaps.py"), path, "--list-over", "25"],
                         capture_output=True, text=True, check=True).stdout
    head = out.splitlines()[0]
    # span 400 us, busy = [0,200] + [260,300] + [305,400] = 335 us, idle 65 us; 10 + 10 us of kernels ran concurrently
    import re
    m = re.search(r"kernels (\d+), span ([\d.]+) ms, some kernel running ([\d.]+) ms, idle ([\d.]+) ms .*: ([\d.]+) ms ran concurrently", head)
    assert m, out
    assert int(m.group(1)) == 5 and float(m.group(2)) == 0.40, out
    assert abs(float(m.group(3)) - 0.335) < 0.006 and abs(float(m.group(4)) - 0.065) < 0.006 and float(m.group(5)) == 0.02, out
    assert "stream 0" in out and "stream 1" in out
    # attribution: 60 us after b_kernel, 5 us after c_kernel, nothing after a_kernel or side_kernel
    lines = {l.split()[0]: l for l in out.splitlines() if l.startswith("  ") and "gaps (" in l}
    assert "b_kernel" in lines and "1 gaps" in lines["b_kernel"].replace("   1 gaps", "1 gaps"), out
    assert "c_kernel" in lines and "a_kernel" not in lines and "side_kernel" not in lines, out
    # the one gap over 25 us, with its neighbours
    listed = [l for l in out.splitlines() if "->" in l]
    assert len(listed) == 1 and "b_kernel" in listed[0] and "c_kernel" in listed[0] and "60.0 us" in listed[0], out
