#!/usr/bin/env python3
"""Derives the compact shape fixture tests/golden/reth_block_23992138_shapes.json.gz from the
reference's captured profile (benchmarks/synthetic/reth-block-23992138-profile.jsonl, schema v2:
benchmarks/synthetic/src/segment_profile.rs:34-62). Only the numeric fields the synthetic AIR
generator consumes are kept (synthetic_runner.rs:111-129 `shape_from_record`):

  segments[i] = {"idx", "deg" (global max constraint degree), "airs": [[log_height, common_main_width,
                 num_constraints, num_interactions, max_constraint_degree, num_distinct_buses,
                 interaction_message_lens, interaction_count_weights], ...]}

Run in the authoring container (needs /root/reference):  python tools/make_reth_shapes.py
"""
import gzip
import json
import os
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/benchmarks/synthetic/reth-block-23992138-profile.jsonl"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "tests", "golden", "reth_block_23992138_shapes.json.gz")

segments = []
for line in open(SRC):
    line = line.strip()
    if not line:
        continue
    seg = json.loads(line)
    assert seg["schema"] == "v2"
    airs = []
    for a in seg["airs"]:
        assert len(a["buses"]) == a["num_interactions"] == len(a["interaction_message_lens"])
        airs.append([a["log_height"], a["width"]["common_main"], a["num_constraints"], a["num_interactions"],
                     a["max_constraint_degree"], len(set(a["buses"])), a["interaction_message_lens"],
                     a["interaction_count_weights"]])
    segments.append({"idx": seg["segment_idx"], "deg": seg["global_max_constraint_degree"], "airs": airs})
doc = {"source": "openvm-org/stark-backend v2.0.1 benchmarks/synthetic/reth-block-23992138-profile.jsonl",
       "generated_by": "tools/make_reth_shapes.py", "segments": segments}
with gzip.GzipFile(DST, "wb", mtime=0) as f:
    f.write(json.dumps(doc, separators=(",", ":")).encode())
print(DST, os.path.getsize(DST), "bytes;", len(segments), "segments,", sum(len(s["airs"]) for s in segments), "AIR records")
