#!/usr/bin/env bash
# kernel trace + idle gaps of one replay segment: tools/profile_replay_segment.sh <outdir> <segment>
OUT=$1; SEG=$2
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o run --output-format rocpd csv -- python3 tools/synthetic_replay.py --segment $SEG > "$OUT/kt.log" 2>&1
KT=$(find "$OUT/kt" -name "*_results.db" | head -1)
python3 tools/rocpd_stats.py "$KT" --second-half --csv "$OUT/kernel_stats.csv" > "$OUT/kernel_stats.txt" 2>&1
python3 tools/rocpd_gaps.py "$KT" --second-half --list-over 40 > "$OUT/idle_gaps.txt" 2>&1
python3 tools/rocpd_gaps.py "$KT" --second-half --timeline 0 1000 > "$OUT/timeline.txt" 2>&1
rm -rf "$OUT/kt"
grep -A30 "calls" "$OUT/kt.log" | head -40
