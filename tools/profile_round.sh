#!/bin/bash
# Reproduces the profile evidence under profiles/<round>/ on a GPU box (run from the repo root):
#   tools/profile_round.sh gpurun_out/prof_r02
# 1. rocprofv3 --kernel-trace of `bench.py --steps 1 --warmup 1` (the second prove is the timed one)
#    -> per-kernel stats csv + idle-gap attribution
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with a trace domain) -> HBM traffic json
# The program follows `--` directly (no env / bash -c hop under the profiler).
set -u
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 1 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-other-form"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o run --output-format rocpd csv -- $BENCH > "$OUT/kt.log" 2>&1
KT=$(find "$OUT/kt" -name "*_results.db" | head -1)
python3 tools/rocpd_stats.py "$KT" --second-half --csv "$OUT/prove_kernel_stats.csv" > "$OUT/prove_kernel_stats.txt" 2>&1
python3 tools/rocpd_gaps.py "$KT" --second-half --list-over 25 > "$OUT/idle_gaps.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o run --output-format rocpd -- $BENCH > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o run --output-format rocpd -- $BENCH > "$OUT/write.log" 2>&1
FE=$(find "$OUT/fetch" -name "*_results.db" | head -1)
WR=$(find "$OUT/write" -name "*_results.db" | head -1)
python3 tools/rocpd_pmc.py "$FE" "$WR" "$OUT/pmc_hbm_traffic.json" > "$OUT/pmc.txt" 2>&1
# 3. two SQ counter passes (instruction mix, waits, LDS conflicts) -> sq_counters.txt + the per-kernel VALU instruction
#    counts bench.py's per-class issue fractions are computed from
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS \
  -d "$OUT/sqa" -o run --output-format rocpd -- $BENCH > "$OUT/sqa.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  -d "$OUT/sqb" -o run --output-format rocpd -- $BENCH > "$OUT/sqb.log" 2>&1
SA=$(find "$OUT/sqa" -name "*_results.db" | head -1)
SB=$(find "$OUT/sqb" -name "*_results.db" | head -1)
python3 tools/rocpd_sq.py "$SA" "$SB" --json "$OUT/sq_valu.json" > "$OUT/sq_counters.txt" 2>&1
# the raw rocpd databases are tens of MB each and gpurun brings back at most 64 MiB: keep the summaries only
[ "${KEEP_RAW:-0}" = 1 ] || rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write" "$OUT/sqa" "$OUT/sqb"
ls -la "$OUT"
