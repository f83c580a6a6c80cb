#!/usr/bin/env bash
# SQ counters + kernel stats of config B (the Keccak-f AIR) for the round-0 kernel: tools/profile_config_b.sh <outdir> [ENV=VAL ...]
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
CMD="python3 tools/prove_config_b.py --reps 2"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o run --output-format rocpd csv -- $CMD > "$OUT/kt.log" 2>&1
KT=$(find "$OUT/kt" -name "*_results.db" | head -1)
python3 tools/rocpd_stats.py "$KT" --second-half --csv "$OUT/kernel_stats.csv" > "$OUT/kernel_stats.txt" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS \
  -d "$OUT/sqa" -o run --output-format rocpd -- $CMD > "$OUT/sqa.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  -d "$OUT/sqb" -o run --output-format rocpd -- $CMD > "$OUT/sqb.log" 2>&1
SA=$(find "$OUT/sqa" -name "*_results.db" | head -1)
SB=$(find "$OUT/sqb" -name "*_results.db" | head -1)
python3 tools/rocpd_sq.py "$SA" "$SB" > "$OUT/sq_counters.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o run --output-format rocpd -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o run --output-format rocpd -- $CMD > "$OUT/write.log" 2>&1
FE=$(find "$OUT/fetch" -name "*_results.db" | head -1)
WR=$(find "$OUT/write" -name "*_results.db" | head -1)
python3 tools/rocpd_pmc.py "$FE" "$WR" "$OUT/pmc_hbm_traffic.json" > "$OUT/pmc.txt" 2>&1
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write" "$OUT/sqa" "$OUT/sqb"
head -12 "$OUT/kernel_stats.txt"; head -8 "$OUT/sq_counters.txt"
