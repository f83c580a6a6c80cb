#!/usr/bin/env bash
# The WHOLE replay (209 segments, unverified) on one box, three libraries interleaved, twice: round 4's last library and the
# mid-round one (built from their commits into stark-backend_amd/build_prev/, see docs/notebook/r05.md §8) against the tree's.
#   git archive <commit> stark-backend_amd include | tar -x -C _old/<commit>; (cd _old/<commit>/stark-backend_amd && OVM_OUT=... ./build.sh)
O=gpurun_out/r05/ab_full
mkdir -p $O
R04=$PWD/stark-backend_amd/build_prev/libopenvm_hip_6511fe3.so
MID=$PWD/stark-backend_amd/build_prev/libopenvm_hip_0718ab7.so
for rep in 1 2; do
  OVM_HIP_LIB=$R04 timeout 900 python tools/synthetic_replay.py --every 1 --out $O/r04_$rep.json > $O/r04_$rep.log 2>&1
  OVM_HIP_LIB=$MID timeout 900 python tools/synthetic_replay.py --every 1 --out $O/mid_$rep.json > $O/mid_$rep.log 2>&1
  timeout 900 python tools/synthetic_replay.py --every 1 --out $O/new_$rep.json > $O/new_$rep.log 2>&1
done
python - $O <<'PY'
import json,sys
o=sys.argv[1]
for n in ('r04','mid','new'):
    print(n, [json.load(open(f"{o}/{n}_{i}.json"))['total_prove_ms'] for i in (1,2)])
PY
{ nproc; lscpu | grep -E "Model name|MHz"; } > $O/box.txt 2>&1
