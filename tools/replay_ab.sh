#!/usr/bin/env bash
# interleaved A/B of the replay sample: tools/replay_ab.sh <outdir> <pairs>
O=$1; N=${2:-4}
mkdir -p $O
PREV=$PWD/stark-backend_amd/build_prev/libopenvm_hip_prev.so
for rep in $(seq 1 $N); do
  OVM_HIP_LIB=$PREV timeout 600 python tools/synthetic_replay.py --every 10 --out $O/prev_$rep.json > $O/prev_$rep.log 2>&1
  timeout 600 python tools/synthetic_replay.py --every 10 --out $O/new_$rep.json > $O/new_$rep.log 2>&1
done
OVM_HIP_LIB=$PREV timeout 600 python tools/synthetic_replay.py --every 10 --profile --out $O/prev_prof.json > $O/prev_prof.log 2>&1
timeout 600 python tools/synthetic_replay.py --every 10 --profile --out $O/new_prof.json > $O/new_prof.log 2>&1
python - $O $N <<'PY'
import json,sys,statistics
o,n=sys.argv[1],int(sys.argv[2])
a=[json.load(open(f"{o}/prev_{i}.json"))['total_prove_ms'] for i in range(1,n+1)]
b=[json.load(open(f"{o}/new_{i}.json"))['total_prove_ms'] for i in range(1,n+1)]
print("prev",a,"median",statistics.median(a)); print("new ",b,"median",statistics.median(b))
pa=json.load(open(f"{o}/prev_prof.json"))['stage_ms']; pb=json.load(open(f"{o}/new_prof.json"))['stage_ms']
for k in list(pa)[:14]: print(f"  {k:28s} {pa[k]:8.1f} -> {pb.get(k,0):8.1f}")
PY
