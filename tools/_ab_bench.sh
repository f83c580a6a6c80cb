#!/usr/bin/env bash
# interleaved A/B of bench.py between the product and a variant library: tools/_ab_bench.sh <outdir> <variant.so> [pairs]
O=$1; V=$2; N=${3:-2}
mkdir -p $O
for rep in $(seq 1 $N); do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-form > $O/prod_$rep.json 2> $O/prod_$rep.err
  OVM_HIP_LIB=$V python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-form > $O/var_$rep.json 2> $O/var_$rep.err
done
python - $O $N <<'PY'
import json,sys
o,n=sys.argv[1],int(sys.argv[2])
for tag in ("prod","var"):
    for i in range(1,n+1):
        try:
            d=json.load(open(f"{o}/{tag}_{i}.json"))
            st=d.get("stage_ms_additive",{}).get("top_level_on_the_callers_stream",{})
            print(tag,i,d["ms_per_step"],d["config"].get("proof_matches_checked_digest"),"hash_rows",st.get("merkle.hash_rows"),"compress",st.get("merkle.compress_layers"),"whir.rs",st.get("whir.rs_commit"),"ntt",st.get("ntt.dif_passes"),"gkr",st.get("gkr.layer_sumchecks"))
        except Exception as e: print(tag,i,"failed",e)
PY
