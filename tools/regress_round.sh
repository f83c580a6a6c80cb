#!/bin/bash
# The round-end regression on ONE GPU box, everything on the library that is in the tree (run from the repo root):
#   tools/regress_round.sh <outdir> <commit hash> [quick]
# 1. the whole `-m gpu` suite              5. config B: prove + stage table, kernel stats, SQ counters, PMC traffic
# 2. configs B and E at full size, bytes   6. configs C and E benches
# 3. tools/profile_round.sh (config D)     7. the 209-segment replay, every proof verified
# 4. bench.py with default flags           8. 500 identical proofs, smoke()
# `quick` skips 2, 7 and 8. The commit hash goes into every log's first line (the box has no .git).
set -u
O=${1:-gpurun_out/regress}
HEAD_HASH=${2:-unknown}
MODE=${3:-full}
mkdir -p "$O"
export TMPDIR=/tmp
stamp() { echo "# commit $HEAD_HASH; library sha256 $(sha256sum stark-backend_amd/libopenvm_hip.so | cut -c1-16); $(date -u +%FT%TZ)"; }

{ stamp; timeout 1500 python -m pytest tests -q -m gpu --durations=15; } > "$O/gputest_all.log" 2>&1
tail -2 "$O/gputest_all.log"
if [ "$MODE" != quick ]; then
  { stamp; OVM_FULL_PARITY_BE=1 timeout 1800 python -m pytest tests/test_gpu_full_size_oneoff.py -q -m gpu -s; } > "$O/gputest_full_size_b_e.log" 2>&1
  tail -2 "$O/gputest_full_size_b_e.log"
  { nproc; grep -E "MemTotal|MemAvailable" /proc/meminfo; lscpu | grep -E "Model name|Socket|Thread"; } > "$O/box_host.txt" 2>&1
fi
timeout 900 bash tools/profile_round.sh "$O/prof" > "$O/profile_round.log" 2>&1
# the ceiling of the permutation's own instruction stream (bench.py: roofline.issue_bound.measured_mix_ceiling_Gperm_s)
[ -x tools/ubench_poseidon2 ] || hipcc --offload-arch=gfx950 -O3 -I stark-backend_amd/csrc -I include tools/ubench_poseidon2.hip -o tools/ubench_poseidon2 > /dev/null 2>&1
timeout 120 ./tools/ubench_poseidon2 "$O/poseidon2_mix_ceiling.json" > "$O/poseidon2_mix_ceiling.txt" 2>&1
timeout 600 python bench.py > "$O/bench_default_flags.json" 2> "$O/bench_default_flags.err"
cat "$O/bench_default_flags.json" | cut -c1-300
timeout 600 python tools/prove_config_b.py --reps 5 --profile > "$O/config_b_keccak.json" 2> "$O/config_b_keccak.err"
timeout 900 bash tools/profile_config_b.sh "$O/prof_b" > "$O/prof_b.log" 2>&1
timeout 600 python bench.py --num-airs 4 --log-rows 20 --no-cpu-baseline > "$O/bench_config_c.json" 2> "$O/bench_config_c.err"
timeout 600 python bench.py --log-rows 24 --log-blowup 2 --no-cpu-baseline --steps 5 --warmup 1 > "$O/bench_config_e.json" 2> "$O/bench_config_e.err"
if [ "$MODE" != quick ]; then
  { stamp; timeout 1500 python tests/replay_verified.py --every 1 --out "$O/synthetic_replay_full_verified.json"; } > "$O/replay_full.log" 2>&1
  tail -3 "$O/replay_full.log"
  { stamp; timeout 600 python tools/stress_identical.py --proofs 500; } > "$O/stress_identical.log" 2>&1
  tail -2 "$O/stress_identical.log"
fi
timeout 300 python tools/synthetic_replay.py --every 10 --out "$O/synthetic_replay_sample21.json" > "$O/replay_sample.log" 2>&1
{ stamp; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"; } > "$O/smoke.log" 2>&1
tail -1 "$O/smoke.log"
python - "$O" <<'PY'
import json, sys, os
o = sys.argv[1]
def j(f):
    try:
        return json.loads(open(os.path.join(o, f)).read().strip().splitlines()[-1])
    except Exception as e:
        return None
b = j("bench_default_flags.json")
if b: print("config D", b["ms_per_step"], "ms", b["value"], b["unit"], "digest ok:", b["config"].get("proof_matches_checked_digest"))
cb = j("config_b_keccak.json")
if cb: print("config B", min(cb["prove_ms"]), {k: v for k, v in cb["stage_ms"].items() if v > 5})
for n in ("c", "e"):
    x = j(f"bench_config_{n}.json")
    if x: print("config", n.upper(), x["ms_per_step"], "ms")
for f in ("synthetic_replay_full_verified.json", "synthetic_replay_sample21.json"):
    try:
        r = json.load(open(os.path.join(o, f))); print(f, r["total_prove_ms"], "ms", r.get("verified"))
    except Exception:
        pass
PY
