#!/bin/bash
# Files the outputs of tools/regress_round.sh (a gpurun_out/ directory) under profiles/<round>/ with the names DESIGN.md,
# README.md and bench.py cite:  tools/file_regression.sh gpurun_out/r05/regress2 profiles/r05 r05
set -eu
R=$1; P=$2; TAG=$3
mkdir -p "$P" "$P/prof_config_b_$TAG"
cp "$R/gputest_all.log" "$P/gputest_all_${TAG}final.log"
[ -f "$R/gputest_full_size_b_e.log" ] && cp "$R/gputest_full_size_b_e.log" "$P/gputest_full_size_b_e_final.log"
[ -f "$R/box_host.txt" ] && cp "$R/box_host.txt" "$P/box_host_${TAG}final.txt"
cp "$R/bench_default_flags.json" "$P/bench_${TAG}_default_flags.json"
cp "$R/config_b_keccak.json" "$P/config_b_keccak_$TAG.json"
cp "$R/bench_config_c.json" "$P/bench_config_c_$TAG.json"
cp "$R/bench_config_e.json" "$P/bench_config_e_$TAG.json"
[ -f "$R/synthetic_replay_full_verified.json" ] && cp "$R/synthetic_replay_full_verified.json" "$P/synthetic_replay_full_${TAG}_verified.json"
cp "$R/synthetic_replay_sample21.json" "$P/synthetic_replay_sample21_$TAG.json"
[ -f "$R/stress_identical.log" ] && cp "$R/stress_identical.log" "$P/stress_identical_${TAG}final.log"
cp "$R/smoke.log" "$P/smoke_${TAG}final.log"
cp "$R/prof/prove_kernel_stats.txt" "$P/prove_${TAG}_kernel_stats.txt"
cp "$R/prof/prove_kernel_stats.csv" "$P/prove_${TAG}_kernel_stats.csv"
cp "$R/prof/idle_gaps.txt" "$P/idle_gaps_$TAG.txt"
cp "$R/prof/pmc_hbm_traffic.json" "$P/pmc_hbm_traffic_$TAG.json"
cp "$R/prof/sq_counters.txt" "$P/sq_counters_$TAG.txt"
cp "$R/prof/sq_valu.json" "$P/sq_valu_$TAG.json"
[ -f "$R/poseidon2_mix_ceiling.json" ] && cp "$R/poseidon2_mix_ceiling.json" "$R/poseidon2_mix_ceiling.txt" "$P/"
[ -f gpurun_out/cpu_baseline_full_size.json ] && cp gpurun_out/cpu_baseline_full_size.json "$P/cpu_baseline_full_size.json"
cp "$R/prof_b/kernel_stats.txt" "$R/prof_b/kernel_stats.csv" "$R/prof_b/sq_counters.txt" "$R/prof_b/pmc_hbm_traffic.json" "$P/prof_config_b_$TAG/"
ls "$P" | wc -l
