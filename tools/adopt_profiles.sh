#!/bin/bash
# Copies the evidence set gpurun_out/<tag> (made by tools/collect_profiles.sh <tag> on the GPU box) into profiles/ under the names
# the docs use: bash tools/adopt_profiles.sh r02_v3   ->  profiles/r02_bench_v3*.json, r02_c2_kernel_stats_v3.csv, r02_v3_pmc_*...
set -e
tag=$1; rnd=${tag%%_*}; ver=${tag##*_}; o=gpurun_out/$tag
cp $o/bench_C2.json profiles/${rnd}_bench_${ver}.json
for w in C1 C3 C5; do cp $o/bench_$w.json profiles/${rnd}_bench_${ver}_$w.json; done
cp $o/bench_C2_coalesce2.json profiles/${rnd}_bench_${ver}_coalesce2.json
cp $o/kernel_stats.csv profiles/${rnd}_c2_kernel_stats_${ver}.csv
[ -f $o/kernel_stats.hash ] && cp $o/kernel_stats.hash profiles/${rnd}_c2_kernel_stats_${ver}.hash
[ -f $o/ubench_mfma_stage.txt ] && cp $o/ubench_mfma_stage.txt profiles/${rnd}_${ver}_ubench_mfma_stage.txt
cp $o/pmc_hbm_traffic.json profiles/${rnd}_${ver}_pmc_hbm_traffic.json
cp $o/pmc_hbm_traffic.txt profiles/${rnd}_${ver}_pmc_hbm_traffic.txt
cp $o/pmc_sq.txt profiles/${rnd}_${ver}_pmc_sq.txt
cp $o/train_step.json profiles/${rnd}_train_step_c4shard_${ver}.json
cp $o/griffin_lim.json profiles/${rnd}_griffin_lim_c2shape_${ver}.json
for f in train_step_atomics train_step_splitbf16 train_step_exact train_step_bptt_per_stage; do [ -f $o/$f.json ] && cp $o/$f.json profiles/${rnd}_${f}_${ver}.json; done
for f in scan_timeline time_manual overlap_scan_ff time_stages bptt_timeline time_front front_timeline decoder_timeline time_layers_native time_stages_native time_train_native chain_timeline; do [ -f $o/$f.txt ] && cp $o/$f.txt profiles/${rnd}_${ver}_$f.txt; done
for f in scan_timeline decoder_timeline; do [ -f $o/$f.json ] && cp $o/$f.json profiles/${rnd}_${ver}_$f.json; done      # bench.py's latency_floor_ms reads these
[ -f $o/bench_C4.json ] && cp $o/bench_C4.json profiles/${rnd}_bench_${ver}_C4.json
[ -f $o/train_kernel_stats.csv ] && cp $o/train_kernel_stats.csv profiles/${rnd}_train_kernel_stats_${ver}.csv
ls profiles | grep "${rnd}_.*${ver}"
