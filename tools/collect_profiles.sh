#!/bin/bash
# Collects the per-round evidence under gpurun_out/ (copy what is kept into profiles/):  bash tools/collect_profiles.sh <tag>
# Counter passes are separate rocprofv3 runs with --pmc only (no trace domains), as MI355X_MICROARCH.md prescribes.
tag=${1:-vXX}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$2" = "core" ]; then
  # the hash-tied part only (after a change to the kernel sources that moves no number, e.g. comments):  bash tools/collect_profiles.sh rNN_vM core
  # order: the counter passes FIRST, so that the bench line finds its traffic profile only if this script is followed by adopt + a second bench run;
  # here the line is printed last and patched by tools/adopt_profiles.sh's caller as before
  B="python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 3 --warmup 0 --lanes 1"
  timeout 400 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_f -o f --output-format csv -- $B > $out/pmc_f.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_w -o w --output-format csv -- $B > $out/pmc_w.log 2>&1
  python tools/pmc_traffic.py $out/pmc_f $out/pmc_w $out/pmc_hbm_traffic > $out/pmc_traffic.log 2>&1
  cp $out/pmc_hbm_traffic.json profiles/${tag%%_*}_${tag##*_}_pmc_hbm_traffic.json      # (on the box only: lets the bench line below find its profile by the source hash)
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/ks -o ks --output-format csv -- python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 5 --warmup 2 --lanes 1 > $out/ks.log 2>&1
  cp $out/ks/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
  python -c "import bench; print(bench.source_hash())" > $out/kernel_stats.hash
  timeout 600 python bench.py --steps 20 --warmup 4 > $out/bench_C2.json 2> $out/bench_C2.err
  rm -rf $out/ks/*kernel_trace.csv $out/pmc_f $out/pmc_w
  tail -c 400 $out/bench_C2.json
  exit 0
fi
[ -x tools/time_stages_native ] && timeout 60 ./tools/time_stages_native > $out/time_stages_native.txt 2>&1     # C ABI only, no Python: seconds
timeout 600 python bench.py --steps 20 --warmup 4 > $out/bench_C2.json 2> $out/bench_C2.err
timeout 300 python bench.py --workload C1 --steps 12 --warmup 3 > $out/bench_C1.json 2>> $out/bench_C2.err      # (with its CPU baseline: one utterance is cheap)
for w in C3 C5; do timeout 300 python bench.py --no-cpu-baseline --workload $w --steps 12 --warmup 3 > $out/bench_$w.json 2>> $out/bench_C2.err; done
timeout 300 python bench.py --no-cpu-baseline --coalesce 2 --steps 24 --warmup 4 > $out/bench_C2_coalesce2.json 2>> $out/bench_C2.err
timeout 300 python tools/bench_train.py > $out/train_step.json 2> $out/train.err
timeout 300 python tools/bench_train.py --deterministic 0 > $out/train_step_atomics.json 2>> $out/train.err      # fp32 atomics instead of the ordered sums (the default since round 4)
timeout 300 python tools/bench_train.py --exact-gemm 0 > $out/train_step_splitbf16.json 2>> $out/train.err
timeout 300 python tools/bench_train.py --exact-gemm 1 > $out/train_step_exact.json 2>> $out/train.err
timeout 300 python tools/bench_train.py --bptt 0 > $out/train_step_bptt_per_stage.json 2>> $out/train.err
timeout 300 python tools/trace_bptt.py 2>&1 | grep -v amdgpu.ids > $out/bptt_timeline.txt
timeout 300 python tools/bench_audio.py > $out/griffin_lim.json 2> $out/audio.err
# round 3: scan timelines (k_bigru_duo vs k_bigru_xcd), manual / simple decoder modes, feed-forward-under-the-scan experiment, training kernel statistics
rm -f $out/scan_timeline.json
# round 5: k_bigru_oct (persist 1 from 9 to 32 rows; 10: wherever it fits) vs k_bigru_duo (11) vs k_bigru_xcd (8); --json feeds bench.py's latency_floor_ms
{ for sh in "32 512 1" "16 512 1" "64 512 1" "8 4000 1" "32 512 11" "16 512 11" "8 4000 10" "32 512 8"; do python tools/trace_bigru.py $sh --json $out/scan_timeline.json; done; } 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|warnings.warn" > $out/scan_timeline.txt
timeout 300 python tools/time_manual.py 2>&1 | grep -v amdgpu.ids > $out/time_manual.txt
timeout 300 python tools/overlap_scan_ff.py 2>&1 | grep -v amdgpu.ids > $out/overlap_scan_ff.txt
timeout 300 python tools/time_stages.py C2 32 64 2>&1 | grep -v amdgpu.ids > $out/time_stages.txt
# round 4: the fused CBHG front (k_cbhg_front): feed-forward time of both stages with the front on / off and per start delay / priority, phase
# timeline of one workgroup (needs the -DTACO_TRACE build next to the library), per-layer timings, the C4 line of bench.py
timeout 300 python tools/time_front.py 2>&1 | grep -v amdgpu.ids > $out/time_front.txt
timeout 300 python tools/time_decoder.py C2:8 --json $out/decoder_timeline.json 2>&1 | grep -v amdgpu.ids > $out/decoder_timeline.txt       # per-phase clocks of one decoder step; 64-row pass at eight rows per group
[ -f multi-speaker-tacotron-tensorflow_amd/csrc/libtaco_hip_trace.so ] && TACO_LIB=$GRAFT_REPO_ROOT/multi-speaker-tacotron-tensorflow_amd/csrc/libtaco_hip_trace.so timeout 200 python tools/trace_front.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|load_library()" > $out/front_timeline.txt
[ -x tools/time_layers_native ] && timeout 120 ./tools/time_layers_native 20 > $out/time_layers_native.txt 2>&1
[ -x tools/time_train_native ] && timeout 120 ./tools/time_train_native > $out/time_train_native.txt 2>&1      # one C4-shard training step through the C ABI, no Python
[ -f multi-speaker-tacotron-tensorflow_amd/csrc/libtaco_hip_trace.so ] && TACO_LIB=$GRAFT_REPO_ROOT/multi-speaker-tacotron-tensorflow_amd/csrc/libtaco_hip_trace.so timeout 200 python tools/trace_chain.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|load_library()" > $out/chain_timeline.txt
timeout 600 python bench.py --workload C4 --steps 10 --warmup 2 > $out/bench_C4.json 2>> $out/bench_C2.err
timeout 400 rocprofv3 --kernel-trace --stats -d $out/tks -o tks --output-format csv -- python tools/bench_train.py --steps 4 --warmup 1 > $out/tks.log 2>&1
cp $out/tks/*kernel_stats.csv $out/train_kernel_stats.csv 2>/dev/null; rm -rf $out/tks
timeout 400 rocprofv3 --kernel-trace --stats -d $out/ks -o ks --output-format csv -- python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 5 --warmup 2 --lanes 1 > $out/ks.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_f -o f --output-format csv -- python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 3 --warmup 0 --lanes 1 > $out/pmc_f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_w -o w --output-format csv -- python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 3 --warmup 0 --lanes 1 > $out/pmc_w.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $out/pmc_sq -o sq --output-format csv -- python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 3 --warmup 0 --lanes 1 > $out/pmc_sq.log 2>&1
python tools/pmc_traffic.py $out/pmc_f $out/pmc_w $out/pmc_hbm_traffic > $out/pmc_traffic.log 2>&1
python tools/pmc_sq.py $out/pmc_sq $out/pmc_sq.txt > /dev/null 2>&1
cp $out/ks/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
python -c "import bench; print(bench.source_hash())" > $out/kernel_stats.hash      # the build the statistics were taken from (bench.py floor_constants)
[ -x tools/ubench_mfma_stage ] && timeout 120 ./tools/ubench_mfma_stage > $out/ubench_mfma_stage.txt 2>&1      # the decoder's exchange in isolation (latency_floor_ms.hardware_terms)
rm -rf $out/ks/*kernel_trace.csv $out/pmc_f $out/pmc_w $out/pmc_sq     # raw traces are large; the reductions above are what is kept
ls -la $out
tail -c 600 $out/bench_C2.json
