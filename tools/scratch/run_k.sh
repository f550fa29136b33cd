out=gpurun_out/r04_k; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|mclk" | head -4
timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks -o ks --output-format csv -- python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 10 --warmup 3 --lanes 1 > $out/ks.log 2>&1; cp $out/ks/*kernel_stats.csv $out/kernel_stats.csv; rm -rf $out/ks; head -6 $out/kernel_stats.csv | cut -c1-110; grep -o '"ms_per_step": [0-9.]*' $out/ks.log
timeout 300 python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 30 --warmup 5 | grep -o '"ms_per_step": [0-9.]*'
rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk" | head -2
