#!/bin/bash
out=gpurun_out/r05_sq; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $out/pmc_sq -o sq --output-format csv -- python tools/bench_train.py --steps 2 --warmup 1 > $out/pmc_sq.log 2>&1
python tools/pmc_sq.py $out/pmc_sq $out/train_pmc_sq.txt "python tools/bench_train.py --steps 2 --warmup 1 (C4 shard; 5 forwards, 4 backward passes)" 30 | tail -34
rm -rf $out/pmc_sq
