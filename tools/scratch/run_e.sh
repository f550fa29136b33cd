out=gpurun_out/r04_e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > $out/counters.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $out/counters.txt | sort -u > $out/sq_counters.txt; wc -l $out/sq_counters.txt
CMD="python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 3 --warmup 0 --lanes 1"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $out/pmc_sq -o sq --output-format csv -- $CMD > $out/pmc_sq.log 2>&1
python tools/pmc_sq.py $out/pmc_sq $out/pmc_sq.txt | head -14
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM -d $out/pmc_sq2 -o sq --output-format csv -- $CMD > $out/pmc_sq2.log 2>&1
python - <<'PY'
import glob, pandas as pd
fs = glob.glob("gpurun_out/r04_e/pmc_sq2/**/*counter_collection.csv", recursive=True)
if fs:
    df = pd.concat([pd.read_csv(f) for f in fs])
    df["k"] = df.Kernel_Name.str.replace(r"\(.*", "", regex=True).str.slice(0, 44)
    p = df.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="sum").fillna(0)
    p = p.sort_values("SQ_WAVE_CYCLES", ascending=False).head(10)
    pd.set_option("display.width", 250); pd.set_option("display.max_columns", 20)
    print(p.to_string())
    open("gpurun_out/r04_e/pmc_sq2.txt", "w").write(p.to_string())
else:
    print(open("gpurun_out/r04_e/pmc_sq2.log").read()[-1500:])
PY
rm -rf $out/pmc_sq $out/pmc_sq2
