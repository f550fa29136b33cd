#!/usr/bin/env python
"""Per-kernel SQ counter summary of one rocprofv3 --pmc pass (wave-cycle split into parked / issue-stalled / active, MFMA busy,
LDS bank conflicts).  Usage:
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU \\
      -d gpurun_out/pmc_sq -o sq --output-format csv -- python bench.py --no-cpu-baseline --steps 3 --warmup 0 --lanes 1
  python tools/pmc_sq.py gpurun_out/pmc_sq profiles/rNN_pmc_sq.txt ["what was profiled" [rows]]"""
import glob, os, sys
import pandas as pd

d, outp = sys.argv[1:3]
label = sys.argv[3] if len(sys.argv) > 3 else "bench.py C2 --lanes 1"      # what was profiled (third argument)
nrows = int(sys.argv[4]) if len(sys.argv) > 4 else 16
df = pd.concat([pd.read_csv(f) for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)])
df["k"] = df.Kernel_Name.str.replace(r"\(.*", "", regex=True).str.slice(0, 52)
p = df.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="sum").fillna(0)
calls = df[df.Counter_Name == df.Counter_Name.iloc[0]].groupby("k").size()
rows = []
for k, r in p.iterrows():
    wc = max(r.get("SQ_WAVE_CYCLES", 0), 1)
    rows.append((wc, k, calls.get(k, 0), 100 * r.get("SQ_WAIT_ANY", 0) / wc, 100 * r.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * r.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                 r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), r.get("SQ_LDS_BANK_CONFLICT", 0), r.get("SQ_INSTS_VALU", 0)))
rows.sort(reverse=True)
with open(outp, "w") as fh:
    fh.write("rocprofv3 --pmc (one pass, SQ block): share of wave-cycles parked (s_waitcnt / barrier) | issue-stalled | issuing; raw MFMA-pipe busy\n"
             "cycles (summed over all dispatches; a 32x32x16 bf16 MFMA = 32, MI355X_MICROARCH.md); kernels ordered by wave-cycles.  %s\n" % label)
    fh.write("%-54s %7s %8s %8s %8s %16s %14s %14s\n" % ("kernel", "calls", "parked%", "stall%", "issue%", "mfma_busy_cyc", "lds_conflicts", "valu_insts"))
    for wc, k, c, a, b, e, m, l, v in rows[:nrows]:
        fh.write("%-54s %7d %8.1f %8.1f %8.1f %16.0f %14.0f %14.0f\n" % (k, c, a, b, e, m, l, v))
print(open(outp).read())
