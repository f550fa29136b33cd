#!/usr/bin/env python
"""HBM-side traffic per forward from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md prescribes:
separate passes, counters in KB of 64-byte... (rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB); gfx950 correction: FETCH_SIZE
doubled.  Usage (on the GPU box, counters only -- no sys/hip/hsa trace domains):
  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f --output-format csv -- python bench.py --no-cpu-baseline --steps 3 --warmup 0 --lanes 1
  rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w --output-format csv -- python bench.py --no-cpu-baseline --steps 3 --warmup 0 --lanes 1
  python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/rNN_pmc_hbm_traffic"""
import glob, json, os, sys
import pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash


def load(d, counter):
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    df = pd.concat([pd.read_csv(f) for f in fs])
    df = df[df.Counter_Name == counter]
    return df


def main():
    fdir, wdir, outp = sys.argv[1:4]
    f, w = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    nfwd_f = int((f.Kernel_Name.str.startswith("k_stop_step")).sum())
    nfwd_w = int((w.Kernel_Name.str.startswith("k_stop_step")).sum())
    fk = f.groupby(f.Kernel_Name.str.slice(0, 46)).Counter_Value.sum() / nfwd_f
    wk = w.groupby(w.Kernel_Name.str.slice(0, 46)).Counter_Value.sum() / nfwd_w
    calls = f.groupby(f.Kernel_Name.str.slice(0, 46)).size() / nfwd_f
    fetch_kb, write_kb = float(fk.sum()), float(wk.sum())
    rec = {"workload": "C2", "kernel_source_hash": source_hash(), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), %d forwards averaged" % nfwd_f,
           "fetch_size_reported_bytes": fetch_kb * 1024, "fetch_size_corrected_bytes": 2 * fetch_kb * 1024, "write_size_bytes": write_kb * 1024,
           "traffic_bytes_per_forward": 2 * fetch_kb * 1024 + write_kb * 1024,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide coalesced stream); counters are L2<->fabric "
                   "requests, Infinity-Cache hits included; WRITE_SIZE uncalibrated"}
    json.dump(rec, open(outp + ".json", "w"), indent=1)
    with open(outp + ".txt", "w") as fh:
        fh.write("rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2), bench.py C2 --lanes 1, per forward (%d forwards averaged), KB as reported\n" % nfwd_f)
        fh.write("total: FETCH_SIZE %.1f MB reported -> %.1f MB corrected (x2, gfx950); WRITE_SIZE %.1f MB\n" % (fetch_kb / 1024, 2 * fetch_kb / 1024, write_kb / 1024))
        for k in fk.sort_values(ascending=False).index:
            fh.write("%-48s calls/fwd %6.0f  fetch %10.1f KB/fwd  write %10.1f KB/fwd\n" % (k, calls.get(k, 0), fk[k], wk.get(k, 0.0)))
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
