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
    # the forward's own kernels only: torch's elementwise / reduce kernels (bench.py's isfinite checks) and the runtime's copy / fill kernels
    # run in the same process but are not part of a forward
    own = lambda k: not (k.startswith("void at::native") or k.startswith("__amd_rocclr"))
    fetch_kb, write_kb = float(sum(v for k, v in fk.items() if own(k))), float(sum(v for k, v in wk.items() if own(k)))
    other_kb = float(sum(2 * v for k, v in fk.items() if not own(k)) + sum(v for k, v in wk.items() if not own(k)))
    rec = {"workload": "C2", "kernel_source_hash": source_hash(), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), %d forwards averaged" % nfwd_f,
           "fetch_size_reported_bytes": fetch_kb * 1024, "fetch_size_corrected_bytes": 2 * fetch_kb * 1024, "write_size_bytes": write_kb * 1024,
           "traffic_bytes_per_forward": 2 * fetch_kb * 1024 + write_kb * 1024,
           "excluded_not_part_of_a_forward_bytes": other_kb * 1024,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide coalesced stream); counters are L2<->fabric "
                   "requests, Infinity-Cache hits included; WRITE_SIZE uncalibrated"}
    json.dump(rec, open(outp + ".json", "w"), indent=1)
    with open(outp + ".txt", "w") as fh:
        fh.write("rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2), bench.py C2 --lanes 1, per forward (%d forwards averaged), KB as reported\n" % nfwd_f)
        fh.write("total of the forward's own kernels (torch's and the runtime's kernels of the same process excluded: %.1f MB): FETCH_SIZE %.1f MiB reported -> %.1f MiB corrected (x2, gfx950); WRITE_SIZE %.1f MiB; traffic = %.3f GB (10^9 bytes) per forward\n"
                 % (other_kb * 1024 / 1e6, fetch_kb / 1024, 2 * fetch_kb / 1024, write_kb / 1024, (2 * fetch_kb + write_kb) * 1024 / 1e9))
        for k in fk.sort_values(ascending=False).index:
            fh.write("%-48s calls/fwd %6.0f  fetch %10.1f KB/fwd  write %10.1f KB/fwd\n" % (k, calls.get(k, 0), fk[k], wk.get(k, 0.0)))
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
