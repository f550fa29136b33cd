import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last encoder pass = kernels between the 3rd-last and ... simply print the last 60 kernels
t0 = None
for r in rows[-75:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - t0) / 1e3 if t0 else 0.0
    t0 = e
    print("%-62s grid %-14s wg %-5s  %8.1f us  gap %6.1f" % (r["Kernel_Name"][:62], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), (e - s) / 1e3, gap))
