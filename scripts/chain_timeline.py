"""Development helper: prints the kernel timeline of the last Cholesky chain found in a
rocprofv3 kernel trace (gpurun_out/prof_<tag>/trace/*/*_kernel_trace.csv)."""
import csv, glob, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "w"
f = sorted(glob.glob(f"gpurun_out/prof_{tag}/trace/*/*_kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_cholg_diag" in r["Kernel_Name"]]
i0 = idx[-5]
t0 = int(rows[i0]["Start_Timestamp"])
prev = None
for r in rows[i0 - 3:i0 + 18]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0][-30:]
    gap = (s - prev) if prev is not None else 0
    print(f"{name:32s} start {s / 1e3:8.2f} dur {(e - s) / 1e3:6.2f} gap {gap / 1e3:6.2f} grid {r['Grid_Size_X']}")
    prev = e
