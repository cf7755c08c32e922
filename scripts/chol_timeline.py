"""Development helper: from a rocprofv3 kernel trace (csv) print the timeline of one dense factorization of the
blocked chain -- start / end (us, relative) and queue of every k_cholg_* launch between two k_schur_reduce."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
red = [i for i, r in enumerate(rows) if "k_schur_reduce" in r["Kernel_Name"]]
a, b = red[-2], red[-1]
t0 = int(rows[a]["End_Timestamp"])
n = 0
for r in rows[a + 1:b]:
    if "k_cholg" not in r["Kernel_Name"]:
        continue
    n += 1
    if n > int(sys.argv[2]) if len(sys.argv) > 2 else 120:
        break
    name = r["Kernel_Name"].split("(")[0].replace("psba::", "").replace("void ", "")
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} {(int(r['End_Timestamp']) - t0) / 1e3:10.1f} q{r.get('Queue_Id', '?'):>3s} grid {r.get('Grid_Size', '?'):>9s} {name}")
