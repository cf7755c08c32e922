"""Turns gpurun_out/prof_<tag>/ (scripts/profile.sh) into the tracked summary profiles/<out>_*.
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads, so it is doubled (MI355X_MICROARCH.md, section HBM)."""
import collections, csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = sys.argv[2] if len(sys.argv) > 2 else tag
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = list(csv.DictReader(open(max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime))))
pmc = {}
for which, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(src, which, "*", "*_counter_collection.csv"))
    if not files:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        pmc.setdefault(k, {})[counter] = sum(v) / len(v)
bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])

cmd = ("scripts/profile.sh", "bench.py --steps 20 --warmup 3 --no-cpu-baseline")
if "--cfg5-full" in sys.argv:
    cmd = ("scripts/profile_cfg5_full.sh", "bench.py --workload cfg5 --cfg5-points 2000000 --steps 3 --warmup 0 --segment 3 "
           "--spread-segments 0 --no-cpu-baseline")
lines = [f"# rocprofv3 summary {out}", "",
         f"Command ({cmd[0]}, on the MI355X box): `rocprofv3 --kernel-trace --stats -f csv -- python3 "
         f"{cmd[1]}`; FETCH_SIZE and WRITE_SIZE from two further "
         "`--pmc` passes of the same command.", "",
         f"Workload: {bench['config']['workload']} ({bench['config']['n_cams']} cameras, {bench['config']['n_pts']} points, "
         f"{bench['config']['n_obs']} observations), {bench['steps']} LM iterations, {bench['damping_tries']} damping tries.",
         "", "| kernel | calls | avg us | total % | FETCH_SIZE KiB (raw) | HBM read MB (x2 corrected) | WRITE_SIZE KiB = HBM write |",
         "|---|---|---|---|---|---|---|"]
for r in stats:
    name = r["Name"]
    short = name.split("(")[0].replace("void ", "")
    p = pmc.get(name, {})
    f, w = p.get("FETCH_SIZE"), p.get("WRITE_SIZE")
    lines.append(f"| `{short}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} | "
                 f"{'' if f is None else f'{f:.0f}'} | {'' if f is None else f'{2 * f * 1024 / 1e6:.1f}'} | "
                 f"{'' if w is None else f'{w:.0f}'} |")
lines += ["", "bench.py line of the traced run (HIP-event timings inside bench.py):", "", "```json", json.dumps(bench, indent=1), "```"]
open(os.path.join(dst, f"{out}_kernel_stats.md"), "w").write("\n".join(lines) + "\n")
import subprocess
try:
    head = subprocess.check_output(["git", "-C", root, "rev-parse", "HEAD"], text=True).strip()
    dirty = bool(subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "psba_amd", "bench.py"], text=True).strip())
    head += "+dirty" if dirty else ""
except Exception:
    head = None
json.dump({"git_head": head, "kernel_stats": stats, "pmc_avg_per_launch_KiB": pmc, "bench": bench},
          open(os.path.join(dst, f"{out}_profile.json"), "w"), indent=1)
print("\n".join(lines[:22]))
