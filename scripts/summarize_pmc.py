"""Turns the counters-only rocprofv3 passes under gpurun_out/prof_<tag>/pmc_*/ (scripts/
profile_pmc_extra.sh, scripts/profile_cfg5.sh) into profiles/<out>_pmc_extra.{md,json}: per kernel,
the average of every counter over the launches of the run (rocprofv3 reports the sum over the
chip's shader engines).  With --cfg5 the kernel-trace statistics of the same directory are added
and the MFMA utilisation of the Cholesky kernels is worked out:
  MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz)
  TFLOP/s            = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flop / kernel duration
(one v_mfma_f64_16x16x4_f64 = 2048 flop = 4 MOPS and holds its SIMD's matrix pipe for 64 cycles)."""
import collections, csv, glob, json, os, sys

tag = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else tag
cfg5 = "--cfg5" in sys.argv
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    if "pmc_fetch" in f or "pmc_write" in f:
        continue
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
counters = sorted({c for d in avg.values() for c in d})
stats = {}
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        stats[r["Name"].split("(")[0].replace("void ", "")] = r
lines = [f"# rocprofv3 PMC passes {out}", "",
         "Counters-only passes (`rocprofv3 --pmc ... -- python3 bench.py ...`, program directly after `--`); "
         "values are per launch, averaged over the launches of the run, summed over the chip as rocprofv3 reports them.", ""]
if cfg5:
    bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
    c = bench["config"]
    lines += [f"Workload: {c['workload']} ({c['n_cams']} cameras, {c['n_pts']} points, {c['n_obs']} observations), "
              f"dense {6 * c['n_cams']} x {6 * c['n_cams']} S; {bench['ms_per_step']:.1f} ms per LM iteration in the traced run.", "",
              "| kernel | calls | avg us | total % | MFMA MOPS_F64 / launch | MFMA busy cycles / launch | MFMA busy of 1024 SIMDs | fp64 MFMA TFLOP/s (of 78.6) |",
              "|---|---|---|---|---|---|---|---|"]
    tot_flop = tot_us = 0.0
    for name, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["Percentage"])):
        a = avg.get(name, {})
        us = float(r["AverageNs"]) / 1e3
        mops, busy = a.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0), a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        frac = busy / (1024 * us * 1e-6 * 2.4e9) if us else 0
        tf = mops * 512 / (us * 1e-6) / 1e12 if us else 0
        if mops and "k_cholg" in name:  # (k_schur_reduce's extra workgroup factors the first block: a few MFMAs, not a chain kernel)
            tot_flop += mops * 512 * int(r["Calls"])
            tot_us += us * int(r["Calls"])
        lines.append(f"| `{name}` | {r['Calls']} | {us:.1f} | {float(r['Percentage']):.2f} | {mops:.3g} | {busy:.3g} | "
                     f"{100 * frac:.1f} % | {tf:.1f} |" if mops else
                     f"| `{name}` | {r['Calls']} | {us:.1f} | {float(r['Percentage']):.2f} | | | | |")
    if tot_us:
        lines += ["", f"Sum of the chain's kernel durations: {tot_us / 1e3:.0f} ms over the run, {tot_flop / (tot_us * 1e-6) / 1e12:.1f} TFLOP/s "
                  f"= {100 * tot_flop / (tot_us * 1e-6) / 78.6e12:.1f} % of the 78.6 TFLOP/s fp64 matrix peak against that sum.  With the "
                  "look-ahead on (the default at this size) the far part of an update runs on the side stream beside the next panel "
                  "and the near part of the next update, so the durations overlap and their sum is longer than the wall time: the "
                  "figure against the sum is a lower bound, not the chain's rate."]
        wall = bench.get("kernels_us", {}).get("cholesky")
        nfact = int(stats.get("psba::k_schur_lds<false>", {"Calls": 0})["Calls"])  # one S, one factorization per damping try
        if wall and nfact:
            per = tot_flop / nfact
            lines += ["", f"Against the wall: {per / 1e9:.1f} GFLOP of MFMA work per factorization (counted by SQ_INSTS_VALU_MFMA_MOPS_F64, "
                      f"{nfact} factorizations in the traced run, timed passes and per-kernel passes together) in {wall / 1e3:.2f} ms between HIP events under the tracer "
                      f"(`kernels_us.cholesky` of the traced bench line) = {per / (wall * 1e-6) / 1e12:.1f} TFLOP/s "
                      f"= {100 * per / (wall * 1e-6) / 78.6e12:.1f} % of peak; the algorithmic n^3/3 = "
                      f"{(6 * c['n_cams']) ** 3 / 3 / 1e9:.1f} GFLOP gives {(6 * c['n_cams']) ** 3 / 3 / (wall * 1e-6) / 1e12:.1f} TFLOP/s "
                      f"= {100 * (6 * c['n_cams']) ** 3 / 3 / (wall * 1e-6) / 78.6e12:.1f} %."]
else:
    lines += ["| kernel | " + " | ".join(c.replace("SQ_", "") for c in counters) + " |", "|---|" + "---|" * len(counters)]
    for k, d in sorted(avg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
        if k.startswith("__amd"):
            continue
        lines.append(f"| `{k}` | " + " | ".join(f"{d[c]:.3g}" if c in d else "" for c in counters) + " |")
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
open(os.path.join(root, "profiles", f"{out}_pmc_extra.md" if not cfg5 else f"{out}_mfma.md"), "w").write("\n".join(lines) + "\n")
json.dump({"pmc_avg_per_launch": avg, "kernel_stats": stats},
          open(os.path.join(root, "profiles", f"{out}_pmc_extra.json" if not cfg5 else f"{out}_mfma.json"), "w"), indent=1)
print("\n".join(lines[:40]))
