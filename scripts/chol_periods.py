"""Development helper: per super-panel of the look-ahead chain, from a timeline of scripts/chol_timeline.py -- the near
update, the far update, the next super-panel's steps and block solve, the period, and which of them set it."""
import sys
rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
ev = [(float(r[0]), float(r[1]), r[2] + r[3], r[-1]) for r in rows]
near = [e for e in ev if "update_wide4" in e[3] and e[2] == "q1"]
far = [e for e in ev if "update_wide4" in e[3] and e[2] != "q1"]
tot = 0.0
for k, n in enumerate(near):
    f = far[k] if k < len(far) else None
    nxt = near[k + 1][0] if k + 1 < len(near) else None
    inside = [e for e in ev if e[0] >= n[1] - 1 and (nxt is None or e[0] < nxt) and e[2] == "q1"]
    steps = [e for e in inside if "panel" in e[3]]
    tb = [e for e in inside if "trsm_block" in e[3]]
    end = max([e[1] for e in steps + tb], default=0.0)
    print(f"J{k:2d} near {n[0]:8.0f} +{n[1] - n[0]:5.0f}  far +{(f[1] - f[0]) if f else 0:5.0f} ends {f[1] if f else 0:8.0f} | steps {len(steps):2d} "
          f"sum {sum(e[1] - e[0] for e in steps):5.0f} trsm_block {sum(e[1] - e[0] for e in tb):5.0f} done {end:8.0f} | period "
          f"{(nxt - n[0]) if nxt else 0:6.0f}")
print("first launch", ev[0][0], "last end", ev[-1][1])
