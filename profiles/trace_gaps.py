"""Kernel-trace reduction: per-kernel busy time and the idle gaps between consecutive kernels over a window of the run.
    python profiles/trace_gaps.py <kernel_trace.csv> [--window 0.6:0.8]"""
import csv, re, sys, collections
path = sys.argv[1]
lo, hi = (float(x) for x in (sys.argv[sys.argv.index("--window") + 1].split(":") if "--window" in sys.argv else ("0.6", "0.8")))
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[int(len(rows) * lo):int(len(rows) * hi)]  # by kernel count: a stretch of steady-state steps
span = rows[-1][1] - rows[0][0]
busy = collections.Counter(); calls = collections.Counter(); gap_after = collections.Counter()
gaps = []
for i, (s, e, n) in enumerate(rows):
    short = re.sub(r"\(.*", "", n).replace("void ", "")
    busy[short] += e - s; calls[short] += 1
    if i + 1 < len(rows):
        g = rows[i + 1][0] - e
        gaps.append(g); gap_after[short] += max(g, 0)
tot_busy = sum(busy.values())
def _short(n): return re.sub(r"\(.*", "", n).replace("void ", "")[:44]
top = sorted(range(len(gaps)), key=lambda i: -gaps[i])[:12]
print("largest gaps, with the kernels either side:")
for i in top:
    print(f"  {gaps[i]/1e3:9.1f} us   after {_short(rows[i][2]):44s} before {_short(rows[i + 1][2])}")
big = sorted(gaps)[-20:]
print(f"largest gaps (us): {[round(g/1e3) for g in big]}")
print(f"window {span/1e6:.2f} ms, {len(rows)} kernels, busy {tot_busy/1e6:.2f} ms ({100*tot_busy/span:.1f} %), gaps {sum(max(g,0) for g in gaps)/1e6:.2f} ms, "
      f"median gap {sorted(gaps)[len(gaps)//2]/1e3:.2f} us")
print(f"{'kernel':60s} {'calls':>7s} {'busy ms':>9s} {'%win':>6s} {'avg us':>8s} {'gap-after avg us':>17s}")
for k, v in busy.most_common(22):
    print(f"{k[:60]:60s} {calls[k]:7d} {v/1e6:9.3f} {100*v/span:6.1f} {v/calls[k]/1e3:8.2f} {gap_after[k]/calls[k]/1e3:17.2f}")
