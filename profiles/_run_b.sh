cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline > $O/prof1.log 2>&1
T=$(find $O/prof1 -name "*kernel_trace.csv" | head -1)
python3 profiles/trace_gaps.py $T > $O/trace_busy_prof1.txt
python3 - $T > $O/chunk_timeline.txt <<'PY'
import csv, sys, re
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")))
rows.sort()
# one steady-state chunk: from an audio_window_kernel to the next
idx = [i for i, r in enumerate(rows) if r[2].startswith("audio_window_kernel")]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = rows[a][0]
print(f"chunk of {b - a} kernels, {(rows[b][0] - t0) / 1e3:.1f} us")
prev_end = t0
for s, e, n in rows[a:b]:
    gap = (s - prev_end) / 1e3
    if gap > 3.0 or n.startswith(("audio", "embed", "llm_rope_cache", "sample_process")):
        print(f"{(s - t0) / 1e3:10.1f} us  gap {gap:7.1f}  {n[:70]}")
    prev_end = e
PY
rm -rf $O/prof1
head -16 $O/trace_busy_prof1.txt; head -80 $O/chunk_timeline.txt
