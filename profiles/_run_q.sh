cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --steps 12 --warmup 4 > $O/prof1.log 2>&1
S=$(find $O/prof1 -name "*kernel_stats.csv" | head -1)
python3 - $S <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "enc_attention" in r["Name"] or "layernorm" in r["Name"]:
        print(r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
rm -rf $O/prof1
