# A/B of library builds on the 64-stream step: per-kernel averages of the kernels matching PATTERN for every lib given
# usage (GPU box): bash profiles/ab_kernel.sh PATTERN STREAMS lib1.so lib2.so ...
PAT=$1; NS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in "$@"; do
  case $L in *=*) export "$L"; echo "  [env $L]"; continue;; esac
  D=$R/gpurun_out/ab_tmp; rm -rf $D
  timeout ${AB_TIMEOUT:-420} rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/profiles/lib_variant_step.py $R/$L $NS 12 2>/dev/null | grep "ms per step"
  S=$(find $D -name "*kernel_stats.csv" | head -1)
  python3 - "$S" "$PAT" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f"    {r['Name'][:70]:70s} calls {r['Calls']:>6s}  avg {float(r['AverageNs']) / 1e3:8.2f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")
PY
  rm -rf $D
done
