set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
for v in intree walkrec; do
  if [ "$v" = intree ]; then unset GCP_LIBRARY; else export GCP_LIBRARY=$GRAFT_REPO_ROOT/variants/$v.so; fi
  rm -rf $O/wr_$v
  rocprofv3 --kernel-trace --stats -d $O/wr_$v -o st --output-format csv -- python3 tools/walk_bench.py cfg3 --iters 10 > $O/wr_$v.log 2>&1
  tail -1 $O/wr_$v.log
  python3 - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
for f in glob.glob(f"gpurun_out/r4/wr_{v}/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "walk" in r["Name"] or "k_pairs" in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:12]:
        print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:8.1f}')
PY
done
