set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
for v in "" walk32 walk64; do
  if [ -z "$v" ]; then unset GCP_LIBRARY; else export GCP_LIBRARY=$GRAFT_REPO_ROOT/variants/$v.so; fi
  timeout -k 10 200 python3 tools/walk_bench.py cfg3 --iters 10 >> $O/walk_ab1.jsonl 2>> $O/walk_ab1.err
  timeout -k 10 200 python3 tools/walk_bench.py cfg2 --iters 10 >> $O/walk_ab1.jsonl 2>> $O/walk_ab1.err
done
unset GCP_LIBRARY
cat $O/walk_ab1.jsonl
rm -rf $O/cut_stats
rocprofv3 --kernel-trace --stats -d $O/cut_stats -o st --output-format csv -- python3 tools/cut_diag.py cfg3 > $O/cut_stats.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r4/cut_stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:25]:
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f}')
PY
