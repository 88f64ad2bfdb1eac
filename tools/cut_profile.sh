# Per-kernel times of the rect cut (rocprofv3 --kernel-trace --stats over tools/cut_diag.py): the one-call cut and the step-by-step one.
#   gpurun -- bash tools/cut_profile.sh
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/cut_profile
mkdir -p $O
for mode in one step; do
  rm -rf $O/cut_$mode
  rocprofv3 --kernel-trace --stats -d $O/cut_$mode -o st --output-format csv -- python3 tools/cut_diag.py cfg3 --only $mode > $O/cut_$mode.log 2>&1
  echo "== $mode"
  python3 - $mode <<'PY'
import csv, glob, sys
mode = sys.argv[1]
for f in glob.glob(f"gpurun_out/cut_profile/cut_{mode}/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) >= 10 and "at::native" not in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = 0.0
    for r in rows[:14]:
        per_call = float(r["TotalDurationNs"]) / 10 / 1e3
        tot += per_call
        print(f'{r["Name"][:80]:80s} calls/cut {int(r["Calls"])/10:5.1f} us/cut {per_call:8.1f}')
    print("sum us per cut", round(tot, 1))
PY
done
