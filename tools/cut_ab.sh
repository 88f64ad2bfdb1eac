# same-box A/B of the cut over library variants:   gpurun -- bash tools/cut_ab.sh cut_base ...
set -e
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
for round in 1 2; do
for v in intree "$@"; do
  if [ "$v" = intree ]; then unset GCP_LIBRARY; else export GCP_LIBRARY=$PWD/variants/$v.so; fi
  O=gpurun_out/cut_ab/$v; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O -o st --output-format csv -- python3 tools/cut_diag.py cfg3 --only one > $O.log 2>&1
  python3 - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
for f in glob.glob(f"gpurun_out/cut_ab/{v}/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) >= 10 and "at::native" not in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / 10 / 1e3
    first = [float(r["AverageNs"]) / 1e3 for r in rows if "k_rect_rows_local" in r["Name"]]
    print(f"{v:12s} k_rect_rows_local {first[0]:7.1f} us   all kernels of a cut {tot:7.1f} us")
PY
done
done
