# HBM-side traffic of the tile-list walk (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) for the product library
# and any variants (tools/build_variant.py):   gpurun -- bash tools/walk_traffic.sh walk64 walksuper
set -e
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
O=gpurun_out/walk_traffic
mkdir -p $O
for v in intree "$@"; do
  if [ "$v" = intree ]; then unset GCP_LIBRARY; else export GCP_LIBRARY=$PWD/variants/$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/${v}_$c
    rocprofv3 --kernel-trace --pmc $c -d $O/${v}_$c -o p --output-format csv -- python3 tools/walk_bench.py cfg3 --iters 3 > $O/${v}_$c.log 2>&1
  done
done
python3 - "$@" <<'PY'
import csv, glob, sys, collections
for v in ["intree"] + sys.argv[1:]:
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"gpurun_out/walk_traffic/{v}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_pairs_scan_boxes<0" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
        for k, vals in acc.items():
            out.setdefault(k, {})[c] = sum(vals) / len(vals)
    for k, d in out.items():
        f, w = d.get("FETCH_SIZE", 0), d.get("WRITE_SIZE", 0)
        print(f"{v:10s} {k:62s} FETCH_SIZE {f/1e6:8.3f} GiB-ish(KiB/1e6) WRITE_SIZE {w/1e6:8.3f}  bytes: fetch {f*1024/1e9:6.3f} GB (x2: {2*f*1024/1e9:6.3f}) write {w*1024/1e9:6.3f} GB")
PY
