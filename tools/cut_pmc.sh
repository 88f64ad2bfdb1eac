# SQ counters of the cut's first pass (k_rect_rows_local): where do its wave cycles go?   gpurun -- bash tools/cut_pmc.sh
set -e
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
O=gpurun_out/cut_pmc
mkdir -p $O
rm -rf $O/p1 $O/p2 $O/p3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/p1 -o p --output-format csv -- python3 tools/cut_diag.py cfg3 --only one > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES -d $O/p2 -o p --output-format csv -- python3 tools/cut_diag.py cfg3 --only one > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p3 -o p --output-format csv -- python3 tools/cut_diag.py cfg3 --only one > $O/p3.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    for f in glob.glob(f"gpurun_out/cut_pmc/{p}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "k_rect_rows_local" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in acc.items():
            print(p, k, f"{v / max(n, 1):.4g}", n)
PY
