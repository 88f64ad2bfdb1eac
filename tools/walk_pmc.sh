#!/bin/bash
# SQ counters of the tile-list walk alone (where do the wave cycles go: parked, issue-stalled, issuing?)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
out=gpurun_out/walk_pmc
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/p1 -o p1 --output-format csv -- python3 tools/walk_bench.py cfg3 --iters 3 > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $out/p2 -o p2 --output-format csv -- python3 tools/walk_bench.py cfg3 --iters 3 > $out/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES -d $out/p3 -o p3 --output-format csv -- python3 tools/walk_bench.py cfg3 --iters 3 > $out/p3.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in ("p1","p2","p3"):
    for f in glob.glob(f"gpurun_out/walk_pmc/{p}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "k_pairs_scan_boxes<0, false, 1>" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        # several rows per dispatch (one per counter); count dispatches from any counter
        for k, (v, n) in acc.items():
            print(p, k, v / max(n, 1), n)
PY
