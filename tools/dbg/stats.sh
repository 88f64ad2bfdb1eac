#!/bin/bash
# usage: tools/dbg/stats.sh <outdir> -- prints per-kernel avg durations of tools/raster_bench.py
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o st --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/raster_bench.py --no-cameras > $OUT/log.txt 2>&1
python3 - <<EOF
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("blend","grad_reduce","sort","tile")):
        print(f'{n[:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us')
EOF
