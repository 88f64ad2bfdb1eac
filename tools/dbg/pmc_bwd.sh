#!/bin/bash
# usage: tools/dbg/pmc_bwd.sh <variant.so> <outdir>
set -e
export GCP_LIBRARY=$1
OUT=$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace -d $OUT -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/raster_bench.py --no-cameras > $OUT/log.txt 2>&1
