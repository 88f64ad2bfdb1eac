#!/bin/bash
# Collect one round's rocprofv3 evidence on the GPU box (run through gpurun), then condense it with
# tools/summarize_profiles.py into profiles/ (tracked):
#   gpurun --timeout 1100 -- 'tools/profile_round.sh r02'
# Counter passes are separate runs with --pmc only (no trace domains besides --kernel-trace), the program itself after `--`.
set -e
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/long_fetch $OUT/long_write $OUT/fn_pmc $OUT/fn_stats
cd /tmp && export TMPDIR=/tmp
echo "== plain bench"; python3 $ROOT/bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
echo "== kernel trace + stats of bench.py"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o st --output-format csv -- python3 $ROOT/bench.py --cpu-sample 0 > $OUT/bench_stats.json 2> $OUT/bench_stats.err
export GCP_BENCH_NO_SHARDED=1
echo "== pmc FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p --output-format csv -- python3 $ROOT/bench.py --cpu-sample 0 --steps 20 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "== pmc WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o p --output-format csv -- python3 $ROOT/bench.py --cpu-sample 0 --steps 20 > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "== long groups: timing, then traffic"
python3 $ROOT/tools/pathological_bench.py > $OUT/long_groups.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/long_fetch -o p --output-format csv -- python3 $ROOT/tools/pathological_bench.py "whole array" "5000" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/long_write -o p --output-format csv -- python3 $ROOT/tools/pathological_bench.py "whole array" "5000" > /dev/null 2>&1
echo "== Function kernels: stats, then SQ counters"
rocprofv3 --kernel-trace --stats -d $OUT/fn_stats -o st --output-format csv -- python3 $ROOT/tools/raster_bench.py --no-cameras > $OUT/fn_bench.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/fn_pmc -o p --output-format csv -- python3 $ROOT/tools/raster_bench.py --no-cameras > /dev/null 2>&1
echo "== rows a5 / a6 (create_alpha_brend, grad_cumsum, and their boxes route): stats, then traffic"
mkdir -p $OUT/wr_stats $OUT/wr_fetch $OUT/wr_write
python3 $ROOT/tools/wrapper_bench.py cfg2 cfg3 --stages > $OUT/wr_bench.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/wr_stats -o st --output-format csv -- python3 $ROOT/tools/wrapper_bench.py cfg3 --profile --iters 4 > $OUT/wr_prof.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/wr_fetch -o p --output-format csv -- python3 $ROOT/tools/wrapper_bench.py cfg3 --profile --iters 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/wr_write -o p --output-format csv -- python3 $ROOT/tools/wrapper_bench.py cfg3 --profile --iters 2 > /dev/null 2>&1
echo "== blend kernels: could two list entries share a visit?"
python3 $ROOT/tools/blend_pairing_stats.py cfg3 > $OUT/pairing.txt 2>&1
python3 $ROOT/tools/blend_pairing_stats.py cfg2 >> $OUT/pairing.txt 2>&1
echo "== done"; ls $OUT
