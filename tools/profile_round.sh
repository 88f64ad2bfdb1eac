#!/bin/bash
# Collect one round's rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r01'
# then, locally:  python tools/summarize_profiles.py gpurun_out/r01 r01
# Kernel trace + stats in one run; FETCH_SIZE and WRITE_SIZE in two separate --pmc runs (never combined with traces).
set -e -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $OUT/bench_plain.json 2> $OUT/plain.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py > $OUT/bench_stats.json 2> $OUT/stats.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $OUT/bench_fetch.json 2> $OUT/fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $OUT/bench_write.json 2> $OUT/write.err
# keep what the summariser needs, drop the bulky per-dispatch traces (gpurun_out/ merges back at most 64 MiB)
find $OUT -name "*kernel_trace.csv" -delete
du -sh $OUT
