set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
for v in intree walksuper; do
  if [ "$v" = intree ]; then unset GCP_LIBRARY; else export GCP_LIBRARY=$GRAFT_REPO_ROOT/variants/$v.so; fi
  timeout -k 10 200 python3 tools/walk_bench.py cfg3 --iters 10 2>&1 | tail -1
  timeout -k 10 200 python3 tools/walk_bench.py cfg2 --iters 10 2>&1 | tail -1
done
