# same-box A/B of csrc/gcp_pixels.hip over library variants:   gpurun -- bash tools/pixels_ab.sh px_nolook px_fwd
set -e
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out/pixels_ab
for round in 1 2; do
for v in intree "$@"; do
  if [ "$v" = intree ]; then unset GCP_LIBRARY; else export GCP_LIBRARY=$PWD/variants/$v.so; fi
  python3 tools/wrapper_bench.py cfg2 cfg3 --only-carry > gpurun_out/pixels_ab/$v.$round.log 2>&1
  python3 - $v gpurun_out/pixels_ab/$v.$round.log <<'PY'
import json, sys
for line in open(sys.argv[2]):
    if line.startswith("{"):
        d = json.loads(line)
        print(f"{sys.argv[1]:10s} {d['workload']}: min(T) {d['create_alpha_brend_min_ms']:.3f}  min(unordered) {d['create_alpha_brend_min_unordered_values_ms']:.3f}  "
              f"+extent {d['create_alpha_brend_min_extent_read_back_ms']:.3f}  first-index {d['create_grad_alphabrend_min_ms']:.3f}  int64 {d['create_alpha_brend_min_int64_ms']:.3f}  "
              f"torch {d['torch_unique_dim0_scatter_amin_ms']:.1f} ms")
PY
done
done
