# same-box A/B of the walk on clustered scenes over library variants:   gpurun -- bash tools/clustered_ab.sh wb16 ...
set -e
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out/clustered_ab
for v in intree "$@"; do
  if [ "$v" = intree ]; then unset GCP_LIBRARY; else export GCP_LIBRARY=$PWD/variants/$v.so; fi
  python3 ${GCP_AB_TOOL:-tools/clustered_bench.py} ${GCP_AB_ARGS:-0 4 8} > gpurun_out/clustered_ab/$v.log 2>&1
  python3 - $v gpurun_out/clustered_ab/$v.log <<'PY'
import json, sys
for line in open(sys.argv[2]):
    if line.startswith("{"):
        d = json.loads(line)
        print(f"{sys.argv[1]:10s} sigma/{d['sigma_divisor']:<4} deepest {d['deepest_tile_list']:6d}  walk {d['walk_ms']:.3f} ms (no drops {d['walk_no_drops_ms']:.3f}, mostly dropped {d['walk_mostly_dropped_ms']:.3f})  call {d['create_alpha_brend_ms']:.3f} ms  blend fwd {d['blend_forward_ms']:.3f} bwd {d['blend_backward_ms']:.3f}")
PY
done
