# A/B of the tile-list walk over library variants (tools/build_variant.py), e.g. the tile shapes of DESIGN.md §3.4:
#   python tools/build_variant.py walk32 -DGCP_TILE_SX=5            # 32 x 8 tiles
#   python tools/build_variant.py walk64 -DGCP_TILE_SX=6            # 64 x 4 tiles
#   python tools/build_variant.py walksuper -DGCP_TILE_SX=5 -DGCP_TILE_SY=4   # 32 x 16 super-tiles, 512-thread blocks
#   gpurun -- bash tools/walk_shape_ab.sh walk32 walk64 walksuper
set -e
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for v in intree "$@"; do
  if [ "$v" = intree ]; then unset GCP_LIBRARY; else export GCP_LIBRARY=$PWD/variants/$v.so; fi
  for cfg in cfg3 cfg2; do timeout -k 10 200 python3 tools/walk_bench.py $cfg --iters 10 2>&1 | tail -1; done
done
