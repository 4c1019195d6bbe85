#!/bin/bash
# Compile-time variants of libloamx.so side by side.
#   here     : tools/variants.sh build <name> "<extra hipcc flags>"   -> loam_amd/lib/variants/<name>.so
#   GPU box  : tools/variants.sh run "<grep pattern of kernel rows>" [bench args]   (every variant, then the default build last)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$ROOT/loam_amd/lib/variants
if [ "$1" = build ]; then
  mkdir -p "$V"
  cd "$ROOT/loam_amd/csrc"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $3 -o "$V/$2.so" loamx_api.hip extract_kernels.hip register_kernels.hip synth_kernels.hip -ldl -Wl,-rpath,/opt/rocm/lib
  echo "built $V/$2.so"
elif [ "$1" = run ]; then
  pat=$2; shift 2
  cd "$ROOT"
  cp loam_amd/lib/libloamx.so /tmp/libloamx_default.so
  for f in "$V"/*.so /tmp/libloamx_default.so; do
    cp "$f" loam_amd/lib/libloamx.so; touch loam_amd/lib/libloamx.so
    echo "== $(basename "$f")"
    bash tools/kbench.sh "$@" | grep "pairs/s\|$pat" | cut -c1-80
  done
fi
