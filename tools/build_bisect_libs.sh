#!/bin/bash
# Builds the forward translation unit of a few historical commits into tools/bisect/<sha>.so (git-ignored,
# travels to the GPU box) so that tools/forward_bisect.py can time them side by side in ONE process.
#     bash tools/build_bisect_libs.sh 46562d3 d4392df c833d28 HEAD
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$R/tools/bisect"
for c in "$@"; do
  sha=$(git -C "$R" rev-parse --short=7 "$c")
  T=$(mktemp -d)
  git -C "$R" archive "$sha" cuembed_amd/csrc include | tar -x -C "$T"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -I"$T/include" -I"$T/cuembed_amd/csrc" \
        "$T/cuembed_amd/csrc/c_api_forward.hip" -o "$R/tools/bisect/$sha.so" &
done
wait
ls -la "$R/tools/bisect"
