#!/bin/bash
# Build the library of another git revision next to the working tree's, for same-box A/B timing:
#   profiles/ab_build.sh [rev=HEAD]   ->  profiles/ab/libmaster_hip_A.so   (git-ignored, travels with gpurun)
#   gpurun -- 'MASTER_HIP_LIB=profiles/ab/libmaster_hip_A.so python bench.py --no-extras; python bench.py --no-extras'
set -e
rev=${1:-HEAD}; root=$(git rev-parse --show-toplevel); tmp=$(mktemp -d)
mkdir -p "$tmp/pkg/csrc" "$tmp/include" "$root/profiles/ab"
for f in $(git ls-tree --name-only "$rev" mastermetastyletransfer_amd/csrc/ | grep -E '\.(hip|h)$'); do git show "$rev:$f" > "$tmp/pkg/csrc/$(basename $f)"; done
git show "$rev:include/master_hip.h" > "$tmp/include/master_hip.h"
objs=""
for f in "$tmp"/pkg/csrc/*.hip; do /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I"$tmp/include" -c "$f" -o "${f%.hip}.o" & objs="$objs ${f%.hip}.o"; done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/profiles/ab/libmaster_hip_A.so" $objs
rm -rf "$tmp"; echo "$root/profiles/ab/libmaster_hip_A.so  <-  $rev"
