#!/bin/bash
# Build the library of another git revision next to the working tree's, for same-box A/B timing:
#   profiles/ab_build.sh [rev=HEAD] ["extra hipcc flags"]   ->  profiles/ab/libmaster_hip_A.so   (git-ignored, travels with gpurun)
#   gpurun -- 'MASTER_HIP_LIB=profiles/ab/libmaster_hip_A.so python bench.py --no-extras; python bench.py --no-extras'
# Every file is compiled with the per-file flags of the WORKING TREE's build.py (r05: this script used to compile with plain -O3, i.e.
# without -amdgpu-mfma-vgpr-form etc. - an A/B against such a library measures the flags, not the revision).
set -e
rev=${1:-HEAD}; extra=$2; root=$(git rev-parse --show-toplevel); tmp=$(mktemp -d)
mkdir -p "$tmp/pkg/csrc" "$tmp/include" "$root/profiles/ab"
for f in $(git ls-tree --name-only "$rev" mastermetastyletransfer_amd/csrc/ | grep -E '\.(hip|h)$'); do git show "$rev:$f" > "$tmp/pkg/csrc/$(basename $f)"; done
git show "$rev:include/master_hip.h" > "$tmp/include/master_hip.h"
objs=""
for f in "$tmp"/pkg/csrc/*.hip; do
  flags=$(cd "$root" && python3 -c "
from mastermetastyletransfer_amd import build as B
print(' '.join([f for f in B.FLAGS if not f.startswith('-I')] + B.EXTRA_FLAGS.get('$(basename $f)', [])))")
  /opt/rocm/bin/hipcc $flags $extra -I"$tmp/include" -c "$f" -o "${f%.hip}.o" & objs="$objs ${f%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/profiles/ab/libmaster_hip_A.so" $objs
rm -rf "$tmp"; echo "$root/profiles/ab/libmaster_hip_A.so  <-  $rev $extra"
