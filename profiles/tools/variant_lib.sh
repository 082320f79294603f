#!/bin/bash
# Build a VARIANT of the library for same-box A/B timing: one source file recompiled with extra flags (probe -D switches), linked with the
# working tree's other objects.   profiles/tools/variant_lib.sh <name> <file.hip> "<extra flags>"  ->  profiles/ab/libmaster_hip_<name>.so
# (git-ignored, travels with gpurun; select with MASTER_HIP_LIB=profiles/ab/libmaster_hip_<name>.so)
set -e
name=$1; src=$2; extra=$3
root=$(cd "$(dirname "$0")/../.." && pwd); csrc=$root/mastermetastyletransfer_amd/csrc
mkdir -p "$root/profiles/ab"
flags=$(cd "$root" && python3 -c "
import os, sys
from mastermetastyletransfer_amd import build as B
print(' '.join([*B.FLAGS, *B.EXTRA_FLAGS.get('$src', [])]))")
obj=$root/profiles/ab/${src%.hip}_$name.o
/opt/rocm/bin/hipcc $flags $extra -c "$csrc/$src" -o "$obj"
others=$(ls "$csrc"/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/profiles/ab/libmaster_hip_$name.so" $others "$obj"
rm -f "$obj"; echo "profiles/ab/libmaster_hip_$name.so  <-  $src $extra"
