#!/bin/bash
# per-kernel register / spill / occupancy summary of one source file, compiled with build.py's flags for it (run from the repo root)
# usage: profiles/tools/kernel_regs.sh mastermetastyletransfer_amd/csrc/mlp_x3.hip [extra hipcc flags] | c++filt
src=$1; shift
flags=$(python3 -c "
import os, sys
sys.path.insert(0, os.getcwd())
from mastermetastyletransfer_amd import build as B
print(' '.join(f for f in [*B.FLAGS, *B.EXTRA_FLAGS.get(os.path.basename('$src'), [])] if f != '-fPIC'))")
/opt/rocm/bin/hipcc $flags "$@" -c --offload-device-only -Rpass-analysis=kernel-resource-usage "$src" -o /dev/null 2>&1 | \
  awk '/remark:.* Name:/ {n=$0; sub(/.*Name: /,"",n); sub(/ \[.*/,"",n); name=n}
       / VGPRs:/ {v=$0; sub(/.* VGPRs: /,"",v); sub(/ .*/,"",v)} / AGPRs:/ {a=$0; sub(/.* AGPRs: /,"",a); sub(/ .*/,"",a)}
       /VGPRs Spill:/ {s=$0; sub(/.*Spill: /,"",s); sub(/ .*/,"",s)} /Occupancy/ {o=$0; sub(/.*SIMD\]: /,"",o); sub(/ .*/,"",o)}
       /LDS Size/ {printf "%-100s vgpr %4s agpr %4s spill %3s occ %s\n", name, v, a, s, o}'
