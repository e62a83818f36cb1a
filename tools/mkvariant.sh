#!/bin/bash
# Build a variant of the library for same-box A/B runs: tools/mkvariant.sh <name> [extra hipcc flags]
# -> tools/probes/lib_<name>.so (git-ignored; travels to the GPU box).  tools/ab.sh runs them.
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function "$@" \
  -o $R/tools/probes/lib_$name.so $R/nim-snappy_amd/csrc/snappy_hip.hip -Wl,-rpath,/opt/rocm/lib && echo built lib_$name.so
