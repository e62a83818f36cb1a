#!/bin/bash
# Same-box A/B of library variants built by tools/mkvariant.sh (boxes differ by 10 % and more, so
# only numbers of one run compare).  usage (via gpurun): tools/ab.sh "<probe command>" <name> <name> ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cmd=$1; shift
cp $R/nim-snappy_amd/libsnappy_hip.so /tmp/lib_keep.so
trap 'cp /tmp/lib_keep.so $R/nim-snappy_amd/libsnappy_hip.so' EXIT INT TERM  # the tracked path gets its library back, whatever happens
for rep in 1 2; do
for v in "$@"; do
  cp $R/tools/probes/lib_$v.so $R/nim-snappy_amd/libsnappy_hip.so
  echo "== $v (rep $rep)"
  (cd $R && eval "$cmd" 2>&1 | grep -v amdgpu.ids)
done
done
cp /tmp/lib_keep.so $R/nim-snappy_amd/libsnappy_hip.so
