# Probe: bulk / tail launch times of the raw splitter for "budget,hops" settings (debug library: tools/build_variants.py dbg)
cd /tmp && export TMPDIR=/tmp
NB=${1:-16384}
for k in "24,16" "32,16" "48,16" "64,16" "255,16"; do
  SNAPPY_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/probes/lib_dbg.so SNAPPY_HIP_SPLIT_KNOBS=$k timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kn -- python3 $GRAFT_REPO_ROOT/tools/probes/raw64.py $NB 2>&1 | grep "^ms" | tail -1 | tr "\n" " "
  echo -n "knobs $k: "
  python3 $GRAFT_REPO_ROOT/tools/raw_timeline.py $GRAFT_REPO_ROOT/gpurun_out/kn | grep "split_bulk\|split_tail" | awk '{printf "%s ", $2} END {print ""}'
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/kn
done
