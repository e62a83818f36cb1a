#!/bin/bash
# round 6, GPU call 1: same-box baseline of the decode probe + a PC-sampling profile of the ring kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r06a
mkdir -p $O
export PROBE_DBG=0
timeout 300 python3 tools/decode_probe.py 16384 T_TEXT,T_HTML > $O/probe_base.txt 2>&1
cat $O/probe_base.txt | grep -v amdgpu.ids | tail -4
timeout 120 /opt/rocm/bin/rocprofv3-avail info --pc-sampling > $O/pcs_avail.txt 2>&1
tail -20 $O/pcs_avail.txt
export SNAPPY_HIP_LIBRARY=$PWD/tools/probes/lib_gline.so
for method in stochastic host_trap; do
  unit=cycles; iv=65536
  if [ $method = host_trap ]; then unit=time; iv=1; fi
  timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $method --pc-sampling-interval $iv \
     --kernel-trace --output-format csv -d $O/pcs_$method -o pcs -- python3 tools/decode_probe.py 8192 T_TEXT > $O/pcs_$method.log 2>&1
  echo "pcs $method rc $?"; tail -5 $O/pcs_$method.log
  find $O/pcs_$method -type f | head; du -sh $O/pcs_$method
  python3 tools/pcs_report.py $O/pcs_$method decode_indexed_kernelILj16384 $O/pcs_${method}_ring.txt 2>&1 | tail -3
  python3 tools/pcs_report.py $O/pcs_$method index_units_kernel $O/pcs_${method}_index.txt 2>&1 | tail -3
  python3 tools/pcs_report.py $O/pcs_$method encode_blocks_kernel $O/pcs_${method}_encode.txt 2>&1 | tail -3
  for f in $(find $O/pcs_$method -name "*pc_sampling*.csv"); do head -5 $f > $O/pcs_${method}_head.txt; s=$(stat -c %s $f); if [ $s -gt 30000000 ]; then rm $f; fi; done
  head -60 $O/pcs_${method}_ring.txt
done
unset SNAPPY_HIP_LIBRARY
