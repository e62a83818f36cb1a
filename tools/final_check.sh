# Everything that is run on the GPU box before a profile set is committed (via gpurun): tools/final_check.sh <tag>
# -> gpurun_out/<tag>/ (test and fuzz logs, the framed decode's kernel timeline) and gpurun_out/prof_<tag>/ (tools/profile_bench.sh).
tag=${1:-r06d}
mkdir -p gpurun_out/$tag
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest_gpu.txt 2>&1; tail -2 gpurun_out/$tag/pytest_gpu.txt
timeout 900 python tools/fuzz_ring_split.py 671 672 673 674 > gpurun_out/$tag/fuzz_ring_split.txt 2>&1; tail -1 gpurun_out/$tag/fuzz_ring_split.txt
timeout 900 python tools/fuzz_mutations.py 4096 681 682 683 > gpurun_out/$tag/fuzz_mutations.txt 2>&1; tail -1 gpurun_out/$tag/fuzz_mutations.txt
timeout 900 python tools/fuzz_roundtrip.py 1024 691 692 693 > gpurun_out/$tag/fuzz_roundtrip.txt 2>&1; tail -1 gpurun_out/$tag/fuzz_roundtrip.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python tools/framed_probe.py 65536 8 2>&1 | grep -i framed
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$tag/tl -- python3 $R/tools/framed_probe.py 65536 3 > /dev/null 2>&1)
python3 tools/framed_timeline.py gpurun_out/$tag/tl > gpurun_out/$tag/framed_timeline.txt; rm -rf gpurun_out/$tag/tl
bash tools/profile_bench.sh $tag > gpurun_out/$tag/profile.log 2>&1; head -4 gpurun_out/$tag/profile.log | cut -c1-400
