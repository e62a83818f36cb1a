tag=${1:-r06d}
mkdir -p gpurun_out/$tag
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest_gpu.txt 2>&1; tail -2 gpurun_out/$tag/pytest_gpu.txt
timeout 900 python tools/fuzz_ring_split.py 671 672 673 674 > gpurun_out/$tag/fuzz_ring_split.txt 2>&1; tail -1 gpurun_out/$tag/fuzz_ring_split.txt
timeout 900 python tools/fuzz_mutations.py 4096 681 682 683 > gpurun_out/$tag/fuzz_mutations.txt 2>&1; tail -1 gpurun_out/$tag/fuzz_mutations.txt
timeout 900 python tools/fuzz_roundtrip.py 1024 691 692 693 > gpurun_out/$tag/fuzz_roundtrip.txt 2>&1; tail -1 gpurun_out/$tag/fuzz_roundtrip.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python tools/framed_probe.py 65536 8 2>&1 | grep -i framed
bash tools/r06_tl.sh > /dev/null 2>&1; cp gpurun_out/r06c/framed_timeline.txt gpurun_out/$tag/framed_timeline.txt
bash tools/profile_bench.sh $tag > gpurun_out/$tag/profile.log 2>&1; head -4 gpurun_out/$tag/profile.log | cut -c1-400
