mkdir -p gpurun_out/r06c
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06c/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r06c/pytest_gpu.txt
timeout 600 python tools/fuzz_ring_split.py > gpurun_out/r06c/fuzz_ring_split.txt 2>&1; tail -2 gpurun_out/r06c/fuzz_ring_split.txt
timeout 600 python tools/fuzz_mutations.py > gpurun_out/r06c/fuzz_mutations.txt 2>&1; tail -2 gpurun_out/r06c/fuzz_mutations.txt
timeout 600 python tools/fuzz_roundtrip.py > gpurun_out/r06c/fuzz_roundtrip.txt 2>&1; tail -2 gpurun_out/r06c/fuzz_roundtrip.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_bench.sh r06c > gpurun_out/r06c/profile.log 2>&1; tail -5 gpurun_out/r06c/profile.log
