# A longer fuzz campaign on the GPU box (via gpurun): tools/fuzz_campaign.sh <first seed> <seeds per fuzzer>  -> gpurun_out/fuzz/
a=${1:-800}; k=${2:-12}
mkdir -p gpurun_out/fuzz
s() { seq $1 $(( $1 + k - 1 )) | tr '\n' ' '; }
timeout 1500 python tools/fuzz_ring_split.py $(s $a) > gpurun_out/fuzz/ring_split_$a.txt 2>&1; grep -c "mismatches 0 |.*mismatches 0" gpurun_out/fuzz/ring_split_$a.txt; grep -v "ring mismatches 0 |.* mismatches 0$" gpurun_out/fuzz/ring_split_$a.txt | grep -v amdgpu.ids | head -3
timeout 1500 python tools/fuzz_mutations.py 4096 $(s $((a+100))) > gpurun_out/fuzz/mutations_$a.txt 2>&1; grep -c "mismatches \[\]" gpurun_out/fuzz/mutations_$a.txt; grep -v "mismatches \[\]" gpurun_out/fuzz/mutations_$a.txt | grep -v amdgpu.ids | head -3
timeout 1500 python tools/fuzz_roundtrip.py 1024 $(s $((a+200))) > gpurun_out/fuzz/roundtrip_$a.txt 2>&1; grep -c "encoder mismatches \[\] decode status errors 0 decode mismatches \[\]" gpurun_out/fuzz/roundtrip_$a.txt; grep -v "encoder mismatches \[\] decode status errors 0 decode mismatches \[\]" gpurun_out/fuzz/roundtrip_$a.txt | grep -v amdgpu.ids | head -3
timeout 1500 python tools/fuzz_encode.py 4096 $(s $((a+300))) > gpurun_out/fuzz/encode_$a.txt 2>&1; grep -c "mismatches \[\]" gpurun_out/fuzz/encode_$a.txt; grep -v "mismatches \[\]" gpurun_out/fuzz/encode_$a.txt | grep -v amdgpu.ids | head -3
