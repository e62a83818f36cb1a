#!/bin/bash
# round 6: the fuzz campaigns of round 5 again, on the shipped library of this round (new seeds)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r06fuzz
mkdir -p $O
( SNAPPY_HIP_ENC_GWAVES=4,1 timeout 900 python3 tools/fuzz_encode.py 4096 601 602 603 604 605 606 607 608 2>&1 | grep -v amdgpu.ids | tail -12 ) > $O/fuzz_encode.txt
( timeout 600 python3 tools/fuzz_mutations.py 4096 611 612 613 614 615 2>&1 | grep -v amdgpu.ids | tail -8 ) > $O/fuzz_mutations.txt
( timeout 600 python3 tools/fuzz_roundtrip.py 1024 621 622 623 624 2>&1 | grep -v amdgpu.ids | tail -6 ) > $O/fuzz_roundtrip.txt
( timeout 900 python3 tools/fuzz_ring_split.py 631 632 633 634 635 636 2>&1 | grep -v amdgpu.ids | tail -14 ) > $O/fuzz_ring_split.txt
tail -3 $O/*.txt
