#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats, then HBM traffic counters in their own
# passes (MI355X_MICROARCH.md: never combine --pmc with other traces; FETCH_SIZE and WRITE_SIZE
# do not fit one pass).  The profiled passes skip the sharded-compress leg (--shard-gib 0): the encoder's persistent grid is the
# same 4 x CUs workgroups for every batch size, so its launches could not be told apart by grid size in the summary.
# usage (via gpurun): tools/profile_bench.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --shard-gib 0 > $OUT/bench_profiled.json 2>> $OUT/bench.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --shard-gib 0 > /dev/null 2>> $OUT/bench.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --shard-gib 0 > /dev/null 2>> $OUT/bench.err
python3 $R/tools/profile_summary.py $OUT > $OUT/summary.md
cat $OUT/summary.md
# keep what is judged (summary, kernel stats, traffic.json, the bench lines); the raw traces stay on the box
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
