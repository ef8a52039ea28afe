#!/bin/bash
# round-2 evidence: bench.py alone (JSON line), then the same command under rocprofv3 --kernel-trace --stats
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd $R
python3 bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
tail -1 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err || { tail -20 $OUT/prof.err; exit 1; }
tail -1 $OUT/bench_under_rocprof.json
python3 tools/summarize_rocprof.py $OUT/prof $OUT/kernel_stats.md "round 2: python3 bench.py --steps 3 --warmup 1 --no-extras, K3" > /dev/null
head -30 $OUT/kernel_stats.md
