#!/bin/bash
# round-4 evidence: the default bench command's K3 part under rocprofv3 --kernel-trace --stats, and the 20-step TTA loop the same way
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04
rm -rf $OUT/prof $OUT/prof_tta; mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err || { tail -20 $OUT/prof.err; exit 1; }
tail -1 $OUT/bench_under_rocprof.json | cut -c1-400
python3 tools/summarize_rocprof.py $OUT/prof $OUT/bench_kernel_stats.md "round 4: python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline, K3 (gemm4k default)" > /dev/null
head -16 $OUT/bench_kernel_stats.md
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tta -- python3 tools/tta_steps.py > $OUT/tta_under_rocprof.log 2> $OUT/prof_tta.err || { tail -20 $OUT/prof_tta.err; exit 1; }
python3 tools/summarize_rocprof.py $OUT/prof_tta $OUT/tta_kernel_stats.md "round 4: tools/tta_steps.py, LoRA-TTA inner steps at 25 200 tokens, 48 blocks" > /dev/null
head -16 $OUT/tta_kernel_stats.md
