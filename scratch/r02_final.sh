#!/bin/bash
# round-2 closing evidence: bench (JSON line + progress), the same command under rocprofv3 --stats, the TTA step profile
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
python3 bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
tail -1 $OUT/bench.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err || { tail -20 $OUT/prof.err; exit 1; }
python3 tools/summarize_rocprof.py $OUT/prof $OUT/kernel_stats.md "round 2 v2: python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline, K3" > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tta -- python3 tools/tta_steps.py 48 720p > $OUT/tta.log 2>&1 || { tail -20 $OUT/tta.log; exit 1; }
grep "s/step" $OUT/tta.log | cut -c1-300
python3 tools/summarize_rocprof.py $OUT/prof_tta $OUT/tta_kernel_stats.md "round 2 v2: 4 LoRA-TTA inner-loop steps, 720p Tc=4 Tt=3 (25 200 tokens), 48 blocks, r=8 qkv+proj, no block checkpointing" > /dev/null
head -14 $OUT/tta_kernel_stats.md
