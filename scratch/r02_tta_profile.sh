#!/bin/bash
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_tta
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 tools/tta_steps.py 48 720p > $OUT/tta.log 2>&1 || { tail -20 $OUT/tta.log; exit 1; }
tail -3 $OUT/tta.log
python3 tools/summarize_rocprof.py $OUT/prof $OUT/kernel_stats.md "round 2 v1: 4 LoRA-TTA inner-loop steps, 720p Tc=4 Tt=3 (25 200 tokens), 48 blocks, r=8 qkv+proj, no block checkpointing" > /dev/null
cat $OUT/kernel_stats.md
