#!/bin/bash
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_vae_enc
rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 scratch/vae_scale.py 720p encode > $OUT/vae.log 2>&1 || { tail -20 $OUT/vae.log; exit 1; }
grep -E "decode|encode" $OUT/vae.log
python3 tools/summarize_rocprof.py $OUT/prof $OUT/kernel_stats.md "round 2: two 49x720p VAE decodes + two encodes (scratch/vae_scale.py 720p encode)" > /dev/null
head -16 $OUT/kernel_stats.md
