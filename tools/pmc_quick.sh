#!/bin/bash
# GPU box: fabric traffic + L2 hit rate of the f16x3 forward (one pass per counter group), short form of profile_mlp.sh.
# Usage: tools/pmc_quick.sh <tag>   -> gpurun_out/pmc_<tag>.json
export TMPDIR=/tmp
TAG=${1:-run}
OUT=gpurun_out/pmcq_$TAG
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-include-regex mlp_fwd_f16x3 --output-format csv -d $OUT/$n -- python3 tools/render_once.py f16x3 2 > $OUT/$n.log 2>&1
done
python3 tools/summarize_pmc.py $OUT > gpurun_out/pmc_$TAG.json
cat gpurun_out/pmc_$TAG.json
