#!/bin/bash
# GPU box: kernel stats + fabric traffic of the HBM-bound kernels (tools/hbm_kernels.py).  Usage: tools/pmc_hbm.sh <tag>
#   -> gpurun_out/hbm_<tag>/{line.json, kernel_stats.csv, pmc.json}
# One rocprofv3 pass per counter (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md); --pmc runs
# carry no trace domains.  FETCH_SIZE is in KiB and counts 64 B per 128-B request on gfx950: x2 (same guide, HBM section).
export TMPDIR=/tmp
TAG=${1:-run}
OUT=gpurun_out/hbm_$TAG
mkdir -p $OUT
python3 tools/hbm_kernels.py > $OUT/line.json 2> $OUT/line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o t -- python3 tools/hbm_kernels.py > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "composite|raygen|patch_gather" --output-format csv -d $OUT/$c -- python3 tools/hbm_kernels.py > $OUT/$c.log 2>&1
done
python3 tools/summarize_hbm_pmc.py $OUT > $OUT/pmc.json
cat $OUT/pmc.json
for c in FETCH_SIZE WRITE_SIZE; do cp $(find $OUT/$c -name '*counter_collection.csv' | head -1) $OUT/${c}_counter_collection.csv; done
rm -rf $OUT/stats $OUT/FETCH_SIZE $OUT/WRITE_SIZE
