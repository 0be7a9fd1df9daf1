#!/bin/bash
# GPU box: the second collection of round 5 (the training iteration after the critical-path work), ONE box and call.  Usage: tools/collect_round5b.sh [tag]
T=${1:-r5c}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
O=gpurun_out/$T
( time timeout 1500 python bench.py > $O/bench_full.json 2> $O/bench_full.err ) 2> $O/bench_wall.txt
python3 tools/chain_probe.py 50 > $O/chain_probe.txt 2>&1
python3 tools/boundary_probe.py 64 > $O/boundary_probe.txt 2>&1
( echo "## strict (the calling stream waits for both optimiser steps)"; python3 tools/linear_timeline.py 50 2>&1 | tail -16
  echo "## defer_results + pipeline_disc_tail (the benchmark loops)"; TP_PIPELINE_DISC=1 TP_DEFER_RESULTS=1 python3 tools/linear_timeline.py 50 2>&1 | tail -16 ) > $O/linear_timeline.txt
python3 tools/launch_counts.py > $O/launch_counts.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/gan4 -o t -- python3 tools/train_bench.py 4 1 12 1 f16x3 > $O/gan4.log 2>&1
python3 tools/launch_histogram.py $O/gan4 > $O/launch_histogram.txt 2>&1
python3 tools/launch_sequence.py $O/gan4 > $O/launch_sequence.txt 2>&1
rm -rf $O/gan4
for i in 1 2; do python3 tools/train_bench.py 4 1 300 1 f16x3 2>&1 | tail -1; python3 tools/train_bench.py 4 0 200 1 f16x3 2>&1 | tail -1; done > $O/train_lines.txt
# one change switched off at a time, alternating with the product configuration
( for e in TP_X=1 TP_NO_DISC_STEP_TAIL=1 TP_X=1 TP_NO_DGRAD_INORM=1 TP_X=1 TP_NO_GEN_SCHEDULE=1 TP_X=1 TP_NO_DEFER=1 TP_X=1 TP_NO_PIPELINE_DISC=1 TP_X=1 "TP_NO_DEFER=1 TP_NO_PIPELINE_DISC=1" TP_X=1 "TP_NO_DEFER=1 TP_NO_PIPELINE_DISC=1 TP_NO_DISC_SPLIT=1" TP_X=1 TP_FEAT_ON_OWN_STREAM=1 TP_X=1 TP_SN_SPLIT=1 TP_X=1 "TP_NO_DISC_STEP_TAIL=1 TP_NO_DGRAD_INORM=1 TP_NO_GEN_SCHEDULE=1 TP_NO_DEFER=1 TP_NO_PIPELINE_DISC=1 TP_NO_DISC_SPLIT=1 TP_FEAT_ON_OWN_STREAM=1" TP_X=1 TP_NO_FEAT_CHAIN=1 TP_X=1 TP_NO_DISC_PAIRS=1 TP_X=1; do
    echo "$e $(env $e python3 tools/train_bench.py 4 1 300 1 f16x3 2>&1 | tail -1 | cut -c1-75)"; done ) > $O/gan_ablations.txt
# where the critical path runs: ten one-thread launches appended to one graph at a time
( echo "## defer_results + pipeline_disc_tail"; for e in TP_X=1 TP_EXTRA_LAUNCHES=D1=10 TP_EXTRA_LAUNCHES=G1=10 TP_EXTRA_LAUNCHES=F=10 TP_EXTRA_LAUNCHES=G2a=10 TP_EXTRA_LAUNCHES=G2b=10 TP_EXTRA_LAUNCHES=D2a=10 TP_EXTRA_LAUNCHES=D2b=10 TP_EXTRA_LAUNCHES=D2b=20 TP_X=1; do
    echo "$e $(env $e python3 tools/train_bench.py 4 1 300 1 f16x3 2>&1 | tail -1 | cut -c1-75)"; done
  echo "## strict"; for e in TP_X=1 TP_EXTRA_LAUNCHES=D1=10 TP_EXTRA_LAUNCHES=G1=10 TP_EXTRA_LAUNCHES=F=10 TP_EXTRA_LAUNCHES=G2a=10 TP_EXTRA_LAUNCHES=G2b=10 TP_EXTRA_LAUNCHES=D2a=10 TP_EXTRA_LAUNCHES=D2b=10 TP_X=1; do
    echo "$e $(env TP_NO_DEFER=1 TP_NO_PIPELINE_DISC=1 $e python3 tools/train_bench.py 4 1 300 1 f16x3 2>&1 | tail -1 | cut -c1-75)"; done
  echo "## graphs left out of the replay (timing only)"; for e in TP_X=1 TP_ABLATE=D2a,D2b TP_ABLATE=D2b TP_ABLATE=D1 TP_ABLATE=F TP_ABLATE=G2a TP_ABLATE=D1,D2a,D2b,G2a,F TP_X=1; do
    echo "$e $(env $e python3 tools/train_bench.py 4 1 300 1 f16x3 2>&1 | tail -1 | cut -c1-75)"; done ) > $O/critical_path_probes.txt
( python3 tools/soak_train.py 20000 2>&1 | tail -9; python3 tools/soak_train.py 3000 2>&1 | tail -3; python3 tools/soak_train.py 3000 2>&1 | tail -3; TP_SOAK_STRICT=1 python3 tools/soak_train.py 3000 2>&1 | tail -3 ) > $O/soak.txt
python3 tools/host_time.py > $O/host_time.txt 2>&1
cat $O/train_lines.txt | cut -c1-100; cat $O/bench_wall.txt; tail -5 $O/soak.txt
