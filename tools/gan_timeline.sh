#!/bin/bash
# GPU box: where the B=4 GAN iteration's time goes OUTSIDE the MLP kernels -- the same replayed iteration with one thing changed
# at a time, on one box, alternating with the product configuration; plus a kernel trace with the hardware queue of every launch.
# Usage: tools/gan_timeline.sh <tag>
T=${1:-r4}
O=gpurun_out/$T
mkdir -p $O
export TMPDIR=/tmp
run() { echo "$1 $(env $1 python3 tools/train_bench.py 4 ${2:-1} 200 1 f16x3 2>&1 | tail -1 | cut -c1-80)"; }
( run TP_X=1
  run GPU_MAX_HW_QUEUES=8
  run GPU_MAX_HW_QUEUES=2
  run TP_X=1
  run TP_BACKWARD_FIRST=1
  run TP_SKIP_DISC_STEP=1
  run TP_NO_BRANCH_OVERLAP=1
  run TP_NO_FEAT_BRANCH=1
  run TP_X=1
  run TP_X=1 0
  run HIP_FORCE_DEV_KERNARG=1
  run DEBUG_HIP_GRAPH_DOT_PRINT=0
  run TP_X=1 ) > $O/gan_timeline.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/gan4 -o t -- python3 tools/train_bench.py 4 1 12 1 f16x3 > $O/gan4.log 2>&1
python3 tools/launch_histogram.py $O/gan4 > $O/launch_histogram.txt 2>&1
python3 tools/launch_sequence.py $O/gan4 > $O/launch_sequence.txt 2>&1
rm -rf $O/gan4
cat $O/gan_timeline.txt
