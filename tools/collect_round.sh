#!/bin/bash
# GPU box: final collection of the round, all from ONE box and call (boxes of the pool differ by several %):
#   bench line; bench under rocprofv3 --kernel-trace --stats + PMC of the eval forward (profile_mlp.sh); B=32 training kernel
#   stats; training PMC (pmc_train.sh); HBM-bound kernels: stats + PMC (pmc_hbm.sh); launch histogram + ordered launches of a
#   replayed B=4 GAN iteration; training lines; C5 multi-object line.          Usage: tools/collect_round.sh [tag]
T=${1:-r5}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
( time timeout 1500 python bench.py > gpurun_out/$T/bench_full.json 2> gpurun_out/$T/bench_full.err ) 2> gpurun_out/$T/bench_wall.txt
bash tools/profile_mlp.sh $T > gpurun_out/$T/profile_mlp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/train32 -o t -- python3 tools/train_bench.py 32 0 30 0 f16x3 > gpurun_out/$T/train32.log 2>&1
cp $(find gpurun_out/$T/train32 -name '*kernel_stats.csv' | head -1) gpurun_out/$T/train32_kernel_stats.csv
bash tools/pmc_train.sh $T > gpurun_out/$T/pmc_train.log 2>&1
bash tools/pmc_hbm.sh $T > gpurun_out/$T/pmc_hbm.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$T/gan4 -o t -- python3 tools/train_bench.py 4 1 12 1 f16x3 > gpurun_out/$T/gan4.log 2>&1
python3 tools/launch_histogram.py gpurun_out/$T/gan4 > gpurun_out/$T/launch_histogram.txt 2>&1
python3 tools/launch_sequence.py gpurun_out/$T/gan4 > gpurun_out/$T/launch_sequence.txt 2>&1
for i in 1 2; do python3 tools/train_bench.py 4 1 200 1 f16x3 2>&1 | tail -1; python3 tools/train_bench.py 4 0 200 1 f16x3 2>&1 | tail -1; python3 tools/train_bench.py 32 0 60 1 f16x3 2>&1 | tail -1; done > gpurun_out/$T/train_lines.txt
timeout 600 python bench.py --config c5 > gpurun_out/$T/c5.json 2> gpurun_out/$T/c5.err
# the GAN loop with one change switched off at a time (same box, alternating with the product configuration)
( for e in TP_X=1 TP_NO_FEAT_CHAIN=1 TP_X=1 TP_NO_DISC_PAIRS=1 TP_X=1 TP_NO_SN_SETS=1 TP_X=1 "TP_NO_FEAT_CHAIN=1 TP_NO_DISC_PAIRS=1 TP_NO_SN_SETS=1" TP_X=1 TP_LINEAR_GRAPHS=0 TP_X=1 TP_FOUR_GRAPHS=0 TP_X=1 TP_PRE_STREAMS=2 "TP_PRE_STREAMS=2 TP_NO_QUEUE_PROBE=1" TP_X=1; do
    echo "$e $(env $e python3 tools/train_bench.py 4 1 200 1 f16x3 2>&1 | tail -1 | cut -c1-75)"; done ) > gpurun_out/$T/gan_ablations.txt
python3 tools/tail_bench.py 4 > gpurun_out/$T/tail_bench.txt 2>&1
python3 tools/host_time.py > gpurun_out/$T/host_time.txt 2>&1
python3 tools/linear_timeline.py 50 > gpurun_out/$T/linear_timeline.txt 2>&1
python3 tools/chain_probe.py 50 > gpurun_out/$T/chain_probe.txt 2>&1
python3 tools/boundary_probe.py 64 > gpurun_out/$T/boundary_probe.txt 2>&1
# fabric traffic of the exact-fp32 forward (profiles/traffic.json: fp32)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex mlp_fwd --output-format csv -d gpurun_out/$T/fp32_$c -- python3 tools/render_once.py fp32 2 > gpurun_out/$T/fp32_$c.log 2>&1
  cp $(find gpurun_out/$T/fp32_$c -name '*counter_collection.csv' | head -1) gpurun_out/$T/fp32_${c}_counter_collection.csv; rm -rf gpurun_out/$T/fp32_$c
done
( for k in 0 1 2 3; do python3 tools/queue_probe.py $k 8; done ) > gpurun_out/$T/queue_probe.txt 2>&1
# the evaluation render with / without the ray-bias variant of the f16x3 kernel, alternating on this box
( for e in TP_X=1 TP_NO_RAY_BIAS=1 TP_X=1 TP_NO_RAY_BIAS=1; do
    echo "$e $(env $e python3 bench.py --no-train --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])")"; done ) > gpurun_out/$T/ray_bias_ab.txt 2>&1
python3 tools/kstats.py gpurun_out/$T/train32 wgrad dgrad mlp_fwd finalize
rm -rf gpurun_out/$T/train32 gpurun_out/$T/gan4
cat gpurun_out/$T/train_lines.txt | cut -c1-100
cat gpurun_out/$T/bench_wall.txt
python3 - <<PY
import json
d=json.load(open("gpurun_out/$T/bench_full.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["traffic"])
t=d["train"]; print(t["full_gan_loop"]["value"], t["nerf_step_b4"]["value"], t["nerf_step_b32"]["value"], t["roofline"])
PY
