#!/bin/bash
# GPU box: rocprofv3 evidence for the fused MLP kernels.  Usage: tools/profile_mlp.sh <tag>   (writes gpurun_out/prof_<tag>/)
# One pass per counter group (SQ: 8 slots, TCC: 4, FETCH_SIZE and WRITE_SIZE do not fit together); never --pmc with traces.
set -u
TAG=${1:-run}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
pass() {  # name, precision, counters...
  local name=$1 prec=$2; shift 2
  rocprofv3 --pmc "$@" --kernel-include-regex "mlp_fwd|rb_" --output-format csv -d $OUT/$name -- python3 tools/render_once.py $prec 2 > $OUT/$name.log 2>&1
}
pass f16x3_sq_a f16x3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
pass f16x3_sq_b f16x3 GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU
pass f16x3_tcc f16x3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass f16x3_fetch f16x3 FETCH_SIZE
pass f16x3_write f16x3 WRITE_SIZE
python3 tools/summarize_pmc.py $OUT > $OUT/summary.json
cat $OUT/summary.json
