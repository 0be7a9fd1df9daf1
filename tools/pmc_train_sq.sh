#!/bin/bash
# GPU box: issue-side counters of the three training MLP kernels at B=32 (MFMA busy, LDS activity / bank conflicts, wait cycles, clock):
# what each of them is bound by besides its bytes (tools/pmc_train.sh has those).   Usage: tools/pmc_train_sq.sh <tag>
#   -> gpurun_out/pmc_train_sq_<tag>.json     (one rocprofv3 --pmc pass per counter group; never --pmc together with a trace)
TAG=${1:-run}
OUT=gpurun_out/pmctsq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
pass() { local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-include-regex "mlp_(fwd_f16x3|dgrad|wgrad_f16x3)" --output-format csv -d $OUT/$name -- python3 tools/train_bench.py 32 0 3 0 f16x3 > $OUT/$name.log 2>&1; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY
pass b GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
python3 - "$OUT" > gpurun_out/pmc_train_sq_$TAG.json <<'PY'
import collections, csv, glob, json, sys
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "recording_forward" if "mlp_fwd_f16x3" in n else ("dgrad" if "dgrad" in n else ("wgrad" if "wgrad_f16x3" in n else None))
        if k:
            res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
for k, d in out.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE"] / 8)
    if "SQ_LDS_BANK_CONFLICT" in d and "SQ_ACTIVE_INST_LDS" in d:
        d["lds_conflict_per_active"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_ACTIVE_INST_LDS"]
print(json.dumps(out, indent=1))
PY
cat gpurun_out/pmc_train_sq_$TAG.json
