#!/bin/bash
# GPU box: fabric traffic + L2 hits of the three training MLP kernels at B=32 (one rocprofv3 --pmc pass per counter group on
# tools/train_bench.py 32 0 3 0 f16x3; per-launch averages).  Usage: tools/pmc_train.sh <tag>  -> gpurun_out/pmc_train_<tag>.json
export TMPDIR=/tmp
TAG=${1:-run}
OUT=gpurun_out/pmct_$TAG
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-include-regex "wgrad_f16x3|wgrad_kernel|dgrad_f16x3|fwd_f16x3" --output-format csv -d $OUT/$n -- python3 tools/train_bench.py 32 0 3 0 f16x3 > $OUT/$n.log 2>&1
done
python3 - "$OUT" > gpurun_out/pmc_train_$TAG.json <<'PY'
import collections, csv, glob, json, sys
res = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = "recording_forward" if "fwd_f16x3" in r["Kernel_Name"] else "dgrad" if "dgrad" in r["Kernel_Name"] else "wgrad"
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        for c, x in v.items():
            res[k][c] = sum(x) / len(x)
            res[k]["launches_averaged"] = len(x)
for k, v in res.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["fetch_bytes"] = v["FETCH_SIZE"] * 1024 * 2          # KiB, x2 on gfx950 (MI355X_MICROARCH.md)
        v["write_bytes"] = v["WRITE_SIZE"] * 1024
print(json.dumps(res, indent=1))
PY
cat gpurun_out/pmc_train_$TAG.json
rm -rf $OUT
