#!/bin/bash
# GPU box: the round's evidence from ONE box and call.  Usage: tools/collect_round6.sh [tag]   (writes gpurun_out/<tag>/)
T=${1:-r6}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
O=gpurun_out/$T
( time timeout 1500 python bench.py > $O/bench_full.json 2> $O/bench_full.err ) 2> $O/bench_wall.txt
# the same command under rocprofv3 --kernel-trace --stats (kernel averages must agree with the line's HIP-event figures)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
f=$(find $O/prof/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
# fabric traffic of the two MLP forward kernels (one counter per pass; KiB, FETCH_SIZE x2 on gfx950) + issue-side counters of the f16x3 one
pass() { local name=$1 prec=$2; shift 2
  rocprofv3 --pmc "$@" --kernel-include-regex "mlp_fwd|rb_" --output-format csv -d $O/prof/$name -- python3 tools/render_once.py $prec 2 > $O/pmc_$name.log 2>&1; }
pass f16x3_sq_a f16x3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
pass f16x3_sq_b f16x3 GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU
pass f16x3_fetch f16x3 FETCH_SIZE
pass f16x3_write f16x3 WRITE_SIZE
pass fp32_fetch fp32 FETCH_SIZE
pass fp32_write fp32 WRITE_SIZE
python3 tools/summarize_pmc.py $O/prof > $O/mlp_pmc_summary.json 2> $O/mlp_pmc_summary.err
python3 - $O/prof > $O/mlp_traffic.json <<'PY'
import collections, csv, glob, json, sys
out = {}
for prec in ("f16x3", "fp32"):
    tot = {}
    for counter, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
        per = collections.defaultdict(float); name = {}
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (sys.argv[1], prec, counter.split("_")[0].lower()), recursive=True):
            for r in csv.DictReader(open(f)):
                if counter in r["Counter_Name"]:
                    per[r["Dispatch_Id"]] += float(r["Counter_Value"]) * scale; name[r["Dispatch_Id"]] = r["Kernel_Name"]
        # the SECOND image's launches: main kernel + (f16x3) its two pre-kernels
        by = collections.defaultdict(list)
        for d in sorted(per, key=int):
            by[name[d].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]].append(per[d])
        tot[counter] = {k: v[-1] for k, v in by.items()}
    out[prec] = dict(fetch_bytes_per_launch=sum(tot["FETCH_SIZE"].values()), write_bytes_per_launch=sum(tot["WRITE_SIZE"].values()), kernels=tot)
print(json.dumps(out, indent=1))
PY
rm -rf $O/prof
bash tools/pmc_train.sh $T > /dev/null 2>&1; cp gpurun_out/pmc_train_$T.json $O/train_b32_pmc.json
# the HBM-bound kernels alone: HIP-event GB/s + fabric bytes per launch (composite forward / backward, ray-gen, patch gather)
bash tools/pmc_hbm.sh $T > /dev/null 2>&1; cp gpurun_out/hbm_$T/pmc.json $O/hbm_kernels_pmc.json 2>/dev/null; cp gpurun_out/hbm_$T/line.json $O/hbm_kernels_events.json 2>/dev/null; rm -f gpurun_out/hbm_$T/*_counter_collection.csv
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
# the training iteration: lines, launch counts, timeline, knobs switched off one at a time, the several-rank form, soak
for i in 1 2 3; do python3 tools/train_bench.py 4 1 400 1 f16x3 2>/dev/null | tail -1 | cut -c1-70; done > $O/train_lines.txt
python3 tools/launch_counts.py > $O/launch_counts.txt 2>&1
python3 tools/linear_timeline.py 50 > $O/linear_timeline_strict.txt 2>&1
( for e in TP_X=1 TP_WGRAD_ALL_CUS=1 TP_X=1 TP_NO_SN_SPLIT=1 TP_X=1 "TP_WGRAD_ALL_CUS=1 TP_NO_SN_SPLIT=1" TP_X=1 TP_NO_DEFER=1 TP_X=1 TP_NO_PIPELINE_DISC=1 TP_X=1 "TP_NO_DEFER=1 TP_NO_PIPELINE_DISC=1" TP_X=1 TP_NO_DISC_STEP_TAIL=1 TP_X=1 TP_NO_GEN_SCHEDULE=1 TP_X=1 TP_NO_FEAT_CHAIN=1 TP_X=1 TP_NO_DISC_PAIRS=1 TP_X=1 TP_LINEAR_GRAPHS=0 TP_X=1; do
    echo "$e $(env $e python3 tools/train_bench.py 4 1 300 1 f16x3 2>/dev/null | tail -1 | cut -c1-60)"; done ) > $O/gan_ablations.txt
python3 - > $O/c4_form.txt 2>&1 <<'PY'
import sys, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import bench, train_dp
dev = torch.device("cuda", 0)
for rep in range(2):
    r = train_dp.measure(dev, 0, 1, global_batch=4, iters=300, warm=10)
    print("one rank   ", json.dumps({k: r[k] for k in ("value", "form", "launches", "launch_counts")}), flush=True)
    print("c4 form    ", json.dumps(bench.c4_form_leg(dev, iters=300)), flush=True)
PY
( python3 tools/soak_train.py 20000 2>&1 | tail -9; python3 tools/soak_train.py 3000 2>&1 | tail -3; python3 tools/soak_train.py 3000 2>&1 | tail -3; TP_SOAK_STRICT=1 python3 tools/soak_train.py 3000 2>&1 | tail -3 ) > $O/soak.txt
python3 tools/host_time.py > $O/host_time.txt 2>&1
cat $O/train_lines.txt; cat $O/bench_wall.txt; tail -4 $O/soak.txt; tail -3 $O/c4_form.txt | cut -c1-200
