"""Collapse the rocprofv3 output of tools/profile_mlp.sh into one JSON: per pass, the counters of the LAST mlp_fwd
dispatch (the second image: caches warm) and its duration; plus the kernel-stats table of the bench run."""
import csv, glob, json, os, sys

out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)
    if name == "stats":
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            rows = list(csv.DictReader(open(f)))
            res["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage") if k in r}
                                   for r in rows[:12]]
        # the stats row of the forward kernel mixes launch sizes (full images, the masked-evaluation leg, ...): average the FULL-IMAGE
        # launches (largest grid / workgroup count of the trace) separately -- that figure is the one to hold against the bench's kernel_ms
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "mlp_fwd_f16x3_kernel<false" in r.get("Kernel_Name", "")]
            if rows:
                work = lambda r: int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
                full = [r for r in rows if work(r) == max(work(x) for x in rows)]
                dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in full]
                # (a persistent grid has the same size for every large launch: keep the slowest class by duration as well)
                big = [x for x in dur if x > 0.5 * max(dur)]
                res["kernel_trace_full_image_launches"] = {"calls": len(big), "avg_ms": sum(big) / len(big), "min_ms": min(big), "max_ms": max(big)}
        continue
    cnt, disp = {}, None
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        if not rows:
            continue
        main = [r for r in rows if "mlp_fwd" in r.get("Kernel_Name", "mlp_fwd")]
        disp = max(int(r["Dispatch_Id"]) for r in main)
        for r in main:
            if int(r["Dispatch_Id"]) == disp:
                cnt[r["Counter_Name"]] = cnt.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                if "Start_Timestamp" in r and "End_Timestamp" in r:
                    cnt["kernel_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        # the ray-bias pre-kernels of the same image (rb_image_bias / rb_ray_bias: the last dispatch of each before the main kernel's)
        for kname in sorted({r["Kernel_Name"] for r in rows if "rb_" in r.get("Kernel_Name", "")}):
            last = max(int(r["Dispatch_Id"]) for r in rows if r["Kernel_Name"] == kname and int(r["Dispatch_Id"]) < disp)
            short = "pre." + ("rb_image_bias" if "image" in kname else "rb_ray_bias")
            for r in rows:
                if int(r["Dispatch_Id"]) == last:
                    key = short + "." + r["Counter_Name"]
                    cnt[key] = cnt.get(key, 0.0) + float(r["Counter_Value"])
    res[name] = cnt
print(json.dumps(res, indent=1))
