"""Collapse the two rocprofv3 --pmc passes of tools/pmc_hbm.sh into per-kernel fabric bytes per launch.
Dispatches are grouped by (kernel, grid size); a group maps to a `roofline_hbm` entry of bench.py: composite fwd / bwd and
ray-gen = the group with most launches (the timed loop), the gather's two sizes by grid.  FETCH_SIZE / WRITE_SIZE are KiB;
FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM)."""
import collections, csv, glob, json, sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (root, counter), recursive=True) + glob.glob("%s/%s_counter_collection.csv" % (root, counter)):
        per_dispatch = collections.defaultdict(float)
        meta = {}
        for r in csv.DictReader(open(f)):
            if counter not in r["Counter_Name"]:
                continue
            per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
            meta[r["Dispatch_Id"]] = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], int(r["Grid_Size"]))
        for d, v in per_dispatch.items():
            acc[meta[d]][counter].append(v)


def pick(sub, rule):
    groups = [(k, v) for k, v in acc.items() if sub in k[0]]
    if not groups:
        return None
    if rule == "most":
        k, v = max(groups, key=lambda kv: len(kv[1].get("WRITE_SIZE", [])))
    elif rule == "small":
        k, v = min(groups, key=lambda kv: kv[0][1])
    else:
        k, v = max(groups, key=lambda kv: kv[0][1])
    f = v.get("FETCH_SIZE", [0.0])
    w = v.get("WRITE_SIZE", [0.0])
    fetch, write = 2 * 1024 * sum(f) / len(f), 1024 * sum(w) / len(w)
    return dict(kernel=k[0], grid=k[1], launches_averaged=len(w), fetch_bytes_per_launch=fetch, write_bytes_per_launch=write,
                total=fetch + write)


out = dict(composite_fwd=pick("composite_fwd", "most"), composite_bwd=pick("composite_bwd", "most"),
           raygen=pick("raygen", "most"), patch_gather=pick("patch_gather", "small"),
           patch_gather_b32_p64=pick("patch_gather", "large"))
print(json.dumps(out, indent=1))
