#!/usr/bin/env python3
"""Update an entry of profiles/traffic.json after a PMC collection and stamp it with the sha of the kernel sources it was
measured on (bench.TRAFFIC_SOURCES); bench.py reports "traffic_stale": true for an entry whose sources changed since.

    tools/traffic_stamp.py f16x3 --fetch 3.0e8 --write 1.76e9 --source "profiles/r4/NN_... (tools/pmc_quick.sh)"
    tools/traffic_stamp.py hbm_kernels.raygen --fetch ... --write ...
    tools/traffic_stamp.py train_b32 --total 1.58e10 --source ...
    tools/traffic_stamp.py --check                      list every entry with its stale flag
The figures must come from THIS tree's build (run the PMC passes first, then stamp in the same commit)."""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                                                                                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("entry", nargs="?")
ap.add_argument("--fetch", type=float)
ap.add_argument("--write", type=float)
ap.add_argument("--total", type=float)
ap.add_argument("--source")
ap.add_argument("--check", action="store_true")
a = ap.parse_args()
path = os.path.join(REPO, "profiles", "traffic.json")
t = json.load(open(path))
if a.check or not a.entry:
    for e in bench.TRAFFIC_SOURCES:
        node, stale = bench.traffic_entry(e)
        print("%-36s %s" % (e, "missing" if node is None else ("no sha" if stale is None else ("STALE" if stale else "current"))))
    raise SystemExit(0)
node = t
for part in a.entry.split("."):
    node = node.setdefault(part, {})
if a.fetch is not None:
    node["fetch_bytes_per_launch"] = a.fetch
if a.write is not None:
    node["write_bytes_per_launch"] = a.write
if a.total is not None:
    node["total_per_step" if a.entry == "train_b32" else "total"] = a.total
elif a.fetch is not None and a.write is not None:
    node["total"] = a.fetch + a.write
if a.source:
    node["source"] = a.source
node["sources_sha16"] = bench.sources_sha16(a.entry)
node["sources"] = [os.path.join(bench.CSRC, n) for n in bench.TRAFFIC_SOURCES[a.entry]]
node.pop("sources_note", None)
json.dump(t, open(path, "w"), indent=1)
print(a.entry, "stamped", node["sources_sha16"])
