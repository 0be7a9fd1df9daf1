"""GPU box: rows of a rocprofv3 --kernel-trace --stats directory whose kernel name contains any of the given substrings
(tools/kstats.py <dir> [substr ...]): average / minimum duration and call count."""
import csv, glob, sys

rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows:
    if len(sys.argv) < 3 or any(s in r["Name"] for s in sys.argv[2:]):
        print("%9.1f us avg  %9.1f us min  x %5s  %s" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Calls"], r["Name"][:90]))
