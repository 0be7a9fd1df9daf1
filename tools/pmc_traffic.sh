export TMPDIR=/tmp
python tools/quick_precision_bench.py 2>&1 | grep "f16x3 ms"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --pmc $c --kernel-include-regex mlp_fwd_f16x3 --output-format csv -d /tmp/pm_$$ -- python3 tools/render_once.py f16x3 2 > /tmp/pm.log 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pm_$$/**/*counter_collection.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
d=max(int(r['Dispatch_Id']) for r in rows)
c=collections.Counter()
for r in rows:
    if int(r['Dispatch_Id'])==d: c[r['Counter_Name']]+=float(r['Counter_Value'])
print(dict(c))
PY
  rm -rf /tmp/pm_$$
done
