for r in 1 2 3; do
for v in "8k:" "16k:TEXPOSE_AMD_LIB=$PWD/texpose_amd/libtexpose_amd_reduce16k.so" "8k_nochain:TP_NO_FEAT_CHAIN=1"; do
  name=${v%%:*}; e=${v#*:}
  out=$(env $e python tools/train_bench.py 4 1 300 1 f16x3 2>/dev/null | tail -1)
  echo "$name $(echo $out | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.1f us  %.0f it/s" % (d["ms_per_iter"]*1e3, d["iters_per_s"]))')"
done; done
echo "--- target wgs"
for r in 1 2; do
for v in "wgs256:" "wgs128:TP_CONV_TARGET_WGS=128" "wgs64:TP_CONV_TARGET_WGS=64" "wgs192:TP_CONV_TARGET_WGS=192"; do
  name=${v%%:*}; e=${v#*:}
  out=$(env $e python tools/train_bench.py 4 1 300 1 f16x3 2>/dev/null | tail -1)
  echo "$name $(echo $out | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.1f us  %.0f it/s" % (d["ms_per_iter"]*1e3, d["iters_per_s"]))')"
done; done
