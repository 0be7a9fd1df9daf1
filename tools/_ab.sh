cd /tmp && export TMPDIR=/tmp
for n in base cur base cur; do
  lib=$GRAFT_REPO_ROOT/texpose_amd/libtexpose_amd_$n.so; [ "$n" = "cur" ] && lib=$GRAFT_REPO_ROOT/texpose_amd/libtexpose_amd.so
  rm -rf /tmp/p_$n
  TEXPOSE_AMD_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$n -o t -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py 32 0 30 0 f16x3 > /dev/null 2>&1
  echo "== $n"; python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/p_$n fwd_f16x3
done
cd $GRAFT_REPO_ROOT; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "bwd or backward or grad or train or split or record" 2>&1 | tail -2
for n in base cur base cur; do
  lib=$GRAFT_REPO_ROOT/texpose_amd/libtexpose_amd_$n.so; [ "$n" = "cur" ] && lib=$GRAFT_REPO_ROOT/texpose_amd/libtexpose_amd.so
  TEXPOSE_AMD_LIB=$lib python3 tools/train_bench.py 4 0 200 1 f16x3 2>&1 | tail -1 | cut -c1-80
done
