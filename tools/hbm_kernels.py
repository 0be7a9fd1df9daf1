"""GPU box: launch the HBM-bound kernels around the MLP (composite fwd / bwd, ray-gen, patch gather) at the sizes of
bench.py's `roofline_hbm` object -- the profiling target of tools/pmc_hbm.sh (rocprofv3 --kernel-trace --stats and one --pmc
pass each for FETCH_SIZE / WRITE_SIZE).  Prints the same object bench.py puts in its line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

print(json.dumps(bench.hbm_rooflines(torch.device("cuda:0"), {})))
