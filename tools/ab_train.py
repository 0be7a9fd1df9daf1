#!/usr/bin/env python3
"""GPU box: A/B of several builds of libtexpose_amd.so on the training step, ON THE SAME DEVICE, interleaved (devices of the
pool differ by several percent).  Usage: tools/ab_train.py libA.so libB.so ... [--rounds 2]
Per build and round: tools/train_bench.py 32 0 60 1 f16x3 (nerf step, B=32, hipGraph) and 4 0 200 1 / 4 1 200 1 (B=4 nerf
step / full GAN loop) in their own processes (TEXPOSE_AMD_LIB selects the build; all builds must have the current C ABI)."""
import json
import os
import subprocess
import sys

rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 2
libs = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--rounds"]
here = os.path.dirname(os.path.abspath(__file__))
cases = (("nerf_b32", ["32", "0", "60", "1", "f16x3"]), ("nerf_b4", ["4", "0", "200", "1", "f16x3"]), ("gan_b4", ["4", "1", "200", "1", "f16x3"]))
res = {(l, c): [] for l in libs for c, _ in cases}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, TEXPOSE_AMD_LIB=os.path.abspath(l))
        for c, args in cases:
            out = subprocess.run([sys.executable, os.path.join(here, "train_bench.py")] + args, env=env, capture_output=True, text=True)
            try:
                res[(l, c)].append(json.loads(out.stdout.strip().splitlines()[-1])["ms_per_iter"])
            except Exception:
                print(l, c, "FAILED", out.stderr[-300:])
for l in libs:
    print("%-36s" % os.path.basename(l), "  ".join("%s best %.3f ms %s" % (c, min(res[(l, c)]), ["%.3f" % v for v in res[(l, c)]]) for c, _ in cases if res[(l, c)]))
