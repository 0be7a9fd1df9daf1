#!/usr/bin/env python3
"""GPU box: A/B timing of several builds of libtexpose_amd.so ON THE SAME DEVICE (devices of the pool differ by several
percent, so numbers from different calls do not compare).  Usage: tools/ab_bench.py libA.so libB.so ... [--rounds 2]
Each build renders a few full 480x640x128 images in its own process (tools/render_once.py), interleaved A B A B."""
import os
import re
import subprocess
import sys

rounds = 2
argv = sys.argv[1:]
if "--rounds" in argv:
    i = argv.index("--rounds")
    rounds = int(argv[i + 1])
    del argv[i:i + 2]
libs = [a for a in argv if not a.startswith("--")]
here = os.path.dirname(os.path.abspath(__file__))
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, TEXPOSE_AMD_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(here, "render_once.py"), "f16x3", "5"], env=env, capture_output=True, text=True)
        ms = [float(x) for x in re.findall(r"image ms ([0-9.]+)", out.stdout)]
        if not ms:
            print(l, "FAILED", out.stderr[-400:])
            continue
        res[l].append(min(ms[1:]))
        for line in out.stdout.splitlines():
            if line.startswith("trace total"):
                print(os.path.basename(l), line)
for l in libs:
    if res[l]:
        print("%-40s best %.2f ms  runs %s" % (os.path.basename(l), min(res[l]), ["%.2f" % v for v in res[l]]))
