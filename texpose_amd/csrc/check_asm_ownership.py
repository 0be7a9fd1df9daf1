#!/usr/bin/env python3
"""Build-time check for mlp_fwd_f16x3.hip: the accumulator sets of the forward kernel live in a[0:255] and are touched
ONLY by the inline-asm blocks of wide_asm.inc.h (likewise the exact-fp32 block kernel of mlp_fwd.hip and fp32_asm.inc.h).  The compiler does not know that, so compiled code of that kernel must
never read or write an AGPR (it would otherwise use them as spill space).  Usage: check_asm_ownership.py file.s [file.s ...]"""
import re
import sys

cur, inasm, bad, stats = None, False, [], {}
text = "\n".join(open(f).read() for f in sys.argv[1:])
for i, l in enumerate(text.split("\n")):
    m = re.match(r"^(_ZN\S+):", l)
    if m:
        cur = m.group(1)
        stats[cur] = [0, 0]
    if cur is None:
        continue
    if "ASMSTART" in l:
        inasm = True
        continue
    if "ASMEND" in l:
        inasm = False
        continue
    if l.startswith(".Lfunc_end"):
        cur = None
        continue
    s = l.split(";")[0]
    if not s.strip():
        continue
    if "scratch_" in s:
        stats[cur][0] += 1
    if inasm:
        stats[cur][1] += 1
    elif (("mlp_fwd_f16x3_kernel" in cur or "mlp_dgrad_f16x3_asm_kernel" in cur or "mlp_fwd_exact_asm_kernel" in cur)
          and re.search(r"\ba\[?\d", s)):
        bad.append("%s:%d: %s" % (cur[:48], i + 1, l.strip()))
for k, (sc, na) in stats.items():
    print("%-70s scratch instructions %4d, inline-asm instructions %5d" % (k[:70], sc, na))
if bad:
    print("COMPILED CODE TOUCHES AGPRs (%d places):" % len(bad))
    print("\n".join(bad[:20]))
    sys.exit(1)
print("ok: compiled code of the forward kernels and the asm data-gradient kernel never touches an AGPR")
