#!/usr/bin/env python3
"""Generates fp32_asm.inc.h: the hand-scheduled blocks of the EXACT-fp32 forward kernel (mlp_fwd.hip, mlp_fwd_exact_asm_kernel).

Why blocks: the compiled exact kernel keeps `h[8]` and `acc[8]` (256 registers) as C++ values, so the trunk feature that both heads
consume (128 more) cannot stay on the CU -- the compiler parks it in scratch / a global slab (141 GB of cache traffic per 480x640x128
image) and spills on top.  Here the two accumulator sets are pinned to the AGPR file and touched ONLY by these blocks (the build
checks that: check_asm_ownership.py), so the compiled code around them owns the 256 VGPRs minus the blocks' fixed ones and can
hold the feature as an ordinary value (v[32:159], 128 registers) from the end of the trunk to the start of the colour head.

Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation in k order: bit-identical to the compiled kernel and
to an fmaf chain), weights = A operand from the LDS chunk (mlp_layout.h: float index ((kstep*2 + t/4)*64 + lane)*4 + t%4), the
previous layer's activations = B operand read STRAIGHT FROM THE OTHER ACCUMULATOR SET (gfx90a+: srcB may be an AGPR): k-step s of
chunk ts contracts register s of tile ts.  bias + ReLU rewrite a set in place (v_accvgpr_read, v_add_f32, v_max_f32,
v_accvgpr_write): the sum is formed first and the bias added last, as the compiled kernel does.

Register map (fixed; the compiled code must stay off these while a value is live in them -- they are clobbers of every block):
  a[0:127]     set P            a[128:255]   set Q
  v[200:223]   A-fragment ring: 3 k-steps x 8 tiles (a slot is refilled two k-steps = 16 MFMAs after its last use)
  v[224:226]   B ring of the extra-input k-steps (per-lane values staged in LDS)
  v[200:231]   bias + ReLU pass: 16 biases, 16 values
  v[232:247]   head accumulator (C / D of the 1..5-row heads: a VGPR tile, so both sets stay intact)
Operands: %[a] VGPR = LDS byte address of this lane's first A fragment of the chunk (chunk buffer + lane * 16);
          %[b] VGPR = LDS byte address of this lane's first staged extra input (k-step stride 1024 B);
          %[bl] VGPR = LDS byte address of this layer's bias block for this lane half.
Hazards: an MFMA result is read by VALU only after 3 x s_nop 7 behind the last MFMA of the set; a VALU write of an MFMA operand is
followed by s_nop 1 before the MFMA (the assembler does not insert either inside inline asm).
"""
import sys

SET = {"P": 0, "Q": 128}
RING, BRING, HACC = 200, 224, 232


def mfma(dst_regs, a_reg, b_op, first):
    c = "0" if first else dst_regs
    return "v_mfma_f32_32x32x2_f32 %s, v%d, %s, %s" % (dst_regs, a_reg, b_op, c)


def a_loads(s, slot):
    """the two ds_read_b128 of k-step s into ring slot `slot`"""
    base = RING + 8 * slot
    return ["ds_read_b128 v[%d:%d], %%[a] offset:%d" % (base + 4 * g, base + 4 * g + 3, (2 * s + g) * 1024) for g in range(2)]


def gen_wide(dst, n_steps, b_of, zero, b_load=None):
    """n_steps k-steps x 8 tiles into set `dst`; b_of(s) = B operand text of k-step s; b_load(s) = optional LDS load of it.
    Loads of k-step s + 2 are issued before the MFMAs of k-step s; LDS returns in order, so the counted wait in front of k-step s
    leaves exactly the younger loads outstanding."""
    per = 2 + (1 if b_load else 0)
    out = []
    for s in range(min(2, n_steps)):
        out += a_loads(s, s % 3) + ([b_load(s)] if b_load else [])
    for s in range(n_steps):
        if s + 2 < n_steps:
            out += a_loads(s + 2, (s + 2) % 3) + ([b_load(s + 2)] if b_load else [])
        younger = per * (min(n_steps, s + 3) - (s + 1))
        out.append("s_waitcnt lgkmcnt(%d)" % younger)
        for t in range(8):
            regs = "a[%d:%d]" % (dst + 16 * t, dst + 16 * t + 15)
            out.append(mfma(regs, RING + 8 * (s % 3) + t, b_of(s), zero and s == 0))
    return out


def gen_gen(src, dst, ts):
    return gen_wide(dst, 16, lambda s: "a%d" % (src + 16 * ts + s), zero=(ts == 0))


def gen_extra(dst, n_steps, zero):
    return gen_wide(dst, n_steps, lambda s: "v%d" % (BRING + s % 3), zero,
                    b_load=lambda s: "ds_read_b32 v%d, %%[b] offset:%d" % (BRING + s % 3, s * 1024))


def gen_act(base, rec=False, mask=False):
    """rec (training): the block also writes the layer's activation record -- the post-ReLU values it holds in v[216:231], tile
    by tile -- in the layout of mlp_layout.h ("Training record": [256 features][32 samples] per wave, 16-byte sample quads
    XOR-swizzled with (f >> 1) & 7).  Register r of a tile is feature (r & 3) + 8 (r >> 2) + 4 h of the tile's 32: the swizzle
    term takes four values over a tile's 16 registers, so four per-lane byte offsets %[o0..o3] (index ((r & 3) >> 1) + 2 ((r >> 2) & 1))
    plus an immediate ((r & 3) + 8 (r >> 2)) * 128 address every store; the tile base s[96:97] advances by 4 KB (fixed SGPRs, clobbers:
    s[96:97] tile base, s[94:95] live mask, s[92:93] saved EXEC).  Only live lanes
    store (EXEC = %[live] != 0 around the stores; the record's padding must stay as the caller zeroed it).
    mask: also the ReLU sign words the data-gradient kernel reads -- bit b of word w <-> tile 2 w + b / 16, register b % 16:
    pushed MSB-first with v_cmp / v_addc, reversed, stored at %[mkoff] + 256 w relative to the record block."""
    out = ["s_nop 7", "s_nop 7", "s_nop 7"]
    if rec:
        out += ["v_cmp_ne_u32_e64 s[94:95], 0, %[live]", "s_mov_b64 s[96:97], %[rbase]"]
    for t in range(8):
        out += ["ds_read_b128 v[%d:%d], %%[bl] offset:%d" % (200 + 4 * g, 203 + 4 * g, (t * 16 + g * 4) * 4) for g in range(4)]
        out += ["v_accvgpr_read_b32 v%d, a%d" % (216 + r, base + 16 * t + r) for r in range(16)]
        out.append("s_waitcnt lgkmcnt(0)")
        out += ["v_add_f32 v%d, v%d, v%d" % (216 + r, 216 + r, 200 + r) for r in range(16)]
        out += ["v_max_f32 v%d, 0, v%d" % (216 + r, 216 + r) for r in range(16)]
        out += ["v_accvgpr_write_b32 a%d, v%d" % (base + 16 * t + r, 216 + r) for r in range(16)]
        if rec:
            if mask:
                for r in range(16):
                    out += ["v_cmp_lt_f32_e32 vcc, 0, v%d" % (216 + r), "v_addc_co_u32_e32 %[mk], vcc, %[mk], %[mk], vcc"]
                if t & 1:
                    out.append("v_bfrev_b32_e32 %[mk], %[mk]")
            out.append("s_and_saveexec_b64 s[92:93], s[94:95]")
            for r in range(16):
                idx = ((r & 3) >> 1) + 2 * ((r >> 2) & 1)
                out.append("global_store_dword %%[o%d], v%d, s[96:97] offset:%d" % (idx, 216 + r, ((r & 3) + 8 * (r >> 2)) * 128))
            if mask and (t & 1):
                out.append("global_store_dword %%[mkoff], %%[mk], %%[rbase] offset:%d" % ((t >> 1) * 256))
            out.append("s_mov_b64 exec, s[92:93]")
            if t < 7:
                out += ["s_add_u32 s96, s96, 0x1000", "s_addc_u32 s97, s97, 0"]
    return out + ["s_nop 1"]


def gen_head(src):
    """1..5-row head: 128 k-steps over all of set `src`, one accumulator tile in v[232:247]; A: one ds_read_b128 per 4 k-steps
    (heads: float index ((kstep/4)*64 + lane)*4 + kstep%4), ring of 3 slots x 4 registers."""
    out = []
    hreg = "v[%d:%d]" % (HACC, HACC + 15)

    def load(q):
        return "ds_read_b128 v[%d:%d], %%[a] offset:%d" % (RING + 4 * (q % 3), RING + 4 * (q % 3) + 3, q * 1024)
    out += [load(0), load(1)]
    for q in range(32):
        if q + 2 < 32:
            out.append(load(q + 2))
        out.append("s_waitcnt lgkmcnt(%d)" % (min(32, q + 3) - (q + 1)))
        for j in range(4):
            s = 4 * q + j
            out.append(mfma(hreg, RING + 4 * (q % 3) + j, "a%d" % (src + s), s == 0))
    return out + ["s_nop 7", "s_nop 7", "s_nop 7"]


SF = 32


def gen_stash(src):
    return ["s_nop 7", "s_nop 7", "s_nop 7"] + ["v_accvgpr_read_b32 v%d, a%d" % (SF + k, src + k) for k in range(128)]


def gen_restore(dst):
    return ["v_accvgpr_write_b32 a%d, v%d" % (dst + k, SF + k) for k in range(128)] + ["s_nop 3"]


def emit_macro(out, name, comment, lines):
    out.append("// %s  (%d instructions)" % (comment, len(lines)))
    out.append("#define %s \\" % name)
    for i, l in enumerate(lines):
        out.append('  "%s\\n\\t"%s' % (l, " \\" if i + 1 < len(lines) else ""))
    out.append("")


def main(path):
    out = ["// GENERATED by gen_fp32_asm.py -- do not edit.  Blocks of the exact-fp32 forward (see the generator's docstring).", "#pragma once", ""]
    for src, dst in (("P", "Q"), ("Q", "P")):
        for ts in range(8):
            emit_macro(out, "TP32_GEN_%s%s_%d" % (src, dst, ts), "chunk %d of a 256 -> 256 part, B = set %s tile %d, into set %s" % (ts, src, ts, dst),
                       gen_gen(SET[src], SET[dst], ts))
    for dst in ("P", "Q"):
        emit_macro(out, "TP32_EXTRA16Z_%s" % dst, "16 extra-input k-steps into set %s, first k-step with C = 0" % dst, gen_extra(SET[dst], 16, True))
        emit_macro(out, "TP32_EXTRA16_%s" % dst, "16 extra-input k-steps into set %s" % dst, gen_extra(SET[dst], 16, False))
        emit_macro(out, "TP32_EXTRA8_%s" % dst, "8 extra-input k-steps into set %s" % dst, gen_extra(SET[dst], 8, False))
        emit_macro(out, "TP32_ACT_%s" % dst, "set %s <- max(set + bias, 0) in place" % dst, gen_act(SET[dst]))
        emit_macro(out, "TP32_ACT_%s_REC" % dst, "set %s <- max(set + bias, 0) in place, recording the values" % dst, gen_act(SET[dst], rec=True))
        emit_macro(out, "TP32_ACT_%s_RECM" % dst, "set %s <- max(set + bias, 0) in place, recording values + ReLU sign words" % dst,
                   gen_act(SET[dst], rec=True, mask=True))
        emit_macro(out, "TP32_HEAD_%s" % dst, "1..5-row head over set %s -> v[232:247]" % dst, gen_head(SET[dst]))
    emit_macro(out, "TP32_STASH_Q", "set Q -> v[32:159]", gen_stash(SET["Q"]))
    emit_macro(out, "TP32_RESTORE_P", "v[32:159] -> set P", gen_restore(SET["P"]))
    out.append('#define TP32_RING_CLOBBERS "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", \\')
    out.append('  "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226"')
    out.append('#define TP32_ACT_CLOBBERS TP32_RING_CLOBBERS, "v227", "v228", "v229", "v230", "v231"')
    out.append('#define TP32_REC_CLOBBERS "s92", "s93", "s94", "s95", "s96", "s97", "vcc"')
    out.append('#define TP32_ALL_AGPRS "a0", "a255"')
    open(path, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "fp32_asm.inc.h")
