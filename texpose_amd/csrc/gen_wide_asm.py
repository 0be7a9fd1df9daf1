#!/usr/bin/env python3
"""Generates wide_asm.inc.h: every piece of the f16x3 MLP forward that touches the two accumulator sets, as
hand-scheduled gfx950 inline-asm blocks with FIXED registers:

    set P = a[0:127], set Q = a[128:255]   (8 tiles of 32 features x 32 samples, 16 registers each)
    v[160:191]  A-fragment ring: 4 (k-step, tile) pair slots x (hi 4 + lo 4 registers)
    v[192:223]  two banks of B operands (xh0 xh1 xl0 xl1, 4 registers each), also scratch of the seeding block
    v[224:227]  conversion temporaries      v[228:230]  LDS read address per ring slot
    v[232:247]  accumulator tile of a narrow head

Why asm: with one wave per SIMD every issue slot that is not an MFMA is exposed, and the kernel needs all 512
registers.  hipcc's schedule of the C++ version clusters the operand-conversion VALU work unevenly, waits for LDS with
lgkmcnt(0), moves the sets between AGPRs and VGPRs and cannot keep a third 128-register set (the trunk feature, held
from L7 to R0) next to them without spilling it.  Here every MFMA gap carries at most four VALU instructions, or one
LDS-DMA piece, or two ds_read_b128; a ring slot is refilled as soon as its last MFMA has issued; waits are counted.

Blocks (C++ wrappers in mlp_fwd_f16x3.hip), each for both set roles:
    WIDE   256 -> 256 layer body: 8 chunks, 384 MFMAs; B operands converted from the source set in the gaps
           (2^-8, ReLU, hi = v & 0xFFFFE000, lo = v - hi, two v_cvt_pkrtz, range guard)
    EXTRA  one chunk of "extra input" k-steps (positional encodings, latents): B operands read from the LDS stage
    HEAD   1..5-row output layer over relu(source set): one chunk, 16 k-steps, one accumulator tile in VGPRs
    INIT   seed a set with bias * 2^8 from LDS
Arithmetic and accumulation order are those of the C++ version (mma_wide16 / part_gen16 / part_head16): results are
bit-identical.

Ring protocol (shared by all blocks): three 32 KiB LDS slots; on entry chunk c is published and its pairs 0..3 are in
the fragment registers, the DMA of c+1 is fully issued, that of c+2 not started; a block issues the 8 DMA pieces of c+2
spread over its MFMA groups, publishes c+1 (counted vmcnt: only the pieces of c+2 issued so far may be in flight,
lgkmcnt(0), s_barrier) 4 pairs before the end of c and leaves with pairs 0..3 of c+1 in the fragment registers.

    python3 gen_wide_asm.py > wide_asm.inc.h        (the Makefile does this)
"""
import os
import sys

NCH = 115
# timing experiments only (results are WRONG): TP_ASM_EXPERIMENT = nodma | nobarrier | noconv | recnostore (recording
# blocks issue no vector stores) | recnolds (no staging-tile writes / reads either)
EXPERIMENT = os.environ.get("TP_ASM_EXPERIMENT", "")
# hi / lo split of a converted operand: "mix" (round 4) = v_cvt_pkrtz of the two values IS the truncation to 11 significand bits,
# and lo = v - float(hi) comes from v_fma_mix_f32 reading hi as fp16 (exact; also right when hi is an fp16 subnormal, where the
# masked fp32 value kept bits the fp16 conversion then dropped): 11 VALU instructions per two elements.  "and" = rounds 2-3:
# hi = v & 0xFFFFE000, lo = v - hi, two conversions: 13 (A/B build: make split_and).
SPLIT = os.environ.get("TP_ASM_SPLIT", "mix")

VB = 160
def F_hi(slot): return VB + 8 * slot
def F_lo(slot): return VB + 8 * slot + 4
XB = [VB + 32, VB + 48]
T = VB + 64
VR = VB + 68
HACC = VB + 72                # head accumulator tile v[232:247]
CLOBBER_V = list(range(XB[0], VR + 3))
BA, BB = 92, 94               # fixed SGPR pairs: global base of the DMA pieces 0..3 / 4..7
CLOBBER_S = [BA, BA + 1, BB, BB + 1]
SET = {"P": 0, "Q": 128}


def vr(lo, n=4):
    return "v[%d:%d]" % (lo, lo + n - 1)


def ar(set_base, tile):
    lo = set_base + 16 * tile
    return "a[%d:%d]" % (lo, lo + 15)


def mfma(dst, a, b, c=None):
    return "v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (dst, a, b, dst if c is None else c)


def conv(src_base, tile, g, bank):
    """11 (13 with SPLIT = "and") VALU instructions: elements 2g, 2g+1 of source tile `tile` -> packed word g of the hi / lo operands in `bank`"""
    xh = XB[bank] + (g >> 2) * 4 + (g & 3)
    xl = XB[bank] + 8 + (g >> 2) * 4 + (g & 3)
    t0, t1, h0, h1 = T, T + 1, T + 2, T + 3
    a = src_base + 16 * tile + 2 * g
    head = ["v_accvgpr_read_b32 v%d, a%d" % (t0, a),
            "v_accvgpr_read_b32 v%d, a%d" % (t1, a + 1),
            "v_mul_f32 v%d, %%[kinv], v%d" % (t0, t0),
            "v_mul_f32 v%d, %%[kinv], v%d" % (t1, t1),
            "v_max_f32 v%d, 0, v%d" % (t0, t0),
            "v_max_f32 v%d, 0, v%d" % (t1, t1)]
    return head + split_ops(t0, t1, h0, h1, xh, xl) + ["v_pk_max_f16 %%[amax], %%[amax], v%d" % xh]


def split_ops(t0, t1, h0, h1, xh, xl):
    """(t0, t1) fp32 -> packed fp16 hi (xh) and lo (xl) words; t0 / t1 keep their values in the "mix" form"""
    if SPLIT == "and":
        return ["v_and_b32 v%d, %%[mask], v%d" % (h0, t0),
                "v_and_b32 v%d, %%[mask], v%d" % (h1, t1),
                "v_sub_f32 v%d, v%d, v%d" % (t0, t0, h0),
                "v_sub_f32 v%d, v%d, v%d" % (t1, t1, h1),
                "v_cvt_pkrtz_f16_f32 v%d, v%d, v%d" % (xh, h0, h1),
                "v_cvt_pkrtz_f16_f32 v%d, v%d, v%d" % (xl, t0, t1)]
    return ["v_cvt_pkrtz_f16_f32 v%d, v%d, v%d" % (xh, t0, t1),
            "v_fma_mix_f32 v%d, -v%d, 1.0, v%d op_sel_hi:[1,0,0]" % (h0, xh, t0),
            "v_fma_mix_f32 v%d, -v%d, 1.0, v%d op_sel:[1,0,0] op_sel_hi:[1,0,0]" % (h1, xh, t1),
            "v_cvt_pkrtz_f16_f32 v%d, v%d, v%d" % (xl, h0, h1)]


# ------------------------------------------------------------------------------------------------ recording variants
# Training (mlp_fwd_f16x3_kernel<true>): the activation record of a layer is written by the block that CONSUMES that layer's
# accumulators -- its conversion already forms h = relu(acc * 2^-8) in t0 / t1 -- instead of a separate pass over the set
# after the layer (round 2: ~34 of 164 us per tile, compiled code reading the tiles back through v_accvgpr_read).  Per
# converted element: one ds_write_b32 into the wave's private 4 KB staging tile (features on rows, as the C++ pass did it)
# and v_cmp + v_addc (sign bit of the ReLU mask word); per 32 x 32 tile: four ds_read_b128 of the transposed tile and four
# nontemporal 16-byte stores one chunk / group later; per two tiles: one mask word.  All of it rides in MFMA issue gaps.
#   %[recw] / %[recr]  VGPR: LDS byte address of this lane's writes / 16-byte reads (mlp_fwd_f16x3.hip: rec_ctx)
#   %[rbase]           SGPR pair: the layer's record block of this wave's group
#   %[mkoff]           VGPR: byte offset (from %[rbase]) of this lane's first mask word
#   %[ro0..1]          VGPR: byte offset of read units 0 / 1 inside a 4 KB record tile (blk_off, XOR swizzle)
# Fixed registers: v231 mask accumulator, s[96:97] running tile base.
MK = VB + 71                  # v231
RB = 96                       # s[96:97]


def rec_wr_off(r):
    return (r >> 2) * 4096 + (r & 3) * 128


def conv_rec(src_base, tile, g, bank, mask=True):
    """conv() + record of the two elements, in issue order: 6 (reads, scale, ReLU: t0 / t1 = h), 2 LDS stores, 4 mask
    instructions (`mask`), 7 (hi / lo split, pack, range guard)"""
    c = conv(src_base, tile, g, bank)
    t0, t1 = T, T + 1
    r0, r1 = 2 * g, 2 * g + 1
    out = c[0:6]
    if EXPERIMENT != "recnolds":
        out += ["ds_write_b32 %%[recw], v%d offset:%d" % (t0, rec_wr_off(r0)),
                "ds_write_b32 %%[recw], v%d offset:%d" % (t1, rec_wr_off(r1))]
    else:
        out += ["s_nop 0", "s_nop 0"]
    if mask:
        out += ["v_cmp_lt_f32_e32 vcc, 0, v%d" % t0,
                "v_addc_co_u32_e32 v%d, vcc, v%d, v%d, vcc" % (MK, MK, MK),
                "v_cmp_lt_f32_e32 vcc, 0, v%d" % t1,
                "v_addc_co_u32_e32 v%d, vcc, v%d, v%d, vcc" % (MK, MK, MK)]
    out += c[6:]
    return out


def tile_regs(set_base, tile, k):
    lo = set_base + 16 * tile + 4 * k
    return "a[%d:%d]" % (lo, lo + 3)


def rec_reads(set_base, tile):
    """the staged tile back out of LDS, transposed, into the accumulator registers of the SOURCE tile itself: they are dead
    from the tile's conversion until it is re-seeded (WIDE: moved behind the stores) or for good (HEAD); LDS loads may
    target AGPRs and vector stores may take their data from them on gfx950 -- no VGPR is spent on the record"""
    if EXPERIMENT == "recnolds":
        return []
    return ["ds_read_b128 %s, %%[recr] offset:%d" % (tile_regs(set_base, tile, k), k * 4096) for k in range(4)]


def rec_store(set_base, tile, k):
    # units 2, 3 sit 16 rows (2 KB) behind units 0, 1 with the same swizzle: two offset registers, an immediate for the rest
    if EXPERIMENT in ("recnostore", "recnolds"):
        return "s_nop 0"
    policy = {"recplain": "", "recsc1": " sc1", "recsc01": " sc0 sc1"}.get(EXPERIMENT, " nt")
    return "global_store_dwordx4 %%[ro%d], %s, s[%d:%d] offset:%d%s" % (k & 1, tile_regs(set_base, tile, k), RB, RB + 1, (k >> 1) * 2048, policy)


def rec_advance():
    return ["s_add_u32 s%d, s%d, 0x1000" % (RB, RB), "s_addc_u32 s%d, s%d, 0" % (RB + 1, RB + 1)]


def rec_mask_store(w4):
    # bits were pushed MSB-first (tile 2 w4 register 0 first): reverse -> bit b <-> tile 2 w4 + b / 16, register b % 16
    if EXPERIMENT in ("recnostore", "recnolds"):
        return []
    return ["v_bfrev_b32_e32 v%d, v%d" % (MK, MK),
            "global_store_dword %%[mkoff], v%d, %%[rbase] offset:%d" % (MK, w4 * 256)]


def split_even(instrs, n):
    k, out, pos = len(instrs), [], 0
    for i in range(n):
        take = (k - pos + (n - i) - 1) // (n - i)
        out.append(instrs[pos:pos + take])
        pos += take
    return out


def ring_prologue(e):
    """slot byte offsets r0 r1 r2 = slots of (current, next, DMA target) chunk; read addresses; DMA LDS bases; dch."""
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 7")                                     # (sets / operands may have been written just before the block)
    e("s_nop 7")
    e("s_lshl_b32 %[r0], %[buf], 15")
    for k, r in ((1, "r1"), (2, "r2")):
        e("s_add_i32 %%[t0], %%[buf], %d" % k)
        e("s_cmp_ge_i32 %[t0], 3")
        e("s_cselect_b32 %%[%s], 3, 0" % r)
        e("s_sub_i32 %%[t0], %%[t0], %%[%s]" % r)
        e("s_lshl_b32 %%[%s], %%[t0], 15" % r)
    for i, r in enumerate(("r0", "r1", "r2")):
        e("v_add_u32 v%d, %%[%s], %%[lane16]" % (VR + i, r))
        e("s_add_i32 %%[m%da], %%[ldswave], %%[%s]" % (i, r))
    e("s_add_i32 %[dch], %[chunk], 2")
    e("s_cmp_ge_i32 %[dch], %[nch]")
    e("s_cselect_b32 %[t0], %[nch], 0")
    e("s_sub_i32 %[dch], %[dch], %[t0]")
    # (%[nch] = number of chunks of the stream the block walks, an SGPR input: 115 forward, 112 forward with per-ray biases, 34
    # transposed stream of the data gradient)


def dma_base():
    return ["s_lshl_b32 %[t0], %[dch], 15",
            "s_add_u32 s%d, %%[stream_lo], %%[t0]" % BA,
            "s_addc_u32 s%d, %%[stream_hi], 0" % (BA + 1),
            "s_add_u32 s%d, s%d, 0x1000" % (BB, BA),
            "s_addc_u32 s%d, s%d, 0" % (BB + 1, BA + 1)]


def dma_piece(e, k, dma_slot):
    if k == 0:
        e("s_mov_b32 m0, %%[m%da]" % dma_slot)
        e("s_nop 0")                                 # (M0 write -> LDS-DMA: one wait state)
    elif k == 4:
        e("s_add_i32 m0, %%[m%da], 0x1000" % dma_slot)
        e("s_nop 0")
    base = "s[%d:%d]" % ((BA, BA + 1) if k < 4 else (BB, BB + 1))
    if EXPERIMENT != "nodma":
        e("global_load_lds_dwordx4 %%[laneoff], %s offset:%d" % (base, (k & 3) * 1024))


def ring_epilogue(e, n_chunks):
    e("s_waitcnt lgkmcnt(0)")
    e("s_add_i32 %%[chunk], %%[chunk], %d" % n_chunks)
    e("s_cmp_ge_i32 %[chunk], %[nch]")
    e("s_cselect_b32 %[t0], %[nch], 0")
    e("s_sub_i32 %[chunk], %[chunk], %[t0]")
    e("s_add_i32 %%[buf], %%[buf], %d" % (n_chunks % 3))
    e("s_cmp_ge_i32 %[buf], 3")
    e("s_cselect_b32 %[t0], 3, 0")
    e("s_sub_i32 %[buf], %[buf], %[t0]")


def refill(e, slot, pair, npairs, cur, nxt):
    addr, off = (VR + cur, pair * 2048) if pair < npairs else (VR + nxt, (pair - npairs) * 2048)
    e("ds_read_b128 %s, v%d offset:%d" % (vr(F_hi(slot)), addr, off))
    e("ds_read_b128 %s, v%d offset:%d" % (vr(F_lo(slot)), addr, off + 1024))


def publish(e, young):
    e("s_waitcnt vmcnt(%d) lgkmcnt(0)" % (0 if EXPERIMENT == "nodma" else young))
    if EXPERIMENT != "nobarrier":
        e("s_barrier")


def chunk_groups(e, npairs, ts, mf, fill, free_after, pieces_per_group, tail=None, head=None, young_extra=0):
    """One chunk of npairs (k-step, tile) pairs = npairs/2 groups of six MFMAs.
      mf(g)          -> the six MFMA lines of group g (they use ring slots (2g)%4 and (2g+1)%4)
      fill(g)        -> five lists of gap instructions (after MFMA 1..5)
      free_after     -> (i, j), i < j: slot a is free after MFMA i, slot b after MFMA j (1-based)
      tail(g)        -> extra lines at the very end of group g;  head(g) -> extra LDS reads right after MFMA 1
    Before a group's first MFMA all LDS reads but the four youngest (the refills issued during the previous group) have
    landed: the refills of the group before that one are this group's fragments."""
    ngroups = npairs // 2
    cur, nxt, dma = ts % 3, (ts + 1) % 3, (ts + 2) % 3
    pub_group = (npairs - 4) // 2
    for g in range(ngroups):
        q = 2 * g
        sa, sb = q % 4, (q + 1) % 4
        lines = mf(g)
        gaps = fill(g)
        e("s_waitcnt lgkmcnt(4)")
        for i in range(6):
            e(lines[i])
            if i == 0:
                for k in range(pieces_per_group):
                    dma_piece(e, g * pieces_per_group + k, dma)
                if head is not None:
                    for ins in head(g):
                        e(ins)
            if i < 5:
                for ins in gaps[i]:
                    e(ins)
            if i + 1 == free_after[0]:
                if g == pub_group:
                    # young_extra: vector-memory operations OTHER than DMA pieces (record stores) issued in this chunk
                    # before this point; they retire in order with the pieces (MI355X_MICROARCH.md: loads, stores and
                    # LDS-DMA count together, in issue order), so the count must be exact: too small only waits longer,
                    # too large would let a piece of the chunk being published still be in flight
                    publish(e, (g + 1) * pieces_per_group + (0 if EXPERIMENT in ("recnostore", "recnolds") else young_extra))
                refill(e, sa, q + 4, npairs, cur, nxt)
            if i + 1 == free_after[1]:
                refill(e, sb, q + 5, npairs, cur, nxt)
        if tail is not None:
            for ins in tail(g):
                e(ins)


def advance_dch():
    return ["s_add_i32 %[dch], %[dch], 1", "s_cmp_eq_u32 %[dch], %[nch]", "s_cselect_b32 %[dch], 0, %[dch]"]


# ------------------------------------------------------------------------------------------------------ WIDE
def gen_wide(src, dst, rec=None):
    """rec: None (plain), "mask" (record + ReLU sign words) or "nomask" (record only: the trunk feature, whose gates the
    backward never needs)"""
    L = []
    e = L.append
    ring_prologue(e)
    cvf = (lambda *a: conv_rec(*a, mask=(rec == "mask"))) if rec else conv
    if rec:
        e("s_mov_b64 s[%d:%d], %%[rbase]" % (RB, RB + 1))
    for g in range(8):
        for ins in cvf(src, 0, g, 0):
            e(ins)
    if rec:
        for ins in rec_reads(src, 0):                # tile 0 (the first chunk's counted waits cover these four reads)
            e(ins)
    for ins in dma_base():
        e(ins)
    for ts in range(8):
        bank = ts & 1

        def mf(g, bank=bank):
            s, t = g >> 2, (2 * g) & 7
            sa, sb = (2 * g) % 4, (2 * g + 1) % 4
            xh, xl = vr(XB[bank] + s * 4), vr(XB[bank] + 8 + s * 4)
            d0, d1 = ar(dst, t), ar(dst, t + 1)
            return [mfma(d0, vr(F_hi(sa)), xh), mfma(d1, vr(F_hi(sb)), xh), mfma(d0, vr(F_hi(sa)), xl),
                    mfma(d1, vr(F_hi(sb)), xl), mfma(d0, vr(F_lo(sa)), xh), mfma(d1, vr(F_lo(sb)), xh)]

        def fill(g, ts=ts, bank=bank):
            if rec:
                # Record traffic of this chunk, all in MFMA issue gaps:
                #   groups 1..4  one 16-byte store each of tile ts (back in its own accumulator registers since the end of the
                #                previous chunk / the prologue; the counted wait at the top of group 1 covers those LDS reads)
                #   group 5      tile base += 4 KB
                #   group 6      tile ts is re-seeded with the next layer's bias (the plain block does that in group 0)
                #   group 7      tile ts + 1 (its last two values were written in this group's second gap) back out of the
                #                staging tile into the hold registers -- LDS operations of a wave execute in order, and the
                #                reads are issued BEFORE the group's ring refills, so the waits' counts are unchanged;
                #                even chunks: the mask word of tiles ts, ts + 1, complete with this group's last v_addc (it
                #                is issued after the publish point: the NEXT chunk counts it)
                # (placement of the four stores -- one per group, all in group 1, or behind the chunk's last DMA piece so that
                # the next publish does not wait for them -- made no difference on one device; nor did plain / sc1 instead of
                # nt stores, which were slower: profiles/r3.  What the stores cost is CLOCK, not issue cycles: 16 KB of HBM
                # writes per chunk and CU lower the clock of the MFMA loop by ~8 %, tools/ubench/store_cost.hip)
                first, mid1, mid2, mid3, last = [], [], [], [], []
                if 1 <= g <= 4:
                    first = [rec_store(src, ts, g - 1)]
                if g == 5:
                    first = rec_advance()
                if g == 7 and ts < 7:
                    rd = rec_reads(src, ts + 1)
                    mid2, mid3 = mid2 + rd[0:2], mid3 + rd[2:4]
                    if rec == "mask" and not (ts & 1):
                        last = last + rec_mask_store(ts >> 1)
                if ts == 7:
                    return [first, mid1, mid2, mid3, last]
                cv = conv_rec(src, ts + 1, g, bank ^ 1, mask=(rec == "mask"))
                if rec == "mask":      # 19: [reads, scale | ReLU, 2 LDS stores, cmp | addc, cmp, addc, 2 and | 2 sub, 2 cvt | max]
                    return [first + cv[0:4], mid1 + cv[4:9], cv[9:14] + mid2, cv[14:18] + mid3, cv[18:19] + last]
                return [first + cv[0:4], mid1 + cv[4:8], cv[8:12] + mid2, cv[12:15] + mid3, last]
            if ts == 7:
                return [[], [], [], [], []]
            cv = conv(src, ts + 1, g, bank ^ 1)
            if EXPERIMENT == "noconv":
                return [[], [], [], [], []]
            return [[], cv[0:4], cv[4:8], cv[8:12], cv[12:13]]

        def tail(g, ts=ts):
            # (tile ts of the SOURCE set -- its operands were converted during the previous chunk -- is seeded with the next
            # layer's bias, that set being the next layer's destination: four LDS reads STRAIGHT INTO the accumulator
            # registers, issued in group 0 BEFORE this group's ring refills so that the counted waits of the following
            # groups cover them.  Until the end of round 2 they went to VGPRs and sixteen v_accvgpr_write followed in the
            # MFMA gaps: ~130 cycles per chunk, a write to the accumulator file next to running MFMAs is not free.)
            out = []
            if ts < 7:
                if 1 <= g <= 3:
                    out.append(advance_dch()[g - 1])
                if g == 7:
                    out += dma_base()
            return out

        def head(g, ts=ts):
            if g != (6 if rec else 0):
                return []
            return ["ds_read_b128 a[%d:%d], %%[nbias] offset:%d" % (src + 16 * ts + 4 * k, src + 16 * ts + 4 * k + 3, ts * 64 + k * 16)
                    for k in range(4)]

        extra = 0
        if rec:        # this chunk's four tile stores + the mask word the previous (even) chunk stored after ITS publish point
            extra = 4 + (1 if (ts & 1) and rec == "mask" else 0)
        chunk_groups(e, 16, ts, mf, fill, (5, 6), 1, tail, head, young_extra=extra)
    ring_epilogue(e, 8)
    return L


# ------------------------------------------------------------------------------------------------------ EXTRA
def gen_extra(dst, ks):
    """one chunk of ks (1 or 2) extra k-steps; B operands at %[stage] (this lane's 16 bytes of k-step 0 hi): hi of k-step
    s at + s*8192, lo at + s*8192 + 4096 (stage layout of mlp_fwd_f16x3.hip)"""
    L = []
    e = L.append
    ring_prologue(e)
    for s in range(ks):
        e("ds_read_b128 %s, %%[stage] offset:%d" % (vr(XB[0] + s * 4), s * 8192))
        e("ds_read_b128 %s, %%[stage] offset:%d" % (vr(XB[0] + 8 + s * 4), s * 8192 + 4096))
    for ins in dma_base():
        e(ins)
    e("s_waitcnt lgkmcnt(0)")

    def mf(g):
        s, t = g >> 2, (2 * g) & 7
        sa, sb = (2 * g) % 4, (2 * g + 1) % 4
        xh, xl = vr(XB[0] + s * 4), vr(XB[0] + 8 + s * 4)
        d0, d1 = ar(dst, t), ar(dst, t + 1)
        return [mfma(d0, vr(F_hi(sa)), xh), mfma(d1, vr(F_hi(sb)), xh), mfma(d0, vr(F_hi(sa)), xl),
                mfma(d1, vr(F_hi(sb)), xl), mfma(d0, vr(F_lo(sa)), xh), mfma(d1, vr(F_lo(sb)), xh)]

    chunk_groups(e, 8 * ks, 0, mf, lambda g: [[], [], [], [], []], (5, 6), 2 // ks)
    ring_epilogue(e, 1)
    return L


# ------------------------------------------------------------------------------------------------------ HEAD
def gen_head(src, rec=False):
    """acc (v[HACC:HACC+15]) = W_head * relu(src set): 16 k-steps of one tile; group ts contracts source tile ts
    (k-steps 2 ts, 2 ts + 1), operands of tile ts+1 converted while its six MFMAs issue.
    rec: also write the activation record of the source set (see "recording variants").  The head accumulator occupies
    v[232:247], and while the transient head runs the compiler owns ~40 VGPRs (the trunk feature sits in v[32:159]): the tile
    travels LDS -> the ACCUMULATOR REGISTERS OF THE SOURCE TILE ITSELF -> memory.  They are dead once the tile has been
    converted (nothing re-seeds the source set of a head: set P is overwritten by the restored feature, set Q by the next
    tile's L1 seed), LDS loads may target AGPRs and vector stores may take their data from them on gfx950."""
    L = []
    e = L.append
    ring_prologue(e)
    if rec:
        e("s_mov_b64 s[%d:%d], %%[rbase]" % (RB, RB + 1))
    for g in range(8):
        for ins in (conv_rec(src, 0, g, 0) if rec else conv(src, 0, g, 0)):
            e(ins)
    def hold_reads(t):
        return rec_reads(src, t)

    if rec:
        for ins in hold_reads(0):                      # tile 0
            e(ins)
    for ins in dma_base():
        e(ins)
    acc = vr(HACC, 16)

    def mf(g):
        bank = g & 1
        sa, sb = (2 * g) % 4, (2 * g + 1) % 4
        xh0, xh1 = vr(XB[bank]), vr(XB[bank] + 4)
        xl0, xl1 = vr(XB[bank] + 8), vr(XB[bank] + 12)
        first = mfma(acc, vr(F_hi(sa)), xh0, "0") if g == 0 else mfma(acc, vr(F_hi(sa)), xh0)
        return [first, mfma(acc, vr(F_hi(sa)), xl0), mfma(acc, vr(F_lo(sa)), xh0),
                mfma(acc, vr(F_hi(sb)), xh1), mfma(acc, vr(F_hi(sb)), xl1), mfma(acc, vr(F_lo(sb)), xh1)]

    def fill(g):
        cv = []
        if g < 7:
            for k in range(8):
                cv += conv_rec(src, g + 1, k, (g & 1) ^ 1) if rec else conv(src, g + 1, k, (g & 1) ^ 1)
        if not rec:
            if g == 7:
                return [[], [], [], [], []]
            # 88 (104) VALU instructions over five gaps: the head is conversion-bound (one tile per six MFMAs)
            return split_even(cv, 5)
        # tile g sits in its own accumulator registers again (read back at the end of the previous group / the prologue,
        # after its conversion): wait for those reads -- at
        # least the two ring refills after them are younger, so "at most two outstanding" covers them --, store the tile,
        # advance the base; odd groups first store the mask word of tiles g - 1, g.  Then this group's conversions of tile
        # g + 1 (with their staging-tile writes), and at the end its read-back.
        store = ["s_waitcnt lgkmcnt(2)"] + [rec_store(src, g, k) for k in range(4)]
        store += rec_advance()
        parts = split_even(cv, 5) if cv else [[], [], [], [], []]
        if g & 1:      # tiles g - 1, g are complete (tile g was converted during the previous group): before this group's pushes
            parts[0] = rec_mask_store(g >> 1) + parts[0]
        parts[1] = store + parts[1]
        if g < 7:
            parts[4] = parts[4] + hold_reads(g + 1)
        return parts

    # record stores issued before the publish point (group 6, after its third MFMA): tiles 0..6 and mask words 0..2
    chunk_groups(e, 16, 0, mf, fill, (3, 6), 1, young_extra=(7 * 4 + 3) if rec else 0)
    ring_epilogue(e, 1)
    e("s_nop 15")                                    # the accumulator tile is read by compiled code after the block
    e("s_nop 7")
    return L


# ------------------------------------------------------------------------------------------------------ HEAD, fp32 VALU form
def gen_head_valu(src, row_offs):
    """Narrow output layer over relu(src set) as fp32 dot products on the vector ALU (ray-bias kernel only): out[row] =
    sum_f relu(acc_f) W[row][f], no operand conversion, no dependent MFMA chain -- and no weight chunk: the 1 / 5 / 3 rows live in
    LDS for the whole kernel (a table in the part of the input stage the ray-bias kernel does not use; mlp_fwd_f16x3.hip), so the
    block stands OUTSIDE the ring protocol (no DMA piece, no publish, no chunk advance).  Inside the ring a 1.3-2.3 k-cycle block
    waited at its publish point for the previous block's DMA pieces (an L2 -> LDS round trip is longer than it): 3.5 k cycles per
    head, measured.
    A lane holds 128 of its sample's 256 features (16 per tile; lane half h the other 128): per source tile 16 v_accvgpr_read +
    16 v_max, then per row eight v_pk_fma_f32 against 16 weights (4 ds_read_b128 at %[hw] + row offset + 32 (4 t + k): the two
    lane halves read the two 16-byte columns of a 32-byte unit, i.e. broadcast reads), two accumulator pairs per row.  Weight reads
    run two row-steps ahead through three 16-register banks: v[208:223], v[232:247] and v[160:175] -- the fragment ring's registers,
    which hold the first four pairs of the CURRENT chunk (fetched by the previous block's tail): they are read again from the
    chunk's slot at the end (the slot stays valid until the ring has advanced twice), the next block's prologue waits for them.
    Results: v[232 + row] = the row's sum over both lanes of a sample, scaled by 2^8 like the accumulators it was formed from."""
    L = []
    e = L.append
    rows = len(row_offs)
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 7")                                     # (the source set may have been written by MFMAs just before the block)
    e("s_nop 7")
    VAL, WA = XB[0], VR + 2
    pairs = [VB + 16 + 2 * i for i in range(8)] + [T, T + 2]
    banks = [XB[1], HACC, VB]
    e("v_mov_b32 v%d, %%[hw]" % WA)
    for i in range(2 * rows):
        e("v_mov_b32 v%d, 0" % pairs[i])
        e("v_mov_b32 v%d, 0" % (pairs[i] + 1))
    steps = [(t, row) for t in range(8) for row in range(rows)]

    def wreads(si):
        t, row = steps[si]
        b = banks[si % 3]
        return ["ds_read_b128 %s, v%d offset:%d" % (vr(b + 4 * k), WA, row_offs[row] + (4 * t + k) * 32) for k in range(4)]

    for si in range(min(2, len(steps))):
        for ins in wreads(si):
            e(ins)
    for t in range(8):
        for r in range(16):
            e("v_accvgpr_read_b32 v%d, a%d" % (VAL + r, src + 16 * t + r))
        for r in range(16):
            e("v_max_f32 v%d, 0, v%d" % (VAL + r, VAL + r))
        for row in range(rows):
            si = t * rows + row
            later = min(len(steps) - 1, si + 1) - si
            e("s_waitcnt lgkmcnt(%d)" % (4 * later))
            b = banks[si % 3]
            for k in range(8):
                a = pairs[2 * row + (k & 1)]
                e("v_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d]" % (a, a + 1, VAL + 2 * k, VAL + 2 * k + 1, b + 2 * k, b + 2 * k + 1, a, a + 1))
            if si + 2 < len(steps):
                for ins in wreads(si + 2):
                    e(ins)
    # sums over the two accumulator pairs, the register pair and the two lanes of a sample (v_permlane32_swap: X.hi <-> Y.lo)
    for row in range(rows):
        a0, a1 = pairs[2 * row], pairs[2 * row + 1]
        e("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (a0, a0 + 1, a0, a0 + 1, a1, a1 + 1))
    for row in range(rows):
        a0 = pairs[2 * row]
        e("v_add_f32 v%d, v%d, v%d" % (a0, a0, a0 + 1))
        e("v_mov_b32 v%d, v%d" % (a0 + 1, a0))
    e("s_nop 1")
    for row in range(rows):
        a0 = pairs[2 * row]
        e("v_permlane32_swap_b32 v%d, v%d" % (a0, a0 + 1))
    e("s_nop 1")
    for row in range(rows):
        a0 = pairs[2 * row]
        e("v_add_f32 v%d, v%d, v%d" % (HACC + row, a0, a0 + 1))
    # the fragment ring again: pairs 0..3 of the current chunk, from its slot (byte offset buf << 15)
    e("s_lshl_b32 %[t0], %[buf], 15")
    e("v_add_u32 v%d, %%[t0], %%[lane16]" % (VR + 0))
    for s_ in range(4):
        refill(e, s_, s_, 16, 0, 1)
    return L


# byte offsets of the head rows in the LDS table, relative to the stage base (mlp_fwd_f16x3.hip: kHeadTab*): sigma | transient 0..4
# | rgb 0, 1 fill k-step 4 of the input stage (8 KB at + 32 KB), rgb 2 sits in slot 6 of the save area (+ 40 KB + 6 KB)
HEADTAB_SIGMA = [32768]
HEADTAB_TRANS = [32768 + 1024 * (1 + r) for r in range(5)]
HEADTAB_RGB = [32768 + 6144, 32768 + 7168, 40960 + 6144]


# ------------------------------------------------------------------------------------------------------ INIT
def gen_init(dst):
    """dst set = 128 floats at LDS address %[bias] (this lane half's bias block of the layer, already * 2^8): 32 LDS reads
    straight into the accumulator registers"""
    L = []
    e = L.append
    e("s_nop 7")
    for k in range(32):
        e("ds_read_b128 a[%d:%d], %%[bias] offset:%d" % (dst + 4 * k, dst + 4 * k + 3, k * 16))
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 3")
    return L


# ------------------------------------------------------------------------------------------------------ STASH / RESTORE
SF = 32                       # the trunk feature is held in v[32:159] from L7 to R0


def gen_stash(src):
    return ["s_nop 7", "s_nop 7"] + ["v_accvgpr_read_b32 v%d, a%d" % (SF + k, src + k) for k in range(128)]


def gen_restore(dst):
    return ["v_accvgpr_write_b32 a%d, v%d" % (dst + k, SF + k) for k in range(128)] + ["s_nop 3"]


def gen_read_tile(src, tile):
    """16 accumulators of one tile -> v[232:247] (training: the activation record is written by compiled code)"""
    return ["s_nop 7", "s_nop 7"] + ["v_accvgpr_read_b32 v%d, a%d" % (HACC + k, src + 16 * tile + k) for k in range(16)]



# ====================================================================================================== DATA GRADIENT
# mlp_dgrad_f16x3_kernel (round 3): the backward chain dh = W^T dz of a head on the same blocks -- transposed f16x3 stream of
# 34 chunks, sets P / Q in ping-pong -- with the GATED conversion of the backward (the recorded ReLU sign words decide, not the
# value) and the dz record written by the block that consumes the set (as the recording forward does).  No seeding: a
# destination tile's first MFMA takes C = 0.  Extra operands:
#   %[g0..3]  VGPR: this layer's four sign words of the lane (bit b of word w <-> tile 2 w + b / 16, register b % 16)
#   %[isc]    VGPR: 2^-k of the sample's chain (0 for a sample past the end): the record holds unscaled gradients
#   %[dzm]    VGPR in/out: running max |dz| of the lane (range of the split-fp16 weight-gradient GEMM)
def conv_gate(src_base, tile, g, bank, xops=True):
    """elements 2g, 2g+1 of source tile `tile`: v = gate ? acc * 2^-8 : 0; record v * isc (staging tile) and max |.|; with
    `xops` also the hi / lo operands in `bank`: 21 (13) instructions"""
    xh = XB[bank] + (g >> 2) * 4 + (g & 3)
    xl = XB[bank] + 8 + (g >> 2) * 4 + (g & 3)
    t0, t1, h0, h1 = T, T + 1, T + 2, T + 3
    a = src_base + 16 * tile + 2 * g
    r0, r1 = 2 * g, 2 * g + 1
    w, b0 = tile >> 1, (tile & 1) * 16 + r0
    out = ["v_accvgpr_read_b32 v%d, a%d" % (t0, a),
           "v_accvgpr_read_b32 v%d, a%d" % (t1, a + 1),
           "v_mul_f32 v%d, %%[kinv], v%d" % (t0, t0),
           "v_mul_f32 v%d, %%[kinv], v%d" % (t1, t1),
           "v_bfe_i32 v%d, %%[g%d], %d, 1" % (h0, w, b0),
           "v_bfe_i32 v%d, %%[g%d], %d, 1" % (h1, w, b0 + 1),
           "v_and_b32 v%d, v%d, v%d" % (t0, t0, h0),
           "v_and_b32 v%d, v%d, v%d" % (t1, t1, h1),
           "v_mul_f32 v%d, %%[isc], v%d" % (h0, t0),
           "v_mul_f32 v%d, %%[isc], v%d" % (h1, t1),
           "ds_write_b32 %%[recw], v%d offset:%d" % (h0, rec_wr_off(r0)),
           "ds_write_b32 %%[recw], v%d offset:%d" % (h1, rec_wr_off(r1)),
           "v_max3_f32 %%[dzm], |v%d|, |v%d|, %%[dzm]" % (h0, h1)]
    if xops:
        out += split_ops(t0, t1, h0, h1, xh, xl)
    return out


def gen_dg_wide(src, dst):
    """256 -> 256 transposed layer: dst = W^T gated(src * 2^-8); records gated(src) * isc as the dz block at %[rbase]"""
    L = []
    e = L.append
    ring_prologue(e)
    e("s_mov_b64 s[%d:%d], %%[rbase]" % (RB, RB + 1))
    for g in range(8):
        for ins in conv_gate(src, 0, g, 0):
            e(ins)
    for ins in rec_reads(src, 0):
        e(ins)
    for ins in dma_base():
        e(ins)
    for ts in range(8):
        bank = ts & 1

        def mf(g, bank=bank, ts=ts):
            s, t = g >> 2, (2 * g) & 7
            sa, sb = (2 * g) % 4, (2 * g + 1) % 4
            xh, xl = vr(XB[bank] + s * 4), vr(XB[bank] + 8 + s * 4)
            d0, d1 = ar(dst, t), ar(dst, t + 1)
            c0 = "0" if ts == 0 and s == 0 else None          # first touch of a destination tile: C = 0, no seeding pass
            return [mfma(d0, vr(F_hi(sa)), xh, c0), mfma(d1, vr(F_hi(sb)), xh, c0), mfma(d0, vr(F_hi(sa)), xl),
                    mfma(d1, vr(F_hi(sb)), xl), mfma(d0, vr(F_lo(sa)), xh), mfma(d1, vr(F_lo(sb)), xh)]

        def fill(g, ts=ts, bank=bank):
            first, mid1, mid2, mid3, last = [], [], [], [], []
            if 1 <= g <= 4:
                first = [rec_store(src, ts, g - 1)]
            elif g == 5:
                first = rec_advance()
            if g == 7 and ts < 7:
                rd = rec_reads(src, ts + 1)
                mid2, mid3 = mid2 + rd[0:2], mid3 + rd[2:4]
            if ts == 7:
                return [first, mid1, mid2, mid3, last]
            cv = conv_gate(src, ts + 1, g, bank ^ 1)       # 19: [reads, scale | 2 bfe, 2 and, mul | mul, 2 LDS stores, max3, and | and, 2 sub, cvt | cvt]
            return [first + cv[0:4], mid1 + cv[4:9], cv[9:14] + mid2, cv[14:18] + mid3, cv[18:19] + last]

        def tail(g, ts=ts):
            out = []
            if ts < 7:
                if 1 <= g <= 3:
                    out.append(advance_dch()[g - 1])
                if g == 7:
                    out += dma_base()
            return out

        chunk_groups(e, 16, ts, mf, fill, (5, 6), 1, tail, None, young_extra=4)
    ring_epilogue(e, 8)
    return L


def gen_dg_narrow(dst):
    """dst = W3^T d: one chunk, one k-step; the 16 + 16 bytes of this lane's B operand (hi / lo) at %[stage] / + 4096"""
    L = []
    e = L.append
    ring_prologue(e)
    e("ds_read_b128 %s, %%[stage] offset:0" % vr(XB[0]))
    e("ds_read_b128 %s, %%[stage] offset:4096" % vr(XB[0] + 8))
    for ins in dma_base():
        e(ins)
    e("s_waitcnt lgkmcnt(0)")

    def mf(g):
        t = (2 * g) & 7
        sa, sb = (2 * g) % 4, (2 * g + 1) % 4
        xh, xl = vr(XB[0]), vr(XB[0] + 8)
        d0, d1 = ar(dst, t), ar(dst, t + 1)
        return [mfma(d0, vr(F_hi(sa)), xh, "0"), mfma(d1, vr(F_hi(sb)), xh, "0"), mfma(d0, vr(F_hi(sa)), xl),
                mfma(d1, vr(F_hi(sb)), xl), mfma(d0, vr(F_lo(sa)), xh), mfma(d1, vr(F_lo(sb)), xh)]

    chunk_groups(e, 8, 0, mf, lambda g: [[], [], [], [], []], (5, 6), 2)
    ring_epilogue(e, 1)
    return L


def gen_dg_finish(src):
    """the last dz block of a head (no layer follows that could carry it): gate + record of set `src`, no MFMA.  Tile t is
    read back (into its own, dead accumulator registers) right after its 16 staging-tile writes; it is stored once the NEXT
    tile's writes have been issued (LDS operations of a wave execute in order: at most those 16 are then outstanding)."""
    L = []
    e = L.append
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 7")
    e("s_nop 7")
    e("s_mov_b64 s[%d:%d], %%[rbase]" % (RB, RB + 1))
    for t in range(8):
        for g in range(8):
            for ins in conv_gate(src, t, g, 0, xops=False):
                e(ins)
        if t > 0:
            e("s_waitcnt lgkmcnt(15)")          # (4-bit counter: the four reads and one write have retired)
            for k in range(4):
                e(rec_store(src, t - 1, k))
            for ins in rec_advance():
                e(ins)
        for ins in rec_reads(src, t):
            e(ins)
    e("s_waitcnt lgkmcnt(0)")
    for k in range(4):
        e(rec_store(src, 7, k))
    return L


def emit_macro(out, name, comment, lines):
    out.append("// " + comment)
    out.append("#define %s \\" % name)
    out.append(" \\\n".join('    "%s\\n"' % l for l in lines))
    out.append("")


def main():
    out = ["// GENERATED by gen_wide_asm.py -- do not edit.  See that file for the schedule.", "#pragma once",
           "#define TP_ASM_CLOBBERS " + ", ".join(['"v%d"' % v for v in CLOBBER_V] + ['"s%d"' % r for r in CLOBBER_S] + ['"a%d"' % a for a in range(256)]),
           "#define TP_ASM_ALL_AGPRS " + ", ".join('"a%d"' % a for a in range(256)),
           '#define TP_ASM_REC_CLOBBERS "v%d", "s%d", "s%d", "vcc"' % (MK, RB, RB + 1)]
    for s, d in (("Q", "P"), ("P", "Q")):
        emit_macro(out, "TP_ASM_WIDE_%s%s" % (s, d), "256 -> 256 layer: set %s -> set %s" % (s, d), gen_wide(SET[s], SET[d]))
        emit_macro(out, "TP_ASM_WIDE_%s%s_REC" % (s, d), "256 -> 256 layer: set %s -> set %s, recording set %s (values + ReLU sign words)" % (s, d, s),
                   gen_wide(SET[s], SET[d], rec="mask"))
    emit_macro(out, "TP_ASM_WIDE_QP_RECNM", "256 -> 256 layer: set Q -> set P, recording set Q (values only: the trunk feature)",
               gen_wide(SET["Q"], SET["P"], rec="nomask"))
    for d in ("P", "Q"):
        for ks in (1, 2):
            emit_macro(out, "TP_ASM_EXTRA%d_%s" % (ks, d), "%d extra k-step(s) into set %s" % (ks, d), gen_extra(SET[d], ks))
        emit_macro(out, "TP_ASM_HEAD_%s" % d, "narrow head over relu(set %s)" % d, gen_head(SET[d]))
        emit_macro(out, "TP_ASM_HEAD_%s_REC" % d, "narrow head over relu(set %s), recording set %s" % (d, d), gen_head(SET[d], rec=True))
        emit_macro(out, "TP_ASM_INIT_%s" % d, "seed set %s with the bias block" % d, gen_init(SET[d]))
        for t in range(8):
            emit_macro(out, "TP_ASM_READ_%s%d" % (d, t), "tile %d of set %s -> v[232:247]" % (t, d), gen_read_tile(SET[d], t))
    emit_macro(out, "TP_ASM_HEADV_P1", "fp32 VALU head, 1 row, over relu(set P)", gen_head_valu(SET["P"], HEADTAB_SIGMA))
    emit_macro(out, "TP_ASM_HEADV_P5", "fp32 VALU head, 5 rows, over relu(set P)", gen_head_valu(SET["P"], HEADTAB_TRANS))
    emit_macro(out, "TP_ASM_HEADV_Q3", "fp32 VALU head, 3 rows, over relu(set Q)", gen_head_valu(SET["Q"], HEADTAB_RGB))
    out.append("#define TP_HEADTAB_OFFSETS {%s}" % ", ".join(str(v) for v in HEADTAB_SIGMA + HEADTAB_TRANS + HEADTAB_RGB))
    emit_macro(out, "TP_ASM_STASH_Q", "set Q -> v[32:159]", gen_stash(SET["Q"]))
    emit_macro(out, "TP_ASM_RESTORE_P", "v[32:159] -> set P", gen_restore(SET["P"]))
    # data gradient: the 34-chunk transposed stream
    emit_macro(out, "TP_ASM_DG_NARROW_P", "data gradient: set P = W3^T d (one k-step)", gen_dg_narrow(SET["P"]))
    emit_macro(out, "TP_ASM_DG_WIDE_PQ", "data gradient: set Q = W^T gated(set P), recording gated(set P)", gen_dg_wide(SET["P"], SET["Q"]))
    emit_macro(out, "TP_ASM_DG_WIDE_QP", "data gradient: set P = W^T gated(set Q), recording gated(set Q)", gen_dg_wide(SET["Q"], SET["P"]))
    emit_macro(out, "TP_ASM_DG_FINISH_P", "data gradient: record gated(set P)", gen_dg_finish(SET["P"]))
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
