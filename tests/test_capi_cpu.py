"""CPU: the C-ABI library loads, exports every symbol include/texpose_amd.h declares, rejects bad
arguments without touching a GPU, and its host-side weight packer produces a stream whose layout
(chunk schedule, MFMA fragment order, extra-input column maps) reproduces the oracle MLP when
consumed the way mlp_fwd.hip consumes it."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import texpose_oracle as O
from texpose_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _raw():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libtexpose_amd.so not built (run __graft_entry__.build())")
    return C.CDLL(_lib.LIB_PATH)


def test_exports_match_header():
    lib = _raw()
    header = open(os.path.join(REPO, "include", "texpose_amd.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|size_t|const char\*)\s+(tp_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_sizes_match_header():
    # the ctypes mirrors must have the C layout (8-byte pointers, natural alignment)
    assert C.sizeof(_lib.RaygenArgs) == 7 * 8 + 6 * 4 + 2 * 4 + 8 + 2 * 8 + 8 + 10 * 4 + 5 * 8          # (+ ndc, depth_param)
    assert C.sizeof(_lib.CompositeArgs) == 5 * 8 + 8 + 4 + 4 + 6 * 8
    assert C.sizeof(_lib.PatchSamplerJob) == 8 + 8 + 8 + 8 + 3 * 4 + 2 * 4 + 4 + 8 + 8 + 8 + 8          # (p and the tail of the float / int run are padded to 8)
    assert C.sizeof(_lib.LatentRowsJob) == 3 * 8 + 3 * 4 + 4 + 3 * 8
    assert C.sizeof(_lib.CompositeBwdArgs) == C.sizeof(_lib.CompositeArgs) + 12 * 8
    assert C.sizeof(_lib.MlpFwdArgs) == 8 * 8 + 3 * 4 + 4 + 5 * 8 + 4 + 4 + 8 + 8 + 8 + 8   # (+ ray_bias, density_noise)
    assert C.sizeof(_lib.MlpWeights) == 32 * 8
    assert C.sizeof(_lib.PatchGatherArgs) == 7 * 8 + 4 * 4 + 8 + 3 * 8 + 2 * 4          # (+ the PatchGAN stacks of the same launch)
    assert C.sizeof(_lib.EvalMetricsArgs) == 3 * 8 + 5 * 4 + 4 + 2 * 8
    assert C.sizeof(_lib.SnWeight) == 8 * 8 + 2 * 4 + 2 * 8 + 4 + 4 + 5 * 8       # (+ accumulate, padding, the second instance)
    assert C.sizeof(_lib.SnStepTail) == 4 * 8 + 4 * 4 + 2 * 4 + 3 * 8 + 2 * 4 + 3 * 8 * 8 + 8 + 4 * 4
    assert C.sizeof(_lib.RmspropTensor) == 5 * 8
    assert C.sizeof(_lib.NerfLossesArgs) == 4 * 8 + 3 * 4 + 4 + 4 * 8
    assert C.sizeof(_lib.Conv4s2Args) == 6 * 8 + 6 * 4 + 4 * 8 + 2 * 4 + 8    # (+ the fused InstanceNorm backward of the data gradient, x_copy)
    assert C.sizeof(_lib.Conv3s1Args) == 7 * 8 + 6 * 4
    assert C.sizeof(_lib.FeatInputsArgs) == 2 * 8 + 7 * 4 + 6 * 4 + 4
    assert C.sizeof(_lib.FeatChainArgs) == 2 * 8 + 8 * 4 + 6 * 4 + 8 * 8 + 2 * 4 + 6 * 8
    assert C.sizeof(_lib.RenderEvalArgs) == C.sizeof(_lib.RaygenArgs) + 3 * 8 + 4 + 4 + 8 + 4 + 4 + 4 * 8 + 8       # (+ packed_ray_bias, padding)


def test_argument_validation_without_gpu():
    lib = _lib.load()
    assert lib.tp_abi_version() == _lib.ABI_VERSION
    a = _lib.RaygenArgs()
    assert lib.tp_raygen(C.byref(a), None) < 0
    assert b"bad sizes" in lib.tp_last_error()
    c = _lib.CompositeArgs()
    assert lib.tp_composite_fwd(C.byref(c), None) < 0
    m = _lib.MlpFwdArgs()
    assert lib.tp_mlp_fwd(C.byref(m), None) < 0
    assert lib.tp_mlp_packed_bytes() == (115 * 8192 + 14 * 256 + 16) * 4
    assert lib.tp_mlp_workspace_bytes(128) == 128 * 256 * 4
    assert lib.tp_mlp_saved_bytes(129) == 8 * (7 * 8192 + 1024 + 6 * 4 * 64) * 4


def test_ops_refuse_cpu_tensors():
    from texpose_amd import ops
    with pytest.raises(_lib.TexposeLibraryError):
        ops.composite_fwd(torch.zeros(1, 2, 3), torch.zeros(1, 2, 4, 3, 2), torch.zeros(1, 2, 4, 2),
                          torch.zeros(1, 2, 4, 1), torch.zeros(1, 2, 4, 1))


# ------------------------------------------------------------------ packed-stream emulation
def _feat_of(t, r, h):
    return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h


def _pack_host(params):
    lib = _lib.load()
    w = _lib.MlpWeights()
    keep = []
    for arr_w, arr_b, name, n in ((w.feat_w, w.feat_b, "mlp_feat", 8), (w.rgb_w, w.rgb_b, "mlp_rgb", 4),
                                  (w.trans_w, w.trans_b, "mlp_trans", 4)):
        for i in range(n):
            wt = np.ascontiguousarray(params[f"{name}.{i}.weight"].numpy())
            bt = np.ascontiguousarray(params[f"{name}.{i}.bias"].numpy())
            keep += [wt, bt]
            arr_w[i], arr_b[i] = wt.ctypes.data, bt.ctypes.data
    out = np.zeros(lib.tp_mlp_packed_bytes() // 4, dtype=np.float32)
    assert lib.tp_mlp_pack_host(C.byref(w), out.ctypes.data) == 0
    return out


class _StreamEmu:
    """Consumes the packed stream exactly in the order / with the operand maps of mlp_fwd.hip, in fp64."""

    def __init__(self, packed):
        self.chunks = packed[:115 * 8192].reshape(115, 8192).astype(np.float64)
        self.bias = packed[115 * 8192:].astype(np.float64)
        self.c = 0

    def wide(self, b_of, n_ksteps):          # b_of(rr, h) -> scalar B operand
        A = self.chunks[self.c].reshape(16, 2, 64, 4)       # [rr][g][lane][sub]
        self.c += 1
        out = np.zeros(256)
        for rr in range(n_ksteps):
            for g in range(2):
                for sub in range(4):
                    t = g * 4 + sub
                    for h in range(2):
                        out[32 * t:32 * t + 32] += A[rr, g, 32 * h:32 * h + 32, sub] * b_of(rr, h)
        return out

    def gen(self, hprev):
        out = np.zeros(256)
        for ts in range(8):
            out += self.wide(lambda rr, h: hprev[_feat_of(ts, rr, h)], 16)
        return out

    def head(self, hprev):
        A = self.chunks[self.c].reshape(32, 64, 4)           # [s4][lane][sub]
        self.c += 1
        out = np.zeros(32)
        for s4 in range(32):
            for sub in range(4):
                s = s4 * 4 + sub
                for h in range(2):
                    out += A[s4, 32 * h:32 * h + 32, sub] * hprev[_feat_of(s >> 4, s & 15, h)]
        return out

    def bias_wide(self, wide):
        out = np.zeros(256)
        for h in range(2):
            for t in range(8):
                for r in range(16):
                    out[_feat_of(t, r, h)] = self.bias[((wide * 2 + h) * 8 + t) * 16 + r]
        return out


def _emulate(packed, x, vu, lt, ll):
    e = _StreamEmu(packed)
    pi32 = float(np.float32(np.pi))

    def enc_b(r, h, blk):
        r = blk * 16 + r
        if r < 30:
            arg = float(np.float32(np.float32(x[r // 10]) * np.float32(pi32 * 2.0 ** (r % 10))))
            return np.cos(arg) if h else np.sin(arg)
        if r == 30:
            return x[1] if h else x[0]
        return 0.0 if h else x[2]

    def x40_b(r, h, blk):
        r = blk * 16 + r
        if r < 12:
            arg = float(np.float32(np.float32(vu[r >> 2]) * np.float32(pi32 * 2.0 ** (r & 3))))
            return np.cos(arg) if h else np.sin(arg)
        if r == 12:
            return vu[1] if h else vu[0]
        if r == 13:
            return x[0] if h else vu[2]
        if r == 14:
            return x[2] if h else x[1]
        if r < 39:
            return ll[(r - 15) + 24 * h]
        return 0.0

    relu = lambda v: np.maximum(v, 0.0)
    acc = e.wide(lambda r, h: enc_b(r, h, 0), 16) + e.wide(lambda r, h: enc_b(r, h, 1), 16)
    hcur = relu(acc + e.bias_wide(0))
    for li in range(1, 8):
        if li == 7:
            sig_raw = e.head(hcur)[0] + e.bias[14 * 256 + 0]
        acc = e.gen(hcur)
        if li == 4:
            acc += e.wide(lambda r, h: enc_b(r, h, 0), 16) + e.wide(lambda r, h: enc_b(r, h, 1), 16)
        hcur = relu(acc + e.bias_wide(li))
    feat = hcur
    acc = e.gen(feat) + e.wide(lambda r, h: lt[r + 8 * h], 8)
    hcur = relu(acc + e.bias_wide(8))
    for li in (9, 10):
        hcur = relu(e.gen(hcur) + e.bias_wide(li))
    th = e.head(hcur)[:5] + e.bias[14 * 256 + 1:14 * 256 + 6]
    acc = e.gen(feat) + sum(e.wide(lambda r, h, b=b: x40_b(r, h, b), 16 if b < 2 else 8) for b in range(3))
    hcur = relu(acc + e.bias_wide(11))
    for li in (12, 13):
        hcur = relu(e.gen(hcur) + e.bias_wide(li))
    rh = e.head(hcur)[:3] + e.bias[14 * 256 + 6:14 * 256 + 9]
    assert e.c == 115
    sp = lambda v: np.log1p(np.exp(v))
    sg = lambda v: 1.0 / (1.0 + np.exp(-v))
    return sg(rh), sg(th[:3]), sp(sig_raw), sp(th[3]), sp(th[4])


def test_packed_stream_reproduces_oracle_mlp():
    params = O.make_params(7)
    packed = _pack_host(params)
    rs = np.random.RandomState(0)
    n = 3
    pts = torch.from_numpy(rs.uniform(-1.2, 1.2, size=(1, n, 1, 3)).astype(np.float32))
    unit = torch.nn.functional.normalize(torch.from_numpy(rs.normal(size=(1, n, 1, 3)).astype(np.float32)), dim=-1)
    lt = torch.from_numpy(rs.normal(size=(1, 16)).astype(np.float32))
    ll = torch.from_numpy(rs.normal(size=(1, 48)).astype(np.float32))
    rgb, den, unc = O.mlp_forward(params, pts, unit, lt, ll)
    for i in range(n):
        rs_, rt_, ss_, st_, u_ = _emulate(packed, pts[0, i, 0].numpy().astype(np.float64),
                                          unit[0, i, 0].numpy().astype(np.float64), lt[0].numpy().astype(np.float64),
                                          ll[0].numpy().astype(np.float64))
        np.testing.assert_allclose(rs_, rgb[0, i, 0, :, 0].numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(rt_, rgb[0, i, 0, :, 1].numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose([ss_, st_], den[0, i, 0].numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(u_, unc[0, i, 0, 0].numpy(), rtol=2e-5, atol=2e-6)
