"""Checks of golden G19 shared by the GPU test (tests/test_gpu_parity.py) and the CPU mirror test (tests/test_host_logic_cpu.py)."""
import torch

from oracle import texpose_oracle as O


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def g19_sub(G, key, t):
    """rel-L2 of `t` against a G13-style packed tensor (full, or strided subsample) + its norm."""
    stride = int(G["stride"])
    t = t.detach().reshape(-1).double().cpu()
    ref = G[key].double() if key in G else G[key + ".sub"].double()
    sub = t if key in G else t[::stride]
    return max(float((sub - ref).norm() / ref.norm()), abs(float(t.norm()) - float(G[key + ".norm"])) / float(G[key + ".norm"]))


def g19c_disc(G, device):
    from texpose_amd.gan_modules import Discriminator
    from texpose_amd.options import default_options
    opt = default_options(H=32, W=32, device=str(device))
    opt.patch_size = 16
    opt.gan.L_nocs = opt.gan.L_normal = int(G["c.L"])
    opt.gan.geo_c2f = [float(v) for v in G["c.range"]]
    cpu = Discriminator(opt)
    O.seed_spectral_module(cpu, int(G["c.seed_d"]))
    sd = cpu.state_dict()
    sd.update({k: G["c.in." + k] for k in sd if "c.in." + k in G})
    disc = Discriminator(opt).to(device)
    disc.load_state_dict(sd)
    disc.train()
    return opt, disc


def g19c_check(G, opt, disc, device, wtol):
    Bd = G["c.x"].shape[0]
    for tag in ("p030", "p100"):
        state = {k: v.clone() for k, v in disc.state_dict().items()}
        disc.progress.data.fill_(float(G[f"c.{tag}.progress"]))
        for q in disc.parameters():
            q.grad = None
        x = G["c.x"].to(device).requires_grad_()
        scale = G["c.scale"].to(device)
        logits = disc(opt, x, scale)
        (gx,) = torch.autograd.grad(logits.sum(), x, create_graph=True)
        reg = gx.pow(2).reshape(Bd, -1).sum(1).mean()
        (torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.ones_like(logits)) + 10.0 * reg).backward()
        torch.testing.assert_close(logits.detach().cpu(), G[f"c.{tag}.logits"], rtol=1e-4, atol=1e-6)
        assert rel_l2(gx, G[f"c.{tag}.gx"]) < 1e-4 and abs(float(reg) - float(G[f"c.{tag}.reg"])) <= 1e-4 * float(G[f"c.{tag}.reg"])
        errs = {n: g19_sub(G, f"c.{tag}.g.{n}", q.grad) for n, q in disc.named_parameters() if q.grad is not None}
        assert len(errs) == 6 and max(errs.values()) < wtol, (tag, errs)
        for k, v in disc.state_dict().items():
            if f"c.{tag}.out.{k}" in G:
                assert rel_l2(v, G[f"c.{tag}.out.{k}"]) < 1e-5, k
        disc.load_state_dict(state)
    return errs


