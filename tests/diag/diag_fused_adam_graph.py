"""GPU box: torch's fused Adam (capturable, tensor lr) eager vs inside a CUDA graph."""
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
def run(graphed, fused):
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(64, 33, device=dev)), torch.nn.Parameter(torch.randn(5, device=dev))]
    lr = torch.tensor(1e-3, device=dev)
    opt = torch.optim.Adam([dict(params=ps[:1], lr=lr), dict(params=ps[1:], lr=lr)], capturable=True, **(dict(fused=True) if fused else {}))
    gs = [torch.randn_like(p) for p in ps]
    def body():
        for p, g in zip(ps, gs):
            p.grad = g * 1.0
        opt.step()
    if graphed:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body(); body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # roll back
        torch.manual_seed(0)
        with torch.no_grad():
            ps[0].copy_(torch.randn(64, 33, device=dev)); ps[1].copy_(torch.randn(5, device=dev))
        for st in opt.state.values():
            for v in st.values():
                if torch.is_tensor(v): v.zero_()
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            body()
        # capture executed nothing
        for _ in range(3): g.replay()
    else:
        for _ in range(3): body()
    torch.cuda.synchronize()
    return [p.detach().clone() for p in ps], [float(st["step"]) for st in opt.state.values()]
for fused in (False, True):
    a, sa = run(False, fused); b, sb = run(True, fused)
    print("fused", fused, "max diff eager vs graph", [float((x - y).abs().max()) for x, y in zip(a, b)], "steps", sa, sb)
