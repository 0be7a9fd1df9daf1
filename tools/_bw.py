import torch, time
x = torch.rand(256*1024*1024//4*4, device='cuda')   # 1 GiB
y = torch.empty_like(x)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
ms=t(lambda: y.copy_(x)); print("copy 1 GiB: %.3f ms  %.0f GB/s (read+write)" % (ms, 2*x.numel()*4/ms/1e6))
ms=t(lambda: x.sum()); print("sum 1 GiB: %.3f ms  %.0f GB/s (read)" % (ms, x.numel()*4/ms/1e6))
ms=t(lambda: y.fill_(1.0)); print("fill 1 GiB: %.3f ms  %.0f GB/s (write)" % (ms, x.numel()*4/ms/1e6))
xs = x[:480*640*128*6]
ms=t(lambda: xs.sum()); print("sum 0.94 GB: %.3f ms  %.0f GB/s (read)" % (ms, xs.numel()*4/ms/1e6))
