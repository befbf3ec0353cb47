import torch, time
def ev(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / n
for mb in (537, 268, 134, 34):
    n = mb * 1000 * 1000 // 2
    x = torch.randn(n, device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
    t = ev(lambda: y.copy_(x)); print(f"{mb} MB copy (r+w): {2*n*2/t/1e12:.2f} TB/s ({t*1e6:.0f} us)")
    t = ev(lambda: torch.relu_(y)); print(f"{mb} MB in-place relu (r+w): {2*n*2/t/1e12:.2f} TB/s")
    xf = x.view(torch.int16)
    t = ev(lambda: xf.max()); print(f"{mb} MB read-only max: {n*2/t/1e12:.2f} TB/s")
    t = ev(lambda: y.zero_()); print(f"{mb} MB write-only: {n*2/t/1e12:.2f} TB/s")
