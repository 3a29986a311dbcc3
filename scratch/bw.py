import torch, time
dev = torch.device("cuda:0")
x = torch.randn(3_000_000 * 48, device=dev)
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
b = x.numel() * 4
ms = t(lambda: x.sum()); print(f"sum (read {b/1e6:.0f} MB): {ms:.4f} ms = {b/ms/1e9:.2f} TB/s")
y = torch.empty_like(x)
ms = t(lambda: y.copy_(x)); print(f"copy (r+w {2*b/1e6:.0f} MB): {ms:.4f} ms = {2*b/ms/1e9:.2f} TB/s")
ms = t(lambda: y.zero_()); print(f"fill (write {b/1e6:.0f} MB): {ms:.4f} ms = {b/ms/1e9:.2f} TB/s")
ms = t(lambda: torch.mul(x, 2.0, out=y)); print(f"mul (r+w): {ms:.4f} ms = {2*b/ms/1e9:.2f} TB/s")
