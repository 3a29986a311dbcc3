"""l1_ssim vs the torch-op chain of base.py:533-541 (pytorch_msssim's op sequence in fp32 on the GPU), forward + backward, at the training image sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-reconstruction_amd")]
import torch
from gsplat.losses import l1_ssim
import torch.nn.functional as F

dev = torch.device("cuda:0")
_c = torch.arange(11, dtype=torch.float32) - 5
_g = torch.exp(-(_c ** 2) / (2 * 1.5 ** 2))
WIN = (_g / _g.sum()).to(dev).reshape(1, 1, 1, -1).repeat(3, 1, 1, 1)


def _filt(x):
    return F.conv2d(F.conv2d(x, WIN.transpose(2, 3), groups=3), WIN, groups=3)


def torch_chain(pred, gt, mask):
    """base.py:533-541 with pytorch_msssim 1.0.0's op sequence (ten depthwise convolutions)."""
    gt = gt * mask[..., None]
    pred = pred * mask[..., None]
    l1 = torch.abs(gt - pred).mean()
    X, Y = gt.permute(2, 0, 1)[None], pred.permute(2, 0, 1)[None]
    mu1, mu2 = _filt(X), _filt(Y)
    s1, s2, s12 = _filt(X * X) - mu1.pow(2), _filt(Y * Y) - mu2.pow(2), _filt(X * Y) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) / (mu1.pow(2) + mu2.pow(2) + 1e-4)) * ((2 * s12 + 9e-4) / (s1 + s2 + 9e-4))
    return l1, torch.flatten(m, 2).mean(-1).mean()


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    for H, W in ((375, 1242), (1280, 1920)):
        gt = torch.rand(H, W, 3, device=dev)
        mask = (torch.rand(H, W, device=dev) > 0.1).float()
        pred = torch.rand(H, W, 3, device=dev, requires_grad=True)

        def ref():
            pred.grad = None
            l1, s = torch_chain(pred, gt, mask)
            (0.8 * l1 + 0.2 * (1 - s)).backward()

        def fused():
            pred.grad = None
            l1, s = l1_ssim(pred, gt, mask)
            (0.8 * l1 + 0.2 * (1 - s)).backward()

        print(f"{H}x{W}: torch chain {t(ref):.3f} ms   fused {t(fused):.3f} ms (fwd + bwd)")
        from gsplat import _backend as B
        B.profiling_start(); fused(); fused(); print({k: round(v[1], 4) for k, v in B.profiling_stop().items()})


if __name__ == "__main__":
    main()
