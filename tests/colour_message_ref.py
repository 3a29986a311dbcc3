"""numpy restatement of the packed colour-gradient message (include/gsplat_mi355x.h, gs_projection_bwd_split /
gs_sh_bwd_packed), used by the tests only: the CPU two-rank test drives gsplat.distributed.GradientExchange's
collectives with it, and the GPU tests compare the kernels' messages with it byte for byte."""
import ctypes

import numpy as np
import torch


def layout(C_layout: int, N: int, cap: int):
    from gsplat import _backend as B

    out = (ctypes.c_int64 * 5)()
    assert B.lib().gs_colour_message_layout(C_layout, N, cap, out) == 0
    return dict(zip(("campos", "words", "prefix", "vals", "total"), (int(v) for v in out)))


def pack(visible: np.ndarray, v_rgb: np.ndarray, campos: np.ndarray, C_layout: int, cap: int) -> np.ndarray:
    """visible [C,N] bool, v_rgb [C,N,3] f32 (clamp-gated), campos [C,3] f32 -> message bytes (uint8 [total]).
    Words: bit n % 64 of word n // 64; prefix: exclusive scan of the word populations over the flattened [C, W];
    vals: the visible pairs' gradients in ascending (c, n) order."""
    C, N = visible.shape
    W = (N + 63) // 64
    L = layout(C_layout, N, cap)
    msg = np.zeros(L["total"], dtype=np.uint8)
    msg[0:4] = np.array([C], dtype=np.int32).view(np.uint8)
    msg[L["campos"]:L["campos"] + C * 12] = campos.astype(np.float32).reshape(-1).view(np.uint8)
    pad = np.zeros((C, W * 64), dtype=bool)
    pad[:, :N] = visible
    bits = pad.reshape(C, W, 64)
    words = (bits.astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(-1, dtype=np.uint64)
    counts = bits.sum(-1).reshape(-1)
    prefix = (np.cumsum(counts) - counts).astype(np.int32)
    msg[L["words"]:L["words"] + C * W * 8] = words.reshape(-1).view(np.uint8)
    msg[L["prefix"]:L["prefix"] + C * W * 4] = prefix.view(np.uint8)
    vals = v_rgb.astype(np.float32)[visible]  # row-major (c, n) order
    assert vals.shape[0] <= cap
    msg[L["vals"]:L["vals"] + vals.size * 4] = vals.reshape(-1).view(np.uint8)
    return msg


def unpack(msg: np.ndarray, C_layout: int, N: int, cap: int):
    """-> (n_cams, campos [n_cams,3], visible [n_cams,N] bool, v_rgb [n_cams,N,3] with zeros where not visible)."""
    L = layout(C_layout, N, cap)
    W = (N + 63) // 64
    C = int(msg[0:4].view(np.int32)[0])
    assert 0 <= C <= C_layout
    campos = msg[L["campos"]:L["campos"] + C * 12].view(np.float32).reshape(C, 3).copy()
    words = msg[L["words"]:L["words"] + C * W * 8].view(np.uint64).reshape(C, W)
    prefix = msg[L["prefix"]:L["prefix"] + C * W * 4].view(np.int32).reshape(C, W)
    bits = ((words[..., None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool)
    visible = bits.reshape(C, W * 64)[:, :N]
    rank_in_word = np.cumsum(bits, -1) - bits
    rows = (prefix[..., None] + rank_in_word).reshape(C, W * 64)[:, :N]
    vals = msg[L["vals"]:L["vals"] + cap * 12].view(np.float32).reshape(cap, 3)
    v_rgb = np.zeros((C, N, 3), dtype=np.float32)
    v_rgb[visible] = vals[rows[visible]]
    return C, campos, visible, v_rgb


def rebuild(messages, C_layout: int, N: int, cap: int, means: torch.Tensor, coeffs: torch.Tensor, degree: int):
    """What gs_sh_bwd_packed computes from a list of messages, by fp64 autograd through the oracle's SH: v_sh [N,K,3]
    and the view-direction part of v_means [N,3]."""
    import oracle

    mu = means.detach().double().requires_grad_(True)
    co = coeffs.detach().double().requires_grad_(True)
    total = torch.zeros((), dtype=torch.float64)
    for msg in messages:
        C, campos, visible, v_rgb = unpack(np.asarray(msg), C_layout, N, cap)
        if C == 0:
            continue
        dirs = mu[None] - torch.from_numpy(campos).double()[:, None, :]
        cols = oracle.spherical_harmonics(degree, dirs, co[None].expand(C, -1, -1, -1), masks=torch.from_numpy(visible))
        total = total + (cols * torch.from_numpy(v_rgb).double()).sum()
    if total.requires_grad:
        total.backward()
    z = lambda t: t.grad if t.grad is not None else torch.zeros_like(t)
    return z(co), z(mu)
