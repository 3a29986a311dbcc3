"""Camera/frame data-parallel helpers for an 8 x MI355X node.

The reference is single-process (`batch_size: 1`, /root/reference/3D-Reconstruction-main/models/trainers/base.py:411);
`rasterization(C > 1)` is by definition the per-camera computation with the Gaussian-parameter gradients summed over
cameras, so cameras/frames shard across ranks with every rank holding all Gaussians, and the only exchange step is one
SUM all-reduce of the dense parameter gradients (RCCL over xGMI; backend "nccl" on ROCm) after backward, plus -- for
densification parity with base.py:279-297 / vanilla.py:154-182 -- a SUM of the per-Gaussian gradient-norm / visibility
counters and a MAX of the normalised radii.  Compute-agnostic: works with gloo on CPU tensors (tests) and RCCL on GPU.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor


def world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_views(n_views: int, rank: int, world_size: int) -> range:
    """Contiguous, balanced split of view indices [0, n_views): the first n_views % world ranks get one more."""
    assert 0 <= rank < world_size
    q, r = divmod(n_views, world_size)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def all_reduce_gradients(params: Iterable[Tensor], group=None, average: bool = False) -> None:
    """SUM (or mean) all-reduce of `.grad` of every parameter, in place.

    One collective per parameter tensor, all launched asynchronously and then waited on together: for the 3DGS
    parameter set these are five large messages (12, 16, 12, 4 and 12*K bytes per Gaussian; 236 MB at N = 1M, K = 16),
    which is the regime where RCCL's ring over the xGMI links is bandwidth- rather than latency-bound, and it avoids the
    flatten/unflatten copies of a bucketed reducer.  Ranks whose cameras saw no Gaussian still contribute zeros, so a
    missing .grad is materialised as zeros to keep the collective sequence identical on every rank."""
    params = list(params)  # (a generator, e.g. model.parameters(), is walked twice below)
    rank, ws = world(group)
    if ws == 1:
        return
    works = []
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        works.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    if average:
        for p in params:
            p.grad.div_(ws)


class _Works:
    """Several asynchronous collectives waited on together."""

    def __init__(self, works):
        self.works = [w for w in works if w is not None]

    def wait(self):
        for w in self.works:
            w.wait()


class GradientExchange:
    """Camera/frame data parallelism around `rasterization()`: inside the `with` block the backward of every
    `rasterization()` call made on this thread returns gradients of the Gaussian parameters (means, quats, scales,
    opacities, colours / SH coefficients) that are already SUMMED OVER ALL RANKS' CAMERAS -- `.grad` after
    `loss.backward()` is the gradient of the whole batch, on every rank, as if one process had rendered every camera.
    Per-camera inputs (viewmats, [C,N,D] colours) keep their local gradients.

        ex = GradientExchange()                      # once, after init_process_group
        with ex:
            rc, ra, meta = rasterization(...)        # this rank's cameras
        loss(rc, ra).backward()                      # exchange happens inside, overlapped with the backward kernels

    Two exchange forms:
      "dense"   one SUM all-reduce per parameter tensor (12 + 16 + 12 + 4 + 12 K bytes per Gaussian: 236 B at K = 16)
      "colour"  (default) for SH colours evaluated inside rasterization(sh_degree=...): the SH-coefficient gradient, 192 of
                those 236 bytes, is rank 1 per (camera, Gaussian) -- basis(view direction) (x) v_rgb -- so the ranks
                all-gather the 12-B colour gradients of their VISIBLE pairs (a visibility bitmap + packed values:
                gs_projection_bwd_split) and every rank rebuilds the full v_sh locally, in the same fixed order
                (gs_sh_bwd_packed: bit-identical on all ranks); only the geometry gradients (44 B per Gaussian) are
                all-reduced, behind the all-gather and while the SH kernel runs.  Calls without in-rasterization SH fall
                back to "dense" for that call.
    Per step and rank at N = 1M Gaussians, 5 cameras, V = 0.98M visible pairs, 8 ranks: dense = 236 MB all-reduced;
    colour = 12.7 MB sent (101 MB gathered) + 44 MB all-reduced.  With world size 1 the block is a no-op.

    The reference has no counterpart (single process, /root/reference/3D-Reconstruction-main/models/trainers/base.py:411);
    the placement of the exchange -- after backward, before optimizer.step() -- is base.py:502-516's."""

    def __init__(self, group=None, mode: str = "auto", force: bool = False):
        assert mode in ("auto", "colour", "dense"), mode
        self.group = group
        self.mode = "colour" if mode == "auto" else mode
        self.rank, self.world = world(group)
        self.active = self.world > 1 or force
        # force: the exchange path, collectives included, runs even on one rank (tests: a one-rank RCCL group puts the
        # real backend under every call the N-rank run makes)
        self.collectives = self.world > 1 or (force and dist.is_available() and dist.is_initialized())
        self._prev = None
        self.bytes_sent = 0  # payload bytes this rank contributed to collectives (diagnostics / tests)

    # ---- context: read by gsplat.cuda._wrapper._ProjectSplats.forward (the autograd node keeps the reference)
    def __enter__(self):
        from .cuda import _wrapper as W

        self._prev = W.current_exchange()
        if self.active:
            W._tls.exchange = self
        return self

    def __exit__(self, *exc):
        from .cuda import _wrapper as W

        if self.active:
            W._tls.exchange = self._prev
        return False

    @property
    def packs_colours(self) -> bool:
        return self.mode == "colour"

    # ---- collectives (called from the autograd nodes)
    def _nccl(self) -> bool:
        return self.collectives and dist.get_backend(self.group) == "nccl"

    def _all_reduce(self, t: Tensor, op, async_op: bool):
        if t.is_cuda and not self._nccl():
            # gloo with device tensors (the one-GPU rehearsal of the N > 1 path; never a measured configuration):
            # staged through the host explicitly -- gloo's own device-tensor path stalls when ranks share a GPU
            h = t.cpu()
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
            return None
        return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)

    def max_over_ranks(self, x: Tensor) -> Tensor:
        """MAX over the ranks of a small device tensor (the visible-pair count that sizes the messages)."""
        if not self.collectives:
            return x
        y = x.clone()
        self._all_reduce(y, dist.ReduceOp.MAX, False)
        return y

    def all_reduce(self, tensors: Sequence[Tensor]) -> _Works:
        """Asynchronous in-place SUM of each tensor; returns a handle whose wait() orders the current stream behind."""
        if not self.collectives:
            return _Works([])
        works = []
        for t in tensors:
            self.bytes_sent += t.numel() * t.element_size()
            works.append(self._all_reduce(t, dist.ReduceOp.SUM, True))
        return _Works(works)

    def all_gather_messages(self, msg: Tensor):
        """All-gather of one equally sized uint8 message per rank.  Returns (messages [world, len], world, own index);
        the current stream is ordered behind the collective."""
        if not self.collectives:
            return msg, 1, 0
        out = torch.empty(self.world, msg.numel(), device=msg.device, dtype=msg.dtype)
        self.bytes_sent += msg.numel()
        if self._nccl():
            dist.all_gather_into_tensor(out, msg, group=self.group)
        elif msg.is_cuda:  # gloo with device tensors (the one-GPU rehearsal): staged through the host explicitly
            host = torch.empty(self.world, msg.numel(), dtype=msg.dtype)
            dist.all_gather(list(host.unbind(0)), msg.cpu(), group=self.group)
            out.copy_(host)
        else:  # gloo on CPU tensors (tests)
            dist.all_gather(list(out.unbind(0)), msg, group=self.group)
        return out, self.world, self.rank


def all_reduce_densification_stats(grad_norm_sum: Tensor, vis_count: Tensor, max_radii: Tensor, group=None) -> None:
    """In place: SUM of accumulated |means2d grad| norms and visibility counts, MAX of radii / max(W, H) -- the three
    per-Gaussian statistics VanillaGaussians.after_train keeps (vanilla.py:154-182)."""
    _, ws = world(group)
    if ws == 1:
        return
    w1 = dist.all_reduce(grad_norm_sum, op=dist.ReduceOp.SUM, group=group, async_op=True)
    w2 = dist.all_reduce(vis_count, op=dist.ReduceOp.SUM, group=group, async_op=True)
    w3 = dist.all_reduce(max_radii, op=dist.ReduceOp.MAX, group=group, async_op=True)
    for w in (w1, w2, w3):
        w.wait()


def rasterize_sharded(render_fn, viewmats: Tensor, Ks: Tensor, group=None, **kwargs):
    """Render this rank's shard of the C cameras with `render_fn(viewmats=..., Ks=..., **kwargs)` (normally
    gsplat.rendering.rasterization with the Gaussian tensors bound).  Returns (outputs of render_fn, view range)."""
    rank, ws = world(group)
    views = shard_views(viewmats.shape[0], rank, ws)
    sl = slice(views.start, views.stop)
    return render_fn(viewmats=viewmats[sl], Ks=Ks[sl], **kwargs), views
