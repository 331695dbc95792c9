"""MI355X-native counterpart of ``open_clip.loss.ClipLoss`` (reference ``src/convert_upload/open_clip/loss.py:19-131``).

Data-parallel InfoNCE: each rank holds ``[b, E]`` L2-normalised image and text embeddings; ONE RCCL all-gather of
the packed ``[b, 2E]`` buffer (``torch.distributed`` backend "nccl" == RCCL over xGMI) yields the rank-ordered global
sets (loss.py:52-61), then the fused HIP kernel computes the local ``[b, N]`` logit strips both ways with labels
``i + b*rank`` (loss.py:93-94,108-110) without materialising them.

Gradients: when a feature tensor or ``logit_scale`` requires grad, the loss is an autograd node whose backward is the
HIP kernel behind ``ov_clip_loss_backward`` (d loss / d features and d loss / d logit_scale); ``openvision_amd.training``
carries the gradient on through the towers.  The gathered-side terms are routed as
``gather_features`` does (loss.py:19-63): own chunk only, or summed over ranks (reduce-scatter) with ``gather_with_grad``.
``use_horovod`` is rejected (RCCL via torch.distributed is the only transport here).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist
from torch import nn

from . import _lib
from ._lib import ptr, stream_ptr, check


_COMM_LOG = None      # diagnostics (bench.py --gpus N): a list that gather_features appends one (start, end) pair per collective to


def record_comm(log) -> None:
    """Diagnostics: pass a list to have every all-gather of gather_features bracketed by a pair of timing marks appended to it --
    ``torch.cuda.Event``s recorded on the current stream for device tensors (the collective is stream-ordered: the current stream waits
    for it), ``time.perf_counter()`` floats on the CPU (gloo rehearsal).  ``None`` turns it off.  Nothing is synchronised here."""
    global _COMM_LOG
    _COMM_LOG = log


def gather_features(image_features: torch.Tensor, text_features: torch.Tensor, local_loss: bool = False,
                    gather_with_grad: bool = False, rank: int = 0, world_size: int = 1, use_horovod: bool = False,
                    group=None, force: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """loss.py:19-63.  One all_gather_into_tensor of the packed [b, 2E] buffer instead of two list gathers;
    the result is identical: rows in rank order (torch.cat(gathered, dim=0), loss.py:60-61)."""
    if use_horovod:
        raise NotImplementedError("horovod transport is not supported; use torch.distributed (RCCL)")
    if world_size == 1 and not force:          # force: run the collective anyway (tests drive the RCCL path in a world of one)
        return image_features, text_features
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("world_size > 1 needs an initialised torch.distributed process group (caller owns init)")
    b, e = image_features.shape
    packed = torch.cat([image_features.detach().float(), text_features.detach().float()], dim=1).contiguous()
    out = torch.empty(world_size * b, 2 * e, dtype=torch.float32, device=packed.device)
    if _COMM_LOG is None:
        dist.all_gather_into_tensor(out, packed, group=group)
    elif packed.is_cuda:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dist.all_gather_into_tensor(out, packed, group=group)
        e1.record()
        _COMM_LOG.append((e0, e1))
    else:
        import time
        t0 = time.perf_counter()
        dist.all_gather_into_tensor(out, packed, group=group)
        _COMM_LOG.append((t0, time.perf_counter()))
    return out[:, :e].contiguous(), out[:, e:].contiguous()


class ClipLoss(nn.Module):
    """Same constructor and call signature as the reference (loss.py:68-83,120-131)."""

    def __init__(self, local_loss: bool = False, gather_with_grad: bool = False, cache_labels: bool = False,
                 rank: int = 0, world_size: int = 1, use_horovod: bool = False, group=None):
        super().__init__()
        self.group = group          # process group of the gather / reduce-scatter (None = the default group, as the reference)
        if use_horovod:
            raise NotImplementedError("horovod transport is not supported; use torch.distributed (RCCL)")
        self.local_loss, self.gather_with_grad, self.cache_labels = local_loss, gather_with_grad, cache_labels
        self.rank, self.world_size, self.use_horovod = rank, world_size, use_horovod
        self.always_collective = False   # tests only: take the world_size > 1 path (gather / reduce-scatter) in a world of one rank
        self._ws: Optional[torch.Tensor] = None
        self.last_terms: Optional[torch.Tensor] = None     # [4, b]: lse_img, diag_img, lse_txt, diag_txt

    @staticmethod
    def _device_scale(logit_scale, device) -> torch.Tensor:
        """The logit multiplier as a 1-element fp32 DEVICE tensor.  It never visits the host (ovhip.h ABI 2): `float(logit_scale)`
        would drain the stream after both towers before the loss could be enqueued."""
        if isinstance(logit_scale, torch.Tensor):
            if not logit_scale.is_cuda:
                return logit_scale.detach().float().reshape(1).to(device, non_blocking=True)
            return logit_scale.detach().float().reshape(1)
        return torch.full((1,), float(logit_scale), dtype=torch.float32, device=device)

    def _loss_strips(self, img, txt, all_img, all_txt, scale: torch.Tensor, label_offset: int) -> torch.Tensor:
        if not img.is_cuda:
            raise _lib.OvhipError("ClipLoss: features must live on an MI355X device (no CPU fallback)")
        lib = _lib.load()
        img, txt = img.detach().float().contiguous(), txt.detach().float().contiguous()
        all_img, all_txt = all_img.detach().float().contiguous(), all_txt.detach().float().contiguous()
        b, e = img.shape
        n = all_img.shape[0]
        nbytes = lib.ov_clip_loss_workspace_bytes(b, n)
        if self._ws is None or self._ws.device != img.device or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=img.device)
        out = torch.empty(1, dtype=torch.float32, device=img.device)
        terms = torch.empty(4, b, dtype=torch.float32, device=img.device)
        check(lib.ov_clip_loss(ptr(img), ptr(txt), ptr(all_img), ptr(all_txt), b, n, e, ptr(scale), int(label_offset),
                               ptr(out), ptr(terms), ptr(self._ws), nbytes, stream_ptr()), "ov_clip_loss")
        self.last_terms = terms
        return out[0]

    def forward(self, image_features, text_features, logit_scale, output_dict: bool = False):
        needs_grad = torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad
                                                     for t in (image_features, text_features, logit_scale))
        if needs_grad:
            if not isinstance(logit_scale, torch.Tensor):
                logit_scale = torch.tensor(float(logit_scale), device=image_features.device)
            loss = _ClipLossFn.apply(self, image_features, text_features, logit_scale)
            return {"contrastive_loss": loss} if output_dict else loss
        if not image_features.is_cuda:
            raise _lib.OvhipError("ClipLoss: features must live on an MI355X device (no CPU fallback)")
        scale = self._device_scale(logit_scale, image_features.device)
        if self.world_size > 1 or self.always_collective:
            all_img, all_txt = gather_features(image_features, text_features, self.local_loss, self.gather_with_grad,
                                               self.rank, self.world_size, self.use_horovod, self.group, self.always_collective)
            if self.local_loss:
                loss = self._loss_strips(image_features, text_features, all_img, all_txt, scale,
                                         image_features.shape[0] * self.rank)
            else:   # global [N,N] logits on every rank (loss.py:111-113): the strips of the full set
                loss = self._loss_strips(all_img, all_txt, all_img, all_txt, scale, 0)
        else:
            loss = self._loss_strips(image_features, text_features, image_features, text_features, scale, 0)
        return {"contrastive_loss": loss} if output_dict else loss


def _sum_over_ranks_own_chunk(full: torch.Tensor, b: int, rank: int, group=None) -> torch.Tensor:
    """Backward of ``torch.distributed.nn.all_gather`` (loss.py:49-50): every rank's [N, 2E] gathered-side gradient summed,
    this rank keeps rows [rank*b, (rank+1)*b).  One reduce-scatter on RCCL; gloo has none, so all-reduce + slice there."""
    if dist.get_backend(group) == "nccl":
        out = torch.empty(b, full.shape[1], dtype=full.dtype, device=full.device)
        dist.reduce_scatter_tensor(out, full.contiguous(), group=group)
        return out
    full = full.contiguous()
    dist.all_reduce(full, group=group)
    return full[rank * b:(rank + 1) * b]


class _ClipLossFn(torch.autograd.Function):
    """ClipLoss as an autograd node.  forward = the fused strip kernel; backward = ov_clip_loss_backward plus the routing of
    the gathered-side gradient that the reference gets from autograd through gather_features (loss.py:19-63)."""

    @staticmethod
    def forward(ctx, mod: "ClipLoss", image_features, text_features, logit_scale):
        ws, rank = mod.world_size, mod.rank
        img, txt = image_features.detach().float().contiguous(), text_features.detach().float().contiguous()
        b = img.shape[0]
        multi = ws > 1 or mod.always_collective
        if multi:
            all_img, all_txt = gather_features(img, txt, mod.local_loss, mod.gather_with_grad, rank, ws, mod.use_horovod, mod.group,
                                               mod.always_collective)
        else:
            all_img, all_txt = img, txt
        if multi and not mod.local_loss:
            x_img, x_txt, off = all_img, all_txt, 0          # every rank evaluates the global loss (loss.py:111-113)
        else:
            x_img, x_txt, off = img, txt, b * rank
        scale = mod._device_scale(logit_scale, img.device)
        loss = mod._loss_strips(x_img, x_txt, all_img, all_txt, scale, off)
        ctx.mod, ctx.off, ctx.b, ctx.multi = mod, off, b, multi
        ctx.in_dtypes = (image_features.dtype, text_features.dtype, logit_scale.dtype)
        ctx.save_for_backward(x_img, x_txt, all_img, all_txt, mod.last_terms, scale)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        mod: "ClipLoss" = ctx.mod
        x_img, x_txt, all_img, all_txt, terms, scale = ctx.saved_tensors
        lib = _lib.load()
        ws, rank, b = mod.world_size, mod.rank, ctx.b
        bx, e = x_img.shape
        n = all_img.shape[0]
        # the gathered side carries gradient when it IS the local tensor (world_size 1), when the own chunk was put back
        # (not local_loss, loss.py:57-59) or when the gather itself is differentiable (gather_with_grad)
        single = not ctx.multi                      # world of one without the collectives: both sides are the same tensors
        gathered_grad = single or not mod.local_loss or mod.gather_with_grad
        d_img, d_txt = torch.empty_like(x_img), torch.empty_like(x_txt)
        d_all = torch.empty(2, n, e, dtype=torch.float32, device=x_img.device) if gathered_grad else None
        d_scale = torch.empty(1, dtype=torch.float32, device=x_img.device)
        grad = grad_out.detach().float().reshape(1).contiguous()        # device scalar: no host round trip
        nbytes = lib.ov_clip_loss_backward_workspace_bytes(bx, n)
        wsb = torch.empty(nbytes + 256, dtype=torch.uint8, device=x_img.device)
        check(lib.ov_clip_loss_backward(ptr(x_img), ptr(x_txt), ptr(all_img), ptr(all_txt), bx, n, e, ptr(scale), ctx.off, ptr(terms),
                                        ptr(grad), ptr(d_img), ptr(d_txt), ptr(d_all[0]) if gathered_grad else None,
                                        ptr(d_all[1]) if gathered_grad else None, ptr(d_scale), ptr(wsb), nbytes, stream_ptr()),
              "ov_clip_loss_backward")
        if single:
            g_img, g_txt = d_img + d_all[0], d_txt + d_all[1]
        elif mod.local_loss:
            g_img, g_txt = d_img, d_txt
            if mod.gather_with_grad:
                own = _sum_over_ranks_own_chunk(torch.cat([d_all[0], d_all[1]], dim=1), b, rank, mod.group)
                g_img, g_txt = g_img + own[:, :e], g_txt + own[:, e:]
        else:
            tot = torch.cat([d_img + d_all[0], d_txt + d_all[1]], dim=1)          # [N, 2E]: both sides are the global set
            own = _sum_over_ranks_own_chunk(tot, b, rank, mod.group) if mod.gather_with_grad else tot[rank * b:(rank + 1) * b]
            g_img, g_txt = own[:, :e], own[:, e:]
        dt_i, dt_t, dt_s = ctx.in_dtypes
        return None, g_img.to(dt_i), g_txt.to(dt_t), d_scale[0].to(dt_s)
