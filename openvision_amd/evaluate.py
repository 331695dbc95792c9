"""On-device consumers of the encode + logits path: zero-shot classification and retrieval recall@k.

Mirrors, on top of the HIP path,
  * ``open_clip.zero_shot_classifier.build_zero_shot_classifier`` (``src/convert_upload/open_clip/zero_shot_classifier.py:21-68``),
  * the zero-shot ``count_correct`` of ``src/evaluators/proj/image_text/discriminative_classifier.py:305-323``,
  * ``image_to_text_retrieval_eval`` / ``text_to_image_retrieval_eval`` of
    ``src/evaluators/proj/image_text/image_text_retrieval.py:24-87`` (Recall@1/5/10).
Tokenisation is outside this path (the hub tokenizer is unavailable offline): callers pass token ids.
Embedding, the similarity matrix (``ov_logits``) and the ranking (``ov_topk``) run in HIP kernels; only the final integer
comparisons / means over a handful of indices use torch tensor ops.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import _lib
from ._lib import ptr, stream_ptr, check
from .model import logits as _logits

RECALL_THRESHOLDS = (1, 5, 10)            # image_text_retrieval.py:21


def topk(x: torch.Tensor, k: int, largest: bool = True):
    """Per-row top-k of a 2-D fp32 device tensor -> (values [rows,k], indices [rows,k] int64); ties -> smaller index."""
    if not x.is_cuda:
        raise _lib.OvhipError("topk: expected a device tensor (no CPU fallback)")
    x = x.detach().float().contiguous()
    idx = torch.empty(x.shape[0], k, dtype=torch.int64, device=x.device)
    val = torch.empty(x.shape[0], k, dtype=torch.float32, device=x.device)
    check(_lib.load().ov_topk(ptr(x), x.stride(0), x.shape[0], x.shape[1], k, int(largest), ptr(idx), ptr(val), stream_ptr()),
          "ov_topk")
    return val, idx


def build_zero_shot_classifier(model, class_tokens: torch.Tensor, num_classes_per_batch: Optional[int] = 10) -> torch.Tensor:
    """class_tokens int64 [C, T, context] (T templates per class) -> zero-shot weights [E, C]
    (zero_shot_classifier.py:51-66: encode_text(normalize=True) -> mean over templates -> renormalise -> transpose)."""
    C_, T_, _ = class_tokens.shape
    lib = _lib.load()
    step = num_classes_per_batch or C_
    outs = []
    for c0 in range(0, C_, step):
        tok = class_tokens[c0:c0 + step]
        emb = model.encode_text(tok.reshape(-1, tok.shape[-1]), normalize=True)
        out = torch.empty(tok.shape[0], emb.shape[1], dtype=torch.float32, device=emb.device)
        check(lib.ov_class_mean_normalize(ptr(emb), ptr(out), tok.shape[0], T_, emb.shape[1], stream_ptr()), "ov_class_mean_normalize")
        outs.append(out)
    return torch.cat(outs, dim=0).T.contiguous()


def zero_shot_predict(model, images: torch.Tensor, classifier: torch.Tensor, k: int = 1):
    """logits = 100 * normalize(encode_image) @ classifier (open_clip zero-shot convention); returns (topk values, indices)."""
    f = model.encode_image(images, normalize=True)
    lg = _logits(f, classifier.T.contiguous(), 100.0)
    return topk(lg, k, largest=True)


def count_correct(best: torch.Tensor, labels: torch.Tensor, mask: Optional[torch.Tensor] = None) -> int:
    """discriminative_classifier.py:305-323: ``best`` [n] predicted class, ``labels`` [n] or [n, m] (-1 padded, "any" match)."""
    if labels.dim() == 1:
        labels = labels[:, None]
    hit = (best[:, None] == labels).sum(dim=1) > 0
    if mask is not None:
        hit = hit & mask.bool()
    return int(hit.sum())


def retrieval_recall(image_emb: torch.Tensor, text_emb: torch.Tensor, text_image_correspondence: Sequence[int],
                     ks: Sequence[int] = RECALL_THRESHOLDS) -> Dict[str, float]:
    """Recall@k both ways from L2-normalised embeddings.  The reference ranks a distance matrix [N_IMAGES, N_TEXTS]
    ascending (image_text_retrieval.py:44,78); similarity descending with the same tie rule is the same ranking."""
    corr = torch.as_tensor(list(text_image_correspondence), dtype=torch.int64, device=image_emb.device)
    sim = _logits(image_emb, text_emb, 1.0)                            # [N_IMAGES, N_TEXTS]
    kmax = max(ks)
    _, img2txt = topk(sim, min(kmax, sim.shape[1]), largest=True)      # per image: best texts
    _, txt2img = topk(sim.T.contiguous(), min(kmax, sim.shape[0]), largest=True)   # per text: best images
    out: Dict[str, float] = {}
    n_img = sim.shape[0]
    for k in ks:
        top_img_of_txt = corr[img2txt[:, :k]]                          # :78-80
        wins = (top_img_of_txt == torch.arange(n_img, device=sim.device)[:, None]).any(dim=1)
        out[f"img2txt/Recall@{k}"] = float(wins.float().mean())
        wins_t = (txt2img[:, :k] == corr[:, None]).any(dim=1)          # :44-48
        out[f"txt2img/Recall@{k}"] = float(wins_t.float().mean())
    return out
