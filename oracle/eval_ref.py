"""ORACLE (tests only) — numpy restatement of the reference's evaluator arithmetic that consumes the encode + logits path.

  zero_shot_classifier   src/convert_upload/open_clip/zero_shot_classifier.py:51-57
  count_correct          src/evaluators/proj/image_text/discriminative_classifier.py:305-323
  retrieval recall@k     src/evaluators/proj/image_text/image_text_retrieval.py:24-87
PARITY PINNING: pinned for the retrieval recall@k and the zero-shot classifier weights -- tests/golden/make_golden.py (gen_eval)
runs the reference's own image_text_retrieval.py (numpy-only, loaded from its file) and zero_shot_classifier.py (torch-only, via
the bare open_clip package) and commits tests/golden/eval.npz; tests/test_eval_oracle.py checks these functions against it.
count_correct stays pinned by hand-checkable cases only: discriminative_classifier.py imports jax / tfds at module level
(parity unpinned for that function).  TEST INFRASTRUCTURE ONLY: nothing under openvision_amd/ imports this module.
"""
import numpy as np

RECALL_THRESHOLDS = (1, 5, 10)


def zero_shot_classifier(class_embeddings: np.ndarray, num_classes: int, num_templates: int) -> np.ndarray:
    """[C*T, E] already L2-normalised text embeddings -> [E, C] (mean over templates, renormalise, transpose)."""
    ce = class_embeddings.reshape(num_classes, num_templates, -1).mean(axis=1)
    ce = ce / np.linalg.norm(ce, axis=1, keepdims=True)
    return ce.T


def count_correct(zimg: np.ndarray, ztxt: np.ndarray, labels: np.ndarray, mask: np.ndarray) -> int:
    best_txt = (zimg @ ztxt.T).argmax(axis=1)
    if labels.ndim == 1:
        labels = labels[..., None]
    matching = (best_txt[:, None] == labels).sum(axis=1)
    return int(np.where(mask, (matching > 0).astype(np.int32), 0).sum())


def text_to_image_retrieval_eval(dist_matrix: np.ndarray, text_image_correspondence):
    per_text_ranks = dist_matrix.argsort(axis=0, kind="stable")
    corr = np.array(text_image_correspondence)
    return {f"Recall@{k}": (per_text_ranks[:k, :] == corr[None]).any(axis=0).mean() for k in RECALL_THRESHOLDS}


def image_to_text_retrieval_eval(dist_matrix: np.ndarray, text_image_correspondence):
    per_image_ranks = dist_matrix.argsort(axis=1, kind="stable")
    corr = np.array(text_image_correspondence)
    out = {}
    for k in RECALL_THRESHOLDS:
        top_k_images = corr[per_image_ranks[:, :k]]
        out[f"Recall@{k}"] = (top_k_images == np.arange(len(per_image_ranks))[:, None]).any(axis=1).mean()
    return out
