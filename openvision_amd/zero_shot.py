"""Counterpart of the fork's ``ov-zero-shot-test.py`` on the MI355X path.

    python -m openvision_amd.zero_shot --use_model DIR --image_dir testcat --prompts "a cat|a dog|a remote control"
    python -m openvision_amd.zero_shot --use_model DIR --image_dir testcat --tokens prompts.npy [--labels a,b,c]

Reproduces the script's observable behaviour: config-dir loading (:37-56), the printed visual-config block (:59-65),
the per-image sorted cosine / probability table (:167-195) and the per-text best image (:198-208).  Prompts are tokenised by
``openvision_amd.tokenizer`` (the reference's WordPiece tokenizer restated; ``<DIR>/vocab.txt`` if present, else the packaged
vocabulary) or given as an int64 ``[n, context_length]`` ``.npy`` of token ids; images are decoded on the host and resized /
normalised on the device (``openvision_amd.preprocess``).
``DIR`` holds ``open_clip_config.json`` and the weights (``open_clip_pytorch_model.bin`` read with
``torch.load(weights_only=True)``, or safetensors; ``openvision_amd.checkpoint``) or, with ``--synthetic``, formula weights.

Token framing: PARITY UNPINNED against the script.  The script tokenises with ``HFTokenizer(cfg.text_cfg.hf_tokenizer_name)``
(``ov-zero-shot-test.py:81``: ``AutoTokenizer`` framing taken from a hub config that is not on disk: [CLS] ... [SEP] then zero
padding).  ``--prompts`` here uses the TRAINING framing of ``CLIPS_Tokenizer`` (``open_clip/tokenizer.py:522-594``: bos 1 ... eos 2,
zero padding, class token 101 last), which is what the text tower's last-token pooling was trained on.
"""
from __future__ import annotations

import argparse
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import config as ovcfg
from . import synth
from .model import create_model, logits


def load_images(image_dir: str, size: int, mean: Sequence[float], std: Sequence[float]) -> Tuple[List[str], torch.Tensor]:
    """Decode on the host, then Resize((S, S)) -> ToTensor -> Normalize (ov-zero-shot-test.py:72-77) on the device with
    ``openvision_amd.preprocess`` (bit-exact against Pillow for RGB files; other modes are converted to RGB before the resize,
    where the script resizes in the file's own mode first)."""
    from PIL import Image
    from .preprocess import preprocess
    names = [n for n in sorted(os.listdir(image_dir)) if n.lower().endswith((".png", ".jpg", ".jpeg", ".webp"))]
    raw = [np.asarray(Image.open(os.path.join(image_dir, n)).convert("RGB"), dtype=np.uint8) for n in names]
    return names, preprocess(raw, size, mean=mean, std=std, resize_mode="squash", interpolation="bilinear")


def describe(model) -> str:
    v = model.visual
    return ("\nVisual Config Used:\n"
            f"  Pool type:             {v.pool_type}\n"
            f"  Final LN after pool:   {v.final_ln_after_pool}\n"
            f"  Attn pool:             {v.attn_pool}\n"
            f"  Projection shape:      {tuple(v.proj.shape)}\n"
            f"  Positional emb shape:  {tuple(v.positional_embedding.shape)}\n"
            f"  Class token shape:     {tuple(v.class_embedding.shape)}\n")


@torch.no_grad()
def zero_shot_table(model, images: torch.Tensor, tokens: torch.Tensor):
    """cosine [n_img, n_txt], probs, argsort(desc) — images are encoded one at a time, as the script does."""
    tf = model.encode_text(tokens, normalize=True)
    rows = [logits(model.encode_image(images[i:i + 1], normalize=True), tf)[0] for i in range(images.shape[0])]
    cos = torch.stack(rows)
    probs = torch.softmax(cos * model.logit_scale.detach().exp(), dim=-1)       # [n_img, n_txt] softmax: plumbing on a 5x9 table
    return cos, probs, cos.argsort(dim=-1, descending=True)


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(description="OpenVision Text-Image Test (MI355X)")
    ap.add_argument("--use_model", required=True)
    ap.add_argument("--image_dir", default="testcat")
    ap.add_argument("--tokens", default="", help=".npy int64 [n, context_length] token ids")
    ap.add_argument("--prompts", default="", help="'|'-separated prompt texts (tokenised here)")
    ap.add_argument("--labels", default="")
    ap.add_argument("--synthetic", action="store_true", help="formula weights instead of open_clip_pytorch_model.bin")
    a = ap.parse_args(argv)
    model_cfg, pp = ovcfg.load_config_dir(a.use_model)
    if a.synthetic:
        model = create_model({k: v for k, v in model_cfg.items() if k in ("embed_dim", "vision_cfg", "text_cfg")},
                             device="cuda:0", state_dict=synth.make_state_dict(model_cfg))
    else:
        from .checkpoint import from_pretrained
        model, pp = from_pretrained(a.use_model, device="cuda:0")     # .bin (weights_only) or safetensors, wrappers unwrapped
    model.use_graphs(8)          # one image per encode call: ~170 launches each, replayed as one hipGraph
    print(describe(model))
    names, imgs = load_images(a.image_dir, model_cfg["vision_cfg"]["image_size"], pp["mean"], pp["std"])
    if bool(a.tokens) == bool(a.prompts):
        ap.error("give exactly one of --tokens / --prompts")
    if a.prompts:
        from .tokenizer import WordPieceTokenizer
        texts = [t.strip() for t in a.prompts.split("|")]
        vocab = os.path.join(a.use_model, "vocab.txt")
        tok = WordPieceTokenizer(vocab if os.path.exists(vocab) else None, context_length=model_cfg["text_cfg"]["context_length"])
        tokens = tok(texts)
        labels = a.labels.split(",") if a.labels else texts
    else:
        tokens = torch.from_numpy(np.load(a.tokens, allow_pickle=False).astype(np.int64))
        labels = a.labels.split(",") if a.labels else [f"prompt {i}" for i in range(tokens.shape[0])]
    cos, probs, order = zero_shot_table(model, imgs.to("cuda:0"), tokens.to("cuda:0"))
    cos, probs, order = cos.cpu(), probs.cpu(), order.cpu()
    print("\n=== Cosine Similarities and Predictions ===")
    for i, n in enumerate(names):
        print(f"\n--- {n} ---")
        for j in order[i].tolist():
            print(f"{labels[j]:<25} cosine: {cos[i, j]:+.4f}  prob: {probs[i, j]:.4%}")
    print("\n=== Best Image Per Text ===")
    best = probs.argmax(dim=0)
    for j, lab in enumerate(labels):
        print(f"{lab:<20} <- {names[int(best[j])]:>25} ({probs[int(best[j]), j]:.4%})")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
