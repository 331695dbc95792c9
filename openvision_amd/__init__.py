"""openvision_amd — MI355X-native (gfx950) encode-and-contrast path of OpenVision (ViT/CLIP forward + InfoNCE).

Python surface mirrors the reference's ``open_clip.model.CLIP`` and ``open_clip.loss.ClipLoss``; all arithmetic runs
in hand-written HIP kernels behind the C ABI of ``libovhip.so`` (see include/ovhip.h).  No CPU fallback."""
from .config import preset, openvision_model_cfg, load_config_dir, PRESETS  # noqa: F401

__all__ = ["CLIP", "ClipLoss", "gather_features", "create_model", "preset", "openvision_model_cfg", "load_config_dir",
           "WordPieceTokenizer"]


def __getattr__(name):   # lazy: importing the package must not require torch.cuda or the built library
    if name in ("CLIP", "create_model", "VisionTransformer", "Transformer", "ResidualAttentionBlock", "LayerNorm",
                "logits", "l2_normalize"):
        from . import model
        return getattr(model, name)
    if name in ("ClipLoss", "gather_features"):
        from . import loss
        return getattr(loss, name)
    if name == "WordPieceTokenizer":
        from . import tokenizer
        return tokenizer.WordPieceTokenizer
    raise AttributeError(name)
