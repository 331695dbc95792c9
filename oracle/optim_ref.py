"""ORACLE (tests only) -- numpy restatement of the reference trainer's parameter update.  TEST INFRASTRUCTURE ONLY.

The reference builds its update as an optax chain (src/optim/build_optax.py:272-278; config: optax_name='scale_by_adam',
optax=dict(mu_dtype='bfloat16', b1=0.9, b2=0.95), wd=0.2 on '.*/kernel$', src/configs/openvision.py:265-289) and applies it at
src/main_clip.py:480-483.  optax is a third-party dependency that is NOT installed here (requirements.txt lists it unpinned), so its
published algorithm is restated: clip_by_global_norm (g * clip / max(||g||, clip)), scale_by_adam (moments, bias correction,
mu cast to mu_dtype after the update), add_decayed_weights (u + wd * p on the mask), scale(lr), scale(-1).
PARITY PINNING: parity unpinned -- no optax run and no fixture of the reference exists for this step; the restatement follows
optax/_src/transform.py (scale_by_adam, add_decayed_weights) and clipping.py of the 0.2.x releases.
"""
import numpy as np


def _bf16(x: np.ndarray) -> np.ndarray:
    """Round fp32 to bfloat16 (nearest even), returned as fp32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def adamw_step(p, g, mu, nu, step, lr, b1=0.9, b2=0.95, eps=1e-8, wd=0.0, grad_scale=1.0, clip_norm=None, gnorm=None):
    """One update of a flat fp32 group.  `gnorm`: the global norm of the (scaled) gradients over ALL groups when clipping."""
    s = np.float32(grad_scale)
    if clip_norm is not None:
        n = np.float32(gnorm if gnorm is not None else np.sqrt(((g.astype(np.float64) * grad_scale) ** 2).sum()))
        s = s * (np.float32(clip_norm) / max(n, np.float32(clip_norm)))      # scale and clip factor folded into one multiplier
    g = g.astype(np.float32) * s
    mu = (np.float32(b1) * mu + np.float32(1.0 - b1) * g).astype(np.float32)          # fp32 here; cast for storage below
    nu = (np.float32(b2) * nu + np.float32(1.0 - b2) * (g * g)).astype(np.float32)
    u = (mu / np.float32(1.0 - b1 ** step)) / (np.sqrt(nu / np.float32(1.0 - b2 ** step)) + np.float32(eps))
    p = (p - np.float32(lr) * (u + np.float32(wd) * p)).astype(np.float32)
    return p, _bf16(mu), nu
