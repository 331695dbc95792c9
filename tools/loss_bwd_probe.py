"""Timing of the InfoNCE forward and backward kernels (ov_clip_loss / ov_clip_loss_backward) at the bench shape
(b = N = 256) and at config #5's per-GPU shape (b = 4096 of N = 32768), E = 768.  FLOP model: forward 4 b N E (two logit
strips), backward 16 b N E (each strip recomputed twice, once per output side, plus the two P.X products per strip)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipops as H

for b, N in [(256, 256), (4096, 32768)]:
    E = 768
    g = torch.Generator(device="cuda").manual_seed(0)
    ai = torch.nn.functional.normalize(torch.randn(N, E, device="cuda", generator=g), dim=-1)
    at = torch.nn.functional.normalize(ai * 0.5 + torch.randn(N, E, device="cuda", generator=g) * 0.05, dim=-1)
    img, txt = ai[:b].contiguous(), at[:b].contiguous()
    s = 1 / 0.07
    _, terms = H.clip_loss(img, txt, ai, at, s, 0)
    reps = 20 if N <= 256 else 3
    for name, fn, flops in (("forward", lambda: H.clip_loss(img, txt, ai, at, s, 0), 4.0 * b * N * E),
                            ("backward local side only", lambda: H.clip_loss_backward(img, txt, ai, at, s, 0, terms, gathered=False), 8.0 * b * N * E),
                            ("backward both sides", lambda: H.clip_loss_backward(img, txt, ai, at, s, 0, terms), 16.0 * b * N * E)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"b {b} N {N} E {E} {name}: {ms:.3f} ms = {flops / ms / 1e9:.1f} TFLOP/s fp32 (peak 157.3)", flush=True)
