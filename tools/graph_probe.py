"""Batch-1 encode_image latency with and without hipGraph replay (CLIP.use_graphs), per preset."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openvision_amd import preset, synth
from openvision_amd.model import create_model
for name in ("vit-tiny-patch16-160", "vit-large-patch14-224"):
    cfg = preset(name)
    m = create_model(cfg, device="cuda", state_dict=synth.make_state_dict(cfg))
    S = cfg["vision_cfg"]["image_size"]
    img = synth.make_images(1, S, seed=1).to("cuda")
    for graphs in (0, 8):
        m.use_graphs(graphs)
        for _ in range(5):
            m.encode_image(img, normalize=True)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            m.encode_image(img, normalize=True)
        torch.cuda.synchronize()
        print(f"{name} batch 1 encode_image, graphs {'on' if graphs else 'off'}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per image", flush=True)
