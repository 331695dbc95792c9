import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openvision_amd import preset, synth
from openvision_amd.model import create_model, _run_blocks
DEV = "cuda:0"
cfg = preset("vit-large-patch14-224")
for variant in ("v1", "sharp"):
    m = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg, 0, variant))
    for seed in (60, 61, 9):
        tok = synth.make_captions(256, seed=seed).to(DEV)
        t = m.encode_text(tok, normalize=True)
        bad = (~torch.isfinite(t)).any(dim=1)
        print(variant, "seed", seed, "text rows with NaN/Inf:", int(bad.sum()), bad.nonzero().flatten()[:10].tolist())
        if bad.any():
            x = m.token_embedding(tok).to(torch.float32) + m.positional_embedding.float()
            blocks = list(m.transformer.resblocks)
            for k in range(1, len(blocks) + 1):
                y = _run_blocks(blocks[:k], x)
                nb = (~torch.isfinite(y)).any(dim=2)
                print("   after block", k - 1, "tokens bad:", int(nb.sum()), "max abs", float(y[torch.isfinite(y)].abs().max()))
                if nb.any():
                    idx = nb.nonzero()[:6].tolist()
                    print("   first bad (image, token):", idx)
                    # single-sample rerun of a bad caption
                    bi = idx[0][0]
                    y1 = _run_blocks(blocks[:k], x[bi:bi + 1])
                    print("   same caption alone -> bad tokens:", int((~torch.isfinite(y1)).any(dim=2).sum()))
                    break
    img = synth.make_structured_images(256, 224, seed=60).to(DEV).to(torch.bfloat16)
    f = m.encode_image(img, normalize=True)
    print(variant, "image rows bad:", int((~torch.isfinite(f)).any(dim=1).sum()))
