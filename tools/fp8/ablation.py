"""Config #5, per-GEMM ablation (run on the GPU box): e4m3 operands on exactly one of the block's four GEMMs at a time, on pairs and on
all four, with FROZEN static scales (repeatable), for both L/14 weight sets:

  * 1 - cos of the embeddings against the REFERENCE's fp32 outputs on the committed golden inputs (tests/golden/large14_224.npz: v1
    weights, 2 pairs; large14_224_sharp.npz: sharp weights, 4 pairs): worst embedding per side;
  * 1 - cos against this build's bf16 path on 256 fresh pairs (median / max per side);
  * images/s of the bench step (bench.py --precision fp8 --fp8-mask m, batch 256, synthetic v1 weights).

    python tools/fp8/ablation.py > gpurun_out/r03_fp8_ablation.md
"""
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openvision_amd import preset, synth                      # noqa: E402
from openvision_amd.model import create_model                  # noqa: E402

DEV = "cuda:0"
NAMES = {1: "QKV", 2: "out_proj", 4: "c_fc", 8: "c_proj"}
MASKS = [0, 1, 2, 4, 8, 3, 12, 13, 14, 15]


def label(m):
    return "bf16 (none)" if m == 0 else " + ".join(v for k, v in NAMES.items() if m & k)


def worst(a, b):
    return float((1 - torch.nn.functional.cosine_similarity(a.float().cpu(), b.float().cpu())).max())


def dist(a, b):
    d = (1 - torch.nn.functional.cosine_similarity(a.float(), b.float())).cpu().numpy()
    return float(np.median(d)), float(d.max())


def main():
    cfg = preset("vit-large-patch14-224")
    speed = {}
    if "--no-speed" not in sys.argv:
        for m in MASKS:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--cpu-seconds", "0"]
            if m:
                cmd += ["--precision", "fp8", "--fp8-mask", str(m)]
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True).stdout
            speed[m] = json.loads(out.strip().splitlines()[-1])["value"]
            sys.stderr.write(f"speed mask {m}: {speed[m]}\n")
    rows = {}
    for variant, gname in (("v1", "large14_224.npz"), ("sharp", "large14_224_sharp.npz")):
        g = np.load(os.path.join(ROOT, "tests", "golden", gname))
        model = create_model(cfg, device=DEV, state_dict=synth.make_state_dict(cfg, 0, variant))
        img = torch.from_numpy(g["images"].astype(np.float32)).to(DEV)
        tok = torch.from_numpy(g["tokens"]).to(DEV)
        ref_i, ref_t = torch.from_numpy(g["image_features"].astype(np.float32)), torch.from_numpy(g["text_features"].astype(np.float32))
        big_i = (synth.make_structured_images(256, 224, seed=77) if variant == "sharp" else synth.make_images(256, 224, seed=77)).to(DEV).to(torch.bfloat16)
        big_t = synth.make_captions(256, seed=77).to(DEV)
        b16_i, b16_t = model.encode_image(big_i, normalize=True), model.encode_text(big_t, normalize=True)
        for m in MASKS:
            if m:
                model.set_precision("fp8", m)
                model.encode_image(big_i); model.encode_text(big_t)            # calibration (row-wise dynamic scales, maxima recorded)
                model.encode_image(img); model.encode_text(tok)
                model.freeze_fp8_scales(delayed=False)                         # frozen: repeatable, batch-composition invariant
            else:
                model.set_precision("bf16")
            fi, ft = model.encode_image(img), model.encode_text(tok)
            bi, bt = model.encode_image(big_i, normalize=True), model.encode_text(big_t, normalize=True)
            rows[(variant, m)] = dict(gi=worst(fi, ref_i), gt=worst(ft, ref_t), bi=dist(bi, b16_i), bt=dist(bt, b16_t))
            sys.stderr.write(f"{variant} mask {m}: {rows[(variant, m)]}\n")
        model.set_precision("bf16")
        del model
        torch.cuda.empty_cache()
    base = speed.get(0)
    print("# Config #5: per-GEMM fp8 (e4m3) ablation, ViT-L/14@224, frozen static scales\n")
    print("`1-cos vs fp32 ref` = worst embedding on the committed golden inputs against the REFERENCE's fp32 outputs (v1: 2 pairs, sharp: 4 pairs);")
    print("`vs bf16 (256)` = median / max over 256 fresh pairs against this build's bf16 path.  Speed: bench step, batch 256, one MI355X.\n")
    print("| e4m3 operands on | img/s | vs bf16 | v1 img 1-cos vs fp32 ref | v1 txt | sharp img 1-cos vs fp32 ref | sharp txt | sharp img vs bf16 (256) med / max | sharp txt vs bf16 (256) med / max |")
    print("|---|---|---|---|---|---|---|---|---|")
    for m in MASKS:
        v, s = rows[("v1", m)], rows[("sharp", m)]
        sp = f"{speed[m]:.0f} | {speed[m] / base - 1:+.1%}" if speed else "- | -"
        print(f"| {label(m)} (mask {m}) | {sp} | {v['gi']:.2e} | {v['gt']:.2e} | {s['gi']:.2e} | {s['gt']:.2e} | {s['bi'][0]:.1e} / {s['bi'][1]:.1e} | {s['bt'][0]:.1e} / {s['bt'][1]:.1e} |")
    print("\n```json")
    print(json.dumps({"speed": speed, "rows": {f"{k[0]}:{k[1]}": v for k, v in rows.items()}}))
    print("```")


if __name__ == "__main__":
    main()
