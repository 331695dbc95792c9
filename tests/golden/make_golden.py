#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in the build container.

Usage (build container only; /root/reference does not exist on the GPU box):
    python tests/golden/make_golden.py [--ref /root/reference]

The reference's PyTorch definition of the OpenVision model is the vendored open_clip 2.26.1 under
``src/convert_upload/open_clip`` (``model.CLIP``, ``loss.ClipLoss``).  Its package ``__init__``
pulls torchvision/ftfy, which are absent offline, so it is imported per SURVEY.md §8c: a stub
``torchvision.ops.misc.FrozenBatchNorm2d`` and a bare ``open_clip`` package whose ``__path__`` points at
the vendored directory (skipping ``__init__.py``).  Only arrays (inputs + the reference's outputs)
are written; weights are NOT stored — they are rebuilt from ``openvision_amd.synth.make_state_dict``.

Files written (np.savez_compressed):
  ops.npz                 LayerNorm / GELU(erf,tanh) / ResidualAttentionBlock of the reference on small inputs
  tiny16_160.npz          Ti/16@160 + text-Ti: tokens after block 0 / last block, features, logits, loss
  preprocess.npz          Pillow resize / ToTensor / Normalize outputs on the testcat PNGs and three synthetic images
  tiny16_160_testcat.npz  the 5 testcat PNGs (resized to 160, normalised) x 9 caption rows: cosine/probs/argsort ('sharp' weights)
  *_sharp.npz             Ti/16, L/14, S/8 on the 'sharp' weights + structured images: separated embeddings, per-block slices
  large14_224.npz         L/14@224 + text-L, B=2: features (fp32 and the reference's bf16 mode), token slices
  small8_384.npz          S/8@384, B=1 (2305 tokens): features, token slices
  eval.npz                recall@k from the reference's image_text_retrieval.py, classifier weights from its zero_shot_classifier.py
  cliploss_ws.npz         ClipLoss(local_loss=True) per-rank losses at world_size 2 and 8 over gloo
  opgrad.npz              autograd through the reference's LayerNorm / nn.Linear / nn.GELU for random upstream gradients
  blockgrad.npz           autograd through the reference ResidualAttentionBlock (Tiny block 0): d input + parameter gradients
  tokenizer.npz           caption token ids [N, 80] from the tokenizers library on the reference's vocabulary, reference framing
  cliploss_grad.npz       autograd gradients of ClipLoss at world_size 1 and per rank at world_size 2
"""
from __future__ import annotations

import argparse
import html
import importlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from openvision_amd import config as ovcfg          # noqa: E402
from openvision_amd import synth                    # noqa: E402


def import_reference(ref_root: str):
    import transformers  # noqa: F401  (must precede the torchvision stub)
    tv = types.ModuleType("torchvision")
    tvo = types.ModuleType("torchvision.ops")
    tvm = types.ModuleType("torchvision.ops.misc")

    class FrozenBatchNorm2d(torch.nn.Module):
        pass

    tvm.FrozenBatchNorm2d = FrozenBatchNorm2d
    tv.ops, tvo.misc = tvo, tvm
    sys.modules.update({"torchvision": tv, "torchvision.ops": tvo, "torchvision.ops.misc": tvm})
    pkg = types.ModuleType("open_clip")
    pkg.__path__ = [os.path.join(ref_root, "src/convert_upload/open_clip")]
    sys.modules["open_clip"] = pkg
    return (importlib.import_module("open_clip.model"), importlib.import_module("open_clip.loss"),
            importlib.import_module("open_clip.transformer"))


def build_ref(m, model_cfg, seed=0, cast_dtype=None, variant="v1"):
    model = m.CLIP(embed_dim=model_cfg["embed_dim"], vision_cfg=dict(model_cfg["vision_cfg"]),
                   text_cfg=dict(model_cfg["text_cfg"]), cast_dtype=cast_dtype)
    sd = synth.make_state_dict(model_cfg, seed, variant)
    model.load_state_dict(sd, strict=True)       # ov-zero-shot-test.py:54
    model.eval()
    if cast_dtype is not None:
        m.convert_weights_to_lp(model, dtype=cast_dtype)   # factory.py:275-296 ('bf16' precision)
    return model


def tokens_after_blocks(model, images, which):
    """Hidden states after selected resblocks of the vision tower (forward hooks on the reference)."""
    outs, hooks = {}, []
    for i in which:
        hooks.append(model.visual.transformer.resblocks[i].register_forward_hook(
            lambda mod, inp, out, i=i: outs.__setitem__(i, out.detach().float().clone())))
    with torch.no_grad():
        model.encode_image(images)
    for h in hooks:
        h.remove()
    return outs


def f32(t):
    return t.detach().float().cpu().numpy().copy()


def gen_ops(m, tr, out):
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(6, 192, generator=g) * 2.0 + 0.3
    ln = tr.LayerNorm(192, eps=1e-6)     # the towers construct every LN with eps=1e-6 (transformer.py:458,491,499,537)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(192, generator=g) * 0.1 + 1)
        ln.bias.copy_(torch.randn(192, generator=g) * 0.1)
    xg = torch.linspace(-6, 6, 193)
    cfg = ovcfg.preset("vit-tiny-patch16-160")
    sd = synth.make_state_dict(cfg, 0)
    blk = tr.ResidualAttentionBlock(192, 3, 4.0, batch_first=True, norm_layer=lambda d: tr.LayerNorm(d, eps=1e-6))
    p = "visual.transformer.resblocks.0."
    blk.load_state_dict({k[len(p):]: v for k, v in sd.items() if k.startswith(p)}, strict=True)
    blk.eval()
    xb = torch.randn(2, 101, 192, generator=g)
    with torch.no_grad():
        np.savez_compressed(
            out, ln_x=f32(x), ln_w=f32(ln.weight), ln_b=f32(ln.bias), ln_y=f32(ln(x)), ln_eps=np.float64(ln.eps),
            gelu_x=f32(xg), gelu_erf=f32(torch.nn.GELU()(xg)), gelu_tanh=f32(torch.nn.GELU(approximate="tanh")(xg)),
            blk_x=f32(xb), blk_y=f32(blk(xb)))


def gen_tiny(m, lossmod, out):
    cfg = ovcfg.preset("vit-tiny-patch16-160")
    model = build_ref(m, cfg)
    img = synth.make_images(4, 160, seed=11)
    tok = synth.make_captions(4, 80, 32000, seed=11)
    hs = tokens_after_blocks(model, img, [0, 11])
    with torch.no_grad():
        fi, ft = model.encode_image(img), model.encode_text(tok)
        ni, nt, s = model(img, tok)
        li, lt = model.get_logits(img, tok)
        loss = lossmod.ClipLoss()(ni, nt, s)
        conv = model.visual.conv1(img)
    np.savez_compressed(out, images=f32(img), tokens=tok.numpy(), conv1=f32(conv), block0=f32(hs[0]), block11=f32(hs[11]),
                        image_features=f32(fi), text_features=f32(ft), image_norm=f32(ni), text_norm=f32(nt),
                        logit_scale_exp=f32(s), logits_per_image=f32(li), logits_per_text=f32(lt), loss=f32(loss))


def load_testcat(ref_root, size, mean, std):
    from PIL import Image
    names = sorted(n for n in os.listdir(os.path.join(ref_root, "testcat")) if n.lower().endswith(".png"))
    arr = []
    for n in names:
        im = Image.open(os.path.join(ref_root, "testcat", n)).convert("RGB")
        im = im.resize((size, size), Image.BILINEAR)         # torchvision Resize on a PIL image
        a = np.asarray(im, dtype=np.float32) / 255.0          # ToTensor
        a = (a - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
        arr.append(a.transpose(2, 0, 1))
    return names, torch.from_numpy(np.stack(arr)).half().float()   # stored as fp16


TOPK_GAP = 5e-2      # margin the fixtures are built with: neighbouring ranks of the reference this far apart


def leading_ranks(row, gap):
    """Number of leading positions of the descending order of `row` that are determined at margin `gap`."""
    srt = np.sort(row)[::-1]
    k = 0
    while k + 1 < len(srt) and srt[k] - srt[k + 1] > gap:
        k += 1
    return k


def gen_testcat(m, ref_root, out):
    """Counterpart of ov-zero-shot-test.py:157-195 on formula weights (the 'sharp' set: distinct inputs give separated
    embeddings).  The HF tokenizer is unavailable offline, so the 9 'prompts' are committed token-id rows in the training
    format, selected so that the reference's own cosine table separates neighbouring ranks of every row by more than
    TOPK_GAP (fixtures are built with margins)."""
    cfg = ovcfg.preset("vit-tiny-patch16-160")
    model = build_ref(m, cfg, variant="sharp")
    pp = ovcfg.DEFAULT_PREPROCESS
    names, img = load_testcat(ref_root, 160, pp["mean"], pp["std"])
    with torch.no_grad():
        feats = []
        for i in range(img.shape[0]):                      # batch = 1 per image, as the script does
            f = model.encode_image(img[i:i + 1])
            feats.append(f / f.norm(dim=-1, keepdim=True))
        feats = torch.cat(feats)
        # 9 caption rows picked greedily from a seeded pool so that, in every image's row, the chosen captions' cosines
        # are pairwise at least TOPK_GAP apart: the reference's whole ranking is then decided with margin
        pool = synth.make_captions(4096, 80, 32000, seed=7)
        pf = model.encode_text(pool)
        pf = pf / pf.norm(dim=-1, keepdim=True)
        pc = (feats @ pf.T).numpy()                        # [5, pool]
        chosen = []
        for j in np.argsort(pc.mean(axis=0)):              # sweep in order of the mean cosine: 1-D interval packing
            if all(np.abs(pc[:, j] - pc[:, c]).min() > TOPK_GAP * 1.02 for c in chosen):
                chosen.append(int(j))
        if len(chosen) < 9:
            raise RuntimeError(f"caption pool gives only {len(chosen)} separated rows")
        chosen = [chosen[i] for i in np.linspace(0, len(chosen) - 1, 9).round().astype(int)]
        chosen = sorted(chosen)
        tok, cos = pool[chosen], feats @ pf[chosen].T
        probs = (model.logit_scale.exp() * cos).softmax(dim=-1)
    print("  testcat: pool rows", chosen, "leading ranks per row", [leading_ranks(r, TOPK_GAP) for r in cos.numpy()])
    np.savez_compressed(out, names=np.array(names), images=img.numpy().astype(np.float16), tokens=tok.numpy(),
                        cosine=f32(cos), probs=f32(probs), argsort=cos.argsort(dim=-1, descending=True).numpy(),
                        best=probs.argmax(dim=-1).numpy(), pool_rows=np.array(chosen), variant=np.array("sharp"))


CLI_PROMPT_ADJ = ("small big red blue green old young happy sad angry sleepy wet dry fluffy striped black white orange grey tiny giant "
                  "funny serious wild tame hungry lazy fast slow quiet loud").split()
CLI_PROMPT_NOUN = ("cat dog bat remote sofa couch blanket kitten puppy tiger lion mouse bird fish car tree house chair table phone laptop "
                   "book cup bottle shoe hat ball flower pizza cake river mountain beach city street window door keyboard guitar piano "
                   "bicycle train plane boat robot dragon wizard castle garden forest desert").split()
CLI_PROMPT_TMPL = ("a photo of a {a} {n}", "a {a} {n} on a couch", "the word {n} written on a {a} sign", "{a} {n}")


def hf_tokenize(ref_root, texts, context_length=80):
    """The reference's caption tokenizer (HuggingFace `tokenizers` BertWordPieceTokenizer on the reference vocabulary, framed as
    CustomTokenizer does: gen_tokenizer below)."""
    from tokenizers import BertWordPieceTokenizer
    tk = BertWordPieceTokenizer.from_file(os.path.join(ref_root, "assets", "bert_base_vocab_bos_eos.txt"))
    rows = []
    for t in texts:
        ids = tk.encode(" ".join(t.strip().split()), add_special_tokens=False).ids[:context_length - 3]
        enc = [1] + ids + [2]
        enc += [0] * (context_length - 1 - len(enc))
        rows.append(enc + [101])
    return torch.tensor(rows, dtype=torch.int64)


CLI_GAP = 0.12      # margin of the zero-shot CLI fixture (see gen_testcat_cli)


def gen_testcat_cli(m, ref_root, outdir):
    """End-to-end fixture of the zero-shot script (ov-zero-shot-test.py:37-56 loading, :167-195 per-image table, :198-208 best image
    per text) whose answer DEPENDS ON THE IMAGE.  The five testcat pictures are one photograph with different captions drawn on it:
    under any generic weights their embeddings sit within cos 0.94-0.99 of each other and every image ranks the prompts alike (the
    older tiny16_160_testcat.npz: five identical rows).  Here (a) the images are committed as 160 x 160 RGB PNG files
    (tests/golden/testcat_160/: the script's Resize((160, 160)) is then the identity, so file -> tensor is ToTensor + Normalize);
    (b) the weights are the 'sharp' formula weights with TWO matrices replaced, both committed in the fixture: visual.proj' = proj (I -
    0.95 c c^T), c = the unit mean of the five reference image embeddings, and text_projection' likewise with the unit mean of the
    prompt pool's embeddings -- a legitimate CLIP state dict that removes most of what the embeddings of a side share, so that they
    spread out (image-image cosines -0.44 .. 0.45 instead of 0.94 .. 0.99); (c) nine real prompts, tokenised by the `tokenizers` library on the reference
    vocabulary, are picked from a seeded pool such that in the REFERENCE's table every column's best image (by probability, as the
    script decides it, and by cosine) wins by >= CLI_GAP = 0.12 in cosine and a factor >= 1.5 in probability, the nine winners cover
    >= 4 different images, every row decides >= 3 leading ranks at margin CLI_GAP, and at least three rows differ in that decided
    prefix.  The margin is 0.12, not the 5e-2 of the other tables: the projections above amplify every error of the towers by the
    factor they spread the embeddings with -- the REFERENCE'S OWN bf16 mode moves this table by up to 0.036 in cosine (measured here,
    stored as `cos_err_ref_bf16`); the GPU test allows CLI_GAP / 2."""
    from PIL import Image
    cfg = ovcfg.preset("vit-tiny-patch16-160")
    model = build_ref(m, cfg, variant="sharp")
    pp = ovcfg.DEFAULT_PREPROCESS
    pdir = os.path.join(outdir, "testcat_160")
    os.makedirs(pdir, exist_ok=True)
    names = sorted(n for n in os.listdir(os.path.join(ref_root, "testcat")) if n.lower().endswith(".png"))
    arr = []
    for n in names:
        im = Image.open(os.path.join(ref_root, "testcat", n)).convert("RGB").resize((160, 160), Image.BILINEAR)
        im.save(os.path.join(pdir, n), optimize=True)
        im = Image.open(os.path.join(pdir, n)).convert("RGB").resize((160, 160), Image.BILINEAR)      # what the script does with the file
        a = (np.asarray(im, dtype=np.float32) / 255.0 - np.asarray(pp["mean"], np.float32)) / np.asarray(pp["std"], np.float32)
        arr.append(a.transpose(2, 0, 1))
    img = torch.from_numpy(np.stack(arr))
    prompts = [t.format(a=a, n=n) for t in CLI_PROMPT_TMPL for a in CLI_PROMPT_ADJ for n in CLI_PROMPT_NOUN]
    ids = hf_tokenize(ref_root, prompts)
    with torch.no_grad():
        f0 = torch.cat([model.encode_image(img[i:i + 1]) for i in range(img.shape[0])])
        c = torch.nn.functional.normalize(torch.nn.functional.normalize(f0, dim=-1).mean(0), dim=0)
        proj = model.visual.proj.detach().clone()
        proj = proj - 0.95 * (proj @ c)[:, None] * c[None, :]
        model.visual.proj.copy_(proj)
        fi = torch.cat([model.encode_image(img[i:i + 1]) for i in range(img.shape[0])])        # batch 1 per image, as the script
        fi = fi / fi.norm(dim=-1, keepdim=True)
        ft = torch.cat([model.encode_text(ids[i:i + 512]) for i in range(0, ids.shape[0], 512)])
        d = torch.nn.functional.normalize(torch.nn.functional.normalize(ft, dim=-1).mean(0), dim=0)
        tproj = model.text_projection.detach().clone()         # the same for the text side: all captions share most of their embedding
        tproj = tproj - 0.95 * (tproj @ d)[:, None] * d[None, :]
        model.text_projection.copy_(tproj)
        ft = torch.cat([model.encode_text(ids[i:i + 512]) for i in range(0, ids.shape[0], 512)])
        ft = ft / ft.norm(dim=-1, keepdim=True)
        scale = float(model.logit_scale.exp())
    cos_all = (fi @ ft.T).numpy().astype(np.float64)

    def table(cols):
        cs = cos_all[:, cols]
        e = np.exp(scale * (cs - cs.max(axis=1, keepdims=True)))
        return cs, e / e.sum(axis=1, keepdims=True)

    def column_ok(cs, pr):
        out = []
        for j in range(cs.shape[1]):
            o = np.argsort(-cs[:, j]); q = np.argsort(-pr[:, j])
            out.append(o[0] == q[0] and cs[o[0], j] - cs[o[1], j] >= CLI_GAP and pr[q[0], j] >= 1.5 * pr[q[1], j])
        return out

    # candidates: columns that win by cosine with margin on their own (the probability condition depends on the whole set)
    srt = np.sort(cos_all, axis=0)
    cand = np.nonzero(srt[-1] - srt[-2] >= CLI_GAP * 1.1)[0]
    rng = np.random.default_rng(20251005)

    def violations(cols):
        """0 = every condition holds; otherwise a count (+ fractional slack) the local search walks down."""
        cs, pr = table(cols)
        v = float(sum(not ok for ok in column_ok(cs, pr)))
        v += max(0, 4 - len(set(pr.argmax(axis=0).tolist())))
        for r in cs:
            srt_ = np.sort(r)[::-1]
            gaps = srt_[:3] - srt_[1:4]
            v += float(np.clip((CLI_GAP * 1.05 - gaps) / CLI_GAP, 0, 1).sum())
        v += max(0, 3 - len({tuple(np.argsort(-cs[r])[:3].tolist()) for r in range(cs.shape[0])}))
        return v

    best = None
    for _ in range(64):                                            # random restarts of a one-swap local search
        cols = rng.choice(cand, size=9, replace=False)
        v = violations(cols)
        for _ in range(4000):
            if v == 0:
                break
            trial = cols.copy()
            trial[rng.integers(9)] = rng.choice(cand)
            if len(set(trial.tolist())) < 9:
                continue
            tv = violations(trial)
            if tv <= v:
                cols, v = trial, tv
        if v == 0:
            cols = np.sort(cols)
            cs, pr = table(cols)
            ks = [leading_ranks(r, CLI_GAP) for r in cs]
            pref = {tuple(np.argsort(-cs[r])[:3].tolist()) for r in range(cs.shape[0])}
            score = (len(pref), min(ks), len(set(pr.argmax(axis=0).tolist())))
            if best is None or score > best[0]:
                best = (score, cols)
    if best is None:
        raise RuntimeError(f"no prompt set satisfies the fixture's conditions ({len(cand)} candidate columns)")
    cols = best[1]
    cs, pr = table(cols)
    chosen = [prompts[j] for j in cols]
    mb = build_ref(m, cfg, variant="sharp", cast_dtype=torch.bfloat16)          # the reference's own 'bf16' precision on the same table
    with torch.no_grad():
        mb.visual.proj.copy_(proj.to(mb.visual.proj.dtype))
        mb.text_projection.copy_(tproj.to(mb.text_projection.dtype))
        bi = torch.cat([mb.encode_image(img[i:i + 1].bfloat16()) for i in range(img.shape[0])]).float()
        bt = mb.encode_text(ids[cols]).float()
        err_bf16 = float(((bi / bi.norm(dim=-1, keepdim=True)) @ (bt / bt.norm(dim=-1, keepdim=True)).T - torch.from_numpy(cs).float()).abs().max())
    print("  testcat_cli: prompts", chosen, "score (distinct top-3 prefixes, min decided ranks, distinct winners)", best[0])
    np.savez_compressed(os.path.join(outdir, "tiny16_160_testcat_cli.npz"), names=np.array(names), prompts=np.array(chosen),
                        tokens=ids[cols].numpy(), visual_proj=f32(proj), text_projection=f32(tproj), cosine=cs.astype(np.float32), probs=pr.astype(np.float32),
                        argsort=np.argsort(-cs, axis=1), best_text_per_image=pr.argmax(axis=1), best_image_per_text=pr.argmax(axis=0),
                        cos_err_ref_bf16=np.float32(err_bf16), gap=np.float32(CLI_GAP), variant=np.array("sharp + visual.proj, text_projection from this file"))


def gen_sharp(m, lossmod, outdir):
    """Discriminating fixtures: 'sharp' formula weights + structured images (synth.make_structured_images), so that a wrong
    row, a wrong image or an input-independent encoder FAILS the 1e-3 cosine gate (on the v1 weights two different images sit
    at cos 0.992 of each other).  Per-block residual-stream slices are kept for the HIP path's block-level comparison."""
    def run(pname, size, nimg, blocks, seed, tail, text=True):
        cfg = ovcfg.preset(pname)
        model = build_ref(m, cfg, variant="sharp")
        img = synth.make_structured_images(nimg, size, seed=seed).half().float()      # stored as fp16
        tok = synth.make_captions(max(nimg, 4), 80, 32000, seed=seed)
        hs = tokens_after_blocks(model, img, blocks)
        res = {"images": img.numpy().astype(np.float16), "tokens": tok.numpy(), "variant": np.array("sharp")}
        with torch.no_grad():
            fi = model.encode_image(img)
            res["image_features"] = f32(fi)
            ni = fi / fi.norm(dim=-1, keepdim=True)
            res["image_image_cos"] = f32(ni @ ni.T)
            if text:
                ft = model.encode_text(tok)
                nt = ft / ft.norm(dim=-1, keepdim=True)
                res.update(text_features=f32(ft), text_text_cos=f32(nt @ nt.T), cosine=f32(ni @ nt.T),
                           argsort=(ni @ nt.T).argsort(dim=-1, descending=True).numpy())
                if nimg == tok.shape[0]:
                    a, b, s = model(img, tok)
                    res["loss"] = f32(lossmod.ClipLoss()(a, b, s))
        for i in blocks:
            res[f"block{i}_head"] = f32(hs[i][:, :4])
            res[f"block{i}_mid"] = f32(hs[i][:, 100:102])
            if tail:
                res[f"block{i}_tail"] = f32(hs[i][:, -2:])
        del model
        # the reference's own bf16 mode (factory.py:275-296) on the same inputs: its distance from the fp32 run IS the bf16 budget
        # of these weights, per block slice and for the embeddings
        mb = build_ref(m, cfg, cast_dtype=torch.bfloat16, variant="sharp")
        hb = tokens_after_blocks(mb, img.to(torch.bfloat16), blocks)
        for i in blocks:
            res[f"block{i}_head_refbf16"] = f32(hb[i][:, :4])
            res[f"block{i}_mid_refbf16"] = f32(hb[i][:, 100:102])
            res[f"block{i}_tail_refbf16"] = f32(hb[i][:, -2:])
        with torch.no_grad():
            res["image_features_refbf16"] = f32(mb.encode_image(img.to(torch.bfloat16)))
            if text:
                res["text_features_refbf16"] = f32(mb.encode_text(tok))
        off = res["image_image_cos"][~np.eye(nimg, dtype=bool)]
        print(f"  {pname}: max cos between different images {off.max():.4f}")
        return res

    np.savez_compressed(os.path.join(outdir, "tiny16_160_sharp.npz"), **run("vit-tiny-patch16-160", 160, 4, [0, 5, 11], 41, True))
    np.savez_compressed(os.path.join(outdir, "large14_224_sharp.npz"),
                        **run("vit-large-patch14-224", 224, 4, [0, 11, 23], 42, True))
    np.savez_compressed(os.path.join(outdir, "small8_384_sharp.npz"),
                        **run("vit-small-patch8-384", 384, 2, [0, 11], 43, True, text=False))


def gen_large(m, lossmod, out):
    cfg = ovcfg.preset("vit-large-patch14-224")
    model = build_ref(m, cfg)
    img = synth.make_images(2, 224, seed=21).half().float()     # stored as fp16: run the reference on the stored values
    tok = synth.make_captions(2, 80, 32000, seed=21)
    hs = tokens_after_blocks(model, img, [0, 23])
    with torch.no_grad():
        fi, ft = model.encode_image(img), model.encode_text(tok)
        ni, nt, s = model(img, tok)
        loss = lossmod.ClipLoss()(ni, nt, s)
    del model
    # the reference's own bf16 mode: bf16 Linear/Conv/MHA/proj, fp32 LN via LayerNormFp32 (model.py:143,396-423)
    mb = build_ref(m, cfg, cast_dtype=torch.bfloat16)
    with torch.no_grad():
        fib = mb.encode_image(img.to(torch.bfloat16))
        ftb = mb.encode_text(tok)
    np.savez_compressed(out, images=img.numpy().astype(np.float16), tokens=tok.numpy(),
                        block0_head=f32(hs[0][:, :4]), block23_head=f32(hs[23][:, :4]), block23_tail=f32(hs[23][:, -2:]),
                        image_features=f32(fi), text_features=f32(ft), image_norm=f32(ni), text_norm=f32(nt),
                        loss=f32(loss), image_features_refbf16=f32(fib), text_features_refbf16=f32(ftb))


def gen_small(m, out):
    cfg = ovcfg.preset("vit-small-patch8-384")
    model = build_ref(m, cfg)
    img = synth.make_images(1, 384, seed=31).half().float()
    hs = tokens_after_blocks(model, img, [0, 11])
    with torch.no_grad():
        fi = model.encode_image(img)
    np.savez_compressed(out, images=img.numpy().astype(np.float16), block0_head=f32(hs[0][:, :4]),
                        block11_head=f32(hs[11][:, :4]), block11_tail=f32(hs[11][:, -2:]), image_features=f32(fi))


def _loss_worker(rank, ws, store, ref_root, feats, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=ws)
    _, lossmod, _ = import_reference(ref_root)
    img, txt, s = feats
    b = img.shape[0] // ws
    fn = lossmod.ClipLoss(local_loss=True, rank=rank, world_size=ws)
    with torch.no_grad():
        l = fn(img[rank * b:(rank + 1) * b], txt[rank * b:(rank + 1) * b], s)
    q.put((rank, float(l)))
    dist.barrier()
    dist.destroy_process_group()


def gen_preprocess(ref_root, out):
    """Pillow (the library torchvision's Resize runs on a PIL image) as the reference of the image transform of
    ov-zero-shot-test.py:72-77 (Resize((S, S)) in the file's own mode, THEN convert("RGB"), ToTensor, Normalize) and of
    open_clip/transform.py:355-392 ('shortest' + CenterCrop, bicubic).  Inputs: the 5 testcat PNGs (RGB view; their alpha is
    255 everywhere, so resizing in RGBA and dropping alpha equals resizing the RGB view) and three synthetic images."""
    from PIL import Image
    pp = ovcfg.DEFAULT_PREPROCESS
    mean, std = np.asarray(pp["mean"], np.float32), np.asarray(pp["std"], np.float32)
    d = {}
    names = sorted(n for n in os.listdir(os.path.join(ref_root, "testcat")) if n.lower().endswith(".png"))
    for i, n in enumerate(names):
        im = Image.open(os.path.join(ref_root, "testcat", n))
        if im.mode == "RGBA":
            assert np.asarray(im)[..., 3].min() == 255
        d[f"cat{i}_in"] = np.asarray(im.convert("RGB"))
        r = im.resize((160, 160), Image.BILINEAR).convert("RGB")            # script order: resize, then convert
        d[f"cat{i}_u8"] = np.asarray(r)
        if i < 2:          # the float stage is elementwise: two images pin it, the uint8 stage is pinned on all five
            x = torch.from_numpy(np.asarray(r).copy()).permute(2, 0, 1).float().div(255)      # ToTensor
            d[f"cat{i}_out"] = ((x - torch.tensor(mean)[:, None, None]) / torch.tensor(std)[:, None, None]).numpy()   # Normalize
    g = np.random.default_rng(5)
    synth_imgs = {"noise": g.integers(0, 256, (97, 131, 3), dtype=np.uint8),
                  "grad": (np.add.outer(np.arange(300), np.arange(211))[..., None] * np.array([1, 2, 3]) % 256).astype(np.uint8),
                  "up": g.integers(0, 256, (40, 56, 3), dtype=np.uint8)}
    for k, img in synth_imgs.items():
        d[f"{k}_in"] = img
        h, w = img.shape[:2]
        # transform.py 'shortest': Resize(224, BICUBIC) on the short edge, CenterCrop(224)
        if w <= h:
            wr, hr = 224, int(224 * h / w)
        else:
            hr, wr = 224, int(224 * w / h)
        r = Image.fromarray(img).resize((wr, hr), Image.BICUBIC)
        top, left = int(round((hr - 224) / 2.0)), int(round((wr - 224) / 2.0))
        r = np.asarray(r)[top:top + 224, left:left + 224]
        d[f"{k}_u8_shortest_bicubic"] = r
        d[f"{k}_u8_squash_bilinear"] = np.asarray(Image.fromarray(img).resize((160, 160), Image.BILINEAR))
    np.savez_compressed(out, names=np.array(names), **d)


def gen_cliploss(lossmod, ref_root, out):
    import torch.multiprocessing as mp
    g = torch.Generator().manual_seed(99)
    n, e = 16, 192
    img = torch.nn.functional.normalize(torch.randn(n, e, generator=g), dim=-1)
    txt = torch.nn.functional.normalize(img * 0.6 + torch.randn(n, e, generator=g) * 0.1, dim=-1)
    s = torch.tensor(1.0 / 0.07)
    res = {"img": f32(img), "txt": f32(txt), "scale": f32(s),
           "loss_ws1": f32(lossmod.ClipLoss()(img, txt, s))}
    ctx = mp.get_context("spawn")
    for ws in (2, 8):
        with tempfile.TemporaryDirectory() as d:
            q = ctx.Queue()
            ps = [ctx.Process(target=_loss_worker, args=(r, ws, os.path.join(d, "store"), ref_root, (img, txt, s), q))
                  for r in range(ws)]
            [p.start() for p in ps]
            got = dict(q.get(timeout=600) for _ in range(ws))
            [p.join() for p in ps]
        res[f"local_losses_ws{ws}"] = np.array([got[r] for r in range(ws)], dtype=np.float64)
    np.savez_compressed(out, **res)


def _lossgrad_worker(rank, ws, store, ref_root, feats, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=ws)
    _, lossmod, _ = import_reference(ref_root)
    img, txt, s = feats
    b = img.shape[0] // ws
    res = {}
    for local_loss in (True, False):
        for gwg in (False, True):
            li = img[rank * b:(rank + 1) * b].clone().requires_grad_(True)
            lt = txt[rank * b:(rank + 1) * b].clone().requires_grad_(True)
            sc = s.clone().requires_grad_(True)
            fn = lossmod.ClipLoss(local_loss=local_loss, gather_with_grad=gwg, rank=rank, world_size=ws)
            try:
                l = fn(li, lt, sc)
                l.backward()
                res[(local_loss, gwg)] = (float(l), f32(li.grad), f32(lt.grad), float(sc.grad))
            except Exception as e:      # gloo has no all_to_all for torch.distributed.nn.all_gather's backward
                res[(local_loss, gwg)] = repr(e)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def gen_lossgrad(lossmod, ref_root, out):
    """Gradients of the reference ClipLoss (loss.py:19-131) by autograd: d loss / d (image_features, text_features,
    logit_scale) at world_size 1, and per rank at world_size 2 over gloo for local_loss x gather_with_grad."""
    import torch.multiprocessing as mp
    g = torch.Generator().manual_seed(7)
    n, e = 48, 192
    img = torch.nn.functional.normalize(torch.randn(n, e, generator=g), dim=-1)
    txt = torch.nn.functional.normalize(img * 0.5 + torch.randn(n, e, generator=g) * 0.08, dim=-1)
    s = torch.tensor(1.0 / 0.07)
    res = {"img": f32(img), "txt": f32(txt), "scale": f32(s)}
    a, b_, c = img.clone().requires_grad_(True), txt.clone().requires_grad_(True), s.clone().requires_grad_(True)
    l = lossmod.ClipLoss()(a, b_, c)
    l.backward()
    res.update(loss_ws1=f32(l), dimg_ws1=f32(a.grad), dtxt_ws1=f32(b_.grad), dscale_ws1=f32(c.grad))
    ws = 2
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as d:
        q = ctx.Queue()
        ps = [ctx.Process(target=_lossgrad_worker, args=(r, ws, os.path.join(d, "store"), ref_root, (img, txt, s), q))
              for r in range(ws)]
        [p.start() for p in ps]
        got = dict(q.get(timeout=600) for _ in range(ws))
        [p.join() for p in ps]
    for (local_loss, gwg) in got[0]:
        tag = f"ws2_local{int(local_loss)}_gwg{int(gwg)}"
        if isinstance(got[0][(local_loss, gwg)], str):
            print("  not captured:", tag, got[0][(local_loss, gwg)][:200])
            continue
        res[tag + "_loss"] = np.array([got[r][(local_loss, gwg)][0] for r in range(ws)], dtype=np.float64)
        res[tag + "_dimg"] = np.stack([got[r][(local_loss, gwg)][1] for r in range(ws)])
        res[tag + "_dtxt"] = np.stack([got[r][(local_loss, gwg)][2] for r in range(ws)])
        res[tag + "_dscale"] = np.array([got[r][(local_loss, gwg)][3] for r in range(ws)], dtype=np.float64)
    np.savez_compressed(out, **res)


def gen_opgrad(tr, out):
    """autograd through the reference's own modules (transformer.py:24-30 LayerNorm; the block's nn.Linear and nn.GELU,
    :232-236) for random upstream gradients: the pins of the operator-level backward restatements in oracle/clip_ref.py."""
    g = torch.Generator().manual_seed(4321)
    res = {}
    x = (torch.randn(37, 192, generator=g) * 1.5 + 0.2).requires_grad_(True)
    ln = tr.LayerNorm(192, eps=1e-6)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(192, generator=g) * 0.1 + 1)
        ln.bias.copy_(torch.randn(192, generator=g) * 0.1)
    dy = torch.randn(37, 192, generator=g)
    ln(x).backward(dy)
    res.update(ln_x=f32(x), ln_w=f32(ln.weight), ln_dy=f32(dy), ln_dx=f32(x.grad), ln_dw=f32(ln.weight.grad), ln_db=f32(ln.bias.grad))
    lin = torch.nn.Linear(128, 192)
    xl = torch.randn(70, 128, generator=g).requires_grad_(True)
    dyl = torch.randn(70, 192, generator=g)
    lin(xl).backward(dyl)
    res.update(lin_x=f32(xl), lin_w=f32(lin.weight), lin_dy=f32(dyl), lin_dx=f32(xl.grad), lin_dw=f32(lin.weight.grad),
               lin_db=f32(lin.bias.grad))
    # the block's attention module (transformer.py:225: nn.MultiheadAttention(d_model, n_head, batch_first=True)) with an identity
    # out-projection and zero biases: d x = dq Wq + dk Wk + dv Wv isolates the softmax-attention backward
    mha = torch.nn.MultiheadAttention(128, 2, batch_first=True)
    with torch.no_grad():
        mha.in_proj_weight.copy_(torch.randn(384, 128, generator=g) * 128 ** -0.5)
        mha.in_proj_bias.zero_()
        mha.out_proj.weight.copy_(torch.eye(128))
        mha.out_proj.bias.zero_()
    xa = torch.randn(2, 37, 128, generator=g).requires_grad_(True)
    dya = torch.randn(2, 37, 128, generator=g)
    mha(xa, xa, xa, need_weights=False)[0].backward(dya)
    res.update(mha_x=f32(xa), mha_w=f32(mha.in_proj_weight), mha_dy=f32(dya), mha_dx=f32(xa.grad), mha_dw=f32(mha.in_proj_weight.grad))
    for name, act in (("erf", torch.nn.GELU()), ("tanh", torch.nn.GELU(approximate="tanh"))):
        a = torch.linspace(-6, 6, 385).requires_grad_(True)
        dh = torch.randn(385, generator=g)
        act(a).backward(dh)
        res.update({f"gelu_{name}_a": f32(a), f"gelu_{name}_dh": f32(dh), f"gelu_{name}_da": f32(a.grad)})
    np.savez_compressed(out, **res)


def gen_blockgrad(tr, out):
    """autograd through the reference ResidualAttentionBlock (transformer.py:210-265) of the Tiny vision tower, formula weights
    of block 0, a random upstream gradient: d input and every parameter gradient (weight matrices: first 6 rows + Frobenius norm)."""
    g = torch.Generator().manual_seed(777)
    cfg = ovcfg.preset("vit-tiny-patch16-160")
    sd = synth.make_state_dict(cfg, 0)
    blk = tr.ResidualAttentionBlock(192, 3, 4.0, batch_first=True, norm_layer=lambda d: tr.LayerNorm(d, eps=1e-6))
    p = "visual.transformer.resblocks.0."
    blk.load_state_dict({k[len(p):]: v for k, v in sd.items() if k.startswith(p)}, strict=True)
    x = torch.randn(2, 101, 192, generator=g).requires_grad_(True)
    dy = torch.randn(2, 101, 192, generator=g)
    blk(x).backward(dy)
    res = {"x": f32(x), "dy": f32(dy), "dx": f32(x.grad)}
    for name, prm in blk.named_parameters():
        gr = prm.grad
        if gr.dim() == 2:
            res["g." + name + ".head"] = f32(gr[:6])
            res["g." + name + ".norm"] = np.float64(gr.double().norm())
        else:
            res["g." + name] = f32(gr)
    np.savez_compressed(out, **res)


def gen_tokenizer(ref_root, out):
    """Token ids of the reference's caption tokenizer: HuggingFace `tokenizers` BertWordPieceTokenizer (the third-party library
    CustomTokenizer wraps, tokenizer.py:522-550) on the reference's vocabulary asset, framed as CustomTokenizer.tokenize /
    pad_and_add_class_token do (bos 1, at most 77 pieces, eos 2, zero padding, class token 101 last).  The vendored tokenizer
    module itself cannot be imported (ftfy missing), so its five framing lines are restated here; captions are chosen so that
    ftfy.fix_text would leave them unchanged."""
    import random
    from tokenizers import BertWordPieceTokenizer
    vocab = os.path.join(ref_root, "assets", "bert_base_vocab_bos_eos.txt")
    tk = BertWordPieceTokenizer.from_file(vocab)
    words = [l.rstrip("\n") for l in open(vocab, encoding="utf-8")]
    texts = ["a photo of a cat", "a photo of a dog", "A Photo Of A CAT.", "a cat sitting on a sofa, looking at the camera",
             "an image of two cats and a remote control", "the quick brown fox jumps over the lazy dog", "", "   ",
             "Hello, world! It's 5:30pm -- really?", "don't can't won't i'm", "naive cafe resume", "na\u00efve caf\u00e9 r\u00e9sum\u00e9",
             "\u00dcber stra\u00dfe \u00e5ngstr\u00f6m", "price: $5.00 + 10% tax = $5.50", "x_y-z/w\\v a|b c^d e~f", "e-mail: someone@example.com",
             "\u65e5\u672c\u8a9e\u306e\u30c6\u30ad\u30b9\u30c8", "a photo of a \u732b and a \u72ac", "\ud55c\uad6d\uc5b4 \ud14d\uc2a4\ud2b8", "\u0440\u0443\u0441\u0441\u043a\u0438\u0439 \u0442\u0435\u043a\u0441\u0442",
             "\u0395\u03bb\u03bb\u03b7\u03bd\u03b9\u03ba\u03ac \u039f\u0394\u039f\u03a3", "emoji \U0001f600 test", "tab\tseparated\nlines\r\nend", "multiple     spaces   here",
             "supercalifragilisticexpialidocious", "a" * 120, "pneumonoultramicroscopicsilicovolcanoconiosis " * 3,
             "word " * 100, "unaffable unbelievably antidisestablishmentarianism", "[UNK] [SEP] [CLS] [MASK] [PAD] literal specials",
             "&amp; &lt;tag&gt; &amp;amp; html entities", "1234567890 3.14159 1,000,000", "UPPER lower MiXeD CaSe", "(parentheses) [brackets] {braces}"]
    rnd = random.Random(2024)
    for _ in range(160):                                    # vocabulary words in random order: long and short captions
        n = rnd.randint(1, 60)
        texts.append(" ".join(rnd.choice(words).replace("##", "") for _ in range(n)))
    clean_ = lambda t: " ".join(html.unescape(html.unescape(t)).strip().split()).strip()     # _clean_whitespace minus ftfy
    rows = []
    for t in texts:
        ids = tk.encode(clean_(t), add_special_tokens=False).ids[:80 - 3]
        enc = [1] + ids + [2]
        if len(enc) < 80 - 1:
            enc += [0] * (80 - 1 - len(enc))
        rows.append(enc + [101])
    np.savez_compressed(out, texts=np.array(texts), ids=np.array(rows, dtype=np.int64))


def gen_eval(ref_root, out):
    """Evaluator arithmetic from the reference's own modules: recall@k of src/evaluators/proj/image_text/image_text_retrieval.py
    (numpy-only, loaded from its file) on a seeded distance matrix without ties, and the classifier weights of
    open_clip/zero_shot_classifier.py:21-68 (torch-only, through the bare open_clip package) driven by a table-lookup stand-in for
    the model / tokenizer pair (encode_text = rows of a seeded, normalised embedding table; the arithmetic under test is the
    reshape / mean / renormalise / transpose / batch-concatenate of :51-66)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_image_text_retrieval", os.path.join(ref_root, "src/evaluators/proj/image_text/image_text_retrieval.py"))
    itr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(itr)
    g = np.random.default_rng(2024)
    n_img, per = 48, 5
    corr = np.repeat(np.arange(n_img), per)
    zi = g.standard_normal((n_img, 64)).astype(np.float32)
    zt = (zi[corr] * 0.22 + g.standard_normal((n_img * per, 64)).astype(np.float32)).astype(np.float32)
    zi /= np.linalg.norm(zi, axis=1, keepdims=True)
    zt /= np.linalg.norm(zt, axis=1, keepdims=True)
    dist = (-(zi @ zt.T)).astype(np.float32)                    # retrieval.py:280-303 feeds -similarity
    # no ties within a row or a column: both argsort orders are unambiguous
    assert all(len(np.unique(r)) == r.size for r in dist) and all(len(np.unique(c)) == c.size for c in dist.T)
    t2i = itr.text_to_image_retrieval_eval(dist, list(corr))
    i2t = itr.image_to_text_retrieval_eval(dist, list(corr))
    res = {"zimg": zi, "ztxt": zt, "dist": dist, "corr": corr.astype(np.int64),
           "t2i": np.array([t2i[f"Recall@{k}"] for k in itr.RECALL_THRESHOLDS], dtype=np.float64),
           "i2t": np.array([i2t[f"Recall@{k}"] for k in itr.RECALL_THRESHOLDS], dtype=np.float64),
           "thresholds": np.array(itr.RECALL_THRESHOLDS)}
    zsc = importlib.import_module("open_clip.zero_shot_classifier")
    C_, T_, E_ = 23, 7, 64
    table = torch.from_numpy(g.standard_normal((C_ * T_, E_)).astype(np.float32))

    class Model:
        def encode_text(self, ids, normalize=False):
            x = table[ids]
            return torch.nn.functional.normalize(x, dim=-1) if normalize else x

    names = [str(c) for c in range(C_)]
    templates = [(lambda c, t=t: f"{c}:{t}") for t in range(T_)]
    tokenizer = lambda texts: torch.tensor([int(x.split(":")[0]) * T_ + int(x.split(":")[1]) for x in texts])
    w = zsc.build_zero_shot_classifier(Model(), tokenizer, names, templates, num_classes_per_batch=10)
    res.update(zs_text_norm=torch.nn.functional.normalize(table, dim=-1).numpy(), zs_weights=w.numpy(),
               zs_classes=np.int64(C_), zs_templates=np.int64(T_))
    np.savez_compressed(out, **res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    m, lossmod, tr = import_reference(a.ref)
    jobs = {
        "ops": lambda: gen_ops(m, tr, os.path.join(HERE, "ops.npz")),
        "tiny": lambda: gen_tiny(m, lossmod, os.path.join(HERE, "tiny16_160.npz")),
        "testcat": lambda: gen_testcat(m, a.ref, os.path.join(HERE, "tiny16_160_testcat.npz")),
        "testcat_cli": lambda: gen_testcat_cli(m, a.ref, HERE),
        "large": lambda: gen_large(m, lossmod, os.path.join(HERE, "large14_224.npz")),
        "sharp": lambda: gen_sharp(m, lossmod, HERE),
        "eval": lambda: gen_eval(a.ref, os.path.join(HERE, "eval.npz")),
        "small": lambda: gen_small(m, os.path.join(HERE, "small8_384.npz")),
        "cliploss": lambda: gen_cliploss(lossmod, a.ref, os.path.join(HERE, "cliploss_ws.npz")),
        "preprocess": lambda: gen_preprocess(a.ref, os.path.join(HERE, "preprocess.npz")),
        "opgrad": lambda: gen_opgrad(tr, os.path.join(HERE, "opgrad.npz")),
        "blockgrad": lambda: gen_blockgrad(tr, os.path.join(HERE, "blockgrad.npz")),
        "tokenizer": lambda: gen_tokenizer(a.ref, os.path.join(HERE, "tokenizer.npz")),
        "lossgrad": lambda: gen_lossgrad(lossmod, a.ref, os.path.join(HERE, "cliploss_grad.npz")),
    }
    for k, fn in jobs.items():
        if a.only and k not in a.only.split(","):
            continue
        print("generating", k, flush=True)
        fn()


if __name__ == "__main__":
    main()
