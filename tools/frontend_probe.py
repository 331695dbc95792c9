"""Timings of the front-end / evaluator kernels with their CPU counterparts beside them:
  * ov_preprocess_image on 64 images of 640 x 480 -> 224 x 224 (device-resident uint8 inputs) vs Pillow resize + numpy normalise
  * ov_topk k = 10 on a [8192, 8192] fp32 similarity matrix (retrieval recall@k) vs torch.topk on the CPU
  * ov_class_mean_normalize on 1000 classes x 7 templates x 768 (zero-shot classifier weights) vs torch on the CPU."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openvision_amd import _lib
from openvision_amd.preprocess import preprocess
from openvision_amd import evaluate


def gpu_time(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8) for _ in range(64)]
dev_imgs = [torch.from_numpy(a).to("cuda:0") for a in imgs]
t = gpu_time(lambda: preprocess(dev_imgs, 224))
nbytes = 64 * (480 * 640 * 3 + 224 * 224 * 3 * 4)
print(f"preprocess 64 x (480x640 -> 224x224, fp32 out): {t * 1e3:.2f} ms = {64 / t:.0f} img/s, {nbytes / t / 1e9:.1f} GB/s of source + output bytes "
      f"(per-image launches: launch-bound at this size)")
from PIL import Image
mean, std = np.array([0.48145466, 0.4578275, 0.40821073], np.float32), np.array([0.26862954, 0.26130258, 0.27577711], np.float32)
t0 = time.perf_counter()
for a in imgs:
    r = np.asarray(Image.fromarray(a).resize((224, 224), Image.BILINEAR), dtype=np.float32) / 255.0
    r = ((r - mean) / std).transpose(2, 0, 1)
tc = time.perf_counter() - t0
print(f"   CPU (Pillow resize + numpy normalise, 1 core): {tc * 1e3:.1f} ms = {64 / tc:.0f} img/s")

x = torch.randn(8192, 8192, device="cuda:0")
t = gpu_time(lambda: evaluate.topk(x, 10))
print(f"topk k=10 on [8192, 8192] fp32: {t * 1e3:.2f} ms = {x.numel() * 4 / t / 1e9:.0f} GB/s of the matrix")
xc = x.cpu()
t0 = time.perf_counter(); torch.topk(xc, 10, dim=1); tc = time.perf_counter() - t0
print(f"   CPU torch.topk ({torch.get_num_threads()} threads): {tc * 1e3:.1f} ms")

lib = _lib.load()
emb = torch.nn.functional.normalize(torch.randn(1000 * 7, 768, device="cuda:0"), dim=-1)
out = torch.empty(1000, 768, device="cuda:0")
fn = lambda: _lib.check(lib.ov_class_mean_normalize(_lib.ptr(emb), _lib.ptr(out), 1000, 7, 768, _lib.stream_ptr()), "ov_class_mean_normalize")
t = gpu_time(fn)
print(f"class_mean_normalize 1000 x 7 x 768: {t * 1e6:.1f} us = {emb.numel() * 4 / t / 1e9:.0f} GB/s")
ec = emb.cpu()
t0 = time.perf_counter(); torch.nn.functional.normalize(ec.view(1000, 7, 768).mean(1), dim=-1); tc = time.perf_counter() - t0
print(f"   CPU torch: {tc * 1e6:.0f} us")
