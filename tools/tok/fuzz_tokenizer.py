"""Fuzz openvision_amd.tokenizer against HuggingFace tokenizers' BertWordPieceTokenizer on the same vocabulary (CPU; needs the
`tokenizers` package, which is what the reference's CustomTokenizer calls).  Usage: python tools/tok/fuzz_tokenizer.py [seed] [n]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openvision_amd.tokenizer import WordPieceTokenizer, clean, DEFAULT_VOCAB
from tokenizers import BertWordPieceTokenizer

ref = BertWordPieceTokenizer.from_file(DEFAULT_VOCAB)
mine = WordPieceTokenizer()
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
words = [l.rstrip("\n") for l in open(DEFAULT_VOCAB, encoding="utf-8")]
ranges = [(0x20, 0x7e), (0xa0, 0x24f), (0x370, 0x3ff), (0x400, 0x4ff), (0x590, 0x6ff), (0x900, 0x97f), (0x2000, 0x206f),
          (0x2190, 0x21ff), (0x3040, 0x30ff), (0x4e00, 0x4fff), (0xac00, 0xacff), (0xff00, 0xffef), (0x1f600, 0x1f64f), (0, 0x1f),
          (0x7f, 0x9f), (0xe000, 0xe010), (0x300, 0x36f), (0xfe00, 0xfe0f), (0x1d400, 0x1d4ff), (0x2e80, 0x2fdf), (0x1100, 0x11ff),
          (0xe0000, 0xe007f)]
seps = [" ", "  ", chr(9), chr(10), ".", ",", "!", "-", "'", '"', "(", ")", chr(0xa0), chr(0x3000), chr(0x200b), chr(0x2003),
        chr(0x85), chr(0xb), chr(0xc), chr(0x1c), chr(0x1f)]


def rnd_text():
    parts = []
    for _ in range(random.randint(0, 12)):
        k = random.random()
        if k < 0.5:
            parts.append(random.choice(words).replace("##", ""))
        elif k < 0.8:
            lo, hi = random.choice(ranges)
            parts.append("".join(chr(random.randint(lo, hi)) for _ in range(random.randint(1, 6))))
        else:
            parts.append(random.choice(seps))
        if random.random() < 0.6:
            parts.append(" ")
    return "".join(parts)


bad = n = 0
for _ in range(N):
    t = rnd_text()
    if any(0xd800 <= ord(c) <= 0xdfff for c in t):
        continue
    n += 1
    a = ref.encode(clean(t), add_special_tokens=False).ids
    b = mine.encode(clean(t))
    if a != b:
        bad += 1
        if bad <= 6:
            print("MISMATCH", [hex(ord(c)) for c in clean(t)][:40], a[:14], b[:14])
print("checked", n, "mismatches", bad)
