"""Caption tokenizer of the OpenVision checkpoints (text half): the reference's ``CLIPS_Tokenizer`` /
``CustomTokenizer`` (``src/convert_upload/open_clip/tokenizer.py:522-594``) — BERT WordPiece, lower-cased, over the vocabulary
``assets/bert_base_vocab_bos_eos.txt`` of the reference (a data file, shipped here unchanged as
``openvision_amd/assets/bert_base_vocab_bos_eos.txt``; the hub's ``vocab.txt`` of a checkpoint directory can be passed instead),
framed as ``[bos=1] + pieces[: L - 3] + [eos=2]``, zero-padded to ``L - 1`` and closed by the class token ``101`` (:534-550).

The WordPiece arithmetic itself lives in a third-party dependency of the reference, HuggingFace ``tokenizers``
(``BertWordPieceTokenizer(lowercase=True)``: BertNormalizer -> BertPreTokenizer -> WordPiece); its published algorithm is
restated here in plain Python (host-side string work: there is nothing for the GPU in it) and pinned bit-exactly against that
library's output on the reference's vocabulary (``tests/golden/tokenizer.npz``).

Not reproduced: the ``ftfy.fix_text`` step of the reference's text cleaning (``tokenizer.py:67-70``; ftfy is not installed here,
so it cannot be pinned).  ``clean`` applies the remaining steps (``html.unescape`` twice, strip, whitespace collapse); captions that
ftfy would rewrite (mojibake, curly quotes, full-width forms, ligatures) may therefore tokenize differently.
"""
from __future__ import annotations

import html
import os
import re
import unicodedata
from typing import Dict, List, Optional, Sequence, Union

import torch

DEFAULT_VOCAB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "bert_base_vocab_bos_eos.txt")
_MAX_WORD_CHARS = 100          # WordPiece max_input_chars_per_word


def clean(text: str) -> str:
    """_clean_whitespace (tokenizer.py:67-89) without the ftfy step."""
    text = html.unescape(html.unescape(text)).strip()
    return " ".join(text.split()).strip()


def _is_whitespace(c: str) -> bool:
    return c in "\t\n\r" or c.isspace() and unicodedata.category(c) in ("Zs", "Zl", "Zp") or c in "\x0b\x0c\x85"


def _is_control(c: str) -> bool:
    if c in "\t\n\r":
        return False
    return unicodedata.category(c) in ("Cc", "Cf", "Co")        # unassigned code points (Cn) are kept, as the library keeps them


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or
            0x2B740 <= cp <= 0x2B81F or 0x2B920 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


def _is_punct(c: str) -> bool:
    cp = ord(c)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(c).startswith("P")


def normalize(text: str) -> str:
    """BertNormalizer(clean_text, handle_chinese_chars, strip_accents (follows lowercase), lowercase)."""
    out = []
    for c in text:
        if c == "\x00" or c == "�" or _is_control(c):
            continue
        out.append(" " if _is_whitespace(c) else c)
    text = "".join(out)
    out = []
    for c in text:
        if _is_cjk(ord(c)):
            out.extend((" ", c, " "))
        else:
            out.append(c)
    text = "".join(out)
    text = "".join(c for c in unicodedata.normalize("NFD", text) if unicodedata.category(c) != "Mn")
    return "".join(c.lower() for c in text)          # per character, as the library does (no final-sigma context rule)


def pre_tokenize(text: str) -> List[str]:
    """BertPreTokenizer: whitespace split, every punctuation character a word of its own."""
    words: List[str] = []
    for chunk in text.split():
        cur = []
        for c in chunk:
            if _is_punct(c):
                if cur:
                    words.append("".join(cur))
                    cur = []
                words.append(c)
            else:
                cur.append(c)
        if cur:
            words.append("".join(cur))
    return words


INNER_CONTEXT_LENGTH = 80      # CustomTokenizer(vocab_file, context_length=80, ...) inside CLIPS_Tokenizer (tokenizer.py:564)


class WordPieceTokenizer:
    """Drop-in for the reference's ``CLIPS_Tokenizer``: ``tok(texts) -> LongTensor [N, context_length]``."""

    def __init__(self, vocab_file: Optional[str] = None, context_length: int = 80, bos_token: int = 1, eos_token: int = 2,
                 class_token: int = 101, pad_token: int = 0, unk_token: str = "[UNK]", prefix: str = "##"):
        self.vocab: Dict[str, int] = {}
        with open(vocab_file or DEFAULT_VOCAB, "r", encoding="utf-8") as f:
            for i, line in enumerate(f):
                self.vocab[line.rstrip("\n")] = i
        if unk_token not in self.vocab:
            raise ValueError(f"vocabulary has no {unk_token} entry")
        self.unk_id, self.prefix = self.vocab[unk_token], prefix
        # the library registers these as special tokens: matched verbatim in the raw text, never normalised or split
        special = [t for t in (unk_token, "[SEP]", "[CLS]", "[PAD]", "[MASK]") if t in self.vocab]
        self._special = re.compile("(" + "|".join(re.escape(t) for t in special) + ")")
        self.context_length = context_length
        self.bos_token, self.eos_token, self.class_token, self.pad_token = bos_token, eos_token, class_token, pad_token

    def _wordpiece(self, word: str) -> List[int]:
        if len(word) > _MAX_WORD_CHARS:
            return [self.unk_id]
        ids, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:                       # greedy longest match first
                piece = word[start:end] if start == 0 else self.prefix + word[start:end]
                if piece in self.vocab:
                    cur = self.vocab[piece]
                    break
                end -= 1
            if cur is None:
                return [self.unk_id]                 # one unknown piece makes the whole word unknown
            ids.append(cur)
            start = end
        return ids

    def encode(self, text: str) -> List[int]:
        """Word-piece ids of one caption, no framing (``encode(text, add_special_tokens=False).ids``)."""
        ids: List[int] = []
        for part in self._special.split(text):
            if part in self.vocab and self._special.fullmatch(part):
                ids.append(self.vocab[part])
                continue
            for w in pre_tokenize(normalize(part)):
                ids.extend(self._wordpiece(w))
        return ids

    def frame(self, ids: Sequence[int], context_length: int) -> List[int]:
        """CustomTokenizer.tokenize + pad_and_add_class_token (tokenizer.py:534-550); the reference truncates to its own
        context_length (80: hard-wired where CLIPS_Tokenizer builds it, tokenizer.py:564) - 3 pieces whatever the OUTER context
        length is, and pads to ``max_length - 1``."""
        out = [self.bos_token] + list(ids[: INNER_CONTEXT_LENGTH - 3]) + [self.eos_token]
        if len(out) < context_length - 1:
            out += [self.pad_token] * (context_length - 1 - len(out))
        return out + [self.class_token]

    def __call__(self, texts: Union[str, Sequence[str]], context_length: Optional[int] = None) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        length = context_length or self.context_length
        rows = [self.frame(self.encode(clean(t)), length) for t in texts]
        if any(len(r) != length for r in rows):
            raise ValueError("context_length shorter than the framed caption (the reference raises on ragged rows as well)")
        return torch.tensor(rows, dtype=torch.long)
