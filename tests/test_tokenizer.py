"""The caption tokenizer (openvision_amd.tokenizer: BERT WordPiece restated in Python) against token ids produced by the library
the reference calls (HuggingFace tokenizers) on the reference's vocabulary with the reference's framing: bit-exact."""
import numpy as np
import torch

from conftest import golden
from openvision_amd.tokenizer import WordPieceTokenizer, clean, normalize, pre_tokenize


def test_token_ids_match_reference_library_bit_exact():
    g = golden("tokenizer.npz")
    tok = WordPieceTokenizer()
    texts = [str(t) for t in g["texts"]]
    got = tok(texts)
    assert got.dtype == torch.long and tuple(got.shape) == (len(texts), 80)
    want = torch.from_numpy(g["ids"])
    bad = (got != want).any(dim=1).nonzero().flatten().tolist()
    assert not bad, [(texts[i], got[i].tolist()[:12], want[i].tolist()[:12]) for i in bad[:3]]


def test_framing_and_edge_cases():
    tok = WordPieceTokenizer()
    row = tok("a photo of a cat")[0].tolist()
    assert row[0] == 1 and row[-1] == 101 and row[6] == 2 and all(v == 0 for v in row[7:-1])          # bos, 5 pieces, eos, pad, class
    empty = tok("")[0].tolist()
    assert empty[:2] == [1, 2] and empty[-1] == 101 and sum(empty[2:-1]) == 0
    long = tok("word " * 200)[0].tolist()
    assert len(long) == 80 and long[0] == 1 and long[78] == 2 and long[79] == 101 and 0 not in long     # 77 pieces kept
    assert tok(["x", "y z"]).shape == (2, 80)
    assert tok("a" * 120)[0, 1].item() == tok.unk_id                                                   # > 100 characters: [UNK]
    assert clean("  a &amp;amp; b \n c ") == "a & b c"
    assert normalize("Café\tOΔΟΣ") == "cafe oδοσ"                     # accents stripped, per-character lower case
    assert pre_tokenize("it's (ok)") == ["it", "'", "s", "(", "ok", ")"]
    assert tok("x", context_length=16).shape == (1, 16)


def test_truncation_follows_the_inner_tokenizer_not_the_outer_context_length():
    """CLIPS_Tokenizer hard-wires its inner CustomTokenizer to context_length 80 (tokenizer.py:564): captions are cut to 77 pieces
    whatever the outer context_length; a shorter outer length then yields a ragged row, which raises (torch.tensor on ragged lists)."""
    import pytest
    from openvision_amd.tokenizer import WordPieceTokenizer
    tk = WordPieceTokenizer(context_length=128)
    row = tk(["word " * 200])[0]
    assert row.shape[0] == 128 and int((row != 0).sum()) == 77 + 3          # bos + 77 pieces + eos + class token, padded to 128
    assert int(row[-1]) == 101 and int(row[78]) == 2
    with pytest.raises(ValueError):
        WordPieceTokenizer(context_length=64)(["word " * 200])
