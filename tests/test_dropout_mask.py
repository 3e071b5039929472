"""Statistics of the factored attention-dropout mask of csrc/k_attn_train.hip (drop_row_word / drop_col_word /
drop_keep), restated in numpy: element (q, k) is kept when the low 32 bits of row_word(q) * col_word(k) (two odd
24-bit words) reach p * 2^32.  The reference draws the mask from torch's Philox stream
(torch.nn.MultiheadAttention(dropout=p), reference src/models/blocks/encoders.py:44-55); what must carry over is
the distribution: keep rate 1-p and no correlation along rows, columns or across 2x2 rectangles."""
import numpy as np
import pytest

G = 0x9E3779B1


def mix32(x):
    x = x.astype(np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def words(seed, ph, tokens):
    idx = (np.arange(tokens, dtype=np.uint64) + ph * tokens) & 0xFFFFFFFF
    row = (mix32((idx * G) & 0xFFFFFFFF ^ seed) >> 8) | 1
    cseed = ((~np.uint64(seed) & 0xFFFFFFFF) * 0x632BE5AB + 0x7F4A7C15) & 0xFFFFFFFF
    col = (mix32((idx * G) & 0xFFFFFFFF ^ cseed) >> 8) | 1
    return row, col


@pytest.mark.parametrize("p", [0.1, 0.25, 0.5])
@pytest.mark.parametrize("seed,ph", [(1, 0), (99, 3), (12345, 1023)])
def test_factored_mask_is_uniform_and_uncorrelated(p, seed, ph):
    T = 560
    row, col = words(seed, ph, T)
    keep = ((row[:, None] * col[None, :]) & 0xFFFFFFFF) >= int(p * 2 ** 32)
    n = T * T
    sd = np.sqrt(p * (1 - p) / n)
    assert abs(keep.mean() - (1 - p)) < 5 * sd
    # row and column keep rates scatter like independent draws
    sd_line = np.sqrt(p * (1 - p) / T)
    assert 0.85 < keep.mean(1).std() / sd_line < 1.15
    assert 0.85 < keep.mean(0).std() / sd_line < 1.15
    k = keep.astype(np.float64) - (1 - p)
    var = p * (1 - p)
    tol = 5 / np.sqrt(n)
    assert abs((k[:, 1:] * k[:, :-1]).mean() / var) < tol          # neighbouring keys
    assert abs((k[1:] * k[:-1]).mean() / var) < tol                # neighbouring queries
    assert abs((k[1:, 1:] * k[:-1, :-1] * k[1:, :-1] * k[:-1, 1:]).mean() / var ** 2) < 3 * tol   # 2x2 rectangles


def test_masks_of_different_heads_and_seeds_differ():
    T = 128
    def mask(seed, ph):
        r, c = words(seed, ph, T)
        return ((r[:, None] * c[None, :]) & 0xFFFFFFFF) >= int(0.5 * 2 ** 32)
    a, b, c = mask(7, 0), mask(7, 1), mask(8, 0)
    for x in (b, c):
        agree = (a == x).mean()
        assert abs(agree - 0.5) < 0.02
