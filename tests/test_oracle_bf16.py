"""Pins the oracle's bf16 restatement (BASELINE configs[3]: half_precision.rs — `half::bf16::from_f32`, dot_product :199-233,
cosine_similarity :237-254, norm_squared :290-311) against the reference's own tests (half_precision_tests.rs, transcribed as data
with file:line) and against independent restatements: the `half` crate's round-bit rule written out differently from the oracle's
add-and-shift, and the sequential f32 sums as plain Python loops.  CPU only — tests/test_gpu_bf16.py compares the HIP kernels with
this oracle."""
import numpy as np
import pytest

from oracle import pyoracle as po

F = np.float32


def half_crate_bf16_bits(x32: np.ndarray) -> np.ndarray:
    """half 2.x `bf16::from_f32` (pinned by the reference's Cargo.lock): NaN keeps its sign / payload top bits and gets the quiet
    bit; otherwise round to nearest even by the round-bit / sticky rule: round up iff bit 15 is set and (any lower bit or bit 16) is."""
    u = x32.view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    round_bit = np.uint64(0x8000)
    up = ((u & round_bit) != 0) & ((u & (3 * round_bit - 1)) != 0)
    r = (u >> 16) + up.astype(np.uint64)
    r = np.where(nan, (u >> 16) | 0x0040, r)
    return (r & 0xFFFF).astype(np.uint16)


def bf16_to_f32(bits16: np.ndarray) -> np.ndarray:
    return (bits16.astype(np.uint32) << 16).view(np.float32)


def test_round_bf16_matches_the_half_crate_rule_bit_for_bit():
    rng = np.random.default_rng(21)
    pats = rng.integers(0, 2 ** 32, 200_000, dtype=np.uint64).astype(np.uint32)
    special = np.array([0x00000000, 0x80000000, 0x7F800000, 0xFF800000, 0x7FC00000, 0x7F800001, 0xFFFFFFFF,   # zeros, infs, NaNs
                        0x00000001, 0x00008000, 0x00018000, 0x00010000, 0x807FFFFF,                             # denormals
                        0x3F808000, 0x3F818000, 0x3F808001, 0x3F807FFF,                                         # ties: to even, up, just above, below
                        0x7F7FFFFF, 0x7F7F8000, 0x7F7F7FFF], dtype=np.uint32)                                   # the largest finite values: up to inf / stays
    x = np.concatenate([pats, special]).view(np.float32)
    got = po.round_bf16(x)
    exp = bf16_to_f32(half_crate_bf16_bits(x))
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    assert np.all((got.view(np.uint32) & 0xFFFF) == 0)                      # a bf16 value: the low 16 bits are clear
    fin = np.isfinite(x) & np.isfinite(got) & (np.abs(x) > 1e-30)
    assert np.max(np.abs(got[fin] - x[fin]) / np.abs(x[fin])) <= 2.0 ** -8   # half an ulp of 8 significant bits
    assert np.isinf(po.round_bf16(np.array([3.4e38], dtype=F)))[0]          # rounds up past the largest bf16
    assert np.isnan(po.round_bf16(np.array([np.nan], dtype=F)))[0]


def test_reference_roundtrip_and_dot_kats():
    orig = np.array([0.1, 0.5, 1.0, -0.5, 0.0], dtype=F)                    # half_precision_tests.rs:116-128
    back = po.round_bf16(orig)
    assert np.all(np.abs(orig - back) < 0.01)
    assert back[1:].tolist() == [0.5, 1.0, -0.5, 0.0]                       # exactly representable values come back unchanged
    assert back[0] == F(0.10009765625)                                      # 0.1 -> 0x3DCD
    # :184-193: [1, 2, 3] . [4, 5, 6] = 32 within 0.5 (all six values are exact in bf16: the oracle returns 32 exactly)
    ids, sc = po.scan_topk_bf16(po.DOT, np.array([[4.0, 5.0, 6.0]], dtype=F), np.array([1.0, 2.0, 3.0], dtype=F), 1)
    assert ids[0, 0] == 0 and sc[0, 0] == F(32.0)
    # the f16 tests' shapes through the bf16 path (:196-205 identical vectors, generate_test_vector(768, 0.0) = sin(0.1 i))
    v = np.sin(np.arange(768, dtype=F) * F(0.1), dtype=F)
    ids, sc = po.scan_topk_bf16(po.COSINE, v[None, :], v, 1)
    assert abs(float(sc[0, 0]) - 1.0) < 0.01
    # :238-270 ranking is preserved: the close vector (seed 0.1) stays more similar than the far one (seed 5.0)
    close, far = np.sin(F(0.1) + np.arange(768, dtype=F) * F(0.1), dtype=F), np.sin(F(5.0) + np.arange(768, dtype=F) * F(0.1), dtype=F)
    ids, sc = po.scan_topk_bf16(po.COSINE, np.stack([far, close]), v, 2)
    assert ids[0].tolist() == [1, 0] and sc[0, 0] > sc[0, 1]


def _py_scores(metric, rows, q):
    """half_precision.rs:199-254 as plain loops over the bf16-rounded values: sequential f32 sums, f32 sqrt, one division"""
    rr, qq = po.round_bf16(rows), po.round_bf16(q)

    def nsq(v):
        s = F(0.0)
        for x in v:
            s = F(s + F(x * x))
        return s
    qn = F(np.sqrt(nsq(qq)))
    out = []
    for r in rr:
        dot = F(0.0)
        for a, b in zip(qq, r):
            dot = F(dot + F(a * b))
        if metric == po.COSINE:
            rn = F(np.sqrt(nsq(r)))
            eps = np.finfo(F).eps
            out.append(F(0.0) if (qn < eps or rn < eps) else F(dot / F(qn * rn)))
        else:
            out.append(dot)
    return np.array(out, dtype=F)


@pytest.mark.parametrize("metric", [po.DOT, po.COSINE])
@pytest.mark.parametrize("dim", [1, 7, 64, 131])
def test_scan_matches_the_plain_loop_restatement(metric, dim):
    rng = np.random.default_rng(100 + dim)
    rows = rng.standard_normal((40, dim)).astype(F)
    rows[3] = 0.0                                   # zero norm: cosine is 0.0 by the EPSILON rule
    rows[5] = rows[4]                               # equal scores: the smaller row first
    q = rng.standard_normal(dim).astype(F)
    exp = _py_scores(metric, rows, q)
    order = np.lexsort((np.arange(40), -exp.astype(np.float64)))[:10]
    ids, sc = po.scan_topk_bf16(metric, rows, q, 10)
    assert ids[0].tolist() == order.tolist()
    assert np.array_equal(sc[0].view(np.uint32), exp[order].view(np.uint32))
    if metric == po.COSINE:
        allids, allsc = po.scan_topk_bf16(metric, rows, q, 40)
        assert allsc[0][allids[0].tolist().index(3)] == F(0.0)


def test_norm_below_epsilon_gives_zero_cosine():
    tiny = np.full((1, 8), 1e-9, dtype=F)           # norm 2.8e-9 < f32::EPSILON (1.19e-7): half_precision.rs:247-249
    q = np.ones(8, dtype=F)
    _, sc = po.scan_topk_bf16(po.COSINE, tiny, q, 1)
    assert sc[0, 0] == F(0.0)
    _, sc = po.scan_topk_bf16(po.COSINE, np.ones((1, 8), dtype=F), tiny[0], 1)
    assert sc[0, 0] == F(0.0)
