"""The error bounds behind the selection stage (sweep_split.hip `select_eps`, `l2_seed_kernel`), checked on the CPU in float64:
whatever order the matrix cores add the products in, a selection score differs from the exact dot product by the roundings of
the operands — bf16 keeps 8 significant bits (|x - hi| <= 2^-8 |x|, round to nearest even), the split remainder
|x - hi - lo| <= 2^-17 |x| — and the bounds the proofs use must cover that for any data, including values that sit exactly
on rounding boundaries.  (The f32 accumulation term 16 dim 2^-24 is on top of these and is not exercised here: the sums below
are exact.)"""
import numpy as np

from oracle import pyoracle as po

EPS1 = 8.2 * 2.0 ** -18                      # level 1: hi.hi + hi.lo + lo.hi
EPS2 = 2.0 * 2.0 ** -8 * 1.002 + 1.6e-5      # level 2: hi.hi


def split(x):
    hi = po.round_bf16(x).astype(np.float32)
    lo = po.round_bf16((x - hi).astype(np.float32)).astype(np.float32)
    return hi.astype(np.float64), lo.astype(np.float64)


def cases(rng):
    for dim in (64, 128, 768):
        yield rng.standard_normal(dim).astype(np.float32), rng.standard_normal(dim).astype(np.float32)
        yield (rng.standard_normal(dim) * 1e-3).astype(np.float32), (rng.standard_normal(dim) * 1e4).astype(np.float32)
        # values just below a rounding boundary: the largest |x - hi| a binade allows, same sign everywhere (errors add up)
        worst = np.float32(1.0 + 2.0 ** -8 - 2.0 ** -20)
        yield np.full(dim, worst, np.float32), np.full(dim, worst, np.float32)
        yield np.full(dim, worst, np.float32), -np.full(dim, np.float32(1.0 + 2.0 ** -8 + 2.0 ** -9 - 2.0 ** -20), np.float32)
        # remainders that are themselves on a boundary of the second rounding
        v = np.float32(1.0 + 2.0 ** -9 + 2.0 ** -17 - 2.0 ** -23)
        yield np.full(dim, v, np.float32), np.full(dim, v, np.float32)
        yield rng.integers(-3, 4, dim).astype(np.float32), rng.integers(-3, 4, dim).astype(np.float32)  # exactly representable


def test_bf16_and_split_rounding_facts():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200_000) * np.exp(rng.uniform(-20, 20, 200_000))).astype(np.float32)
    hi, lo = split(x)
    x64 = x.astype(np.float64)
    assert np.all(np.abs(x64 - hi) <= 2.0 ** -8 * np.abs(x64))           # 8 significant bits, nearest
    assert np.all(np.abs(x64 - hi - lo) <= 2.0 ** -17 * np.abs(x64))     # the remainder's own rounding
    # 2^-9 (the unit roundoff one might assume for "bfloat16") is NOT a bound: values exist beyond it
    assert np.max(np.abs(x64 - hi) / np.abs(x64)) > 2.0 ** -9


def test_selection_scores_stay_inside_the_bounds_the_proofs_use():
    rng = np.random.default_rng(1)
    worst1 = worst2 = 0.0
    for x, q in cases(rng):
        xh, xl = split(x)
        qh, ql = split(q)
        exact = float(np.dot(x.astype(np.float64), q.astype(np.float64)))
        scale = float(np.linalg.norm(x.astype(np.float64)) * np.linalg.norm(q.astype(np.float64)))
        if scale == 0.0:
            continue
        a1 = float(np.dot(xh, qh) + np.dot(xh, ql) + np.dot(xl, qh))     # level 1 (the lo.lo term is dropped)
        a2 = float(np.dot(xh, qh))                                        # level 2
        worst1 = max(worst1, abs(a1 - exact) / scale)
        worst2 = max(worst2, abs(a2 - exact) / scale)
        assert abs(a1 - exact) <= EPS1 * scale, (abs(a1 - exact) / scale, EPS1)
        assert abs(a2 - exact) <= EPS2 * scale, (abs(a2 - exact) / scale, EPS2)
    # the adversarial cases come close to the bounds — and beyond what a 2^-9-per-rounding analysis (3.1 * 2^-18) would allow
    assert worst1 > 3.1 * 2.0 ** -18, worst1
    assert worst2 > 0.5 * EPS2, worst2


def test_euclidean_augmentation_bound():
    # s = q.v - |v|^2 / 2 as the dot product of (v, -h_hi, -h_lo) and (q, 1, 1): error <= eps2 |q| |v| + 2^-16 h
    rng = np.random.default_rng(2)
    for dim in (128, 768):
        for scale_v in (1.0, 7.5, 1e-2):
            v = (rng.standard_normal(dim) * scale_v).astype(np.float32)
            q = rng.standard_normal(dim).astype(np.float32)
            vh, _ = split(v)
            qh, _ = split(q)
            nn = np.float32(np.sqrt(np.float32(np.dot(v.astype(np.float64), v.astype(np.float64)))))
            h = np.float32(-0.5) * nn * nn
            hh, hl = split(np.array([h], np.float32))
            approx = float(np.dot(vh, qh) + hh[0] + hl[0])
            v64, q64 = v.astype(np.float64), q.astype(np.float64)
            exact = float(np.dot(q64, v64) - 0.5 * np.dot(v64, v64))
            bound = EPS2 * np.linalg.norm(q64) * np.linalg.norm(v64) + (2.0 ** -16 + 1e-6) * 0.5 * float(np.dot(v64, v64))
            assert abs(approx - exact) <= bound, (dim, scale_v, abs(approx - exact), bound)
            # and the identity the proof turns back into a distance
            d2 = float(np.dot(q64 - v64, q64 - v64))
            assert abs(d2 - (float(np.dot(q64, q64)) - 2.0 * exact)) <= 1e-9 * max(d2, 1.0)


def test_measured_residual_bound_of_level_2():
    # sweep_split.hip select_eps_q: |x.q - hx.hq| <= (rho_x + rho_q + 3 rho_x rho_q) |x| |q| with the MEASURED residual ratios
    # rho = |v - bf16(v)| / |v| — for any data (adversarial boundary values, denormal-range rows, mixed scales), and about
    # half the constant bound on Gaussian data (what the benchmark and typical embeddings look like)
    rng = np.random.default_rng(3)

    def rho(v):
        h, _ = split(v)
        v64 = v.astype(np.float64)
        n = float(np.linalg.norm(v64))
        return (float(np.linalg.norm(v64 - h)) / n if n > 0.0 else 0.0), h

    extra = [
        (np.full(768, 1e-40, np.float32), rng.standard_normal(768).astype(np.float32)),           # denormal row: rho far above 2^-8
        ((rng.standard_normal(768) * np.exp(rng.uniform(-30, 30, 768))).astype(np.float32), rng.standard_normal(768).astype(np.float32)),
    ]
    ratios = []
    for x, q in list(cases(rng)) + extra:
        rx, xh = rho(x)
        rq, qh = rho(q)
        x64, q64 = x.astype(np.float64), q.astype(np.float64)
        scale = float(np.linalg.norm(x64) * np.linalg.norm(q64))
        if scale == 0.0:
            continue
        err = abs(float(np.dot(xh, qh)) - float(np.dot(x64, q64)))
        bound = (rx + rq + 3.0 * rx * rq) * scale
        assert err <= bound * (1.0 + 1e-12), (err / scale, rx, rq)
        ratios.append((rx, rq))
    g = [rho(rng.standard_normal(768).astype(np.float32))[0] for _ in range(200)]
    assert max(g) < 0.5 * 2.0 ** -8, max(g)       # Gaussian rows: ~0.41 x 2^-8
    assert max(r for r, _ in ratios) > 0.05       # ... while the denormal row shows why the constant was not a bound for ALL data


def test_normalised_cosine_image_bound():
    """Round 6 (sweep_split.hip seln_rows_kernel / seln_prep_queries_kernel): Cosine batches select over bf16 images of the NORMALISED
    vectors, u = bf16(fl(v / fl|v|)), w = bf16(fl(q / fl|q|)); the selection score u.w must stay within
        (rho_u + rho_w + 3 rho_u rho_w) + 4e-6
    of the exact cosine, with rho measured on the normalised vectors (what the kernels store), for Gaussian rows, mixed scales,
    boundary values, tiny and huge norms.  (The f32 accumulation term is on top and not exercised: the sums here are exact.)"""
    rng = np.random.default_rng(4)

    def norm_f32(v):   # the canonical norm is an f32 chain: any f32-rounded norm within a few ulps serves the bound
        return np.float32(np.sqrt(np.float32(np.dot(v.astype(np.float64), v.astype(np.float64)))))

    def image(v):
        n = norm_f32(v)
        u = (v / n).astype(np.float32) if n != 0 else np.zeros_like(v)
        h = po.round_bf16(u).astype(np.float32)
        u64 = u.astype(np.float64)
        nu = float(np.linalg.norm(u64))
        return h.astype(np.float64), (float(np.linalg.norm(u64 - h)) / nu if nu > 0 else 0.0)

    worst = 0.0
    data = list(cases(rng)) + [((rng.standard_normal(768) * 1e-18).astype(np.float32), (rng.standard_normal(768) * 1e15).astype(np.float32)),
                               ((rng.standard_normal(768) * np.exp(rng.uniform(-8, 8, 768))).astype(np.float32), rng.standard_normal(768).astype(np.float32))]
    for x, q in data:
        x64, q64 = x.astype(np.float64), q.astype(np.float64)
        nx, nq = float(np.linalg.norm(x64)), float(np.linalg.norm(q64))
        if nx == 0.0 or nq == 0.0:
            continue
        exact = float(np.dot(x64, q64)) / (nx * nq)
        u, ru = image(x)
        w, rw = image(q)
        approx = float(np.dot(u, w))
        bound = (ru + rw + 3.0 * ru * rw) + 4e-6
        worst = max(worst, abs(approx - exact) / bound)
        assert abs(approx - exact) <= bound, (abs(approx - exact), bound, ru, rw)
    assert worst > 0.3   # the boundary cases come close: the bound is not loose by an order of magnitude
    # what the tighter bound buys on the benchmark data: no row norm enters the kernel's quick test.  The Cosine instance bounds a
    # lane's 32 rows by the smallest of their norms; on N(0,1) x 768 that slack is as large as 2 delta itself
    norms = np.linalg.norm(rng.standard_normal((4096, 768)), axis=1)
    slack = float(np.mean(1.0 - norms.reshape(-1, 32).min(axis=1) / norms.reshape(-1, 32).mean(axis=1)))
    assert 0.03 < slack < 0.08


def test_wide_selection_keeps_every_row_of_the_exact_top_k():
    """sweep_wide.hip's argument, on numbers: approximate scores within delta of the exact ones (any perturbation), tau = (k-th best
    approximate score over ANY subset holding k rows) - 2 delta  =>  every row of the exact top k passes `approx > tau`, and the k-th
    best exact score among the survivors stands more than delta above tau (the proof by construction)."""
    rng = np.random.default_rng(6)
    for trial in range(200):
        n, k = int(rng.integers(200, 3000)), int(rng.integers(11, 129))
        exact = rng.standard_normal(n) * 0.036
        delta = float(rng.uniform(1e-4, 1e-2))
        approx = exact + rng.uniform(-delta, delta, n)
        subset = rng.choice(n, size=int(rng.integers(k, n + 1)), replace=False)     # the rows seen so far (a seed sample, a list)
        a_k = np.sort(approx[subset])[::-1][k - 1]
        tau = a_k - 2.0 * delta * 1.01 - abs(a_k) * 1e-6
        survivors = np.nonzero(approx > tau)[0]
        top = np.argsort(-exact, kind="stable")[:k]
        assert set(top.tolist()) <= set(survivors.tolist())
        e_k = np.sort(exact[survivors])[::-1][k - 1]
        assert e_k > tau + delta
