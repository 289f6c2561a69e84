"""Host models of the top-k mechanics round 5 added to csrc/sweep.hip, statement for statement, against a plain sort — the CPU tier's
view of code whose parity tests need a GPU (tests/test_gpu_round5_parity.py, test_gpu_sweep.py, the fuzzers):

  * wave_min_key: the smallest 64-bit key of a wave as TWO 32-bit minima — the score word, then the row word among the lanes that tie;
  * wave_k_smallest<R> / wave_k_smallest_dup<R>: k times { the lane's smallest of its R registers, the wave's smallest of those, drop
    it } — lane e ends with the e-th smallest; the _dup form drops ONE instance of a repeated key (the lowest lane's first register);
  * block_k_smallest: four wave lists, then wave 0 over the 4 k <= 64 survivors (one per lane);
  * the carry of sweep_bits_fused: a further batch extracts from { previous list (lane e holds the e-th) + the new registers };
  * heads first (merge_topk_heads, fused_tail_merge): with B the k-th smallest of the lists' smallest keys, every key of the answer is
    <= B and sits in one of the <= k lists whose smallest key is <= B — also when fewer than k lists hold anything (B = "invalid");
  * merge_topk_heads' step in (list, entry) coordinates: the list of key i + 256 from the list of key i without a division.
"""
import numpy as np
import pytest

INV = np.uint64(0xFFFFFFFFFFFFFFFF)


def wave_min_key(m):
    hi = (m >> np.uint64(32)).astype(np.uint32)
    h = hi.min()
    lo = np.where(hi == h, (m & np.uint64(0xFFFFFFFF)).astype(np.uint32), np.uint32(0xFFFFFFFF)).min()
    return (np.uint64(h) << np.uint64(32)) | np.uint64(lo)


def wave_k_smallest(regs, k, dup):
    """regs: [64][R] uint64 (consumed). -> out[64]: lane e < k holds the e-th smallest (INV where nothing is left)"""
    out = np.full(64, INV, np.uint64)
    for e in range(k):
        mloc = regs.min(axis=1)
        wm = wave_min_key(mloc)
        out[e] = wm
        if wm == INV:
            break
        if dup:
            lane = int(np.nonzero(mloc == wm)[0][0])
            r = int(np.nonzero(regs[lane] == wm)[0][0])
            regs[lane, r] = INV
        else:
            regs[regs == wm] = INV
    return out


def block_k_smallest(regs256, k, dup=False):
    wl = [wave_k_smallest(regs256[w * 64:(w + 1) * 64].copy(), k, dup)[:k] for w in range(4)]
    lane = np.arange(64)
    wsrc, esrc = lane // k, lane % k
    m2 = np.array([wl[w][e] if w < 4 else INV for w, e in zip(wsrc, esrc)], np.uint64)
    return wave_k_smallest(m2.reshape(64, 1), k, dup)[:k]


def keys_of(rng, n, dup_frac=0.0, score_bits=8):
    """n keys (score << 32 | row): few distinct scores, so ties on the score word are the rule"""
    score = rng.integers(0, 1 << score_bits, n, dtype=np.uint64)
    row = rng.permutation(1 << 20)[:n].astype(np.uint64)
    k = (score << np.uint64(32)) | row
    if dup_frac:
        idx = rng.integers(0, n, int(n * dup_frac))
        k[idx] = k[rng.integers(0, n, idx.size)]
    return k


@pytest.mark.parametrize("seed", range(6))
def test_wave_min_key_is_the_64_bit_minimum(seed):
    rng = np.random.default_rng(seed)
    for _ in range(200):
        m = keys_of(rng, 64, score_bits=int(rng.integers(1, 12)))
        m[rng.random(64) < rng.random()] = INV
        assert wave_min_key(m) == m.min()


@pytest.mark.parametrize("R", [1, 4, 8, 16])
@pytest.mark.parametrize("k", [1, 3, 10, 16])
def test_wave_and_block_extraction_equal_a_sort(R, k):
    rng = np.random.default_rng(100 * R + k)
    for trial in range(25):
        n_valid = int(rng.integers(0, 256 * R + 1)) if trial % 3 else int(rng.integers(0, 2 * k + 1))
        flat = np.full(256 * R, INV, np.uint64)
        flat[rng.permutation(256 * R)[:n_valid]] = keys_of(rng, n_valid)          # distinct keys (a row sits in one list)
        regs = flat.reshape(256, R)
        want = np.sort(flat)[:k]
        assert np.array_equal(wave_k_smallest(regs[:64].copy(), k, False)[:k], np.sort(regs[:64].ravel())[:k])
        assert np.array_equal(block_k_smallest(regs, k), want)
        assert np.array_equal(block_k_smallest(regs, k, dup=True), want)          # the _dup form agrees on distinct keys


@pytest.mark.parametrize("k", [1, 10, 16])
def test_repeated_keys_keep_their_places_in_the_dup_form(k):
    rng = np.random.default_rng(7 + k)
    for _ in range(40):
        n_valid = int(rng.integers(1, 2049))
        flat = np.full(2048, INV, np.uint64)
        flat[rng.permutation(2048)[:n_valid]] = keys_of(rng, n_valid, dup_frac=0.3, score_bits=3)
        want = np.sort(flat)[:k]                                                   # merge_topk_select: ties by position, same values
        assert np.array_equal(block_k_smallest(flat.reshape(256, 8), k, dup=True), want)


def test_a_further_batch_carries_the_list_on():
    rng = np.random.default_rng(5)
    for k in (1, 10, 16):
        allkeys = keys_of(rng, 64 * 16 * 3)
        best = np.full(64, INV, np.uint64)
        for b in range(3):                                                         # three batches of 16 registers per lane
            regs = np.concatenate([best.reshape(64, 1), allkeys[b * 1024:(b + 1) * 1024].reshape(64, 16)], axis=1)
            best = wave_k_smallest(regs, k, False)
            best[k:] = INV
        assert np.array_equal(best[:k], np.sort(allkeys)[:k])


@pytest.mark.parametrize("k,kin", [(10, 10), (16, 16), (10, 1), (4, 12), (1, 10)])
def test_heads_first_bound_keeps_the_answer_and_at_most_k_lists(k, kin):
    rng = np.random.default_rng(31 * k + kin)
    for trial in range(60):
        nl = int(rng.integers(1, 1025))
        lists = np.full((nl, kin), INV, np.uint64)
        fill = rng.integers(0, kin + 1, nl) if trial % 4 else np.where(rng.random(nl) < 3.0 / nl, kin, 0)   # (sometimes < k non-empty lists)
        allk = keys_of(rng, int(fill.sum()))
        pos = 0
        for l in range(nl):
            lists[l, :fill[l]] = allk[pos:pos + fill[l]]                           # unsorted inside a list: the kernel takes minima
            pos += fill[l]
        heads = lists.min(axis=1)
        nh = int((heads != INV).sum())
        bound = np.sort(heads)[k - 1] if nh > k else INV
        under = (heads != INV) & (heads <= bound)
        assert under.sum() <= k or nh <= k
        cand = lists[under].ravel()
        cand = cand[(cand != INV) & (cand <= bound)]
        assert cand.size <= k * kin
        want = np.sort(lists.ravel())[:k]
        got = np.sort(cand)[:k]
        got = np.concatenate([got, np.full(k - got.size, INV, np.uint64)])
        assert np.array_equal(got, want)


@pytest.mark.parametrize("kin", [1, 3, 10, 12, 16, 64, 255, 256, 257, 300])
def test_list_entry_stepping_matches_the_division(kin):
    dl, de = 256 // kin, 256 % kin
    for tid in (0, 1, 63, 200, 255):
        l, e = tid // kin, tid % kin
        for step in range(40):
            i = tid + 256 * step
            assert (l, e) == (i // kin, i % kin)
            l += dl
            e += de
            if e >= kin:
                e -= kin
                l += 1
