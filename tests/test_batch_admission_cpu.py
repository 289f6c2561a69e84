"""The argument behind CandList::admit_batch (velesdb_amd/csrc/vdb_hnsw_device.hpp), replayed on the CPU: admitting a chunk of
evaluated neighbours at once — the ef smallest keys of (list U the candidates that pass the chunk-start test) — ends in the same
list as the reference's one-by-one loop (native/graph.rs:500-511: `if results.len() < ef || dist < furthest { push; if len > ef
{ pop furthest } }`) whenever no candidate's distance ties with another key's; with ties the strict compare decides in arrival
order, which is why the kernel sends such chunks to the one-by-one loop."""
import numpy as np


def truncate(lst, ef):
    # the device list keeps entries past position ef - 1 only while their distance equals the ef-th's (exact ties)
    if len(lst) <= ef:
        return lst
    far = lst[ef - 1][0]
    keep = ef
    while keep < len(lst) and not (lst[keep][0] > far):
        keep += 1
    return lst[:keep]


def sequential(lst, ef, cands):
    lst = list(lst)
    for d, node in cands:
        size = min(len(lst), ef)
        far = lst[size - 1][0] if size else float("inf")
        if size < ef or d < far:                     # graph.rs:503 (strict)
            lst.append((d, node))
            lst.sort()
            lst = truncate(lst, ef)
    return lst


def batch(lst, ef, cands):
    size = min(len(lst), ef)
    far = lst[size - 1][0] if size else float("inf")
    acc = [(d, n) for d, n in cands if size < ef or d < far]     # the chunk-start test
    merged = sorted(list(lst) + acc)
    for i in range(len(merged) - 1):                              # a tie between neighbours of the merged order, one of them new
        if merged[i][0] == merged[i + 1][0] and (merged[i] in acc or merged[i + 1] in acc):
            return None                                           # -> one by one
    return truncate(merged, ef)


def test_batch_admission_equals_the_sequential_loop_without_ties():
    rng = np.random.default_rng(7)
    for it in range(400):
        ef = int(rng.integers(1, 40))
        n_old = int(rng.integers(0, 60))
        old = sorted((float(d), i) for i, d in enumerate(rng.random(n_old)))
        old = truncate(old, ef)
        m = int(rng.integers(1, 65))
        cands = [(float(d), 1000 + j) for j, d in enumerate(rng.random(m) * rng.choice([0.3, 1.0, 3.0]))]
        b = batch(old, ef, cands)
        assert b is not None                                       # random doubles: no ties
        assert b == sequential(old, ef, cands), (it, ef, n_old, m)


def test_ties_are_detected_and_do_differ():
    # two candidates at the furthest distance of a full list: the strict compare rejects both one by one; the batch rule would
    # have taken them — the detector must send the chunk to the loop
    old = [(0.1, 1), (0.2, 2), (0.3, 3)]
    cands = [(0.3, 10), (0.25, 11)]
    assert sequential(old, 3, cands) == [(0.1, 1), (0.2, 2), (0.25, 11)]
    cands2 = [(0.25, 11), (0.25, 12), (0.25, 13)]
    assert batch(old, 3, cands2) is None
    # and a tie that involves only OLD entries is none of the batch's business
    old2 = [(0.1, 1), (0.2, 2), (0.2, 3)]
    assert batch(old2, 3, [(0.15, 20)]) == sequential(old2, 3, [(0.15, 20)])
