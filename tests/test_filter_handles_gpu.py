"""Device-resident filters (vk_filter_*, csrc/filter_build.hip, csrc/filter_set.cc).

The reference filters an HNSW search with a functor per visited candidate (InlineVectorFilter, src/query/search.cc:103-134)
and holds, for a predicate, the EntriesFetchers of its terms (search.cc:301-399; tag.cc:383-455): lists of keys.  The
library builds the allow-bitmap ON THE DEVICE from such id lists / id runs.  Pinned here:
  * the bitmap built from ids (any order, duplicates, ids beyond nbits) and from runs equals oracle.allow_bitmap bit for
    bit, its population count included; combine() = the set operation;
  * a search with the handle answers exactly like the same search with the host bitmap (HNSW: one filter per query in one
    launch; FLAT; blocking, submitted, batch entry points) -- and like the CPU oracle;
  * the cache: hit under the same key and epoch, miss after the epoch moved, bounded by filter-cache-entries;
  * a request keeps its filter alive: the creator may release the handle while the search is in flight."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


@pytest.fixture(scope="module")
def graphs(vsa, oracle):
    rng = np.random.default_rng(505)
    n, dim = 20_000, 48
    x = rng.standard_normal((n, dim)).astype(np.float32)
    h = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=16, ef_construction=100, build_threads=1)
    h.add_batch(x)
    h.flush()
    f = vsa.Index("FLAT", dim, "L2", initial_cap=n)
    f.add_batch(x)
    f.flush()
    oh = oracle.HNSW.from_product_index(h.save_raw, dim, "L2", 16, ef_construction=100)
    of = oracle.Flat(dim, "L2", max_elements=n)
    of.add_many(x)
    return {"n": n, "dim": dim, "x": x, "h": h, "f": f, "oh": oh, "of": of, "rng": rng}


def test_id_lists_and_runs_build_the_oracle_bitmap(vsa, oracle, graphs):
    g, n, rng = graphs["h"], graphs["n"], graphs["rng"]
    for nbits in (n, n - 37, 64, 65, 1):
        ids = rng.integers(0, n + 500, size=5000, dtype=np.uint64)          # duplicates, and ids beyond nbits (ignored)
        fl = g.make_filter(nbits, labels=ids)
        want = oracle.allow_bitmap(np.unique(ids[ids < nbits]), nbits)
        assert fl.read().tolist() == want[: (nbits + 63) // 64].tolist()
        assert fl.info() == (nbits, int(np.unique(ids[ids < nbits]).size))
    # runs: inclusive, overlapping, touching word edges, reaching past the end; on top of a host bitmap
    runs = np.array([[0, 0], [63, 64], [100, 4000], [3990, 4100], [n - 5, n + 50], [128, 191], [7000, 6999]], dtype=np.uint64)
    base = oracle.allow_bitmap(np.array([5, 9000, n - 1], dtype=np.uint64), n)
    fl = g.make_filter(n, runs=runs, base_bits=base)
    members = {5, 9000, n - 1}
    for lo, hi in runs.tolist():
        members.update(range(lo, min(hi, n - 1) + 1))
    want = oracle.allow_bitmap(np.array(sorted(members), dtype=np.uint64), n)
    assert fl.read().tolist() == want.tolist() and fl.info()[1] == len(members)
    # combine = the set operations (composed predicates over cached terms)
    a_ids = np.flatnonzero(rng.random(n) < 0.3).astype(np.uint64)
    b_ids = np.flatnonzero(rng.random(n) < 0.3).astype(np.uint64)
    fa, fb = g.make_filter(n, labels=a_ids), g.make_filter(n, labels=b_ids)
    for op, fn in (("and", np.intersect1d), ("or", np.union1d), ("andnot", np.setdiff1d)):
        fc = g.combine_filters(fa, fb, op)
        w = fn(a_ids, b_ids)
        assert fc.read().tolist() == oracle.allow_bitmap(w, n).tolist() and fc.info()[1] == w.size, op
    # ... and a batch of them in one launch (vk_filter_combine_batch): the same bitmaps and counts as one by one
    terms = [g.make_filter(n, labels=np.flatnonzero(rng.random(n) < p).astype(np.uint64)) for p in (0.05, 0.2, 0.5, 0.9)]
    pairs = [(terms[i % 4], terms[(i * 7 + 1) % 4], ("and", "or", "andnot")[i % 3]) for i in range(97)]
    batch = g.combine_filters_batch(pairs)
    assert len(batch) == 97
    for (a, b, op), fc in zip(pairs, batch):
        one = g.combine_filters(a, b, op)
        assert fc.read().tolist() == one.read().tolist() and fc.info() == one.info(), op
    assert g.combine_filters_batch([]) == []
    with pytest.raises(vsa.VkError):
        g.combine_filters_batch([(fa, fb), (fa, g.make_filter(n - 1, labels=a_ids))])   # one pair of different sizes fails the call
    with pytest.raises(vsa.VkError):
        g.combine_filters(fa, g.make_filter(n - 1, labels=a_ids), "and")    # different sizes
    with pytest.raises(vsa.VkError):
        graphs["f"].combine_filters(fa, fb, "and")                           # another index's filters


@pytest.mark.parametrize("algo", ["h", "f"])
def test_search_with_a_handle_equals_search_with_the_host_bitmap_and_the_oracle(vsa, oracle, graphs, algo):
    g, o, n, dim, rng = graphs[algo], graphs["o" + algo], graphs["n"], graphs["dim"], graphs["rng"]
    k, ef = 10, 96
    Q = rng.standard_normal((48, dim)).astype(np.float32)
    sets = [np.flatnonzero(rng.random(n) < p).astype(np.uint64) for p in (0.5, 0.1, 0.01)] + [np.arange(0, n, 7, dtype=np.uint64)]
    bitmaps = [oracle.allow_bitmap(s, n) for s in sets]
    handles = [g.make_filter(n, labels=rng.permutation(s)) for s in sets]
    # one filter per query, mixed with unfiltered queries, in one call
    which = [None if i % 5 == 4 else i % len(sets) for i in range(Q.shape[0])]
    D1, L1, N1 = g.search_batch_filter_handles(Q, k, [None if w is None else handles[w] for w in which], ef=ef)
    D2, L2, N2 = g.search_batch_filters(Q, k, [None if w is None else bitmaps[w] for w in which], [n] * Q.shape[0], ef=ef)
    assert N1.tolist() == N2.tolist() and L1.tolist() == L2.tolist() and D1.view(np.uint32).tolist() == D2.view(np.uint32).tolist()
    for i in range(0, Q.shape[0], 3):
        w = which[i]
        if algo == "h":
            od, ol = o.search(Q[i], k, ef=ef, allow=None if w is None else bitmaps[w], allow_nbits=n)
        else:
            od, ol = o.search(Q[i], k, allow=None if w is None else bitmaps[w], allow_nbits=n)
            # (bruteforce.h's loop can under-fill behind a filter; the product returns the exact k best allowed rows)
            if w is not None:
                row_d = ((graphs["x"][sets[w].astype(np.int64)] - Q[i]) ** 2).sum(1)
                assert set(L1[i, :N1[i]].tolist()) == set(sets[w][np.argsort(row_d, kind="stable")[:k]].tolist())
                continue
        assert L1[i, :N1[i]].tolist() == ol.tolist() and D1[i, :N1[i]].view(np.uint32).tolist() == od.view(np.uint32).tolist()
    # single-query entry points, dispatcher off and on (blocking + submitted); the creator drops its handle while in flight
    for coalesce in (0, 64):
        g.set_coalescing(coalesce, 300)
        try:
            for i in (0, 1, 2, 7):
                w = which[i]
                d, l = g.search_filter(Q[i], k, None if w is None else handles[w], ef=ef)
                assert l.tolist() == L1[i, :N1[i]].tolist() and d.view(np.uint32).tolist() == D1[i, :N1[i]].view(np.uint32).tolist()
            if coalesce:
                done = threading.Semaphore(0)
                pend = []
                for i in range(Q.shape[0]):
                    w = which[i]
                    tmp = None if w is None else g.make_filter(n, labels=sets[w])
                    pend.append(g.submit_filter(Q[i], k, lambda st: done.release(), tmp, ef=ef))
                    if tmp is not None:
                        pend[-1].allow = None     # (the binding's own reference)
                        tmp.release()             # the request holds the filter until it completes
                for _ in pend:
                    assert done.acquire(timeout=60)
                for i, p in enumerate(pend):
                    d, l = p.result()
                    assert p.status == 0 and l.tolist() == L1[i, :N1[i]].tolist() and d.view(np.uint32).tolist() == D1[i, :N1[i]].view(np.uint32).tolist()
        finally:
            g.set_coalescing(0, 0)


def test_cache_hits_under_one_epoch_and_is_bounded(vsa, graphs):
    g, n = graphs["h"], graphs["n"]
    s0 = g.stats()
    assert g.filter_cache_get(b"@tag:{red}", 7) is None
    f = g.make_filter(n, labels=np.arange(0, n, 3, dtype=np.uint64))
    g.filter_cache_put(b"@tag:{red}", 7, f)
    hit = g.filter_cache_get(b"@tag:{red}", 7)
    assert hit is not None and hit.read().tolist() == f.read().tolist()
    assert g.filter_cache_get(b"@tag:{red}", 8) is None              # a write phase later: stale, dropped
    assert g.filter_cache_get(b"@tag:{red}", 7) is None
    s1 = g.stats()
    assert s1.filter_cache_hits - s0.filter_cache_hits == 1 and s1.filter_cache_misses - s0.filter_cache_misses == 3
    g.set_option("filter-cache-entries", 4)
    try:
        for i in range(10):
            g.filter_cache_put(b"key%d" % i, 1, f)
        st = g.stats()
        assert st.filter_cache_entries == 4 and st.filter_cache_bytes == 4 * ((n + 63) // 64 + 1) * 8
        assert g.filter_cache_get(b"key0", 1) is None and g.filter_cache_get(b"key9", 1) is not None
        g.set_option("filter-cache-entries", 0)                      # off: nothing is kept
        g.filter_cache_put(b"off", 1, f)
        assert g.filter_cache_get(b"off", 1) is None
    finally:
        g.set_option("filter-cache-entries", 256)


def test_a_filter_of_another_index_fails_alone(vsa, graphs):
    """one member's failure used to be reported to every member of its batch.  A request whose filter belongs to
    another index is refused at the ABI; inside a batch an argument error re-runs the members one by one."""
    h, f, n, dim = graphs["h"], graphs["f"], graphs["n"], graphs["dim"]
    foreign = f.make_filter(n, labels=np.arange(10, dtype=np.uint64))
    q = graphs["x"][0]
    with pytest.raises(vsa.VkError) as e:
        h.search_filter(q, 5, foreign)
    assert e.value.code == vsa.VK_ERR_INVALID and "another index" in e.value.msg


def test_released_bitmaps_are_recycled_and_a_recycled_block_starts_clean(vsa, oracle, graphs):
    """r05: a released filter's device memory goes to a per-device pool (hipFree would wait for every search in flight on the
    device) and the next filter of that size takes it over -- with whatever the previous owner left in it.  A block that held
    ALL ONES must come back as exactly the new id list / the new combination (bits, slack word, count); device memory does
    not grow over a thousand create / combine / release rounds."""
    g, n, rng = graphs["h"], graphs["n"], graphs["rng"]
    ones = np.full((n + 63) // 64, ~np.uint64(0), dtype=np.uint64)
    few = np.array([3, 64, n - 1], dtype=np.uint64)
    a_ids = np.flatnonzero(rng.random(n) < 0.2).astype(np.uint64)
    b_ids = np.flatnonzero(rng.random(n) < 0.2).astype(np.uint64)
    fa, fb = g.make_filter(n, labels=a_ids), g.make_filter(n, labels=b_ids)
    want_few = oracle.allow_bitmap(few, n).tolist()
    want_and = oracle.allow_bitmap(np.intersect1d(a_ids, b_ids), n).tolist()
    import torch
    free0 = None
    for r in range(1000):
        full = g.make_filter(n, base_bits=ones)
        assert full.info() == (n, n)                 # (stray bits past nbits in the caller's last word are not counted)
        full.release()
        f1 = g.make_filter(n, labels=few)            # takes the block that held all ones
        if r % 100 == 0:
            assert f1.read().tolist() == want_few and f1.info() == (n, 3)
        f1.release()
        fc = g.combine_filters(fa, fb, "and")
        if r % 100 == 0:
            assert fc.read().tolist() == want_and and fc.info()[1] == np.intersect1d(a_ids, b_ids).size
            # a search through the recycled bitmap equals the search through the host bitmap
            q = graphs["x"][r % n]
            d1, l1 = g.search_filter(q, 5, fc, ef=64)
            d2, l2 = g.search(q, 5, ef=64, allow=np.array(want_and, dtype=np.uint64), allow_nbits=n)
            assert l1.tolist() == l2.tolist() and d1.view(np.uint32).tolist() == d2.view(np.uint32).tolist()
        fc.release()
        if r == 10:
            free0 = torch.cuda.mem_get_info()[0]
    assert torch.cuda.mem_get_info()[0] >= free0 - (8 << 20), (free0, torch.cuda.mem_get_info()[0])
