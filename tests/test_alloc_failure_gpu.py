"""Device memory that cannot be had.  The row table of an index grows with its content (capacity is the logical limit the
reference's block_size arithmetic reports, vector_flat.cc:164-176; device memory is taken as rows arrive), so the failure
comes from a WRITE or from vk_index_device_rows.  The reference's counterpart: hnswlib throws "Not enough memory" from
resizeIndex / addPoint (hnswalg.h:758-777, bruteforce.h:44-48) and the wrappers turn every exception into
absl::InternalError (vector_flat.cc:165-176, vector_hnsw.cc:186-197).  Here: the call fails with VK_ERR_INTERNAL, the index
answers as before, the thread's NEXT calls are not charged with the failure (the HIP runtime keeps a per-thread last error
that a later launch check would otherwise pick up), and once memory is there again the same write goes through."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIM = 16000              # 64 000-byte rows


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _same(gd, gl, od, ol):
    assert gl.tolist() == ol.tolist()
    assert gd.view(np.uint32).tolist() == od.view(np.uint32).tolist()


def _occupy(torch, leave_bytes):
    """take the device's free memory down to about leave_bytes (blocks of 1 GiB, then 64 MiB, then 4 MiB)"""
    import gc
    gc.collect()                 # (an index of an earlier test that dies in the middle of this one would give its memory back)
    torch.cuda.empty_cache()
    held = []
    for block in (1 << 30, 64 << 20, 4 << 20):
        while torch.cuda.mem_get_info()[0] > leave_bytes + block:
            try:
                held.append(torch.empty(block, dtype=torch.uint8, device="cuda"))
            except torch.OutOfMemoryError:      # (free in pieces smaller than this block: go on with the next size)
                break
    return held


def test_the_row_table_cannot_be_had(vsa, oracle):
    """vk_index_device_rows (the bulk-load door of an EMPTY index) for 512 GB of rows: refused; the index takes rows the usual way"""
    rng = np.random.default_rng(2)
    n, k = 1500, 10
    x = rng.standard_normal((n, DIM)).astype(np.float32)
    Q = rng.standard_normal((8, DIM)).astype(np.float32)
    g = vsa.Index("FLAT", DIM, "L2", initial_cap=9_000_000)
    with pytest.raises(vsa.VkError) as e:
        g.device_rows(8_000_000)
    assert e.value.code == vsa.VK_ERR_INTERNAL, e.value
    assert g.stats().count == 0
    # the very next calls of this thread: writes, single query (scan), batch (matrix cores)
    g.add_batch(x)
    assert g.add(n, x[0]) == vsa.VK_OK
    d, l = g.search(x[0], 2)
    assert sorted(l.tolist()) == [0, n] and d.tolist() == [0.0, 0.0]
    o = oracle.Flat(DIM, "L2", max_elements=n + 1)
    o.add_many(np.concatenate([x, x[:1]]))
    for q in Q[:3]:
        _same(*g.search(q, k), *o.search(q, k))
    d, l, c = g.search_batch(Q, k)
    for i, q in enumerate(Q):
        _same(d[i, :c[i]], l[i, :c[i]], *o.search(q, k))


@pytest.mark.parametrize("algo", ["FLAT", "HNSW"])
def test_a_write_under_memory_pressure(vsa, oracle, algo):
    import torch
    rng = np.random.default_rng(3)
    n0, n1, k = 1000, 24_000, 10                      # 64 MB of rows, then 1.5 GB more
    x = rng.standard_normal((n0 + n1, DIM)).astype(np.float32)
    Q = rng.standard_normal((6, DIM)).astype(np.float32)
    g = vsa.Index(algo, DIM, "L2", initial_cap=n0 + n1, m=8, ef_construction=32, ef_runtime=64)
    g.add_batch(x[:n0])
    g.flush()
    before = [g.search(q, k) for q in Q]
    held = _occupy(torch, 512 << 20)                  # half a GiB left: the table cannot grow by 1.5 GB
    try:
        with pytest.raises(vsa.VkError) as e:
            g.add_batch(x[n0:], labels=np.arange(n0, n0 + n1, dtype=np.uint64))
            g.flush()
            g.search(Q[0], k)
        assert e.value.code == vsa.VK_ERR_INTERNAL, e.value
    finally:
        del held
        torch.cuda.empty_cache()
    # memory is back.  The failed write may have left some of its rows in the index -- a batch is not a transaction, in the
    # reference neither -- but what is in the index is whole: FLAT answers as before among the old labels (exact search), the
    # HNSW graph with the rows behind it answers like the oracle walking the same graph
    def check_hnsw():
        o = oracle.HNSW.from_product_index(g.save_raw, DIM, "L2", 8, ef_construction=32)
        for q in Q:
            _same(*g.search(q, k, ef=64), *o.search(q, k, ef=64))
    if algo == "FLAT":
        assert g.stats().count == n0            # (the run that found no room was given back whole)
        for q, (d0, l0) in zip(Q, before):
            _same(*g.search(q, k), d0, l0)
    else:
        check_hnsw()
    # ... and the same write goes through now (labels already in are updated in place: same row, same answer)
    g.add_batch(x[n0:], labels=np.arange(n0, n0 + n1, dtype=np.uint64))
    g.flush()
    st = g.stats()
    assert st.count == n0 + n1 and st.deleted == 0
    if algo == "FLAT":
        o = oracle.Flat(DIM, "L2", max_elements=n0 + n1)
        o.add_many(x)
        for q in Q[:2]:
            _same(*g.search(q, k), *o.search(q, k))
    else:
        check_hnsw()
        for i in (0, n0 - 1, n0, n0 + n1 // 2, n0 + n1 - 1):   # (iid Gaussian rows of 16 000 dimensions: no self-retrieval claim)
            assert np.array_equal(g.get_row(i), x[i])


def test_max_label_after_a_bulk_add(vsa):
    """vk_index_stats.max_label (GetMaxInternalLabel after a load, vector_base.cc:480-481) through the strided-copy path of
    vk_index_add_batch, which r06 found not keeping it"""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((600, 32)).astype(np.float32)
    for algo in ("FLAT", "HNSW"):
        g = vsa.Index(algo, 32, "L2", initial_cap=700, m=8, ef_construction=32)
        g.add_batch(x, labels=np.arange(1000, 1600, dtype=np.uint64))
        g.flush()
        assert g.stats().max_label == 1599, algo
        assert g.add(5000, x[0]) == vsa.VK_OK
        g.remove(5000)
        g.flush()
        assert g.stats().max_label == 5000, algo     # the largest label EVER held


def test_staged_adds_under_memory_pressure(vsa, oracle):
    """single adds are acknowledged when they are STAGED (host memory); the device is asked for room when the bulk is linked.
    No room: the flush fails, every acknowledged add is still there (readable, removable, updatable), and the same flush
    links them once memory is back"""
    import torch
    dim, k = 8192, 10                                 # 32 KB rows: the device build takes them
    n0, n1 = 20_000, 30_000                           # 655 MB in the graph, 983 MB staged behind it
    rng = np.random.default_rng(7)
    lat = rng.standard_normal((n0 + n1, 24)).astype(np.float32)
    x = lat @ rng.standard_normal((24, dim)).astype(np.float32) + 0.05 * rng.standard_normal((n0 + n1, dim)).astype(np.float32)
    Q = x[rng.integers(0, n0 + n1, 6)] + 0.01 * rng.standard_normal((6, dim)).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n0 + n1, m=8, ef_construction=32, ef_runtime=64)
    g.add_batch(x[:n0])
    g.flush()
    held = _occupy(torch, 256 << 20)
    try:
        for i in range(n0, n0 + n1):
            assert g.add(i, x[i]) == vsa.VK_OK
        st = g.stats()
        assert st.count == n0 + n1 and st.staged_ops >= n1
        with pytest.raises(vsa.VkError) as e:
            g.flush()
        assert e.value.code == vsa.VK_ERR_INTERNAL, e.value
        st = g.stats()
        assert st.count == n0 + n1 and st.staged_ops >= n1, (st.count, st.staged_ops)     # nothing left the staging area
        for i in (n0, n0 + 7, n0 + n1 - 1):
            assert np.array_equal(g.get_row(i), x[i])
        assert g.remove(n0 + 5) == vsa.VK_OK          # a delete and an update of rows that wait
        assert g.add(n0 + 6, x[0]) == vsa.VK_OK
        with pytest.raises(vsa.VkError):              # a search links what is staged first: the same refusal, no answer over half an index
            g.search(Q[0], k)
    finally:
        del held
        torch.cuda.empty_cache()
    g.flush()
    st = g.stats()
    assert st.count == n0 + n1 - 1 and st.deleted == 0 and st.staged_ops == 0 and st.staged_adds_device > 0
    assert np.array_equal(g.get_row(n0 + 6), x[0]) and np.array_equal(g.get_row(n0 + n1 - 1), x[-1])
    assert g.get_row(n0 + 5) is None and not g.contains(n0 + 5)
    o = oracle.HNSW.from_product_index(g.save_raw, dim, "L2", 8, ef_construction=32)
    for q in Q:
        _same(*g.search(q, k, ef=64), *o.search(q, k, ef=64))
    hits = 0
    for i in rng.integers(n0, n0 + n1, 50):
        if i in (n0 + 5, n0 + 6):
            continue
        d, l = g.search(x[i], 1, ef=64)
        hits += l.tolist() == [i]
    assert hits >= 45, hits                           # the staged rows are linked INTO the graph, not just stored


def test_a_bulk_that_fails_part_of_the_way_loses_nothing(vsa, oracle):
    """The room for the rows is taken before a bulk leaves the staging area; what can still fail is the scratch of a device
    batch.  The failpoint option makes the second batch of the bulk fail behind its registration: its elements are in the
    graph with empty lists, the batches behind it never got there.  Nothing acknowledged may be lost: the registered ones are
    linked by the host builder, the others go back to staging, the error goes to the caller of the flush."""
    dim, k = 64, 10
    n0, n1 = 20_000, 20_000
    rng = np.random.default_rng(11)
    lat = rng.standard_normal((n0 + n1, 16)).astype(np.float32)
    x = lat @ rng.standard_normal((16, dim)).astype(np.float32) + 0.05 * rng.standard_normal((n0 + n1, dim)).astype(np.float32)
    Q = x[rng.integers(0, n0 + n1, 16)] + 0.01 * rng.standard_normal((16, dim)).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n0 + n1, m=8, ef_construction=64, ef_runtime=64)
    g.add_batch(x[:n0])
    g.flush()
    for i in range(n0, n0 + n1):
        assert g.add(i, x[i]) == vsa.VK_OK
    g.set_option("hnsw-failpoint-batch", 2)
    with pytest.raises(vsa.VkError) as e:
        g.flush()
    assert e.value.code == vsa.VK_ERR_INTERNAL and "failpoint" in e.value.msg
    st = g.stats()
    # the first batch (n0 / 32 = 625 rows) is linked, the second registered; the rest waits again
    assert st.count == n0 + n1 and 0 < st.staged_ops < n1, (st.count, st.staged_ops)
    for i in range(n0, n0 + n1, 97):
        assert np.array_equal(g.get_row(i), x[i]) and g.contains(i)
    assert g.remove(n0 + n1 - 1) == vsa.VK_OK and g.remove(n0) == vsa.VK_OK        # one that waits again, one that was linked
    g.set_option("hnsw-failpoint-batch", 0)
    g.flush()
    st = g.stats()
    assert st.count == n0 + n1 - 1 and st.deleted == 1 and st.staged_ops == 0
    chunks = g.save()
    deg = np.array([int(np.frombuffer(c[:4], np.uint32)[0] & 0xFFFF) for c in chunks[1:1 + n0 + n1 - 1]])
    assert (deg == 0).sum() == 0, int((deg == 0).sum())       # nobody was left registered and unlinked
    o = oracle.HNSW.from_product_index(g.save_raw, dim, "L2", 8, ef_construction=64)
    for q in Q:
        _same(*g.search(q, k, ef=64), *o.search(q, k, ef=64))
    hits = 0
    probe = [int(i) for i in rng.integers(n0 + 1, n0 + n1 - 1, 200)]
    for i in probe:
        d, l = g.search(x[i], 1, ef=64)
        hits += l.tolist() == [i]
    assert hits >= 190, hits


def test_flat_single_adds_under_memory_pressure(vsa, oracle):
    """AddRecord one key at a time: the rows wait in pinned host memory; the search that has to publish them is refused while
    the table cannot grow, and answers over all of them once it can"""
    import torch
    rng = np.random.default_rng(13)
    n0, n1, k = 1000, 12_000, 10
    x = rng.standard_normal((n0 + n1, DIM)).astype(np.float32)
    Q = rng.standard_normal((4, DIM)).astype(np.float32)
    g = vsa.Index("FLAT", DIM, "L2", initial_cap=n0 + n1)
    g.add_batch(x[:n0])
    g.flush()
    held = _occupy(torch, 384 << 20)
    try:
        for i in range(n0, n0 + n1):
            assert g.add(i, x[i]) == vsa.VK_OK
        assert g.stats().count == n0 + n1
        with pytest.raises(vsa.VkError) as e:
            g.search(Q[0], k)
        assert e.value.code == vsa.VK_ERR_INTERNAL, e.value
        assert g.remove(n0 + 3) == vsa.VK_OK
    finally:
        del held
        torch.cuda.empty_cache()
    o = oracle.Flat(DIM, "L2", max_elements=n0 + n1)
    keep = np.array([i for i in range(n0 + n1) if i != n0 + 3], dtype=np.uint64)
    o.add_many(x[keep], keep)
    for q in Q:
        _same(*g.search(q, k), *o.search(q, k))
    d, l, c = g.search_batch(Q, k)
    for i, q in enumerate(Q):
        _same(d[i, :c[i]], l[i, :c[i]], *o.search(q, k))
    assert g.stats().count == n0 + n1 - 1


def test_filter_builds_under_memory_pressure(vsa, oracle):
    """vk_filter_create / combine when the bitmap cannot be had: VK_ERR_INTERNAL, the handles that exist keep working, and the
    same build passes once memory is back"""
    import torch
    rng = np.random.default_rng(17)
    n, dim, k = 4000, 64, 10
    x = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal(dim).astype(np.float32)
    g = vsa.Index("FLAT", dim, "L2", initial_cap=n)
    g.add_batch(x)
    g.flush()
    ids = rng.choice(n, 500, replace=False).astype(np.uint64)
    small = g.make_filter(n, labels=ids)
    o = oracle.Flat(dim, "L2", max_elements=n)
    o.add_many(x)
    bits = oracle.allow_bitmap(ids, n)
    want = o.search(q, k, allow=bits, allow_nbits=n)
    held = _occupy(torch, 192 << 20)
    big_bits = 1 << 34                                # a 2 GiB bitmap
    try:
        with pytest.raises(vsa.VkError) as e:
            g.make_filter(big_bits, labels=ids)
        assert e.value.code == vsa.VK_ERR_INTERNAL, e.value
        _same(*g.search_filter(q, k, small), *want)   # (the next launches of this thread are not charged with it)
        both = g.combine_filters(small, small, "and")
        assert both.info()[1] == 500
    finally:
        del held
        torch.cuda.empty_cache()
    big = g.make_filter(big_bits, labels=ids)
    assert big.info() == (big_bits, 500)
    _same(*g.search_filter(q, k, big), *want)


@pytest.mark.parametrize("algo", ["FLAT", "HNSW"])
def test_search_scratch_under_memory_pressure(vsa, algo):
    """the per-call scratch of a large batch (query block, partial lists, visited sets) cannot be had: the batch is refused or
    served -- never a fault -- and the same batch gives the same answer as before once memory is back"""
    import torch
    rng = np.random.default_rng(19)
    n, dim, nq, k = 20_000, 2048, 4096, 64
    x = rng.standard_normal((n, dim)).astype(np.float32)
    Q = rng.standard_normal((nq, dim)).astype(np.float32)
    g = vsa.Index(algo, dim, "L2", initial_cap=n, m=8, ef_construction=32, ef_runtime=128)
    g.add_batch(x)
    g.flush()
    d1, l1, c1 = g.search_batch(Q[:8], k)             # (small batch first: its scratch exists before the squeeze)
    held = _occupy(torch, 8 << 20)
    refused = 0
    try:
        for _ in range(2):
            try:
                g.search_batch(Q, k)
            except vsa.VkError as e:
                assert e.code == vsa.VK_ERR_INTERNAL, e
                refused += 1
        d2, l2, c2 = g.search_batch(Q[:8], k)         # what fits the scratch it already has still runs
        assert l2.tolist() == l1.tolist() and d2.view(np.uint32).tolist() == d1.view(np.uint32).tolist()
    finally:
        del held
        torch.cuda.empty_cache()
    D, L, N = g.search_batch(Q, k)
    assert N.min() == k
    assert L[:8].tolist() == l1.tolist() and D[:8].view(np.uint32).tolist() == d1.view(np.uint32).tolist()
    D2, L2, _ = g.search_batch(Q, k)
    assert L2.tolist() == L.tolist() and D2.view(np.uint32).tolist() == D.view(np.uint32).tolist()
    print(f"{algo}: {refused} of 2 large batches refused under pressure")


@pytest.mark.parametrize("algo", ["FLAT", "HNSW"])
def test_sharded_write_under_memory_pressure(vsa, oracle, algo):
    """the same through the sharded index (four logical shards on the one device): a shard that finds no room gives its run
    back, the router forgets the labels no shard holds, and the same batch passes -- with the same answers as one unsharded
    FLAT index -- once memory is back"""
    import torch
    rng = np.random.default_rng(23)
    n0, n1, k = 2000, 24_000, 10
    x = rng.standard_normal((n0 + n1, DIM)).astype(np.float32)
    Q = rng.standard_normal((4, DIM)).astype(np.float32)
    g = vsa.Index(algo, DIM, "L2", initial_cap=n0 + n1, m=8, ef_construction=32, ef_runtime=64, shard_devices=[0, 0, 0, 0])
    g.add_batch(x[:n0])
    g.flush()
    before = [g.search(q, k) for q in Q]
    held = _occupy(torch, 256 << 20)
    try:
        with pytest.raises(vsa.VkError) as e:
            g.add_batch(x[n0:], labels=np.arange(n0, n0 + n1, dtype=np.uint64))
            g.flush()
            g.search(Q[0], k)
        assert e.value.code == vsa.VK_ERR_INTERNAL, e.value
    finally:
        del held
        torch.cuda.empty_cache()
    st = g.stats()
    assert n0 <= st.count <= n0 + n1
    if algo == "FLAT":
        allow = oracle.allow_bitmap(np.arange(n0, dtype=np.uint64), n0 + n1)
        for q, (d0, l0) in zip(Q, before):
            _same(*g.search(q, k, allow=allow, allow_nbits=n0 + n1), d0, l0)
    held_now = sum(g.contains(i) for i in range(n0, n0 + n1, 13))
    g.add_batch(x[n0:], labels=np.arange(n0, n0 + n1, dtype=np.uint64))
    g.flush()
    st = g.stats()
    assert st.count == n0 + n1 and st.deleted == 0, (st.count, held_now)
    for i in (0, n0 - 1, n0, n0 + n1 // 2, n0 + n1 - 1):
        assert np.array_equal(g.get_row(i), x[i])
    if algo == "FLAT":
        o = oracle.Flat(DIM, "L2", max_elements=n0 + n1)
        o.add_many(x)
        for q in Q[:2]:
            _same(*g.search(q, k), *o.search(q, k))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("algo", ["FLAT", "HNSW"])
def test_coalesced_callers_under_memory_pressure(vsa, algo):
    """the reader pool's blocking Search() calls merged into device batches while the batch's scratch cannot be had: every
    caller gets an answer or VK_ERR_INTERNAL -- nobody hangs, nobody gets another caller's answer -- and the same callers all
    get their answers once memory is back"""
    import threading
    import torch
    rng = np.random.default_rng(29)
    n, dim, k, callers, each = 20_000, 2048, 1000, 96, 6      # (k = 1000: tens of MB of partial lists per batch)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    Q = rng.standard_normal((callers * each, dim)).astype(np.float32)
    g = vsa.Index(algo, dim, "L2", initial_cap=n, m=8, ef_construction=32, ef_runtime=1500)
    g.add_batch(x)
    g.flush()
    g.set_coalescing(callers, 3000)                     # (no search yet: the contexts have no scratch of their own)

    def run(out):
        def work(t):
            for j in range(each):
                i = t * each + j
                try:
                    out[i] = g.search_one(Q[i], k)
                except vsa.VkError as e:
                    out[i] = e
        th = [threading.Thread(target=work, args=(t,)) for t in range(callers)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    held = _occupy(torch, 0)
    got = [None] * len(Q)
    try:
        run(got)
    finally:
        del held
        torch.cuda.empty_cache()
    refused = 0
    for i, r in enumerate(got):
        assert r is not None, i
        if isinstance(r, vsa.VkError):
            assert r.code == vsa.VK_ERR_INTERNAL, r
            refused += 1
    again = [None] * len(Q)
    run(again)
    g.set_coalescing(0, 0)
    want = [g.search_one(q, k) for q in Q[:24]]         # one query per call, no merging
    for i, r in enumerate(again):
        assert not isinstance(r, vsa.VkError), (i, r)
        assert len(r[1]) == k
        if i < len(want):
            assert r[1].tolist() == want[i][1].tolist() and r[0].view(np.uint32).tolist() == want[i][0].view(np.uint32).tolist()
        if not isinstance(got[i], vsa.VkError):
            assert got[i][1].tolist() == r[1].tolist()
    print(f"{algo}: {refused} of {len(Q)} coalesced calls refused under pressure")
