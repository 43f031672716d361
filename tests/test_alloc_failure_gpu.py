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
    """take the device's free memory down to about leave_bytes (blocks of 1 GiB, then 64 MiB)"""
    held = []
    for block in (1 << 30, 64 << 20):
        while torch.cuda.mem_get_info()[0] > leave_bytes + block:
            held.append(torch.empty(block, dtype=torch.uint8, device="cuda"))
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
