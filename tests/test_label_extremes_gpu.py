"""Labels are the reference's size_t labeltype (hnswlib.h:141), in valkey-search VectorBase's counter from zero.  Any 64-bit
value must work -- beyond 2^32, beyond 2^63, one below the maximum -- except UINT64_MAX itself, which is the padding label of
result lists ((+inf, UINT64_MAX) past the count): r06 found a FLAT row under that label missing from its own neighbourhood, and
the label is refused at the door now (add, add_batch, the device bulk load, a stream being loaded)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LABELS = np.array([2**64 - 2, 2**64 - 3, 2**63, 2**63 - 1, 2**32, 2**32 - 1, 0, 1, 2**40 + 7, 12345678901234567], dtype=np.uint64)


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


@pytest.mark.parametrize("algo", ["FLAT", "HNSW"])
@pytest.mark.parametrize("metric", ["L2", "IP"])
def test_extreme_labels(vsa, oracle, algo, metric):
    rng = np.random.default_rng(41)
    n, dim, k = 600, 40, 12
    x = rng.standard_normal((n, dim)).astype(np.float32)
    labels = rng.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    labels[: len(LABELS)] = LABELS
    assert len(set(labels.tolist())) == n
    x[1] = x[0]                                   # the two largest labels at the same distance from everything: tie by label
    Q = rng.standard_normal((20, dim)).astype(np.float32)
    Q[0] = x[0]
    g = vsa.Index(algo, dim, metric, initial_cap=n, m=8, ef_construction=64, ef_runtime=64)
    g.add_batch(x, labels=labels)
    g.flush()
    if algo == "FLAT":
        o = oracle.Flat(dim, metric, max_elements=n)
        o.add_many(x, labels)
        for q in Q:
            od, ol = o.search(q, k)
            gd, gl = g.search(q, k)
            assert gl.tolist() == ol.tolist() and gd.view(np.uint32).tolist() == od.view(np.uint32).tolist()
        D, L, N = g.search_batch(Q, k)            # (the matrix-core path)
        for i, q in enumerate(Q):
            od, ol = o.search(q, k)
            assert N[i] == k and L[i].tolist() == ol.tolist() and D[i].view(np.uint32).tolist() == od.view(np.uint32).tolist()
        d, l = g.search(Q[0], n + 5)              # more than there is: every row once, the padding never
        assert len(l) == n and sorted(l.tolist()) == sorted(labels.tolist())
    else:
        o = oracle.HNSW.from_product_index(g.save_raw, dim, metric, 8, ef_construction=64)
        for q in Q:
            od, ol = o.search(q, k, ef=64)
            gd, gl = g.search(q, k, ef=64)
            assert gl.tolist() == ol.tolist() and gd.view(np.uint32).tolist() == od.view(np.uint32).tolist()
        d, l = g.search(Q[0], 2, ef=64)
        assert sorted(l.tolist()) == sorted(LABELS[:2].tolist())
    for j in range(len(LABELS)):
        assert np.array_equal(g.get_row(int(LABELS[j])), x[j]) and g.contains(int(LABELS[j]))
    assert g.stats().max_label == 2**64 - 2
    g.remove(int(LABELS[0]))
    g.flush()
    d, l = g.search(Q[0], 1)
    assert l.tolist() == [int(LABELS[1])]
    chunks = g.save()
    h = vsa.Index.load(chunks, algo, dim, metric, m=8, ef_construction=64, ef_runtime=64)
    d2, l2 = h.search(Q[0], 5)
    d1, l1 = g.search(Q[0], 5)
    assert l1.tolist() == l2.tolist() and d1.view(np.uint32).tolist() == d2.view(np.uint32).tolist()


@pytest.mark.parametrize("algo", ["FLAT", "HNSW"])
def test_the_padding_label_is_refused(vsa, algo):
    rng = np.random.default_rng(43)
    n, dim = 300, 16
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = vsa.Index(algo, dim, "L2", initial_cap=n + 10, m=8, ef_construction=32)
    g.add_batch(x)
    assert g.add(2**64 - 1, x[0]) == vsa.VK_ERR_INVALID
    bad = np.arange(1000, 1005, dtype=np.uint64)
    bad[3] = 2**64 - 1
    with pytest.raises(vsa.VkError) as e:
        g.add_batch(x[:5], labels=bad)
    assert e.value.code == vsa.VK_ERR_INVALID and "reserved" in e.value.msg
    g.flush()
    assert g.stats().count == n and not g.contains(1000)          # nothing of the refused batch got in
    # a stream that carries it (not something the reference can write: its labels count up from zero)
    chunks = g.save()
    victim = next(j for j in range(1, len(chunks)) if len(chunks[j]) >= dim * 4 + 8 and chunks[j][-8:] == (7).to_bytes(8, "little"))
    forged = list(chunks)
    forged[victim] = chunks[victim][:-8] + b"\xff" * 8
    with pytest.raises(vsa.VkError) as e:
        vsa.Index.load(forged, algo, dim, "L2", m=8, ef_construction=32)
    assert "reserved" in e.value.msg
