"""hnsw-allow-replace-deleted through the C ABI against the CPU oracle.

Reference: `addPoint(data, label, replace_deleted)` third_party/hnswlib/hnswalg.h:1278-1340, switched on by the config
`hnsw-allow-replace-deleted` (src/valkey_search_options.cc:149-152) at src/indexes/vector_hnsw.cc:99-100,153-154 and passed on
every AddRecordImpl (:182-183); ResizeIfFull does not grow while a tombstoned slot is vacant (:231-236).

What can be pinned: WHICH vacant slot a new label takes is `*deleted_elements.begin()` of a std::unordered_set -- the
container's choice, not the algorithm's -- so the product chooses, the choice is read back from its own save stream and
replayed into an independently driven oracle (which refuses a slot that is not vacant).  updatePoint iterates unordered_sets
too (link lists compared as sets, like tests/test_host_graph.py).  Answers are compared bit for bit against the oracle
searching the product's saved graph, and against the independently driven oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _same(gd, gl, od, ol):
    assert gl.tolist() == ol.tolist()
    assert gd.view(np.uint32).tolist() == od.view(np.uint32).tolist()


def _links_as_sets(e):
    out = {}
    for i in range(len(e["levels"])):
        out[(i, 0)] = frozenset(e["l0"][i, 1:1 + (int(e["l0"][i, 0]) & 0xFFFF)].tolist())
        for lv in range(1, int(e["levels"][i]) + 1):
            out[(i, lv)] = frozenset(e["upper"][(i, lv)].tolist())
    return out


@pytest.mark.parametrize("metric,dim,M,efc", [("L2", 48, 12, 80), ("COSINE", 96, 16, 100)])
def test_interleaved_remove_add_update_search(vsa, oracle, metric, dim, M, efc):
    rng = np.random.default_rng(606)
    n, extra = 2500, 64
    pool = rng.standard_normal((n + 3000, dim)).astype(np.float32)
    if metric == "COSINE":
        pool /= np.linalg.norm(pool, axis=1, keepdims=True)
        pool = np.stack([oracle.normalize(v)[0] for v in pool])
    g = vsa.Index("HNSW", dim, metric, initial_cap=n + extra, m=M, ef_construction=efc, ef_runtime=10, build_threads=1,
                  allow_replace_deleted=True)
    o = oracle.HNSW(dim, metric, max_elements=n + extra, M=M, ef_construction=efc, ef=10, allow_replace_deleted=True)
    for i in range(n):
        assert g.add(i, pool[i]) == 0
    o.add_many(pool[:n])
    live = set(range(n))
    next_label = n
    Q = rng.standard_normal((16, dim)).astype(np.float32)
    if metric == "COSINE":
        Q = np.stack([oracle.normalize(v)[0] for v in Q])

    def check(tag):
        saved = oracle.HNSW.from_product_index(g.save_raw, dim, metric, M, ef_construction=efc)
        agree = 0
        for ef in (10, 64):
            D, L, N = g.search_batch(Q, 10, ef=ef)
            for qi, q in enumerate(Q):
                sd, sl = saved.search(q, 10, ef=ef)
                _same(D[qi][:N[qi]], L[qi][:N[qi]], sd, sl)            # bit for bit on the saved graph
                assert set(sl.tolist()) <= live, tag
                od, ol = o.search(q, 10, ef=ef)
                agree += sl.tolist() == ol.tolist() and sd.view(np.uint32).tolist() == od.view(np.uint32).tolist()
        # the independently driven oracle: same graph up to the unordered_set iteration order inside updatePoint
        a, b = saved.export_graph(), o.export_graph()
        assert a["labels"].tolist() == b["labels"].tolist() and a["deleted"].tolist() == b["deleted"].tolist(), tag
        assert a["levels"].tolist() == b["levels"].tolist(), tag
        assert (a["entry_point"], a["max_level"]) == (b["entry_point"], b["max_level"]), tag
        la, lb = _links_as_sets(a), _links_as_sets(b)
        same = sum(la[k] == lb[k] for k in la)
        assert same >= 0.97 * len(la), (tag, same, len(la))
        assert agree >= 0.9 * 2 * len(Q), (tag, agree)
        return saved, same, len(la), agree

    check("built")
    for rnd in range(5):
        # -- remove
        dead = [int(v) for v in rng.choice(sorted(live), 40 + 25 * rnd, replace=False)]
        for lab in dead:
            assert g.remove(lab) == 0 and o.mark_delete(lab) == 0
            live.discard(lab)
        assert g.remove(dead[0]) != 0                                   # "already deleted"
        st = g.stats()
        assert st.deleted == len(o.vacant()) == o.deleted_count
        check("removed %d" % rnd)                                       # searches see the tombstones (hnswalg.h:515-524)
        # -- new labels take the vacant slots (one of them is a deleted label coming back: un-deleted in place)
        count_before = g.stats().count
        back = dead[3]
        adds = [(back, rng.standard_normal(dim).astype(np.float32))]
        vac_before = o.deleted_count
        n_new = len(dead) - 10 if rnd % 2 == 0 else vac_before + 7      # odd rounds: more labels than vacancies -> growth
        for _ in range(n_new):
            adds.append((next_label, pool[next_label]))
            next_label += 1
        # ... interleaved with updates of live labels
        for lab in rng.choice(sorted(live), 6, replace=False):
            adds.insert(int(rng.integers(1, len(adds))), (int(lab), rng.standard_normal(dim).astype(np.float32)))
        if metric == "COSINE":
            adds = [(lab, oracle.normalize(row)[0]) for lab, row in adds]
        for lab, row in adds:
            assert g.add(lab, row) == 0
            live.add(lab)
        saved = oracle.HNSW.from_product_index(g.save_raw, dim, metric, M, ef_construction=efc)
        slot_of = {int(l): i for i, l in enumerate(saved.export_graph()["labels"].tolist())}
        for lab, row in adds:
            assert o.add_into(row, lab, slot_of[lab]) == 0, oracle.last_error()   # (refuses a slot that is not vacant)
        grew = max(0, 1 + n_new - vac_before)
        assert g.stats().count == count_before + grew == o.count
        assert g.stats().deleted == o.deleted_count == max(0, vac_before - 1 - n_new)
        _, same, total, agree = check("added %d" % rnd)
        print("round %d: %d/%d link lists equal as sets, %d/%d answers equal the independently driven oracle's"
              % (rnd, same, total, agree, 2 * len(Q)))
    # every label that was replaced is gone for good, the ones that came back are live
    for lab in range(n):
        assert g.contains(lab) == (lab in live)


def test_full_index_takes_a_vacant_slot_without_growing(vsa, oracle):
    """ResizeIfFull (vector_hnsw.cc:227-236): at capacity an add succeeds iff a tombstoned slot is vacant; the capacity
    stays.  Without a vacancy the add fails with the capacity error the caller answers with resize + retry."""
    rng = np.random.default_rng(607)
    n, dim = 600, 32
    x = rng.standard_normal((n + 8, dim)).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=8, ef_construction=40, build_threads=1, allow_replace_deleted=True,
                  block_size=128)
    for i in range(n):
        assert g.add(i, x[i]) == 0
    assert g.stats().capacity == n
    assert g.remove(17) == 0
    assert g.add(n, x[n]) == 0
    st = g.stats()
    assert (st.capacity, st.count, st.deleted) == (n, n, 0)
    assert not g.contains(17) and g.contains(n)
    d, l = g.search(x[n], 1, ef=50)
    assert l.tolist() == [n]
    assert g.get_row(n).view(np.uint32).tolist() == x[n].view(np.uint32).tolist()
    # no vacancy: "The number of elements exceeds the specified limit" -- the caller resizes by its block size and retries
    # (vector_hnsw.cc:186-192; the adaptor's AddRecordImpl does exactly that)
    assert g.add(n + 1, x[n + 1]) == vsa.VK_ERR_CAPACITY
    g.resize(n + 128)
    assert g.add(n + 1, x[n + 1]) == 0
    st = g.stats()
    assert st.count == n + 1 and st.capacity == n + 128


def test_loaded_index_reuses_its_tombstones(vsa, oracle):
    """LoadIndex with allow_replace_deleted_ set refills deleted_elements from the tombstone bits (hnswalg.h:1110-1117)."""
    rng = np.random.default_rng(608)
    n, dim, M = 800, 40, 8
    x = rng.standard_normal((n + 40, dim)).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n + 40, m=M, ef_construction=40, build_threads=1, allow_replace_deleted=True)
    for i in range(n):
        assert g.add(i, x[i]) == 0
    dead = list(range(100, 120))
    for lab in dead:
        assert g.remove(lab) == 0
    chunks = g.save()
    h = vsa.Index.load(chunks, "HNSW", dim, "L2", initial_cap=n + 40, m=M, ef_construction=40, build_threads=1,
                       allow_replace_deleted=True)
    assert h.stats().deleted == 20
    for j in range(25):
        assert h.add(n + j, x[n + j]) == 0
    st = h.stats()
    assert st.deleted == 0 and st.count == n + 5
    saved = oracle.HNSW.from_product_index(h.save_raw, dim, "L2", M, ef_construction=40)
    labels = saved.export_graph()["labels"].tolist()
    assert sorted(labels[100:120]) == list(range(n, n + 20)) and labels[n:] == list(range(n + 20, n + 25))
    for q in rng.standard_normal((8, dim)).astype(np.float32):
        _same(*h.search(q, 10, ef=64), *saved.search(q, 10, ef=64))


def test_bulk_adds_fill_vacancies_first(vsa, oracle):
    """A bulk of new labels that arrives while tombstoned slots are vacant cannot be linked as one device bulk (every
    insert may reuse a slot): it takes the host builder, fills the vacancies and grows by the rest."""
    rng = np.random.default_rng(609)
    n, dim, M = 20000, 32, 8
    A = rng.standard_normal((dim, 8)).astype(np.float32)
    x = (rng.standard_normal((n + 5000, 8)).astype(np.float32) @ A.T + 0.05 * rng.standard_normal((n + 5000, dim))).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n + 5000, m=M, ef_construction=60, allow_replace_deleted=True)
    g.add_batch(x[:n])
    dead = rng.choice(n, 300, replace=False)
    for lab in dead:
        assert g.remove(int(lab)) == 0
    # (a multi-threaded build hands out slots in arrival order: where the dead labels sit is read off the graph)
    before = np.array(oracle.HNSW.from_product_index(g.save_raw, dim, "L2", M, ef_construction=60).export_graph()["labels"])
    dead_slots = np.flatnonzero(np.isin(before, dead))
    assert len(dead_slots) == 300
    g.add_batch(x[n:], np.arange(n, n + 5000, dtype=np.uint64))
    st = g.stats()
    assert st.deleted == 0 and st.count == n + 5000 - 300
    saved = oracle.HNSW.from_product_index(g.save_raw, dim, "L2", M, ef_construction=60)
    labels = np.array(saved.export_graph()["labels"])
    assert np.all(labels[dead_slots] >= n) and len(set(labels.tolist())) == st.count == len(labels)
    f = vsa.Index("FLAT", dim, "L2", initial_cap=n + 5000)
    keep = np.setdiff1d(np.arange(n + 5000), dead)
    f.add_batch(x[keep], keep.astype(np.uint64))
    Q = x[rng.choice(keep, 64, replace=False)] + 0.01 * rng.standard_normal((64, dim)).astype(np.float32)
    D, L, N = g.search_batch(Q, 10, ef=128)
    _, Lf, _ = f.search_batch(Q, 10)
    hit = 0
    for qi, q in enumerate(Q):
        _same(D[qi][:N[qi]], L[qi][:N[qi]], *saved.search(q, 10, ef=128))
        hit += len(set(L[qi].tolist()) & set(Lf[qi].tolist()))
    assert hit / 640.0 >= 0.9


def test_reference_cases_through_the_abi(vsa):
    """testing/vector_test.cc:583-617 AllowReplaceDeletedNoLabelReuse (ten keys, two removed, five new ones: 13 nodes, a search
    for 13 finds 13) and :973-1001 (a known label is updated in its own slot while another slot is tombstoned)."""
    from conftest import reference_vectors
    v, w = reference_vectors(10, 100, 10.0), reference_vectors(5, 100, 20.0)
    g = vsa.Index("HNSW", 100, "L2", initial_cap=15000, m=16, ef_construction=20, ef_runtime=20, build_threads=1, allow_replace_deleted=True)
    for i in range(10):
        assert g.add(i, v[i]) == 0
    assert g.remove(8) == 0 and g.remove(9) == 0
    st = g.stats()
    assert (st.count, st.deleted, st.max_label) == (10, 2, 9)
    for i in range(5):
        assert g.add(10 + i, w[i]) == 0
    st = g.stats()
    assert (st.count, st.deleted) == (13, 0)                     # "Verifies we reused tombstoned hnsw nodes"
    d, l = g.search(w[0], 13)
    assert sorted(l.tolist()) == [0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14]
    h = vsa.Index("HNSW", 100, "L2", initial_cap=1000, m=16, ef_construction=20, build_threads=1, allow_replace_deleted=True)
    assert h.add(0, v[0]) == 0 and h.add(1, v[1]) == 0 and h.remove(0) == 0
    assert h.add(1, v[0]) == 0
    st = h.stats()
    assert (st.count, st.deleted) == (2, 1) and not h.contains(0) and h.contains(1)
    assert h.get_row(1).tolist() == v[0].tolist()


@pytest.mark.parametrize("allow", [True, False])
def test_reload_with_tombstones_then_new_labels(vsa, allow):
    """integration/test_hnsw_allow_replace_deleted.py: ten vectors, the two highest labels deleted, SAVE + restart, five new
    vectors -- no add error under either setting (the id counter resumes behind the largest label the stream holds, tombstoned
    or not: GetMaxInternalLabel, vector_hnsw.cc:387-394), 13 documents, and KNN 13 finds all 13."""
    rows = np.array([[float(i) + 0.1 * d for d in range(4)] for i in range(10)], np.float32)
    new = np.array([[100.0 + i + 0.1 * d for d in range(4)] for i in range(5)], np.float32)
    g = vsa.Index("HNSW", 4, "L2", initial_cap=1024, m=16, ef_construction=200, ef_runtime=10, build_threads=1, allow_replace_deleted=allow)
    for i in range(10):
        assert g.add(i, rows[i]) == 0
    assert g.remove(8) == 0 and g.remove(9) == 0
    chunks = g.save()
    h = vsa.Index.load(chunks, "HNSW", 4, "L2", initial_cap=1024, m=16, ef_construction=200, ef_runtime=10, build_threads=1,
                       allow_replace_deleted=allow)
    st = h.stats()
    assert (st.count, st.deleted, st.max_label) == (10, 2, 9)
    for i in range(5):
        assert h.add(int(st.max_label) + 1 + i, new[i]) == 0          # inc_id_ = GetMaxInternalLabel() + 1 (vector_base.cc:480-481)
    st = h.stats()
    assert (st.count, st.deleted) == ((13, 0) if allow else (15, 2))
    d, l = h.search(np.array([50.0, 50.1, 50.2, 50.3], np.float32), 13, ef=13)
    assert sorted(l.tolist()) == [0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14]
