"""K9 reachable from AddRecordImpl (VERDICT r04 missing #3).  IndexSchema feeds the vector index ONE key at a time
(src/index_schema.cc:755-791 -> VectorBase::AddRecord -> AddRecordImpl -> addPoint, hnswalg.h:1278-1340); r04 linked each
such call on the host (10M x 768: 1013 s) and built on the device only for vk_index_add_batch, which nothing on the
reference side calls.  Now single adds of new labels are staged and linked in bulk at vk_index_flush (the write -> read
phase switch) -- on the device when >= 4096 wait.  Pinned:
  * 1M x 768 single adds from 16 native writer threads + flush take at most 1.5x the time of one add_batch of the same rows,
    every staged row went through the device build, recall = add_batch's (same builder, same relaxation);
  * at 200k x 768 (where the host builder is affordable inside the suite): recall within half a percent of the host (hnswlib-order) build over 2048 queries,
    and the CPU oracle, loaded from the graph the product SAVED, reproduces the product's answers id for id and bit for bit;
  * addPoint semantics while staged: the same label again = the later row; remove of a staged label; capacity error and the
    resize-and-retry loop (vector_hnsw.cc:238-271); contains / get_row / count see staged rows; a search flushes."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def latent(n, dim, seed, rank=32):
    A = np.random.default_rng(1000).standard_normal((dim, rank)).astype(np.float32)
    r = np.random.default_rng(seed)
    x = r.standard_normal((n, rank)).astype(np.float32) @ A.T + 0.05 * r.standard_normal((n, dim)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def recall(g, gt, Q, k=10, ef=128):
    hit = 0
    for lo in range(0, len(Q), 512):
        D, L, N = g.search_batch(Q[lo:lo + 512], k, ef=ef)
        hit += sum(len(set(L[i, :N[i]].tolist()) & set(gt[lo + i].tolist())) for i in range(L.shape[0]))
    return hit / float(k * len(Q))


def exact(flat, Q, k):
    return np.concatenate([flat.search_batch(Q[lo:lo + 256], k)[1] for lo in range(0, len(Q), 256)])


def test_single_adds_reach_the_device_build(vsa, oracle):
    dim, k = 768, 10
    Q = latent(2048, dim, 99)
    # ---- 1M rows: single adds from 16 threads + flush against one add_batch
    n = 1_000_000
    x = latent(n, dim, 3)
    flat = vsa.Index("FLAT", dim, "IP", initial_cap=n)
    flat.add_batch(x)
    gt = exact(flat, Q, k)
    del flat
    gb = vsa.Index("HNSW", dim, "IP", initial_cap=n, m=16, ef_construction=200, ef_runtime=128)
    t0 = time.perf_counter()
    gb.add_batch(x)
    gb.flush()
    t_batch = time.perf_counter() - t0
    r_batch = recall(gb, gt, Q)
    del gb
    gs = vsa.Index("HNSW", dim, "IP", initial_cap=n, m=16, ef_construction=200, ef_runtime=128)
    t0 = time.perf_counter()
    failed, t_adds = vsa.probe_add_single(gs, x, threads=16)
    gs.flush()
    t_single = time.perf_counter() - t0
    st = gs.stats()
    assert failed == 0 and st.count == n and st.staged_ops == 0
    # (the first 16384 points are linked by the host builder as they come: the device build needs a graph to extend)
    # (... and the writers keep staging while a bulk is being linked, so the LAST bulk -- the flush's -- can be a remainder below
    #  4096 rows, which the host builder's threads link.  Only that one: a writer that waited behind a bulk and finds fewer
    #  rows than a device bulk takes goes back to staging -- before r06 each of the queued writers linked the handful it found,
    #  up to 3 800 rows of this million in bulks of tens, and one run in eight or so crossed the 4096 of this assertion)
    assert st.staged_adds >= n - 16384 - 64 and st.staged_adds - st.staged_adds_device < 4096, (st.staged_adds, st.staged_adds_device)
    r_single = recall(gs, gt, Q)
    print(f"1M x 768: add_batch {t_batch:.1f} s (recall {r_batch:.4f}); 16 threads x single adds {t_adds:.1f} s + flush = {t_single:.1f} s (recall {r_single:.4f}); "
          f"staged {st.staged_adds}, of them linked by the host builder {st.staged_adds - st.staged_adds_device}")
    assert t_single <= 1.5 * t_batch, (t_single, t_batch)
    assert r_single >= r_batch - 0.005, (r_single, r_batch)      # (the same builder: 2048 queries apart)
    del gs, x
    # ---- 200k rows: against the host build, and the oracle on the saved graph
    n = 200_000
    x = latent(n, dim, 4)
    flat = vsa.Index("FLAT", dim, "IP", initial_cap=n)
    flat.add_batch(x)
    gt = exact(flat, Q, k)
    del flat
    gh = vsa.Index("HNSW", dim, "IP", initial_cap=n, m=16, ef_construction=200, ef_runtime=128, options={"hnsw-device-build": 0})
    gh.add_batch(x)
    r_host = recall(gh, gt, Q)
    del gh
    gs = vsa.Index("HNSW", dim, "IP", initial_cap=n, m=16, ef_construction=200, ef_runtime=128)
    failed, _ = vsa.probe_add_single(gs, x, threads=16)
    assert failed == 0
    r_single = recall(gs, gt, Q)          # (the search links what is staged: no explicit flush)
    assert gs.stats().staged_adds_device > 0
    print(f"200k x 768: host build recall {r_host:.4f}, single adds -> device build {r_single:.4f}")
    assert r_single >= r_host - 0.005, (r_single, r_host)        # north_star: recall >= reference at identical ef (2048 queries)
    o = oracle.HNSW.from_product_index(gs.save_raw, dim, "IP", 16, ef_construction=200)
    D, L, N = gs.search_batch(Q[:64], k, ef=128)
    for i in range(64):
        od, ol = o.search(Q[i], k, ef=128)
        assert L[i, :N[i]].tolist() == ol.tolist() and D[i, :N[i]].view(np.uint32).tolist() == od.view(np.uint32).tolist()


def test_addpoint_semantics_while_staged(vsa):
    dim = 32
    rng = np.random.default_rng(8)
    n0 = 20_000
    x = rng.standard_normal((n0 + 6000, dim)).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n0 + 5000, m=8, ef_construction=64, ef_runtime=64)
    g.add_batch(x[:n0])
    g.flush()
    # staged: visible to count / contains / get_row at once, to searches after the implicit flush
    for i in range(n0, n0 + 5000):
        assert g.add(i, x[i]) == vsa.VK_OK
    st = g.stats()
    assert st.count == n0 + 5000 and st.staged_ops >= 5000 and st.staged_adds == 5000
    assert g.contains(n0 + 17) and g.get_row(n0 + 17).tolist() == x[n0 + 17].tolist()
    # the same label again while staged: the later row is the one that is linked
    assert g.add(n0 + 17, x[n0 + 5017]) == vsa.VK_OK and g.stats().count == n0 + 5000
    assert g.get_row(n0 + 17).tolist() == x[n0 + 5017].tolist()
    # a staged label removed again never enters the graph
    assert g.remove(n0 + 18) == vsa.VK_OK and not g.contains(n0 + 18) and g.stats().count == n0 + 4999
    # capacity: "The number of elements exceeds the specified limit" -> the caller resizes and retries (vector_hnsw.cc:238-271)
    assert g.add(n0 + 5000, x[n0 + 5000]) == vsa.VK_OK          # (18 left a free place)
    assert g.add(n0 + 5001, x[n0 + 5001]) == vsa.VK_ERR_CAPACITY
    assert b"exceeds the specified limit" in vsa.lib().vk_last_error()
    g.resize(n0 + 6000)
    assert g.add(n0 + 5001, x[n0 + 5001]) == vsa.VK_OK
    # an update of a LINKED label is not staged: it is the host builder's updatePoint
    assert g.add(5, x[n0 + 5005]) == vsa.VK_OK and g.stats().staged_adds == 5002
    d, l = g.search(x[n0 + 5017], 1, ef=64)                      # the search links what was staged (4096+ rows: on the device)
    assert l.tolist() == [n0 + 17] and d[0] == 0.0
    st = g.stats()
    assert st.staged_ops == 0 and st.count == n0 + 5001 and st.staged_adds_device == 5001
    d, l = g.search(x[n0 + 18], 5, ef=64)
    assert n0 + 18 not in l.tolist()
    d, l = g.search(x[n0 + 5005], 1, ef=64)
    assert l.tolist() == [5]
    # fewer than 4096 staged rows go to the host builder; staging can be switched off
    for i in range(n0 + 5002, n0 + 5100):
        assert g.add(i, x[i]) == vsa.VK_OK
    g.flush()
    st2 = g.stats()
    assert st2.staged_adds == st.staged_adds + 98 and st2.staged_adds_device == st.staged_adds_device
    g.set_option("hnsw-stage-adds", 0)
    assert g.add(n0 + 5100, x[n0 + 5100]) == vsa.VK_OK and g.stats().staged_adds == st2.staged_adds


def test_a_delete_is_not_lost_while_its_bulk_is_being_linked(vsa):
    """drain_pending() swaps the staged rows out and links them later; in between, a remove / contains /
    get_row / second add of a label whose add() had already returned found the label NOWHERE -- the delete was lost and the
    vector linked behind it (a ghost for a deleted key).  Four writer threads stage single adds with a small staging area (a
    bulk is linked every 8192 rows, on the device, by whichever writer fills it, while the others keep adding) and remove every fifth label
    right after adding it; a fifth of the rest is added a second time with another row.  Afterwards: every removed label is
    gone (not contained, no distance, never a search result), every other label is there with its LAST row, the count is right."""
    import threading
    dim, n0, per, T = 16, 20_000, 12_000, 4
    rng = np.random.default_rng(81)
    base = rng.standard_normal((n0, dim)).astype(np.float32)
    new = rng.standard_normal((T * per, dim)).astype(np.float32)
    again = rng.standard_normal((T * per, dim)).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n0 + T * per, m=8, ef_construction=40, ef_runtime=64, options={"hnsw-stage-max": 8192})
    g.add_batch(base)
    g.flush()
    removed, readded, errors = [set() for _ in range(T)], [set() for _ in range(T)], []

    def writer(t):
        try:
            for i in range(per):
                j = t * per + i
                lab = n0 + j
                assert g.add(lab, new[j]) == vsa.VK_OK
                if j % 5 == 0:
                    assert g.remove(lab) == vsa.VK_OK, lab
                    assert not g.contains(lab), lab
                    removed[t].add(lab)
                elif j % 5 == 1:
                    assert g.contains(lab), lab
                    assert g.add(lab, again[j]) == vsa.VK_OK          # the same label again: an update, never a second element
                    readded[t].add(lab)
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=writer, args=(t,)) for t in range(T)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]
    g.flush()
    gone = set().union(*removed)
    twice = set().union(*readded)
    st = g.stats()
    assert st.staged_ops == 0 and st.staged_adds_device > 0
    # (a label removed while staged never entered the graph; one removed after it was linked is a tombstone)
    assert st.count - st.deleted == n0 + T * per - len(gone), (st.count, st.deleted, len(gone))
    probe = rng.permutation(T * per)[:3000]
    for j in probe:
        lab = n0 + int(j)
        if lab in gone:
            assert not g.contains(lab) and g.distance(lab, new[j]) is None, lab
        else:
            want = again[j] if lab in twice else new[j]
            assert g.contains(lab) and g.get_row(lab).tolist() == want.tolist(), lab
    qs = np.stack([new[int(j)] for j in probe[:512]])
    D, L, N = g.search_batch(qs, 5, ef=64)
    assert not set(L.ravel().tolist()) & gone, "a deleted key came back from a search"
    # the live probes find themselves (an approximate index: nearly all of them, vector_test.cc's self-retrieval bar)
    live = [(r, n0 + int(j)) for r, j in enumerate(probe[:512]) if n0 + int(j) not in gone and n0 + int(j) not in twice]
    hit = sum(int(L[r, 0] == lab and D[r, 0] == 0.0) for r, lab in live)
    assert hit >= 0.97 * len(live), (hit, len(live))
