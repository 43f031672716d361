"""vk_index_add_batch with everything a caller can put into one batch: new labels, labels the index holds (updates), a label
twice in the batch (the later row wins: addPoint of an existing label is an update, bruteforce.h:66-83, hnswalg.h:1278-1340),
labels removed earlier coming back, and the batch that hits the capacity limit part of the way (addPoint throws there:
everything before it is in, the rest is not).  FLAT takes runs of new labels by one strided copy and the rest through the
staging log -- the order of effects must be the order of the rows.  Checked against the oracle fed one row at a time."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OFFSET = int(os.environ.get("VK_SWEEP_OFFSET", "0"))      # other random batches: VK_SWEEP_OFFSET=<n>


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _same(gd, gl, od, ol):
    assert gl.tolist() == ol.tolist()
    assert gd.view(np.uint32).tolist() == od.view(np.uint32).tolist()


@pytest.mark.parametrize("shards", [0, 3])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_flat_mixed_batches(vsa, oracle, seed, shards):
    rng = np.random.default_rng(100 + seed + 1000 * OFFSET)
    dim, k, cap = 24, 10, 2500          # (reached in the fourth or fifth batch)
    kw = dict(shard_devices=[0] * shards) if shards else {}
    g = vsa.Index("FLAT", dim, "L2", initial_cap=cap, **kw)
    o = oracle.Flat(dim, "L2", max_elements=cap)
    held = set()
    latest = {}
    next_label = 0
    hit_capacity = 0
    for step in range(8):
        n = int(rng.integers(300, 1200))
        labels = np.empty(n, dtype=np.uint64)
        for i in range(n):
            r = rng.random()
            if r < 0.55 or not held:
                labels[i] = next_label
                next_label += 1
            elif r < 0.8:
                labels[i] = rng.choice(list(held)) if rng.random() < 0.5 else labels[rng.integers(0, i)] if i else next_label
            else:
                labels[i] = rng.integers(0, max(next_label, 1))      # may be held, removed or never seen
        rows = rng.standard_normal((n, dim)).astype(np.float32)
        # the oracle, row by row (stops at the capacity error like addPoint does)
        want_err = False
        for i in range(n):
            if int(labels[i]) not in held and len(held) >= cap:
                want_err = True
                break
            o.add(rows[i], int(labels[i]))
            held.add(int(labels[i]))
            latest[int(labels[i])] = rows[i]
        try:
            g.add_batch(rows, labels=labels)
            assert not want_err
        except vsa.VkError as e:
            assert want_err and e.code == vsa.VK_ERR_CAPACITY, e
            hit_capacity += 1
        g.flush()
        assert g.stats().count == len(held)
        for lab in rng.choice(list(held), min(len(held), 120), replace=False):
            if rng.random() < 0.5:
                g.remove(int(lab))
                o.remove(int(lab))
                held.discard(int(lab))
        Q = rng.standard_normal((6, dim)).astype(np.float32)
        for q in Q[:3]:
            _same(*g.search(q, k), *o.search(q, k))
        D, L, N = g.search_batch(Q, k)
        for i, q in enumerate(Q):
            od, ol = o.search(q, k)
            _same(D[i, :N[i]], L[i, :N[i]], od, ol)
        for lab in rng.choice(list(held), 20, replace=False):
            assert np.array_equal(g.get_row(int(lab)), latest[int(lab)])
    assert hit_capacity >= 1 or OFFSET != 0        # (the sizes are tuned so that the default seeds reach the limit)


@pytest.mark.parametrize("shards", [0, 3])
@pytest.mark.parametrize("seed", [1, 2])
def test_hnsw_mixed_batches(vsa, oracle, seed, shards):
    """the same for HNSW (updates relink, removed labels come back as updates of their tombstone): count, tombstones, the row
    behind every label, and the answers of the saved graph walked by the oracle"""
    rng = np.random.default_rng(200 + seed + 1000 * OFFSET)
    dim, k, cap = 24, 10, 2200          # (reached around the fifth batch; a tombstone keeps its slot)
    kw = dict(shard_devices=[0] * shards) if shards else {}
    g = vsa.Index("HNSW", dim, "L2", initial_cap=cap, m=8, ef_construction=48, ef_runtime=48, **kw)
    live, dead = set(), set()
    latest = {}
    next_label = 0
    hit_capacity = 0
    for step in range(7):
        n = int(rng.integers(300, 1200))
        labels = np.empty(n, dtype=np.uint64)
        for i in range(n):
            r = rng.random()
            if r < 0.55 or not live:
                labels[i] = next_label
                next_label += 1
            elif r < 0.8:
                labels[i] = rng.choice(list(live)) if rng.random() < 0.5 else labels[rng.integers(0, i)] if i else next_label
            else:
                labels[i] = rng.integers(0, max(next_label, 1))
        rows = rng.standard_normal((n, dim)).astype(np.float32)
        want_err = False
        for i in range(n):
            lab = int(labels[i])
            if lab not in live and lab not in dead and len(live) + len(dead) >= cap:   # (a tombstone keeps its slot)
                want_err = True
                break
            live.add(lab)
            dead.discard(lab)                          # addPoint of a deleted label brings it back (hnswalg.h:1300-1316)
            latest[lab] = rows[i]
        try:
            g.add_batch(rows, labels=labels)
            assert not want_err
        except vsa.VkError as e:
            assert want_err and e.code == vsa.VK_ERR_CAPACITY, e
            hit_capacity += 1
        g.flush()
        st = g.stats()
        assert (st.count, st.deleted) == (len(live) + len(dead), len(dead)), (st.count, st.deleted, len(live), len(dead))
        for lab in rng.choice(list(live), min(len(live), 100), replace=False):
            if rng.random() < 0.5:
                assert g.remove(int(lab)) == vsa.VK_OK
                live.discard(int(lab))
                dead.add(int(lab))
        g.flush()
        wrong = [lab for lab in live if not np.array_equal(g.get_row(lab), latest[lab])]
        assert not wrong, (step, len(wrong), wrong[:5], [int((labels == w).sum()) for w in wrong[:5]])
        if not shards:
            o = oracle.HNSW.from_product_index(g.save_raw, dim, "L2", 8, ef_construction=48)
            for q in rng.standard_normal((5, dim)).astype(np.float32):
                _same(*g.search(q, k, ef=48), *o.search(q, k, ef=48))
        # (self-retrieval is NOT guaranteed under this load: updatePoint's repair and the tombstones leave a per cent or two of
        #  the nodes without a useful way in -- the single-threaded oracle fed the same operations loses 4 .. 53 of 400 .. 2500
        #  live labels at ef = 200.  What is asserted is the order of magnitude.)
        probe = rng.choice(list(live), min(len(live), 300), replace=False)
        lost = sum(g.search(latest[int(lab)], 1, ef=200)[1].tolist() != [int(lab)] for lab in probe)
        assert lost <= max(3, len(probe) // 15), (step, lost, len(probe))
        for lab in list(dead)[:20]:
            assert not g.contains(int(lab))
    assert hit_capacity >= 1 or OFFSET != 0        # (the sizes are tuned so that the default seeds reach the limit)
