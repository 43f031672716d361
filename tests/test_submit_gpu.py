"""The non-blocking search entry, run-time options and cumulative statistics through the C ABI on a real device.

vk_index_search_submit is the shape of query::SearchAsync (src/query/search.cc:886-910): single-query requests are queued
(max-query-queue-depth, src/valkey_search_options.cc:231-234), the dispatcher keeps `batches-in-flight` device batches
going, a callback completes each request.  Every answer must be the oracle's (ids and distance bits), however the requests
were batched."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _unit(x):
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


@pytest.fixture(scope="module")
def flat(vsa, oracle):
    rng = np.random.default_rng(31)
    n, dim = 120_000, 96
    x = _unit(rng.standard_normal((n, dim)).astype(np.float32))
    Q = _unit(rng.standard_normal((192, dim)).astype(np.float32))
    ix = vsa.Index("FLAT", dim, "COSINE", initial_cap=n, options={"filter-prepass-rows": 1024, "filter-min-rows": 32768})
    ix.add_batch(x)
    o = oracle.Flat(dim, "COSINE", max_elements=n)
    o.add_many(x)
    ref = [o.search(q, 10) for q in Q]
    rd = np.stack([r[0] for r in ref])
    rl = np.stack([r[1] for r in ref])
    return ix, Q, rd, rl


def test_submitted_requests_get_the_oracles_answers(vsa, flat):
    ix, Q, rd, rl = flat
    ix.set_coalescing(64, 300)
    try:
        r = vsa.probe_submit(ix, Q, 10, total=6000, producers=4, window=512, ref=(rd, rl))
        assert r.completed == 6000 and r.mismatches == 0 and r.errors == 0 and r.rejected == 0
        assert r.device_batches < 6000 // 8 and r.mean_batch > 8        # requests did travel together
        assert r.max_batches_in_flight == 2                              # ... in two batches at a time (the default)
        st = ix.stats()
        assert st.submitted >= 6000 and st.queued_now == 0 and st.searches >= 6000
        assert sum(st.latency_hist) >= 6000 and st.latency_sum_ns > 0
        # one batch in flight when asked; same answers
        ix.set_option("batches-in-flight", 1)
        r1 = vsa.probe_submit(ix, Q, 10, total=2000, producers=2, window=256, ref=(rd, rl))
        assert r1.completed == 2000 and r1.mismatches == 0 and r1.errors == 0
        ix.set_option("batches-in-flight", 2)
        # the blocking entry shares the lanes: same answers
        rb = vsa.probe_blocking(ix, Q, 10, threads=48, calls=20, ref=(rd, rl))
        assert rb.completed == 960 and rb.mismatches == 0 and rb.errors == 0 and rb.mean_batch > 2
    finally:
        ix.set_coalescing(0, 0)


def test_python_callbacks_and_a_full_queue(vsa, flat):
    ix, Q, rd, rl = flat
    with pytest.raises(vsa.VkError) as e:                        # submit needs coalescing on
        ix.submit(Q[0], 10, lambda s: None)
    assert e.value.code == vsa.VK_ERR_INVALID
    ix.set_coalescing(32, 2000)
    try:
        done = threading.Semaphore(0)
        hs = [ix.submit(Q[i], 10, lambda s: done.release()) for i in range(40)]
        for _ in hs:
            assert done.acquire(timeout=30)
        for i, h in enumerate(hs):
            d, l = h.result()
            assert h.status == vsa.VK_OK and l.tolist() == rl[i].tolist() and d.view(np.uint32).tolist() == rd[i].view(np.uint32).tolist()
        # a queue of depth 4 under a burst: some submissions are refused with VK_ERR_BUSY, the accepted ones complete
        ix.set_option("max-query-queue-depth", 4)
        ix.set_coalescing(4, 50_000)                             # (a long window: the queue stays full while the burst arrives)
        ok, busy = [], 0
        for i in range(64):
            try:
                ok.append((i, ix.submit(Q[i], 10, lambda s: done.release())))
            except vsa.VkError as err:
                assert err.code == vsa.VK_ERR_BUSY
                busy += 1
        assert busy > 0 and len(ok) + busy == 64
        for _ in ok:
            assert done.acquire(timeout=30)
        for i, h in ok:
            assert h.status == vsa.VK_OK and h.result()[1].tolist() == rl[i].tolist()
        assert ix.stats().rejected >= busy
    finally:
        ix.set_option("max-query-queue-depth", 100000)
        ix.set_coalescing(0, 0)


def test_hnsw_submissions_with_filters_and_tokens(vsa, oracle):
    """HNSW: submitted requests with their own filter bitmaps and cancellation tokens; a token that is up when the batch
    forms is answered VK_ERR_CANCELLED (vector_hnsw.cc:327-329) without a search"""
    rng = np.random.default_rng(8)
    n, dim, M = 20_000, 48, 12
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=80, ef_runtime=64, build_threads=1)
    g.add_batch(x)
    g.flush()
    o = oracle.HNSW.from_product_index(g.save_raw, dim, "L2", M, ef_construction=80)
    Q = rng.standard_normal((96, dim)).astype(np.float32)
    bits = oracle.allow_bitmap(np.flatnonzero(rng.random(n) < 0.3), n)
    g.set_coalescing(48, 1000)
    try:
        done = threading.Semaphore(0)
        flags = [C.c_int(1 if i % 11 == 3 else 0) for i in range(96)]
        hs = [g.submit(Q[i], 10, lambda s: done.release(), ef=64, allow=bits if i % 2 else None, allow_nbits=n,
                       cancel=flags[i], partial_ok=False) for i in range(96)]
        for _ in hs:
            assert done.acquire(timeout=60)
        for i, h in enumerate(hs):
            if i % 11 == 3:
                assert h.status == vsa.VK_ERR_CANCELLED and h.n[0] == 0
                continue
            e_d, e_l = o.search(Q[i], 10, ef=64, allow=bits if i % 2 else None, allow_nbits=n)
            d, l = h.result()
            assert h.status == vsa.VK_OK and l.tolist() == e_l.tolist() and d.view(np.uint32).tolist() == e_d.view(np.uint32).tolist(), i
        st = g.stats()
        assert st.search_errors[vsa.VK_ERR_CANCELLED] >= 8 and st.total_n_eval > 0 and st.total_n_hops > 0
    finally:
        g.set_coalescing(0, 0)
    # tombstones show as reclaimable bytes (hnswalg.h:1199)
    for lab in range(0, 500):
        assert g.remove(lab) == 0
    st = g.stats()
    assert st.deleted == 500 and st.tombstoned_bytes == 500 * (64 * 4 + (2 * M + 1) * 4)


def test_options_are_named_ranged_and_forwarded_to_shards(vsa):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((4000, 32)).astype(np.float32)
    ix = vsa.Index("FLAT", 32, "L2", initial_cap=4000, shard_devices=[0, 0], options={"filter-cap": 4096})
    ix.add_batch(x)
    assert ix.get_option("filter-cap") == 4096 and ix.get_option("batches-in-flight") == 2
    assert ix.get_option("max-query-queue-depth") == 100000          # the reference's default (valkey_search_options.cc:231-234)
    ix.set_option("shard-ef-pct", 50)
    assert ix.get_option("shard-ef-pct") == 50
    for bad in (("no-such-option", 1), ("batches-in-flight", 0), ("batches-in-flight", 99), ("shard-ef-pct", 5000)):
        with pytest.raises(vsa.VkError) as e:
            ix.set_option(*bad)
        assert e.value.code == vsa.VK_ERR_INVALID
    with pytest.raises(vsa.VkError):
        ix.get_option("nope")
    # kernel timing is opt-in: no HIP event pairs on the search path unless asked for
    assert ix.get_option("kernel-timing") == 0
    d, l, n = ix.search_batch(x[:8], 5)
    assert (l[:, 0] == np.arange(8)).all()
    st = ix.stats()
    assert st.searches == 8 and st.search_calls == 1 and sum(st.search_errors) == 0
    # params are range-checked too
    with pytest.raises(vsa.VkError):
        vsa.Index("HNSW", 32, "L2", initial_cap=100, shard_devices=[0, 0], shard_ef_pct=100000)
