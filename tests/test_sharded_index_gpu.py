"""The multi-GPU index inside the library (vk_index_params.n_shards / shard_devices): one vk_index over several
sub-indexes, queries broadcast, per-shard top-k gathered and merged by (distance,label) on the serving device -- the
role of the cluster fan-out + SearchPartitionResultsTracker::AddResult (src/query/fanout.cc:162-175) in one process.
A one-GPU box runs it with LOGICAL shards (the same device named several times): every code path but the peer copies.
FLAT: the S-shard answer must be bit-identical to the single-index answer and to the oracle; HNSW (one graph per
shard): identical to the (distance,label) merge of the answers of the same graphs held by separate indexes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _same(a, b):
    (ad, al, an), (bd, bl, bn) = a, b
    assert an.tolist() == bn.tolist()
    for i in range(len(an)):
        assert al[i, :an[i]].tolist() == bl[i, :bn[i]].tolist(), i
        assert ad[i, :an[i]].view(np.uint32).tolist() == bd[i, :bn[i]].view(np.uint32).tolist(), i


@pytest.mark.parametrize("metric,dtype,shards", [("COSINE", "f32", 3), ("L2", "f32", 2), ("IP", "bf16", 4), ("COSINE", "f32", 8)])
def test_flat_sharded_equals_single_and_oracle(vsa, oracle, metric, dtype, shards):
    rng = np.random.default_rng(shards * 7 + len(metric))
    n, dim = 40_000, 96
    x = rng.standard_normal((n, dim)).astype(np.float32)
    if metric == "COSINE":
        x = x / np.linalg.norm(x, axis=1, keepdims=True)
    labels = rng.permutation(n * 3)[:n].astype(np.uint64)
    one = vsa.Index("FLAT", dim, metric, initial_cap=n, dtype=dtype)
    one.add_batch(x, labels)
    sh = vsa.Index("FLAT", dim, metric, initial_cap=n, dtype=dtype, shard_devices=[0] * shards)
    assert sh.shard_count() == shards and one.shard_count() == 0
    sh.add_batch(x, labels)
    assert sh.stats().count == n
    Q = rng.standard_normal((70, dim)).astype(np.float32)
    bits = oracle.allow_bitmap(labels[rng.random(n) < 0.2], int(labels.max()) + 1)
    nb = int(labels.max()) + 1
    for nq in (1, 7, 70):
        for k in (1, 10, 100):
            _same(sh.search_batch(Q[:nq], k), one.search_batch(Q[:nq], k))
        _same(sh.search_batch(Q[:nq], 10, allow=bits, allow_nbits=nb), one.search_batch(Q[:nq], 10, allow=bits, allow_nbits=nb))
    _same(sh.search_batch(Q[:3], 1500), one.search_batch(Q[:3], 1500))         # beyond the device merge: host merge
    if dtype == "f32":
        o = oracle.Flat(dim, metric, max_elements=n)
        o.add_many(x, labels)
        D, L, N = sh.search_batch(Q[:8], 10)
        for i in range(8):
            od, ol = o.search(Q[i], 10)
            assert L[i].tolist() == ol.tolist() and D[i].view(np.uint32).tolist() == od.view(np.uint32).tolist()


def test_flat_sharded_mutations_and_point_reads(vsa, oracle):
    rng = np.random.default_rng(4)
    n, dim = 6000, 40
    x = rng.standard_normal((n, dim)).astype(np.float32)
    sh = vsa.Index("FLAT", dim, "L2", initial_cap=n + 10, shard_devices=[0, 0, 0])
    o = oracle.Flat(dim, "L2", max_elements=n + 10)
    for i in range(300):                                    # one by one: the emptiest shard takes the next new label
        assert sh.add(i, x[i]) == 0
    sh.add_batch(x[300:], np.arange(300, n, dtype=np.uint64))
    o.add_many(x)
    for lab in rng.choice(n, 700, replace=False):
        assert sh.remove(int(lab)) == 0
        o.remove(int(lab))
    for lab in range(0, 200, 3):                            # updates of live labels, re-adds of removed ones
        v = rng.standard_normal(dim).astype(np.float32)
        assert sh.add(lab, v) == 0
        o.add(v, lab)
    assert sh.stats().count == o.count
    Q = rng.standard_normal((9, dim)).astype(np.float32)
    D, L, N = sh.search_batch(Q, 25)
    for i in range(9):
        od, ol = o.search(Q[i], 25)
        assert L[i].tolist() == ol.tolist() and D[i].view(np.uint32).tolist() == od.view(np.uint32).tolist()
    keys = rng.choice(n + 50, 400, replace=False).astype(np.uint64)   # some unknown, some removed
    gd, gl = sh.search_labels(Q[0], 10, keys)
    rows, labs = [], []
    for kk in keys:
        r = sh.get_row(int(kk))
        assert (r is not None) == sh.contains(int(kk))
        if r is not None:
            rows.append(r)
            labs.append(kk)
            assert sh.distance(int(kk), Q[0]).view(np.uint32) == oracle.distance("L2", Q[0], r).view(np.uint32)
    ed, el = oracle.prefilter_topk("L2", Q[0], np.stack(rows), np.array(labs, np.uint64), 10)
    assert gl.tolist() == el.tolist() and gd.view(np.uint32).tolist() == ed.view(np.uint32).tolist()
    # the limit the caller sees is the index's, not a shard's
    small = vsa.Index("FLAT", dim, "L2", initial_cap=10, shard_devices=[0, 0])
    for i in range(10):
        assert small.add(i, x[i]) == 0
    assert small.add(10, x[10]) == vsa.VK_ERR_CAPACITY
    small.resize(12)
    assert small.add(10, x[10]) == 0 and small.stats().capacity == 12


def test_flat_sharded_persistence_any_shard_count(vsa, oracle):
    rng = np.random.default_rng(5)
    n, dim = 5000, 33
    x = rng.standard_normal((n, dim)).astype(np.float32)
    labels = (np.arange(n, dtype=np.uint64) * 3 + 1)
    sh = vsa.Index("FLAT", dim, "IP", initial_cap=n, shard_devices=[0, 0, 0])
    sh.add_batch(x, labels)
    chunks = sh.save()                                       # ONE stream in the reference's layout
    assert len(chunks) == 1 + n and all(len(c) == dim * 4 + 8 for c in chunks[1:])
    Q = rng.standard_normal((5, dim)).astype(np.float32)
    want = sh.search_batch(Q, 20)
    _same(vsa.Index.load(chunks, "FLAT", dim, "IP").search_batch(Q, 20), want)                        # plain index
    _same(vsa.Index.load(chunks, "FLAT", dim, "IP", shard_devices=[0] * 5).search_batch(Q, 20), want)  # other shard count
    plain = vsa.Index("FLAT", dim, "IP", initial_cap=n)
    plain.add_batch(x, labels)
    _same(vsa.Index.load(plain.save(), "FLAT", dim, "IP", shard_devices=[0, 0]).search_batch(Q, 20), want)


def _merge(parts, k):
    """(distance,label) merge of per-index answers for one query"""
    allp = sorted((float(d), int(l)) for D, L in parts for d, l in zip(D, L))
    return allp[:k]


def test_hnsw_sharded_is_the_merge_of_its_graphs(vsa, oracle):
    rng = np.random.default_rng(6)
    n, dim, S, M = 3000, 48, 3, 8
    x = rng.standard_normal((n, dim)).astype(np.float32)
    kw = dict(m=M, ef_construction=60, ef_runtime=40, build_threads=1)
    sh = vsa.Index("HNSW", dim, "L2", initial_cap=n, shard_devices=[0] * S, **kw)
    sh.add_batch(x)                                          # rows [s*n/S, (s+1)*n/S) to shard s
    singles = []
    for s in range(S):
        lo, hi = s * n // S, (s + 1) * n // S
        g = vsa.Index("HNSW", dim, "L2", initial_cap=max(1024, -(-n // S)), **kw)
        g.add_batch(x[lo:hi], np.arange(lo, hi, dtype=np.uint64))
        singles.append(g)
    dead = rng.choice(n, 200, replace=False)
    for lab in dead:
        assert sh.remove(int(lab)) == 0
        s = next(s for s in range(S) if s * n // S <= lab < (s + 1) * n // S)
        assert singles[s].remove(int(lab)) == 0
    assert sum(g.stats().deleted for g in singles) == 200 == sh.stats().deleted
    bits = oracle.allow_bitmap(np.flatnonzero(rng.random(n) < 0.3), n)
    Q = rng.standard_normal((20, dim)).astype(np.float32)
    for ef, allow in ((0, None), (100, None), (100, bits)):
        D, L, N = sh.search_batch(Q, 10, ef=ef, allow=allow, allow_nbits=n if allow is not None else None)
        for i in range(len(Q)):
            parts = [g.search(Q[i], 10, ef=ef, allow=allow, allow_nbits=n if allow is not None else None) for g in singles]
            want = _merge(parts, 10)
            assert [int(v) for v in L[i, :N[i]]] == [l for _, l in want]
            assert [float(v) for v in D[i, :N[i]]] == [d for d, _ in want]
    assert not sh.contains(int(dead[0])) and sh.distance(int(dead[0]), Q[0]) is None
    # container round trip: the same graphs come back
    chunks = sh.save()
    back = vsa.Index.load(chunks, "HNSW", dim, "L2", initial_cap=n, shard_devices=[0] * S, **kw)
    _same(back.search_batch(Q, 10, ef=100), sh.search_batch(Q, 10, ef=100))
    assert back.contains(7) == sh.contains(7) and back.stats().count == sh.stats().count
    # a plain single-graph stream loads into a sharded index by re-inserting its rows (tombstoned ones are gone)
    re = vsa.Index.load(singles[0].save(), "HNSW", dim, "L2", initial_cap=n, shard_devices=[0, 0], **kw)
    assert re.stats().count == singles[0].stats().count - singles[0].stats().deleted
    d, l = re.search(x[5], 1, ef=50)
    assert (l.tolist() == [5]) == singles[0].contains(5) or not singles[0].contains(5)


def test_sharded_device_buffer_entry_point(vsa, oracle):
    import torch
    rng = np.random.default_rng(8)
    n, dim, k, nq = 30_000, 64, 10, 33
    x = rng.standard_normal((n, dim)).astype(np.float32)
    sh = vsa.Index("FLAT", dim, "L2", initial_cap=n, shard_devices=[0, 0, 0, 0])
    # bulk load shard by shard, rows written on the device
    dev = torch.device("cuda", 0)
    for s in range(4):
        lo, hi = s * n // 4, (s + 1) * n // 4
        ptr, stride = sh.shard_device_rows(s, hi - lo)
        assert stride == dim * 4
        import ctypes as C
        t = torch.from_numpy(x[lo:hi]).to(dev)
        torch.cuda.synchronize()
        from bench import device_view
        device_view(ptr, (hi - lo, dim), dev).copy_(t)
        torch.cuda.synchronize()
        sh.shard_commit_device_rows(s, hi - lo, np.arange(lo, hi, dtype=np.uint64))
    one = vsa.Index("FLAT", dim, "L2", initial_cap=n)
    one.add_batch(x)
    Q = rng.standard_normal((nq, dim)).astype(np.float32)
    dq = torch.from_numpy(Q).to(dev)
    od = torch.empty(nq, k, device=dev, dtype=torch.float32)
    ol = torch.empty(nq, k, device=dev, dtype=torch.int64)
    on = torch.empty(nq, device=dev, dtype=torch.int32)
    st = torch.cuda.Stream(device=dev)
    for _ in range(3):                                       # back to back without a host sync in between
        sh.search_batch_device(dq.data_ptr(), nq, k, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
    D, L, N = one.search_batch(Q, k)
    assert (ol.cpu().numpy().view(np.uint64) == L).all() and (od.cpu().numpy().view(np.uint32) == D.view(np.uint32)).all()
    assert (on.cpu().numpy() == k).all()


@pytest.mark.parametrize("algo,shards", [("FLAT", 3), ("FLAT", 8), ("HNSW", 2)])
def test_rccl_all_gather_of_the_per_shard_lists_equals_the_peer_copies(vsa, oracle, algo, shards):
    """Option shard-gather = 1: the per-shard top-k lists travel by an in-library RCCL all-gather (ncclCommInitAll in this
    one process, one communicator per DISTINCT device, two ncclAllGather per fan-out) instead of one-shot peer copies.  On a
    one-GPU box the communicator has one rank and every device group holds all the shards -- the code path (library loaded
    at run time, communicators, send slots per shard, group call, stream ordering, merge over [ranks][lists]) is the one
    eight GPUs run; what a single device cannot show is the xGMI traffic.  Answers must not change, and a request for the
    collective is never served by the copies (vk_index_stats.rccl_gathers counts)."""
    rng = np.random.default_rng(77 + shards)
    n, dim = (30_000, 64) if algo == "FLAT" else (6_000, 32)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x = x / np.linalg.norm(x, axis=1, keepdims=True)
    kw = dict(m=16, ef_construction=100, ef_runtime=64, build_threads=1) if algo == "HNSW" else {}
    sh = vsa.Index(algo, dim, "COSINE", initial_cap=n, shard_devices=[0] * shards, **kw)
    sh.add_batch(x)
    Q = rng.standard_normal((70, dim)).astype(np.float32)
    Q = Q / np.linalg.norm(Q, axis=1, keepdims=True)
    bits = oracle.allow_bitmap(np.arange(0, n, 3, dtype=np.uint64), n)
    ref = {(nq, k): sh.search_batch(Q[:nq], k) for nq in (1, 7, 70) for k in (1, 10, 100)}
    ref_f = sh.search_batch(Q, 10, allow=bits, allow_nbits=n)
    assert sh.stats().rccl_gathers == 0
    sh.set_option("shard-gather", 1)
    assert sh.get_option("shard-gather") == 1
    for (nq, k), want in ref.items():
        _same(sh.search_batch(Q[:nq], k), want)
    _same(sh.search_batch(Q, 10, allow=bits, allow_nbits=n), ref_f)
    st = sh.stats()
    assert st.rccl_gathers == len(ref) + 1 and st.fanout_calls >= 2 * (len(ref) + 1)
    # the device-buffer entry point (what bench.py times) and a single query
    d1, l1 = sh.search(Q[0], 10)
    assert l1.tolist() == ref[(1, 10)][1][0].tolist()
    sh.set_option("shard-gather", 0)
    _same(sh.search_batch(Q[:7], 10), ref[(7, 10)])
    assert sh.stats().rccl_gathers == st.rccl_gathers + 1
    if algo == "FLAT":
        o = oracle.Flat(dim, "COSINE", max_elements=n)
        o.add_many(x)
        D, L, N = ref[(7, 10)]
        for i in range(7):
            od, ol = o.search(Q[i], 10)
            assert L[i].tolist() == ol.tolist() and D[i].view(np.uint32).tolist() == od.view(np.uint32).tolist()


def test_a_second_device_without_peer_access_is_refused_loudly(vsa):
    """Two-device readiness: an index over devices that cannot reach each other directly would stage every broadcast and
    gather through host memory.  vk_index_create refuses it (VK_ERR_NO_DEVICE with the reason) unless shard-allow-staged
    is set; on this box there is one device, so what can be checked is that naming a device that does not exist fails
    loudly too, and that the single-device sharded index does not need the switch."""
    import pytest as _pt
    nd = vsa.lib().vk_device_count()
    with _pt.raises(vsa.VkError) as e:
        vsa.Index("FLAT", 16, "L2", initial_cap=1024, shard_devices=[0, nd])
    assert e.value.code in (vsa.VK_ERR_NO_DEVICE, vsa.VK_ERR_INVALID, vsa.VK_ERR_INTERNAL) and e.value.msg
    ok = vsa.Index("FLAT", 16, "L2", initial_cap=1024, shard_devices=[0, 0])
    assert ok.get_option("shard-allow-staged") == 0 and ok.shard_count() == 2
