"""Filtered / tombstoned HNSW searches never truncate the frontier.

hnswlib's candidate_set is an unbounded heap (hnswalg.h:367-370, :502-524): with a selective filter the result list
fills slowly and the frontier grows to about ef / selectivity entries.  The device keeps it in HBM: a first launch
with at most 64k entries per wave, and -- for the queries that outgrow that -- a second launch whose frontier is
sized by the graph (a node enters the frontier at most once).  Whatever the path, ids, distance bits and the
layer-0 work counters must be the oracle's on the SAME graph, and vk_index_stats must say what happened."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


class _Env:
    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _oracle_of(oracle, g, dim, metric, M, efc):
    return oracle.HNSW.from_product_index(g.save_raw, dim, metric, M, ef_construction=efc)


def _check_batch(g, o, Q, k, ef, bits, nbits):
    D, L, N = g.search_batch(Q, k, ef=ef, allow=bits, allow_nbits=nbits)
    ne = nh = 0
    for i in range(len(Q)):
        od, ol, e, h = o.search(Q[i], k, ef=ef, allow=bits, allow_nbits=nbits, stats=True)
        assert L[i, :N[i]].tolist() == ol.tolist(), i
        assert D[i, :N[i]].view(np.uint32).tolist() == od.view(np.uint32).tolist(), i
        ne += e
        nh += h
    st = g.stats()
    assert (st.last_n_eval, st.last_n_hops) == (ne, nh)
    assert st.last_frontier_dropped == 0
    return st


@pytest.mark.parametrize("ef,k", [(64, 10), (256, 50), (700, 20)])     # register lists and the LDS list (ef > 512)
@pytest.mark.parametrize("redo_bytes", [None, 1])                       # default budget / a single wave in the second launch
def test_forced_overflow_is_answered_by_the_graph_sized_frontier(vsa, oracle, ef, k, redo_bytes):
    rng = np.random.default_rng(5150)
    n, dim, M = 6000, 64, 8
    x = rng.standard_normal((n, dim)).astype(np.float32)
    env = {"VK_HNSW_GPOOL_CAP": 128}
    if redo_bytes:
        env["VK_HNSW_REDO_BYTES"] = redo_bytes
    with _Env(**env):
        g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=40, build_threads=4)
    g.add_batch(x)
    for lab in rng.choice(n, n // 10, replace=False):
        assert g.remove(int(lab)) == 0
    g.flush()
    o = _oracle_of(oracle, g, dim, "L2", M, 40)
    assert o.count == n and o.deleted_count == n // 10
    bits = oracle.allow_bitmap(np.flatnonzero(rng.random(n) < 0.05), n)
    Q = rng.standard_normal((40, dim)).astype(np.float32)
    st = _check_batch(g, o, Q, k, ef, bits, n)
    assert st.last_frontier_redo > 0          # 128 entries cannot hold ef / 0.05 of them
    # tombstones alone (no filter) take the same path; and an unfiltered index without tombstones never does
    st = _check_batch(g, o, Q[:8], k, ef, None, None)
    h = vsa.Index("HNSW", dim, "L2", initial_cap=1000, m=M, ef_construction=40)
    h.add_batch(x[:1000])
    h.search_batch(Q, k, ef=ef)
    assert h.stats().last_frontier_redo == 0 and h.stats().last_frontier_dropped == 0


def test_large_graph_one_percent_filter_and_tombstones_equal_the_oracle(vsa, oracle):
    """>= 200k nodes: the regime where the first launch's cap (64k entries) is below the node count, so both launches
    are live.  Device-assisted build, 5 % tombstones, 1 % and 0.2 % filters, ef = 256."""
    rng = np.random.default_rng(77)
    n, dim, M = 200_000, 16, 16
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=100)
    g.add_batch(x)
    for lab in rng.choice(n, n // 20, replace=False):
        assert g.remove(int(lab)) == 0
    g.flush()
    o = _oracle_of(oracle, g, dim, "L2", M, 100)
    assert o.count == n
    Q = rng.standard_normal((48, dim)).astype(np.float32)
    redo = 0
    for sel in (0.01, 0.002):
        bits = oracle.allow_bitmap(np.flatnonzero(rng.random(n) < sel), n)
        st = _check_batch(g, o, Q, 10, 256, bits, n)
        redo += st.last_frontier_redo
    # (whether a frontier passes 64k entries at these selectivities is the data's business; the next block makes sure
    # the second launch runs at this size)
    with _Env(VK_HNSW_GPOOL_CAP=4096):
        chunks = g.save()
        g2 = vsa.Index.load(chunks, "HNSW", dim, "L2", m=M, ef_construction=100, initial_cap=n)
    bits = oracle.allow_bitmap(np.flatnonzero(rng.random(n) < 0.002), n)
    st = _check_batch(g2, o, Q, 10, 256, bits, n)
    assert st.last_frontier_redo > 0


def test_device_buffer_path_runs_both_launches(vsa, oracle):
    """vk_index_search_batch_device (the shard leg of the multi-GPU path) has no host in the loop: the second launch is
    enqueued unconditionally behind the first and picks up whatever it abandoned."""
    import torch
    rng = np.random.default_rng(99)
    n, dim, M = 5000, 32, 8
    x = rng.standard_normal((n, dim)).astype(np.float32)
    with _Env(VK_HNSW_GPOOL_CAP=128):
        g = vsa.Index("HNSW", dim, "IP", initial_cap=n, m=M, ef_construction=40, build_threads=4)
    g.add_batch(x)
    g.flush()
    o = _oracle_of(oracle, g, dim, "IP", M, 40)
    allowed = np.flatnonzero(rng.random(n) < 0.03)
    bits = oracle.allow_bitmap(allowed, n)
    nq, k = 300, 10
    Q = rng.standard_normal((nq, dim)).astype(np.float32)
    dev = torch.device("cuda", 0)
    dq = torch.from_numpy(Q).to(dev)
    db = torch.from_numpy(bits.view(np.int64)).to(dev)
    od = torch.empty(nq, k, device=dev, dtype=torch.float32)
    ol = torch.empty(nq, k, device=dev, dtype=torch.int64)
    on = torch.empty(nq, device=dev, dtype=torch.int32)
    g.search_batch_device(dq.data_ptr(), nq, k, od.data_ptr(), ol.data_ptr(), on.data_ptr(), ef=128,
                          d_allow=db.data_ptr(), allow_nbits=n)
    torch.cuda.synchronize()
    D, L, N = od.cpu().numpy(), ol.cpu().numpy().view(np.uint64), on.cpu().numpy()
    for i in range(nq):
        e_d, e_l = o.search(Q[i], k, ef=128, allow=bits, allow_nbits=n)
        assert L[i, :N[i]].tolist() == e_l.tolist(), i
        assert D[i, :N[i]].view(np.uint32).tolist() == e_d.view(np.uint32).tolist(), i
        assert np.all(np.isinf(D[i, N[i]:]))
