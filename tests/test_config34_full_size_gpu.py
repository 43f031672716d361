"""BASELINE.json configs[3] and configs[4] AT THEIR STATED SIZE on one MI355X (288 GB), as eight LOGICAL shards of one
vk_index (vk_index_params.n_shards = 8, every shard on device 0 -- every code path of the multi-GPU index but the peer
copies, which need a second device):

  configs[3]  FLAT 80M x 768, bf16 rows (122.9 GB), IP, k=10, batch=256, 8 shards of 10M rows
  configs[4]  HNSW M=16 efC=200 over 10M x 768 f32 cosine, 8 graphs of 1.25M rows, 10 % TAG filter, efSearch=256, k=10
              (+ the pre-filter branch below 0.1 % selectivity, planner.cc:21-45)

What is checked is what the oracle can reach at this size: the merged answer equals the per-shard answers merged by the
oracle's (distance, label) merge (fanout.cc:162-175 with the total order of bruteforce.h's heap); restricted to a sample of
rows it equals the oracle's scan of the sample; the HNSW answer equals the oracle's search of every shard's OWN saved graph,
merged; answers are sorted, complete and idempotent."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

D, K, B, S = 768, 10, 256, 8


def _queries(dev, nq, seed):
    import torch
    from bench import make_queries
    A = torch.randn(D, 32, generator=torch.Generator(device=dev).manual_seed(1234), device=dev)
    return make_queries(A, nq, D, dev, seed)


def test_config3_flat_80m_bf16_ip_eight_shards(oracle):
    import torch
    import _pkg
    from bench import _fill_shard
    vsa = _pkg.vsa
    dev = torch.device("cuda", 0)
    n = 10_000_000
    N = n * S
    free, _total = torch.cuda.mem_get_info(dev)
    assert free > N * D * 2 + (24 << 30), f"needs {N * D * 2 >> 30} GiB of HBM for the rows, {free >> 30} GiB free"
    ix = vsa.Index("FLAT", D, "IP", initial_cap=N, dtype="bf16", shard_devices=[0] * S)
    assert ix.shard_count() == S
    tabs = [_fill_shard(ix, s, s * n, n, D, dev, True) for s in range(S)]
    torch.cuda.empty_cache()
    st = ix.stats()
    assert st.count == N and st.device_bytes >= N * D * 2
    Q = _queries(dev, B, 4242)
    hq = Q.cpu().numpy()

    # the timed path of bench.py: device buffers in and out, one call for the batch
    od = torch.empty(B, K, device=dev, dtype=torch.float32)
    ol = torch.empty(B, K, device=dev, dtype=torch.int64)
    on = torch.empty(B, device=dev, dtype=torch.int32)
    ix.search_batch_device(Q.data_ptr(), B, K, od.data_ptr(), ol.data_ptr(), on.data_ptr())
    torch.cuda.synchronize()
    gd, gl = od.cpu().numpy(), ol.cpu().numpy().view(np.uint64)
    assert (on.cpu().numpy() == K).all()
    # ... equals the host entry point, which is idempotent
    Dh, Lh, Nh = ix.search_batch(hq, K)
    assert (Nh == K).all() and (Lh == gl).all() and (Dh.view(np.uint32) == gd.view(np.uint32)).all()
    D2, L2, _ = ix.search_batch(hq, K)
    assert (L2 == Lh).all() and (D2.view(np.uint32) == Dh.view(np.uint32)).all()
    # every shard took the bf16 matrix-core filter for every one of the three batches so far
    assert all(ix.shard_stats(s).filter_batches >= 3 for s in range(S)) and ix.stats().filter_batches >= 3
    # ascending by (distance, label), labels distinct and inside the index
    for i in range(B):
        pairs = list(zip(Dh[i].tolist(), Lh[i].tolist()))
        assert pairs == sorted(pairs) and len(set(Lh[i].tolist())) == K and int(Lh[i].max()) < N

    # per-shard answers (the index restricted to one shard's labels by a filter) merged by the ORACLE == the merged answer
    per_d = np.full((S, B, K), np.inf, np.float32)
    per_l = np.full((S, B, K), np.iinfo(np.uint64).max, np.uint64)
    per_n = np.zeros((S, B), np.uint32)
    words = (N + 63) // 64
    for s in range(S):
        bits = np.zeros(words, np.uint64)
        bits[s * n // 64:(s + 1) * n // 64] = ~np.uint64(0)           # (n is a multiple of 64)
        ds, ls, ns = ix.search_batch(hq, K, allow=bits, allow_nbits=N)
        assert (ns == K).all() and (ls >= s * n).all() and (ls < (s + 1) * n).all()
        per_d[s], per_l[s], per_n[s] = ds, ls, ns
    for i in range(B):
        md, ml = oracle.merge_topk(per_d[:, i, :], per_l[:, i, :], per_n[:, i], K)
        assert ml.tolist() == Lh[i].tolist(), i
        assert md.view(np.uint32).tolist() == Dh[i].view(np.uint32).tolist(), i
    # the single-query scan kernel on every shard agrees with the batched path
    for i in range(0, B, 32):
        d1, l1 = ix.search(hq[i], K)
        assert l1.tolist() == Lh[i].tolist() and d1.view(np.uint32).tolist() == Dh[i].view(np.uint32).tolist()

    # restricted to a sample of rows spread over all eight shards: the oracle's IP scan over the (rounded) sample rows
    per = 8_000
    rng = np.random.default_rng(17)
    sample, rows = [], []
    for s in range(S):
        loc = np.sort(rng.choice(n, per, replace=False))
        sample.append(loc.astype(np.uint64) + np.uint64(s * n))
        rows.append(tabs[s][torch.from_numpy(loc).to(dev), :D].float().cpu().numpy())
    sample = np.concatenate(sample)
    rows = np.ascontiguousarray(np.concatenate(rows))
    o = oracle.Flat(D, "IP", max_elements=len(sample))
    o.add_many(rows, sample)
    bits = oracle.allow_bitmap(sample, N)
    Df, Lf, Nf = ix.search_batch(hq[:48], K, allow=bits, allow_nbits=N)
    for i in range(48):
        e_d, e_l = o.search(hq[i], K)
        assert Lf[i, :Nf[i]].tolist() == e_l.tolist(), i
        assert Df[i, :Nf[i]].view(np.uint32).tolist() == e_d.view(np.uint32).tolist(), i
    # self-retrieval across shard boundaries: a stored row queried with itself comes back first (IP over unit rows)
    for r in (0, n - 1, n, 3 * n + 12345, N - 1):
        row = tabs[r // n][r % n, :D].float().cpu().numpy()
        d1, l1 = ix.search(row, 1)
        assert int(l1[0]) == r
    del tabs, ix
    torch.cuda.empty_cache()


def test_config4_hnsw_10m_tag_filter_eight_shards(oracle):
    import torch
    import _pkg
    from bench import gen_rows
    O = oracle
    vsa = _pkg.vsa
    dev = torch.device("cuda", 0)
    N, ef = 10_000_000, 256
    host = np.empty((N, D), np.float32)
    for lo, x in gen_rows(0, N, D, dev):
        host[lo:lo + x.shape[0]] = x.cpu().numpy()
    torch.cuda.empty_cache()
    h = vsa.Index("HNSW", D, "COSINE", initial_cap=N, m=16, ef_construction=200, ef_runtime=ef, shard_devices=[0] * S)
    h.add_batch(host)
    h.flush()
    st = h.stats()
    assert st.count == N and h.shard_count() == S
    nq = 256
    hq = _queries(dev, nq, 7070).cpu().numpy()
    tag = np.arange(3, N, 10, dtype=np.uint64)                          # "@tag:{t3}": 10 % of the rows -> inline filter
    bits = O.allow_bitmap(tag, N)
    Dg, Lg, Ng = h.search_batch(hq, K, ef=ef, allow=bits, allow_nbits=N)
    st = h.stats()
    assert st.last_frontier_dropped == 0
    assert (Ng == K).all() and (Lg % 10 == 3).all()
    for i in range(nq):
        pairs = list(zip(Dg[i].tolist(), Lg[i].tolist()))
        assert pairs == sorted(pairs) and len(set(Lg[i].tolist())) == K
    D2, L2, _ = h.search_batch(hq, K, ef=ef, allow=bits, allow_nbits=N)
    assert (L2 == Lg).all() and (D2.view(np.uint32) == Dg.view(np.uint32)).all()
    # the device-buffer entry point (what bench.py times) gives the same answer
    Qd = torch.from_numpy(hq).to(dev)
    d_bits = torch.from_numpy(bits.view(np.int64)).to(dev)
    od = torch.empty(nq, K, device=dev, dtype=torch.float32)
    ol = torch.empty(nq, K, device=dev, dtype=torch.int64)
    on = torch.empty(nq, device=dev, dtype=torch.int32)
    h.search_batch_device(Qd.data_ptr(), nq, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), ef=ef, d_allow=d_bits.data_ptr(), allow_nbits=N)
    torch.cuda.synchronize()
    assert (ol.cpu().numpy().view(np.uint64) == Lg).all() and (od.cpu().numpy().view(np.uint32) == Dg.view(np.uint32)).all()

    # the oracle on every shard's OWN saved graph, merged by (distance, label)
    graphs = O.HNSW.shards_from_product_index(h.save_raw, D, "COSINE", 16, ef_construction=200)
    assert len(graphs) == S and sum(g.count for g in graphs) == N
    sample = list(range(0, nq, nq // 48))[:48]
    for i in sample:
        pd = np.full((S, K), np.inf, np.float32)
        pl = np.full((S, K), np.iinfo(np.uint64).max, np.uint64)
        pn = np.zeros(S, np.uint32)
        for s, g in enumerate(graphs):
            e_d, e_l = g.search(hq[i], K, ef=ef, allow=bits, allow_nbits=N)
            pd[s, :len(e_d)], pl[s, :len(e_l)], pn[s] = e_d, e_l, len(e_l)
        md, ml = O.merge_topk(pd, pl, pn, K)
        assert Lg[i, :len(ml)].tolist() == ml.tolist(), i
        assert Dg[i, :len(md)].view(np.uint32).tolist() == md.view(np.uint32).tolist(), i
    del graphs
    # recall@10 of the filtered search against the exact filtered answer (oracle scan of the tagged rows), a few queries
    exact = O.Flat(D, "COSINE", max_elements=len(tag))
    exact.add_many(host[3::10], tag, borrowed=True)
    hit = 0
    for i in sample[:16]:
        _, e_l = exact.search(hq[i], K)
        hit += len(set(e_l.tolist()) & set(Lg[i].tolist()))
    assert hit / (16 * K) >= 0.95, hit / (16 * K)
    del exact
    # below 0.1 % selectivity the planner pre-filters (planner.cc:21-45): exact kNN over the key list, on the sharded index
    few = np.arange(7, N, 2003, dtype=np.uint64)                        # 0.05 % of the rows
    for i in sample[:8]:
        pd_, pl_ = h.search_labels(hq[i], K, few)
        e_d, e_l = O.prefilter_topk("COSINE", hq[i], host[few.astype(np.int64)], few, K)
        assert pl_.tolist() == e_l.tolist() and pd_.view(np.uint32).tolist() == e_d.view(np.uint32).tolist()
    del h
    torch.cuda.empty_cache()
