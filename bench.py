#!/usr/bin/env python3
"""bench.py -- kNN queries/sec on the FLAT path (BASELINE.json configs[1]):
FLAT index, 10M x 768 fp32 COSINE, k=10, batch=256 queries per step.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  The index (fixed total size) is sharded by contiguous row ranges over
the N ranks ("strong" scaling: total work fixed); a step = every rank scans its shard for
the same 256 queries (rows resident in HBM, queries resident in HBM), one RCCL all-gather
of the per-shard top-k, and the (distance,label) merge -- so the N-GPU answer is
bit-identical to the 1-GPU answer.  Rank 0 prints ONE JSON line.

PyTorch is plumbing here: synthetic data generation on the device, the RCCL process group,
and HIP events.  The search itself is libvkindex.so called through its C ABI.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
import _pkg  # noqa: E402

vsa = _pkg.vsa

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TF = 157.3     # MI355X_MICROARCH.md: FP32 matrix peak (dense)


class _DevMem:
    """Expose a raw device allocation (the index's HBM row table) to torch without a copy."""

    def __init__(self, ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


def device_view(ptr, shape, device):
    return torch.as_tensor(_DevMem(ptr, shape), device=device)


def device_view_typed(ptr, shape, device, typestr):
    return torch.as_tensor(_DevMem(ptr, shape, typestr), device=device)


def effective_cpus() -> int:
    """CPUs this process may really use: affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, round(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, round(q / p)))
        except (OSError, ValueError):
            pass
    return n


def gen_rows(n_begin, n_rows, dim, device, chunk=65536, latent=32, noise=0.05, seed=1234):
    """Rank-32 latent model x = A z + 0.05 eps, L2-normalised;
    chunk c is seeded by (seed, c) so any shard of any world size sees the same rows."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    A = torch.randn(dim, latent, generator=g, device=device, dtype=torch.float32)
    c0, c1 = n_begin // chunk, (n_begin + n_rows + chunk - 1) // chunk
    for c in range(c0, c1):
        g.manual_seed(seed * 1000003 + c + 1)
        z = torch.randn(chunk, latent, generator=g, device=device, dtype=torch.float32)
        e = torch.randn(chunk, dim, generator=g, device=device, dtype=torch.float32)
        x = z @ A.T + noise * e
        x = torch.nn.functional.normalize(x, dim=1)
        lo = max(n_begin, c * chunk)
        hi = min(n_begin + n_rows, (c + 1) * chunk)
        yield lo, x[lo - c * chunk: hi - c * chunk].contiguous()


GATHER_CEILING_GBS = 6300.0   # profiles/r01_gather_ceiling_10Mx768.log: random 3 KiB rows of a 30 GB table, quad per row


def make_queries(A, nq, D, device, seed):
    """queries from the same rank-32 latent model as the rows, disjoint seed"""
    qg = torch.Generator(device=device)
    qg.manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(nq, 32, generator=qg, device=device) @ A.T +
                                         0.05 * torch.randn(nq, A.shape[0], generator=qg, device=device), dim=1).contiguous()


def timed(fn, reps):
    """mean milliseconds of fn() over reps, HIP events on the current (work) stream"""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def recall_of(got, gt, K):
    return float(np.mean([len(set(a.tolist()) & set(b.tolist())) for a, b in zip(got, gt)])) / K


def exact_ground_truth(flat_ix, hq, K, n_rows, total_rows):
    """exact top-K of the first n_rows rows from the FLAT index (labels == row numbers), 256 queries per call"""
    bits = nb = None
    if n_rows < total_rows:
        from oracle import oracle as O
        bits, nb = O.allow_bitmap(np.arange(n_rows, dtype=np.uint64), n_rows), n_rows
    out = np.empty((hq.shape[0], K), np.uint64)
    for i in range(0, hq.shape[0], 256):
        _, L, _ = flat_ix.search_batch(hq[i:i + 256], K, allow=bits, allow_nbits=nb)
        out[i:i + 256] = L
    return out


def hnsw_leg(args, flat_ix, host_rows, A, device, stream_ptr, total_rows):
    """BASELINE.json configs[2]: HNSW M=16 efC=200 over the first --hnsw-rows rows (default: all 10M), efSearch=128,
    k=10: device-assisted bulk build (K9), device search, recall@10 against the exact FLAT answer, the CPU oracle
    searching the SAME graph on all quota threads (ids compared), the layer-0 work counters and the useful bytes they
    imply, and the matched-recall point (smallest ef of the sweep with recall@10 >= 0.95).  Shape of the reference's
    own harness: testing/vector_test.cc:138-197."""
    from oracle import oracle as O
    Nh, D, K, ef = host_rows.shape[0], args.dim, args.k, args.hnsw_ef
    t_leg = time.perf_counter()
    # the two ingest routes over the same 1M rows: one vk_index_add_batch, and what IndexSchema does -- one vk_index_add per
    # key from the writer pool (src/index_schema.cc:755-791; here 16 native threads), linked in bulk on the device at flush
    routes = None
    if Nh >= 1_000_000 and not args.no_serving:
        sub = host_rows[:1_000_000]
        t0 = time.perf_counter()
        hb = vsa.Index("HNSW", D, "COSINE", initial_cap=len(sub), m=16, ef_construction=200, ef_runtime=ef)
        hb.add_batch(sub)
        hb.flush()
        t_b = time.perf_counter() - t0
        del hb
        t0 = time.perf_counter()
        hs = vsa.Index("HNSW", D, "COSINE", initial_cap=len(sub), m=16, ef_construction=200, ef_runtime=ef)
        failed, t_adds = vsa.probe_add_single(hs, sub, threads=16)
        hs.flush()
        t_s = time.perf_counter() - t0
        sst = hs.stats()
        routes = {"rows": len(sub), "add_batch_s": round(t_b, 2), "single_adds_16_threads_s": round(t_s, 2), "of_which_staging_s": round(t_adds, 2),
                  "single_over_batch": round(t_s / t_b, 3), "failed": failed, "staged": int(sst.staged_adds), "linked_on_device": int(sst.staged_adds_device)}
        del hs
    t0 = time.perf_counter()
    h = vsa.Index("HNSW", D, "COSINE", initial_cap=Nh, m=16, ef_construction=200, ef_runtime=ef)
    h.add_batch(host_rows)
    h.flush()
    build_s = time.perf_counter() - t0
    nq = args.hnsw_queries
    Qh = make_queries(A, nq, D, device, 9090)
    hq = Qh.cpu().numpy()
    gt = exact_ground_truth(flat_ix, hq, K, Nh, total_rows)
    od = torch.empty(nq, K, device=device, dtype=torch.float32)
    ol = torch.empty(nq, K, device=device, dtype=torch.int64)
    on = torch.empty(nq, device=device, dtype=torch.int32)

    vis_modes = {}
    VIS = {0: ("hnsw_search_kernel", "one bit per node in memory"), 1: ("hnsw_search_hash_kernel", "hash table in HBM"),
           2: ("hnsw_search_hash_kernel", "buckets in HBM, counts in LDS"), 3: ("hnsw_search_ldsvis_kernel", "LDS, 12 KB per wave, spill to memory"),
           5: ("hnsw_search_ldsvis_kernel", "LDS, 32 KB per wave, spill to memory")}

    def point(ef_s, reps=5):
        ms = timed(lambda: h.search_batch_device(Qh.data_ptr(), nq, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), ef=ef_s,
                                                 stream=stream_ptr()), reps)
        gl = ol.cpu().numpy().view(np.uint64).copy()
        vis_modes[ef_s] = int(h.stats().last_visited_mode)         # (of the launches just timed: a 1024-query call takes another kernel)
        _D, _L, _N = h.search_batch(hq[:1024], K, ef=ef_s)        # host path once: fills the work counters
        st = h.stats()
        useful = (st.last_n_eval * (D * 4 + 4) + st.last_n_hops * 132) / 1024.0 * nq
        gbs = useful / (ms * 1e-3) / 1e9
        return gl, {"ef": ef_s, "gpu_qps": round(nq / (ms * 1e-3), 1), "ms_per_batch": round(ms, 3),
                    "recall_at_10": round(recall_of(gl, gt, K), 4), "visited_mode": vis_modes[ef_s],
                    "n_eval_per_query": round(st.last_n_eval / 1024.0, 1), "n_hops_per_query": round(st.last_n_hops / 1024.0, 1),
                    "useful_gbs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                    "frac_of_gather_ceiling": round(gbs / GATHER_CEILING_GBS, 4)}

    gl, head = point(ef)
    # matched recall: the smallest ef of the sweep whose recall@10 reaches 0.95 (SURVEY.md 8e asks for both points)
    sweep, matched = [head], None
    for ef_s in (192, 256, 384, 512, 544, 576, 640, 768, 1024, 1536, 2048, 3072, 4096):
        if ef_s <= ef:
            continue
        _, pt = point(ef_s, reps=2)
        sweep.append(pt)
        if pt["recall_at_10"] >= 0.95:
            matched = pt
            break
    # one query at a time (latency kernel)
    t1 = time.perf_counter()
    for i in range(64):
        h.search(hq[i], K, ef=ef)
    lat_ms = (time.perf_counter() - t1) / 64 * 1e3
    # single-query traffic through the dispatcher (submit: 8 producers, 4 batches of 8192 outstanding; blocking: 256 callers)
    try:
        serving = serving_leg(h, hq, K, head["gpu_qps"], max_batch=nq, wait_us=2000, producers=8, window=4 * nq, total=16 * nq,
                              threads=256, calls=64, ef=ef)
    except Exception as e:   # noqa: BLE001
        serving = {"error": f"{type(e).__name__}: {e}"}
    # CPU: the oracle searches the very same graph (SaveIndex chunk stream straight into its C loader), one query per
    # thread on every quota thread, each thread with its own visited list like hnswlib's pool
    t1 = time.perf_counter()
    o = O.HNSW.from_product_index(h.save_raw, D, "COSINE", 16, ef_construction=200)
    export_s = time.perf_counter() - t1
    threads = effective_cpus()
    views = [o] + [o.view() for _ in range(threads - 1)]
    ncq = min(nq, threads * args.hnsw_cpu_queries_per_thread)
    o.search(hq[0], K, ef=ef)

    def cpu_chunk(t):
        return [(i, views[t].search(hq[i], K, ef=ef)) for i in range(t, ncq, threads)]

    t1 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(cpu_chunk, range(threads)))
    cdt = time.perf_counter() - t1
    cpu_res = dict(kv for p in parts for kv in p)
    cpu_recall = recall_of([cpu_res[i][1] for i in range(ncq)], gt[:ncq], K)
    same = sum(int(cpu_res[i][1].tolist() == gl[i][:len(cpu_res[i][1])].tolist()) for i in range(ncq))
    del views
    return {"workload": f"HNSW {Nh}x{D} fp32 COSINE M=16 efC=200 efSearch={ef} k={K} (BASELINE.json configs[2])",
            "rows": Nh, "M": 16, "ef_construction": 200, "k": K, "queries_per_batch": nq,
            "build_s": round(build_s, 2), "build_inserts_per_s": round(Nh / build_s, 1), "build": "device-assisted (K9)", "build_routes_1M": routes,
            **head, "single_query_ms": round(lat_ms, 3),
            "roofline": {"bound": "hbm", "achieved": head["useful_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": head["frac_of_hbm_peak"], **hnsw_pmc_traffic(Nh, nq, ef),
                         "kernel": VIS[vis_modes[ef]][0], "visited_set": VIS[vis_modes[ef]][1] + " (vk_index_stats.last_visited_mode)",
                         "bytes": "n_eval*(D*4+4) + n_hops*132 per query, counted by the kernel",
                         "gather_ceiling_gbs": GATHER_CEILING_GBS, "gather_ceiling_source": "profiles/r01_gather_ceiling_10Mx768.log"},
            "matched_recall_0.95": matched, "ef_sweep": sweep, "single_query_serving": serving,
            "cpu": {"kind": "port", "threads": threads, "qps": round(ncq / cdt, 1), "qps_per_thread": round(ncq / cdt / threads, 1),
                    "queries": ncq, "ef": ef, "recall_at_10": round(cpu_recall, 4), "same_graph": True,
                    "ids_identical_to_gpu": f"{same}/{ncq}", "graph_export_s": round(export_s, 2)},
            "leg_s": round(time.perf_counter() - t_leg, 1)}


def hybrid_leg(args, flat_ix, host_rows, A, device, stream_ptr, total_rows):
    """One shard of BASELINE.json configs[4]: HNSW over 1.25M rows + a TAG-like filter as an allow-bitmap over labels
    (planner.cc:21-45 picks the path): inline filter at 10 % selectivity with efSearch=256, and the pre-filter branch
    below 0.1 %.  Recall against the exact filtered FLAT answer; the oracle answers a sample on the same graph."""
    from oracle import oracle as O
    Nh, D, K, ef_h = host_rows.shape[0], args.dim, args.k, 256
    t_leg = time.perf_counter()
    t0 = time.perf_counter()
    h = vsa.Index("HNSW", D, "COSINE", initial_cap=Nh, m=16, ef_construction=200, ef_runtime=ef_h)
    h.add_batch(host_rows)
    h.flush()
    build_s = time.perf_counter() - t0
    nq = args.hybrid_queries
    Qh = make_queries(A, nq, D, device, 7070)
    hq = Qh.cpu().numpy()
    tag_bits = O.allow_bitmap(np.arange(3, Nh, 10, dtype=np.uint64), Nh)           # "tag t3": 10 % of the rows
    d_bits = torch.from_numpy(tag_bits.view(np.int64)).to(device)
    gt_f = np.empty((nq, K), np.uint64)
    for i in range(0, nq, 256):
        _, L, _ = flat_ix.search_batch(hq[i:i + 256], K, allow=tag_bits, allow_nbits=Nh)
        gt_f[i:i + 256] = L
    od = torch.empty(nq, K, device=device, dtype=torch.float32)
    ol = torch.empty(nq, K, device=device, dtype=torch.int64)
    on = torch.empty(nq, device=device, dtype=torch.int32)
    ms_f = timed(lambda: h.search_batch_device(Qh.data_ptr(), nq, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), ef=ef_h,
                                               d_allow=d_bits.data_ptr(), allow_nbits=Nh, stream=stream_ptr()), 3)
    glf = ol.cpu().numpy().view(np.uint64).copy()
    recall_f = recall_of(glf, gt_f, K)
    Dh, Lh, Nn = h.search_batch(hq[:512], K, ef=ef_h, allow=tag_bits, allow_nbits=Nh)
    st = h.stats()
    # the oracle on the same graph, same filter: ids, distance bits (a sample)
    o = O.HNSW.from_product_index(h.save_raw, D, "COSINE", 16, ef_construction=200)
    ns = min(64, nq)
    same = 0
    for i in range(ns):
        e_d, e_l = o.search(hq[i], K, ef=ef_h, allow=tag_bits, allow_nbits=Nh)
        same += int(Lh[i, :Nn[i]].tolist() == e_l.tolist() and Dh[i, :Nn[i]].view(np.uint32).tolist() == e_d.view(np.uint32).tolist())
    # pre-filter (<= 0.001 * N keys): exact kNN over the key list, one call per query like CalcBestMatchingPrefilteredKeys
    keys = np.sort(np.random.default_rng(77).choice(Nh, max(1, Nh // 2000), replace=False)).astype(np.uint64)
    key_bits = O.allow_bitmap(keys, Nh)
    h.search_labels(hq[0], K, keys)      # (first use of the gather kernel in a process: module load, pinned buffers)
    t1 = time.perf_counter()
    pre = [h.search_labels(hq[i], K, keys) for i in range(256)]
    pre_dt = time.perf_counter() - t1
    _, gt_p, ngt = flat_ix.search_batch(hq[:256], K, allow=key_bits, allow_nbits=Nh)
    pre_ok = all(set(pre[i][1].tolist()) == set(gt_p[i, :ngt[i]].tolist()) for i in range(256))
    return {"workload": f"one shard of BASELINE.json configs[4]: HNSW {Nh}x{D} fp32 COSINE M=16 efC=200 + TAG filter, efSearch={ef_h}, k={K}",
            "rows": Nh, "build_s": round(build_s, 2), "queries_per_batch": nq,
            "inline_filter": {"selectivity": 0.1, "ef": ef_h, "gpu_qps": round(nq / (ms_f * 1e-3), 1), "ms_per_batch": round(ms_f, 3),
                              "recall_at_10": round(recall_f, 4),
                              "n_eval_per_query": round(st.last_n_eval / 512.0, 1), "n_hops_per_query": round(st.last_n_hops / 512.0, 1),
                              "frontier_redo_queries": int(st.last_frontier_redo), "frontier_dropped": int(st.last_frontier_dropped),
                              "oracle_same_graph_bit_identical": f"{same}/{ns}"},
            "pre_filter": {"keys": int(len(keys)), "selectivity": round(len(keys) / Nh, 5),
                           "qps_single_caller": round(256 / pre_dt, 1), "exact": bool(pre_ok)},
            "leg_s": round(time.perf_counter() - t_leg, 1)}


def bf16_ip_leg(args, A, device, stream_ptr, local_rank):
    """One shard of BASELINE.json configs[3]: FLAT 10Mx768, bf16 row storage, IP metric (rows L2-normalised by the
    generator, so IP == cosine here -- stated, not assumed by the kernel), k=10, batch=256.  Parity: the answer
    restricted by a filter to the first rows must equal the oracle's IP answer over the RNE-rounded rows, bit for bit."""
    from oracle import oracle as O
    N, D, B, K = args.bf16_rows, args.dim, args.batch, args.k
    t_leg = time.perf_counter()
    ix = vsa.Index("FLAT", D, "IP", initial_cap=N, device_id=local_rank, dtype="bf16", options={"kernel-timing": 1})
    base_ptr, stride = ix.device_rows(N)
    table = device_view_typed(base_ptr, (N, stride // 2), device, "<i2").view(torch.bfloat16)
    if stride != D * 2:
        table[:, D:] = 0
    for lo, x in gen_rows(0, N, D, device):
        table[lo: lo + x.shape[0], :D] = x        # round to nearest even, as the library's ingest does
    torch.cuda.synchronize()
    ix.commit_device_rows(N, np.arange(N, dtype=np.uint64))
    Q = make_queries(A, B, D, device, 4242)
    od = torch.empty(B, K, device=device, dtype=torch.float32)
    ol = torch.empty(B, K, device=device, dtype=torch.int64)
    on = torch.empty(B, device=device, dtype=torch.int32)
    ms = timed(lambda: ix.search_batch_device(Q.data_ptr(), B, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=stream_ptr()),
               args.steps)
    ms1 = timed(lambda: ix.search_batch_device(Q.data_ptr(), 1, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=stream_ptr()), 5)
    S = min(100_000, N)
    host = np.ascontiguousarray(table[:S, :D].float().cpu().numpy())
    o = O.Flat(D, "IP", isa="skylake", max_elements=S)
    o.add_many(host, np.arange(S, dtype=np.uint64), borrowed=True)
    bits = O.allow_bitmap(np.arange(S, dtype=np.uint64), S)
    hq = Q.cpu().numpy()
    ok = True
    bd, bl, bn = ix.search_batch(hq[:32], K, allow=bits, allow_nbits=S)
    for i in range(32):
        e_d, e_l = o.search(hq[i], K)
        ok = ok and bl[i, :bn[i]].tolist() == e_l.tolist() and bd[i, :bn[i]].view(np.uint32).tolist() == e_d.view(np.uint32).tolist()
    for i in range(2):
        gd, gl = ix.search(hq[i], K, allow=bits, allow_nbits=S)
        e_d, e_l = o.search(hq[i], K)
        ok = ok and gl.tolist() == e_l.tolist() and gd.view(np.uint32).tolist() == e_d.view(np.uint32).tolist()
    scan_bytes = N * stride
    return {"workload": f"one shard of BASELINE.json configs[3]: FLAT {N}x{D} bf16 rows, IP, k={K}, batch={B}",
            "rows": N, "gpu_qps": round(B / (ms * 1e-3), 1), "ms_per_step": round(ms, 3),
            "hbm_gbs_algorithmic": round(scan_bytes / (ms * 1e-3) / 1e9, 1),
            "tflops": round(2.0 * N * D * B / (ms * 1e-3) / 1e12, 2),
            "single_query": {"ms": round(ms1, 4), "hbm_gbs": round(scan_bytes / (ms1 * 1e-3) / 1e9, 1),
                             "hbm_frac": round(scan_bytes / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "parity_vs_oracle": "bit-exact" if ok else "MISMATCH", "leg_s": round(time.perf_counter() - t_leg, 1)}


def hnsw_sharded_leg(args, flat_ix, table, A, device, stream_ptr, world, rank, r0, dist, gather, local_rank):
    """N > 1: one independent HNSW graph per shard (as one graph per cluster shard in the reference), every rank
    searches its graph for the same query batch, per-shard top-k lists are all-gathered and merged by
    (distance, label).  Ground truth = the exact FLAT answer over the same rows, merged the same way."""
    from oracle import oracle as O
    Nh, D, K, ef = min(args.hnsw_rows // world, table.shape[0]), args.dim, args.k, args.hnsw_ef
    host_rows = np.ascontiguousarray(table[:Nh, :D].float().cpu().numpy())
    t0 = time.perf_counter()
    h = vsa.Index("HNSW", D, "COSINE", initial_cap=Nh, m=16, ef_construction=200, ef_runtime=ef, device_id=local_rank)
    h.add_batch(host_rows, np.arange(r0, r0 + Nh, dtype=np.uint64))
    h.flush()
    build_s = time.perf_counter() - t0
    nq = min(args.hnsw_queries, 2048)
    qg = torch.Generator(device=device)
    qg.manual_seed(9090)
    Qh = torch.nn.functional.normalize(torch.randn(nq, 32, generator=qg, device=device) @ A.T +
                                       0.05 * torch.randn(nq, D, generator=qg, device=device), dim=1).contiguous()
    od = torch.empty(nq, K, device=device, dtype=torch.float32)
    ol = torch.empty(nq, K, device=device, dtype=torch.int64)
    on = torch.empty(nq, device=device, dtype=torch.int32)
    ad = torch.empty(world * nq, K, device=device, dtype=torch.float32)
    al = torch.empty(world * nq, K, device=device, dtype=torch.int64)
    fd = torch.empty(nq, K, device=device, dtype=torch.float32)
    fl = torch.empty(nq, K, device=device, dtype=torch.int64)
    fn = torch.empty(nq, device=device, dtype=torch.int32)

    def merged(search):
        search()
        gather(ad, od)
        gather(al, ol)
        vsa.merge_topk_device(ad.data_ptr(), al.data_ptr(), world, nq, K, fd.data_ptr(), fl.data_ptr(), fn.data_ptr(),
                              local_rank, stream_ptr())

    def hnsw_search():
        h.search_batch_device(Qh.data_ptr(), nq, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), ef=ef, stream=stream_ptr())

    # exact answer over the same rows: this rank's FLAT shard restricted to its first Nh rows
    bits = O.allow_bitmap(np.arange(r0, r0 + Nh, dtype=np.uint64), r0 + Nh)
    d_bits = torch.from_numpy(bits.view(np.int64)).to(device)

    def flat_search():
        flat_ix.search_batch_device(Qh.data_ptr(), nq, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(),
                                    d_allow=d_bits.data_ptr(), allow_nbits=r0 + Nh, stream=stream_ptr())

    merged(flat_search)
    torch.cuda.synchronize()
    gt = fl.cpu().numpy().copy()
    merged(hnsw_search)
    dist.barrier()
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        merged(hnsw_search)
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    if args.backend == "nccl":
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    got = fl.cpu().numpy()
    recall = float(np.mean([len(set(a.tolist()) & set(b.tolist())) for a, b in zip(got, gt)])) / K
    return {"shards": world, "rows_per_shard": Nh, "rows": Nh * world, "M": 16, "ef_construction": 200, "ef": ef, "k": K,
            "queries_per_batch": nq, "build_s_per_shard": round(build_s, 2), "gpu_qps": round(nq * reps / float(dt.item()), 1),
            "recall_at_10": round(recall, 4), "merge": "all-gather + (distance,label) merge on every rank"}


def _fill_shard(ix, shard, r0, n, D, sdev, bf16):
    """rows [r0, r0 + n) of the synthetic table generated on the shard's own device straight into its HBM row table"""
    esz = 2 if bf16 else 4
    ptr, stride = ix.shard_device_rows(shard, n)
    tab = (device_view_typed(ptr, (n, stride // 2), sdev, "<i2").view(torch.bfloat16) if bf16 else device_view(ptr, (n, stride // 4), sdev))
    if stride != D * esz:
        tab[:, D:] = 0
    for lo, x in gen_rows(r0, n, D, sdev):
        tab[lo - r0: lo - r0 + x.shape[0], :D] = x
    torch.cuda.synchronize(sdev)
    ix.shard_commit_device_rows(shard, n, np.arange(r0, r0 + n, dtype=np.uint64))
    return tab


def _final_rows(ix, rows):
    """rows the candidate filter's MAIN pass walked in the most recent batch -- the launch the library's kernel timing brackets
    (vk_index_stats.last_filter_final_rows: all rows, or, with option filter-two-pass, the rows behind the early pass's share);
    of shard 0 for a sharded index (equal shards)"""
    try:
        st = ix.shard_stats(0) if ix.shard_count() > 1 else ix.stats()
    except Exception:   # noqa: BLE001
        st = ix.stats()
    v = int(st.last_filter_final_rows)
    return v if 0 < v <= rows else rows


def _shard_filter_ms(ix, before, after=None):
    """the SLOWEST shard's final-pass time per launch: max over shards of a shard's own delta(ns) / delta(batches)
    (vk_index_shard_stats; a difference of maxima over shards would not be any shard's time).  With after=None returns the
    per-shard snapshot to pass back in as `before`."""
    snap = [ix.shard_stats(s) for s in range(ix.shard_count())]
    if before is None:
        return snap
    best, n = None, 0
    for b, a in zip(before, snap):
        db = a.filter_batches - b.filter_batches
        if db:
            ms = (a.filter_kernel_ns - b.filter_kernel_ns) / 1e6 / db
            best = ms if best is None else max(best, ms)
            n = max(n, db)
    return best, n


def _timed_steps(step, steps, warmup):
    """HIP events on the current stream + the wall clock around `steps` calls (after `warmup` untimed ones)"""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, (time.perf_counter() - t0) / steps * 1e3


def lib_multi_gpu(args, world, rank, dist, device):
    """--gpus N through the PRODUCT's multi-GPU path: ONE vk_index with n_shards = N (shard s on HIP device s;
    vk_index_params.shard_devices), built and searched from ONE process like valkey-server would -- `python bench.py
    --gpus N` needs no launcher; under torch.distributed.run rank 0 does the work and the other ranks only keep the launch
    contract (barriers, max-over-ranks timing).  A step = one vk_index_search_batch_device on the sharded index: queries
    broadcast by peer copy, every shard scans its rows on its own device and stream (one enqueue thread per shard),
    per-shard top-k lists gathered on device 0, (distance,label) merge.  The headline is BASELINE.json configs[1] with
    the 10M rows dealt over the N GPUs (strong scaling); configs[3] (N x 10M bf16 rows, IP) and configs[4] (one HNSW
    graph of 1.25M rows per GPU + a TAG filter) run as legs of the same line.  Returns the JSON dict on rank 0 (None
    elsewhere), or "fallback" if the sharded index cannot be set up and there are ranks to fall back to."""
    N, D, B, K = args.rows, args.dim, args.batch, args.k
    bf16 = args.dtype == "bf16"
    esz = 2 if bf16 else 4
    devs = [0] * world if args.same_device else list(range(world))
    multi_proc = dist is not None
    state = {}
    ok = torch.ones(1, device=device if (multi_proc and args.backend == "nccl") else "cpu")
    if rank == 0:
        try:
            if vsa.lib().vk_device_count() < (1 if args.same_device else world):
                raise RuntimeError(f"this process sees {vsa.lib().vk_device_count()} HIP devices, needs {world}")
            # The library REFUSES devices without peer access (an index that looks multi-GPU and gathers through host memory).
            # A benchmark run on such a box should still produce its line -- loudly labelled: the option is set for every index
            # of this process and the line says so (config.peer_access).
            distinct = sorted(set(devs))
            no_peer = [(a, b) for a in distinct for b in distinct if a != b and not torch.cuda.can_device_access_peer(a, b)]
            state["peer_access"] = not no_peer
            state["peer_matrix"] = [[1 if a == b or torch.cuda.can_device_access_peer(a, b) else 0 for b in distinct] for a in distinct]
            if no_peer:
                os.environ["VK_SHARD_ALLOW_STAGED"] = "1"
                print(f"[bench] WARNING: no peer access between devices {no_peer[:4]}...: the sharded index stages its broadcast "
                      f"and gather through host memory (VK_SHARD_ALLOW_STAGED=1 set for this run)", file=sys.stderr, flush=True)
            t_build = time.time()
            ix = vsa.Index("FLAT", D, "COSINE", initial_cap=N, dtype=args.dtype, shard_devices=devs)
            tabs = []
            for s_i, dv in enumerate(devs):
                r0, r1 = s_i * N // world, (s_i + 1) * N // world
                tabs.append(_fill_shard(ix, s_i, r0, r1 - r0, D, torch.device("cuda", dv), bf16))
            state["build_s"] = time.time() - t_build
            gA = torch.Generator(device=device)
            gA.manual_seed(1234)
            A = torch.randn(D, 32, generator=gA, device=device, dtype=torch.float32)
            Q = make_queries(A, B, D, device, 4242)
            od = torch.empty(B, K, device=device, dtype=torch.float32)
            ol = torch.empty(B, K, device=device, dtype=torch.int64)
            on = torch.empty(B, device=device, dtype=torch.int32)
            ws = torch.cuda.Stream(device=device)
            torch.cuda.set_stream(ws)
            state.update(ix=ix, A=A, Q=Q, od=od, ol=ol, on=on, ws=ws, tabs=tabs)

            def step():
                ix.search_batch_device(Q.data_ptr(), B, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=ws.cuda_stream)

            for _ in range(max(1, args.warmup)):
                step()
            torch.cuda.synchronize()
            state["step"] = step
        except Exception as e:   # noqa: BLE001
            ok.zero_()
            state["error"] = f"{type(e).__name__}: {e}"
    if multi_proc:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) == 0.0:
        if not multi_proc:
            raise RuntimeError(f"bench: the library's multi-GPU path is unavailable: {state.get('error')}")
        if rank == 0:
            print(f"bench: library multi-GPU path unavailable ({state.get('error')}); falling back to one process per GPU",
                  file=sys.stderr)
        state.clear()
        torch.cuda.empty_cache()
        return "fallback"

    def barrier():
        if multi_proc:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    st0 = state["ix"].stats() if rank == 0 else None
    t0 = time.perf_counter()
    if rank == 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            state["step"]()
        e1.record()
    barrier()
    dt = time.perf_counter() - t0
    if multi_proc:
        t = torch.tensor([dt], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    out = None
    if rank == 0:
        ix, A, Q, od, ol, tabs = state["ix"], state["A"], state["Q"], state["od"], state["ol"], state["tabs"]
        st1 = ix.stats()
        dev_ms = e0.elapsed_time(e1) / args.steps
        # per-shard kernel times: a second region of K steps with the library's event pairs around each shard's main pass
        # (option kernel-timing: the timed region above runs without those stream markers)
        ix.set_option("kernel-timing", 1)
        for _ in range(2):
            state["step"]()
        torch.cuda.synchronize()
        sh0 = _shard_filter_ms(ix, None)
        for _ in range(args.steps):
            state["step"]()
        torch.cuda.synchronize()
        sh1 = [ix.shard_stats(s_i) for s_i in range(ix.shard_count())]
        ix.set_option("kernel-timing", 0)
        per_shard_ms = [round((a.filter_kernel_ns - b.filter_kernel_ns) / 1e6 / (a.filter_batches - b.filter_batches), 4)
                        if a.filter_batches != b.filter_batches else None for b, a in zip(sh0, sh1)]
        res_d, res_l = od.cpu().numpy(), ol.cpu().numpy().view(np.uint64)
        n_local = N // world
        stride = ((D + 63) // 64) * 64 * esz
        scan_bytes = n_local * stride                  # algorithmic bytes of one pass over ONE GPU's shard
        # the dominant kernel per shard: the candidate filter; its duration = the SLOWEST shard's HIP events (the library
        # records them around the launches on each shard's stream; vk_index_stats of a sharded index reports the maximum)
        timed = [m for m in per_shard_ms if m]
        filt_ms, filt_n = (max(timed), args.steps) if timed else (None, 0)
        fan_calls = st1.fanout_calls - st0.fanout_calls
        fan_us = (st1.fanout_enqueue_ns - st0.fanout_enqueue_ns) / 1e3 / fan_calls if fan_calls else None
        kern_ms = filt_ms if filt_ms else dev_ms
        kern_bytes = (_final_rows(ix, n_local) if filt_ms else n_local) * stride     # the main pass's rows (bench.py, N = 1)
        # per GPU, the whole timed step: one pass over ONE GPU's shard / (HIP-event time of the K timed steps / K) -- fan-out,
        # every shard's launches, gather and merge inside; the slowest shard's main pass alone is `dominant_kernel`
        roofline = {"bound": "hbm", "achieved": round(scan_bytes / (dev_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(scan_bytes / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                    "per_gpu": True, "priced": "whole step", "algorithmic_bytes": scan_bytes, "step_ms_on_stream": round(dev_ms, 4),
                    "dominant_kernel": ({"kernel": "flat_filter_bdma_kernel (slowest shard)", "achieved": round(kern_bytes / (kern_ms * 1e-3) / 1e9, 2),
                                         "frac": round(kern_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "per_launch_ms": round(kern_ms, 4),
                                         "launches_timed": int(filt_n), "algorithmic_bytes": kern_bytes,
                                         "measured": "the library's HIP event pairs (option kernel-timing), K steps behind the timed region"}
                                        if filt_ms else None),
                    "per_shard_main_pass_ms": per_shard_ms,
                    "aggregate_gbs": round(world * scan_bytes / (dev_ms * 1e-3) / 1e9, 1),
                    "host_fanout_enqueue_us_per_step": round(fan_us, 1) if fan_us else None}

        def leg(fn, *a):
            try:
                return fn(*a)
            except Exception as e:   # noqa: BLE001
                import traceback
                return {"error": f"{type(e).__name__}: {e}", "where": traceback.format_exc().strip().splitlines()[-3:]}

        # ---- the gather, A/B: one-shot peer copies (the default, timed above) against the in-library RCCL all-gather of the
        #      per-shard top-k (option shard-gather = 1; north_star's collective) -- same steps, same answer required
        def gather_ab():
            g0 = ix.stats().rccl_gathers
            ix.set_option("shard-gather", 1)
            try:
                for _ in range(max(2, args.warmup)):
                    state["step"]()
                torch.cuda.synchronize()
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record()
                for _ in range(args.steps):
                    state["step"]()
                a1.record()
                torch.cuda.synchronize()
                same = bool((ol.cpu().numpy().view(np.uint64) == res_l).all() and (od.cpu().numpy().view(np.uint32) == res_d.view(np.uint32)).all())
                return {"peer_copies_ms_per_step": round(dev_ms, 4), "rccl_all_gather_ms_per_step": round(a0.elapsed_time(a1) / args.steps, 4),
                        "rccl_ranks": len(set(devs)), "lists_per_rank": max(devs.count(d) for d in set(devs)),
                        "rccl_gathers": int(ix.stats().rccl_gathers - g0), "answers_identical": same,
                        "note": "one communicator per distinct device inside the one process (ncclCommInitAll); two ncclAllGather "
                                "(distances f32, labels u64) of [lists_per_rank][batch][k] per step in one group call"}
            finally:
                ix.set_option("shard-gather", 0)
        gather = leg(gather_ab)

        # ---- CPU baseline (same oracle leg as N = 1) + parity of the merged answer on the queries it scanned ----
        cpu = parity = None
        host_rows = None
        need_host = not args.no_cpu_baseline or args.hnsw_rows > 0 or args.hybrid_rows > 0
        if need_host and not bf16:
            host_rows = np.empty((N, D), np.float32)
            for s_i, tab in enumerate(tabs):
                r0 = s_i * N // world
                host_rows[r0: r0 + tab.shape[0]] = tab[:, :D].cpu().numpy()
        if not args.no_cpu_baseline and host_rows is not None:
            from oracle import oracle as O
            S = N if args.cpu_rows <= 0 else min(args.cpu_rows, N)
            flat = O.Flat(D, "COSINE", isa="skylake", max_elements=S)
            flat.add_many(host_rows[:S], np.arange(S, dtype=np.uint64), borrowed=True)
            hq = Q.cpu().numpy()
            threads = effective_cpus()
            nqt = threads * args.cpu_queries_per_thread
            t1 = time.perf_counter()
            with ThreadPoolExecutor(threads) as ex:
                res = list(ex.map(lambda i: flat.search(hq[i % B], K), range(nqt)))
            cdt = time.perf_counter() - t1
            full = S == N
            cpu = {"value": round(nqt / cdt * S / N, 4), "unit": "queries/s", "cores": threads, "kind": "port",
                   "seconds": round(cdt, 2), "host_gbs": round(nqt * S * D * 4 / cdt / 1e9, 1),
                   "sample": f"oracle FLAT scan ({O.cpu_path()} clone of the SimSIMD skylake order), {nqt} queries, each a pass over "
                             f"{'all' if full else 'the first'} {S} rows, one query per thread on {threads} threads (the container's CPU quota)"
                             + ("" if full else f", scaled linearly to {N} rows")}
            if full:   # the merged N-shard answer of the timed step against the CPU path: ids and distance bits
                okp = True
                for i in range(min(B, len(res))):
                    e_d, e_l = res[i]
                    okp = okp and res_l[i].tolist() == e_l.tolist() and res_d[i].view(np.uint32).tolist() == e_d.view(np.uint32).tolist()
                parity = "bit-exact" if okp else "MISMATCH"
            del flat
        verify = None
        if args.verify_merge:   # test aid (small --rows): the N-shard answer must be bit-identical to the 1-shard answer
            full_ix = vsa.Index("FLAT", D, "COSINE", initial_cap=N, device_id=0, dtype=args.dtype)
            fp, fstride = full_ix.device_rows(N)
            ft = (device_view_typed(fp, (N, fstride // 2), device, "<i2").view(torch.bfloat16) if bf16
                  else device_view(fp, (N, fstride // 4), device))
            if fstride != D * esz:
                ft[:, D:] = 0
            for lo, x in gen_rows(0, N, D, device):
                ft[lo: lo + x.shape[0], :D] = x
            torch.cuda.synchronize()
            full_ix.commit_device_rows(N, np.arange(N, dtype=np.uint64))
            fd, fl, fn = full_ix.search_batch(Q.cpu().numpy(), K)
            same = bool((fl == res_l).all() and (fd.view(np.uint32) == res_d.view(np.uint32)).all())
            verify = "bit-identical" if same else "MISMATCH"
            del full_ix
        hnsw = hybrid = None
        if args.hnsw_sharded and args.hnsw_rows > 0 and host_rows is not None:
            hnsw = leg(sharded_hnsw_leg, args, ix, host_rows[:min(args.hnsw_rows, N)], A, device, state["ws"], devs, N)
        if args.hybrid_rows > 0 and host_rows is not None:
            hybrid = leg(sharded_hybrid_leg, args, ix, host_rows[:min(args.hybrid_rows * world, N)], A, device, state["ws"], devs, N)
        host_rows = None
        cfg3 = None
        if args.bf16_rows > 0 and not bf16:
            state.pop("tabs", None)
            tabs = None
            cfg3 = leg(sharded_bf16_ip_leg, args, A, device, state["ws"], devs)
        qps = B * args.steps / dt
        out = {"metric": BASELINE_METRIC if not bf16 else "kNN queries/sec, FLAT 10Mx768 bf16-stored cosine k=10 batch=256",
               "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "row_storage": args.dtype, "data": "synthetic",
               "config": {"workload": f"FLAT {N}x{D} {'bf16 rows' if bf16 else 'fp32'} COSINE k={K} batch={B} (BASELINE.json configs[1]), "
                                      f"rows dealt over {world} GPUs",
                          "rows_per_gpu": n_local, "sharding": f"rows/{world}", "recall_at_10": 1.0, "parity_vs_oracle": parity,
                          "peer_access": state.get("peer_access", True), "peer_access_matrix": state.get("peer_matrix"),
                          "parallelism": f"one vk_index with n_shards={world} in ONE process"
                                         + (" (rank 0 of the launcher's ranks; the others idle)" if multi_proc else "")
                                         + ": queries broadcast by peer copy, one enqueue thread per shard, per-shard top-k "
                                           "gathered on device 0, (distance,label) merge",
                          "devices": devs, "verify_merge": verify},
               "roofline": roofline, "cpu_baseline": cpu, "gather": gather, "hnsw": hnsw, "config4_sharded_hybrid": hybrid,
               "config3_sharded_bf16_ip": cfg3, "build_s": round(state["build_s"], 2)}
        print(json.dumps(out))
        if verify is not None:
            print(json.dumps({"verify_merge": verify, "shards": world, "rows": N}))
            assert verify == "bit-identical"
    if multi_proc:
        dist.barrier()
        dist.destroy_process_group()
    return out


def sharded_bf16_ip_leg(args, A, device, ws, devs):
    """BASELINE.json configs[3]: FLAT, N GPUs x 10M rows of 768 bf16 (weak scaling: 80M rows on 8 GPUs), IP, k=10,
    batch=256, rows generated per shard on its own device (L2-normalised by the generator, so IP == cosine here).  QPS
    over the whole index, per-GPU HBM roofline from the slowest shard's filter kernel, and a parity spot check: the
    answer restricted by a filter to the first rows must equal the oracle's IP answer over the RNE-rounded rows."""
    from oracle import oracle as O
    n, D, B, K, S = args.bf16_rows, args.dim, args.batch, args.k, len(devs)
    t_leg = time.perf_counter()
    ix = vsa.Index("FLAT", D, "IP", initial_cap=n * S, dtype="bf16", shard_devices=devs, options={"kernel-timing": 1})
    first = None
    for s_i, dv in enumerate(devs):
        tab = _fill_shard(ix, s_i, s_i * n, n, D, torch.device("cuda", dv), True)
        if s_i == 0:
            first = np.ascontiguousarray(tab[:min(100_000, n), :D].float().cpu().numpy())
        del tab
    build_s = time.perf_counter() - t_leg
    Q = make_queries(A, B, D, device, 4242)
    od = torch.empty(B, K, device=device, dtype=torch.float32)
    ol = torch.empty(B, K, device=device, dtype=torch.int64)
    on = torch.empty(B, device=device, dtype=torch.int32)
    step = lambda: ix.search_batch_device(Q.data_ptr(), B, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=ws.cuda_stream)   # noqa: E731
    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    sh0 = _shard_filter_ms(ix, None)
    ms, wall_ms = _timed_steps(step, args.steps, 0)
    fms, _fb = _shard_filter_ms(ix, sh0)
    Sn = first.shape[0]
    o = O.Flat(D, "IP", isa="skylake", max_elements=Sn)
    o.add_many(first, np.arange(Sn, dtype=np.uint64), borrowed=True)
    bits = O.allow_bitmap(np.arange(Sn, dtype=np.uint64), Sn)
    hq = Q.cpu().numpy()
    okp = True
    bd, bl, bn = ix.search_batch(hq[:32], K, allow=bits, allow_nbits=Sn)
    for i in range(32):
        e_d, e_l = o.search(hq[i], K)
        okp = okp and bl[i, :bn[i]].tolist() == e_l.tolist() and bd[i, :bn[i]].view(np.uint32).tolist() == e_d.view(np.uint32).tolist()
    stride = ((D + 63) // 64) * 64 * 2
    kern = fms if fms else ms
    kb = (_final_rows(ix, n) if fms else n) * stride          # the main pass's rows of a shard
    return {"workload": f"BASELINE.json configs[3]: FLAT {n * S}x{D} bf16 rows over {S} GPUs ({n} per GPU), IP, k={K}, batch={B}",
            "scaling": "weak", "rows": n * S, "rows_per_gpu": n, "gpu_qps": round(B / (wall_ms * 1e-3), 1), "ms_per_step": round(wall_ms, 3),
            "step_ms_on_stream": round(ms, 3),
            "roofline": {"bound": "hbm", "per_gpu": True, "algorithmic_bytes": kb, "step_algorithmic_bytes": n * stride,
                         "achieved": round(kb / (kern * 1e-3) / 1e9, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(kb / (kern * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "kernel": "flat_filter_bfmma_dma_kernel (slowest shard)" if fms else "whole step", "per_launch_ms": round(kern, 4)},
            "aggregate_scan_gbs": round(S * n * stride / (ms * 1e-3) / 1e9, 1),
            "parity_vs_oracle": "bit-exact" if okp else "MISMATCH", "build_s": round(build_s, 2),
            "leg_s": round(time.perf_counter() - t_leg, 1)}


def _exact_gt(flat_ix, hq, K, bits, nbits):
    out = np.empty((hq.shape[0], K), np.uint64)
    for i in range(0, hq.shape[0], 256):
        _, L, _ = flat_ix.search_batch(hq[i:i + 256], K, allow=bits, allow_nbits=nbits)
        out[i:i + 256] = L
    return out


def sharded_hybrid_leg(args, flat_ix, host_rows, A, device, ws, devs, total_rows):
    """BASELINE.json configs[4]: one HNSW graph (M=16, efC=200) of --hybrid-rows rows per GPU inside ONE sharded index + TAG
    filters of 10 % selectivity (the inline-filter branch of planner.cc:21-45), efSearch=256, k=10 -- the way FT.SEARCH traffic
    brings them: EVERY QUERY CARRIES ITS OWN PREDICATE, and the filter is built INSIDE the timed step from what the query
    layer holds for it, the posting list of the tag (tag.cc:383-455 -> search.cc:301-399), through vk_filter_create (device
    scatter of the id list) and the (predicate, epoch) cache.  Three steps are timed:
      shared_tags_cold   16 tags over 4096 queries, a write phase before every step (epoch bump): all 16 filters are rebuilt
                         from their 10 %-of-N id lists inside the step; each query looks its predicate up
      shared_tags_warm   the same without the write phase: 4096 cache hits
      distinct_per_query 1024 queries, each with a predicate nobody shares (tag_a OR tag_b, 1024 distinct pairs): the step's
                         predicates combined from cached terms in ONE vk_filter_combine_batch call inside the step
    Reported per step: QPS, recall@10 against the exact filtered FLAT answer per query, the layer-0 work counters, the useful
    bytes they imply and their fraction of the HBM peak (through the host entry point: query upload and result copy-out are
    inside), and -- a sample -- the CPU oracle searching the SAME graphs with the SAME bitmaps (ids compared, timed)."""
    from oracle import oracle as O
    Nh, D, K, ef_h, S = host_rows.shape[0], args.dim, args.k, 256, len(devs)
    t_leg = time.perf_counter()
    h = vsa.Index("HNSW", D, "COSINE", initial_cap=Nh, m=16, ef_construction=200, ef_runtime=ef_h, shard_devices=devs)
    h.add_batch(host_rows)
    h.flush()
    build_s = time.perf_counter() - t_leg
    nq = min(args.hybrid_queries, 4096)
    Qh = make_queries(A, nq, D, device, 7070)
    hq = Qh.cpu().numpy()
    T = 16
    tag_ids = [np.flatnonzero(np.random.default_rng(4000 + t).random(Nh) < 0.1).astype(np.uint64) for t in range(T)]   # posting lists
    tag_bits = [O.allow_bitmap(ids, Nh) for ids in tag_ids]
    which = np.arange(nq) % T
    gt = np.empty((nq, K), np.uint64)
    for t in range(T):                                                      # exact filtered answers, one scan group per tag
        sel = np.flatnonzero(which == t)
        gt[sel] = _exact_gt(flat_ix, hq[sel], K, tag_bits[t], Nh)
    epoch = [1]
    t_filters = [0.0]
    n_built = [0]

    def filters_of_step():
        """what the adaptor's BuildFilter does per FT.SEARCH (include/vk_vector_adaptor.h): cache lookup under the predicate's
        text and the write-phase epoch; on a miss the posting list goes to the device"""
        t0 = time.perf_counter()
        out = []
        for i in range(nq):
            key = b"@tag:{t%d}" % which[i]
            f = h.filter_cache_get(key, epoch[0])
            if f is None:
                f = h.make_filter(Nh, labels=tag_ids[which[i]])
                h.filter_cache_put(key, epoch[0], f)
                n_built[0] += 1
            out.append(f)
        t_filters[0] += time.perf_counter() - t0
        return out

    last = {}

    def step(cold):
        if cold:
            epoch[0] += 1                                                   # a write phase happened: cached filters are stale
        fl = filters_of_step()
        last["out"] = h.search_batch_filter_handles(hq, K, fl, ef=ef_h)

    def timed_leg(fn, steps):
        fn()                                                                # warm-up (contexts, first-use allocations)
        t_filters[0], n_built[0] = 0.0, 0
        st0 = h.stats()
        totals[0] = (st0.total_n_eval, st0.total_n_hops)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        dt = (time.perf_counter() - t0) / steps
        return dt, t_filters[0] / steps, n_built[0] / steps

    totals = [None]

    def work(n_queries, dt_search, steps_run):
        """layer-0 work of the steps just timed, from the index's running totals (the shards' kernels add to them on the device)"""
        st = h.stats()
        ev, hp = st.total_n_eval - totals[0][0], st.total_n_hops - totals[0][1]
        ev, hp = ev / steps_run, hp / steps_run
        useful = ev * (D * 4 + 4) + hp * 132
        gbs = useful / dt_search / 1e9
        return {"n_eval_per_query": round(ev / n_queries, 1), "n_hops_per_query": round(hp / n_queries, 1),
                "useful_gbs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / (HBM_PEAK_GBS * len(set(devs))), 4)}

    legs = {}
    for name, cold in (("shared_tags_cold", True), ("shared_tags_warm", False)):
        dt, dt_f, built = timed_leg(lambda: step(cold), 2)
        _D, L, _N = last["out"]
        legs[name] = {"queries": nq, "distinct_predicates": T, "gpu_qps": round(nq / dt, 1), "ms_per_step": round(dt * 1e3, 2),
                      "of_which_filters_ms": round(dt_f * 1e3, 2), "filters_built_per_step": built,
                      "ids_per_filter": int(np.mean([len(x) for x in tag_ids])), "recall_at_10": round(recall_of(L, gt, K), 4),
                      **work(nq, dt - dt_f, 2)}
    glw = last["out"][1].copy()
    # What S graphs need to give back the ONE graph's recall: every shard searched with a fraction of the query's ef (option
    # shard-ef-pct; the reference's cluster mode sends the full ef to every shard, which is what the two steps above time)
    if S > 1:
        for pct in (50, 25):
            try:
                h.set_option("shard-ef-pct", pct)
                dt, dt_f, built = timed_leg(lambda: step(False), 2)
                _D, L, _N = last["out"]
                legs[f"shared_tags_warm_shard_ef_{pct}pct"] = {"queries": nq, "per_shard_ef": max(K, ef_h * pct // 100), "gpu_qps": round(nq / dt, 1),
                                                               "ms_per_step": round(dt * 1e3, 2), "recall_at_10": round(recall_of(L, gt, K), 4),
                                                               **work(nq, dt - dt_f, 2)}
            finally:
                h.set_option("shard-ef-pct", 100)
    # every query its own predicate: tag_a OR tag_b over cached terms, combined on the device inside the step
    nd = min(1024, nq)
    pairs = [(i % T, (i // T + 1 + i) % T) for i in range(nd)]
    pairs = [(a, b if b != a else (b + 1) % T) for a, b in pairs]
    terms = [h.filter_cache_get(b"@tag:{t%d}" % t, epoch[0]) for t in range(T)]

    def step_distinct():
        t0 = time.perf_counter()
        fl = h.combine_filters_batch([(terms[a], terms[b]) for a, b in pairs], "or")      # (one launch for the step's 1024 predicates)
        t_filters[0] += time.perf_counter() - t0
        n_built[0] += nd
        last["out"] = h.search_batch_filter_handles(hq[:nd], K, fl, ef=ef_h)

    dt, dt_f, built = timed_leg(step_distinct, 2)
    _D, Ld, _N = last["out"]
    gtd = np.empty((nd, K), np.uint64)
    uniq = sorted(set(pairs))
    if len(uniq) <= 256:                                                    # (one exact scan per distinct pair)
        for a, b in uniq:
            sel = np.array([i for i, p in enumerate(pairs) if p == (a, b)])
            gtd[sel] = _exact_gt(flat_ix, hq[sel], K, tag_bits[a] | tag_bits[b], Nh)
        rec_d = round(recall_of(Ld, gtd, K), 4)
    else:
        rec_d = None
    legs["distinct_per_query"] = {"queries": nd, "distinct_predicates": len(uniq), "predicate": "tag_a OR tag_b, combined on the device from cached terms",
                                  "gpu_qps": round(nd / dt, 1), "ms_per_step": round(dt * 1e3, 2), "of_which_filters_ms": round(dt_f * 1e3, 2),
                                  "filters_built_per_step": built, "recall_at_10": rec_d, **work(nd, dt - dt_f, 2)}
    # the CPU oracle on the very same graphs with the very same bitmaps (a sample): ids, and its rate on the box's cores
    cpu = None
    try:
        t1 = time.perf_counter()
        graphs = O.HNSW.shards_from_product_index(h.save_raw, D, "COSINE", 16, ef_construction=200, ef=ef_h)
        export_s = time.perf_counter() - t1
        threads = effective_cpus()
        ncq = min(nq, threads * 4)
        views = [[g if t == 0 else g.view() for g in graphs] for t in range(threads)]

        def cpu_chunk(t):
            out = []
            for i in range(t, ncq, threads):
                parts = [g.search(hq[i], K, ef=ef_h, allow=tag_bits[which[i]], allow_nbits=Nh) for g in views[t]]
                dd = np.full((len(parts), K), np.inf, np.float32)
                ll = np.full((len(parts), K), np.iinfo(np.uint64).max, np.uint64)
                cc = np.zeros(len(parts), np.uint64)
                for s_, (pd, pl) in enumerate(parts):
                    dd[s_, :len(pd)], ll[s_, :len(pl)], cc[s_] = pd, pl, len(pd)
                out.append((i, O.merge_topk(dd, ll, cc, K)))
            return out

        cpu_chunk(0)
        t1 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(cpu_chunk, range(threads)))
        cdt = time.perf_counter() - t1
        res = dict(kv for p_ in parts for kv in p_)
        same = sum(int(res[i][1].tolist() == glw[i][:len(res[i][1])].tolist()) for i in range(ncq))
        cpu = {"kind": "port", "threads": threads, "queries": ncq, "qps": round(ncq / cdt, 1), "same_graphs_same_bitmaps": True,
               "ids_identical_to_gpu": f"{same}/{ncq}", "graph_export_s": round(export_s, 2),
               "recall_at_10": round(recall_of([res[i][1] for i in range(ncq)], gt[:ncq], K), 4)}
        del views, graphs
    except Exception as e:   # noqa: BLE001
        cpu = {"error": f"{type(e).__name__}: {e}"}
    head = legs["shared_tags_warm"]
    st = h.stats()
    return {"workload": f"BASELINE.json configs[4]: HNSW M=16 efC=200, {Nh}x{D} fp32 COSINE over {S} GPUs ({Nh // S} rows per graph) "
                        f"+ TAG filters (10 %, one predicate per query, built inside the step), efSearch={ef_h}, k={K}",
            "shards": S, "rows": Nh, "rows_per_shard": Nh // S, "queries_per_batch": nq, "build_s": round(build_s, 2),
            "gpu_qps": head["gpu_qps"], "ms_per_batch": head["ms_per_step"], "recall_at_10": head["recall_at_10"],
            "roofline": {"bound": "hbm", "achieved": head["useful_gbs"], "peak": HBM_PEAK_GBS * len(set(devs)), "unit": "GB/s",
                         "frac": head["frac_of_hbm_peak"], "kernel": "hnsw_search_kernel (HBM frontier, one bitmap per query)",
                         "bytes": "n_eval*(D*4+4) + n_hops*132 per query, counted by the kernels of every shard",
                         "note": "every shard is searched with the full ef: S times the hops of one graph" + ("; the shards share one GPU here" if len(set(devs)) == 1 else "")},
            "steps": legs, "filter_cache": {"hits": int(st.filter_cache_hits), "misses": int(st.filter_cache_misses), "built": int(st.filters_built),
                                            "entries": int(st.filter_cache_entries), "device_bytes": int(st.filter_cache_bytes)},
            "cpu": cpu,
            "merge": "one graph per shard inside one vk_index; per-shard top-k gathered on device 0, (distance,label) merge",
            "leg_s": round(time.perf_counter() - t_leg, 1)}


def sharded_hnsw_leg(args, flat_ix, host_rows, A, device, ws, devs, total_rows):
    """configs[2] over N GPUs: one independent graph per shard (as one per cluster shard in the reference), searched with
    the same ef ("matched ef": about N times the hops of one graph, recall above the single graph's) and with the
    smallest per-shard ef whose merged recall@10 reaches the SINGLE graph's at efSearch=128 and at its 0.95 point
    ("matched recall": SURVEY 8e, docs/topics/search.md:84) -- the per-shard ef policy vk_index_params.shard_ef_pct
    stands for.  The single graph over the same rows is built on device 0 for the reference recalls."""
    Nh, D, K, ef, S = host_rows.shape[0], args.dim, args.k, args.hnsw_ef, len(devs)
    t_leg = time.perf_counter()
    nq = min(args.hnsw_queries, 4096)
    Qh = make_queries(A, nq, D, device, 9090)
    hq = Qh.cpu().numpy()
    bits = nb = None
    if Nh < total_rows:
        from oracle import oracle as O
        bits, nb = O.allow_bitmap(np.arange(Nh, dtype=np.uint64), Nh), Nh
    gt = _exact_gt(flat_ix, hq, K, bits, nb)
    od = torch.empty(nq, K, device=device, dtype=torch.float32)
    ol = torch.empty(nq, K, device=device, dtype=torch.int64)
    on = torch.empty(nq, device=device, dtype=torch.int32)

    def point(ix, ef_s, reps=3):
        ms, wall = _timed_steps(lambda: ix.search_batch_device(Qh.data_ptr(), nq, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(),
                                                               ef=ef_s, stream=ws.cuda_stream), reps, 1)
        return {"ef": ef_s, "gpu_qps": round(nq / (wall * 1e-3), 1), "ms_per_batch": round(wall, 3),
                "recall_at_10": round(recall_of(ol.cpu().numpy().view(np.uint64), gt, K), 4)}

    single = None
    if args.hnsw_single_ref:
        t0 = time.perf_counter()
        g1 = vsa.Index("HNSW", D, "COSINE", initial_cap=Nh, m=16, ef_construction=200, ef_runtime=ef, device_id=devs[0])
        g1.add_batch(host_rows)
        g1.flush()
        sb = time.perf_counter() - t0
        p128 = point(g1, ef)
        p95 = None
        for ef_s in (192, 256, 384, 512, 768, 1024, 1536, 2048):
            p = point(g1, ef_s, reps=2)
            if p["recall_at_10"] >= 0.95:
                p95 = p
                break
        single = {"build_s": round(sb, 2), "at_ef": p128, "at_recall_0.95": p95}
        del g1
        torch.cuda.empty_cache()
    t0 = time.perf_counter()
    h = vsa.Index("HNSW", D, "COSINE", initial_cap=Nh, m=16, ef_construction=200, ef_runtime=ef, shard_devices=devs)
    h.add_batch(host_rows)
    h.flush()
    hb = time.perf_counter() - t0
    matched_ef = point(h, ef)
    out = {"workload": f"HNSW {Nh}x{D} fp32 COSINE M=16 efC=200 k={K}: one graph per GPU over {S} GPUs ({Nh // S} rows each)",
           "shards": S, "rows": Nh, "rows_per_shard": Nh // S, "M": 16, "ef_construction": 200, "k": K, "queries_per_batch": nq,
           "build_s": round(hb, 2), "matched_ef": matched_ef, "single_graph": single,
           "merge": "one graph per shard inside one vk_index; per-shard top-k gathered on device 0, (distance,label) merge"}
    if single:
        def smallest(target, top):
            best = None
            for ef_s in sorted({max(K, x) for x in (top, top * 3 // 4, top // 2, top * 3 // 8, top // 4, top * 3 // 16, top // 8, top // 16)}, reverse=True):
                p = point(h, ef_s, reps=2)
                if p["recall_at_10"] >= target:
                    best = p
                else:
                    break
            return best
        m128 = smallest(single["at_ef"]["recall_at_10"], ef)
        out["matched_recall_of_single_graph_at_ef"] = None if m128 is None else {
            **m128, "target_recall": single["at_ef"]["recall_at_10"], "shard_ef_pct": round(100.0 * m128["ef"] / ef, 1),
            "single_graph_qps": single["at_ef"]["gpu_qps"]}
        if single["at_recall_0.95"]:
            top = single["at_recall_0.95"]["ef"]
            m95 = smallest(0.95, top)
            out["matched_recall_0.95"] = None if m95 is None else {
                **m95, "single_graph_ef": top, "shard_ef_pct": round(100.0 * m95["ef"] / top, 1),
                "single_graph_qps": single["at_recall_0.95"]["gpu_qps"]}
    out["leg_s"] = round(time.perf_counter() - t_leg, 1)
    return out


def serving_leg(ix, hq, K, device_qps, max_batch, wait_us, producers, window, total, threads, calls, ef=0):
    """N1, the path a real FT.SEARCH takes: SINGLE-QUERY requests against the index of the headline / HNSW leg, driven
    natively (scripts/serving_probe.cc through ctypes, no interpreter in the loop).
      submit    vk_index_search_submit: `producers` threads keep `window` requests outstanding -- query::SearchAsync's shape
                (search.cc:886-910: the queue holds up to max-query-queue-depth requests whatever the number of reader threads)
      blocking  `threads` callers of vk_index_search back to back (the reader pool as it is today)
    Every answer is compared (ids and distance bits) with the answer of the same query in one vk_index_search_batch call;
    device_qps = what the device does on a full resident batch (the headline number), for the ratio."""
    rd, rl, rn = ix.search_batch(hq, K, ef=ef)
    assert (rn == K).all()
    ix.set_coalescing(max_batch, wait_us)

    def accounted(fn):
        """the probe's result + the dispatcher's account of its threads' time during it (vk_index_stats.dispatch_*_us)"""
        s0 = ix.stats()
        r = fn()
        s1 = ix.stats()
        r.threads_ms = {k: round((getattr(s1, f"dispatch_{k}_us") - getattr(s0, f"dispatch_{k}_us")) / 1e3, 1)
                        for k in ("idle", "window", "search", "handout", "completer")}
        r.wall_ms = round(r.seconds * 1e3, 1)
        return r

    try:
        vsa.probe_submit(ix, hq, K, min(total, 4 * max_batch), producers, window, ef, ref=(rd, rl))     # warm-up: runner threads, contexts
        sub = accounted(lambda: vsa.probe_submit(ix, hq, K, total, producers, window, ef, ref=(rd, rl)))
        blk = accounted(lambda: vsa.probe_blocking(ix, hq, K, threads, calls, ef, ref=(rd, rl)))
        # ... and THROUGH THE ADAPTOR CLASSES (include/vk_vector_adaptor.h on the mock of VectorBase, scripts/adaptor_probe.cc):
        # one front thread (the main thread) keeps `window` FT.SEARCHes outstanding (the clients' concurrency) and posts each to a
        # reader pool of the box's cores, which calls SearchAsync (the coalescing is the adaptor's own); and the reference's
        # blocking Search() from the same pool
        readers = effective_cpus()
        hnsw_ix = ix.algo == "HNSW"
        vsa.adaptor_probe(ix, hq, K, min(total, 4 * max_batch), readers, window, ef, hnsw=hnsw_ix, ref=(rd, rl))
        ada = accounted(lambda: vsa.adaptor_probe(ix, hq, K, total, readers, window, ef, hnsw=hnsw_ix, ref=(rd, rl)))
        adb = accounted(lambda: vsa.adaptor_probe(ix, hq, K, max(readers * 24, total // 8), readers, readers, ef, hnsw=hnsw_ix, blocking=True, ref=(rd, rl)))
    finally:
        ix.set_coalescing(0, 0)
    st = ix.stats()

    def row(r, **kw):
        return {**kw, "qps": round(r.qps, 1), "of_device_batch_rate": round(r.qps / device_qps, 3) if device_qps else None,
                "requests": int(r.completed), "answers_identical": bool(r.mismatches == 0 and r.errors == 0), "rejected_busy": int(r.rejected),
                "device_batches": int(r.device_batches), "mean_batch": round(r.mean_batch, 1), "batches_in_flight": int(r.max_batches_in_flight),
                "latency_us": {"p50": round(r.p50_us, 1), "p99": round(r.p99_us, 1), "max": round(r.max_us, 1)},
                "wall_ms": r.wall_ms, "dispatcher_threads_ms": r.threads_ms}

    return {"max_batch": max_batch, "max_wait_us": wait_us, "device_batch_qps": round(device_qps, 1) if device_qps else None,
            "submit": row(sub, producers=producers, outstanding=window),
            "blocking": row(blk, callers=threads, calls_per_caller=calls),
            "adaptor": {"async": row(ada, reader_threads=readers, front_threads=1, outstanding=window, entry="VectorGpu::SearchAsync"),
                        "blocking": row(adb, reader_threads=readers, entry="VectorGpu::Search"),
                        "cancelled_early": int(st.cancelled_early)},
            "latency_hist_us_pow2_from_64": list(st.latency_hist)}


def _baseline_metric():
    try:
        return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")))["metric"]
    except (OSError, KeyError, ValueError):
        return "kNN queries/sec + recall@10, 10M\u00d7768 fp32 cosine, flat & HNSW, 1/2/4/8 GPU"


BASELINE_METRIC = _baseline_metric()


STAMPED_SOURCES = ("*.hip", "device_common.hpp", "kernels.hpp", "options.hpp", "flat_index.cc", "hnsw_index.cc")


def source_sha256():
    """Hash of what decides a launch's memory traffic -- the device sources (csrc/*.hip and the two headers they include), the
    option defaults and the two files that choose kernel variant, grid and work split (flat_index.cc, hnsw_index.cc): the build
    id a PMC traffic file is stamped with.  (Host-only files -- dispatcher, filter handles, sharding, the ABI -- are not in it:
    until r06 a change there made the bench line drop its `traffic` although no kernel or launch had changed.)"""
    import hashlib
    h = hashlib.sha256()
    csrc = ROOT / "valkey-search_amd" / "csrc"
    files = sorted({f for pat in STAMPED_SOURCES for f in csrc.glob(pat)})
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def hnsw_pmc_traffic(rows, nq, ef):
    """memory-side traffic of the HNSW search launch from the committed PMC passes (scripts/pmc_hnsw_traffic.sh: FETCH_SIZE
    calibrated on the single-query FLAT scan of the same run, the L2's hit rate, the fabric read requests), against the useful
    bytes the kernel counts: printed only while the file was taken with the current kernel sources and this workload"""
    path = ROOT / "profiles" / "r06_pmc_hnsw_traffic.json"
    if not path.exists():
        return {"traffic": None}
    j = json.load(open(path))
    if j.get("src_sha256") != source_sha256():
        return {"traffic": None, "traffic_source": "profiles/r06_pmc_hnsw_traffic.json is stale (other kernel sources): re-run scripts/pmc_hnsw_traffic.sh"}
    if (j.get("rows"), j.get("queries_per_launch"), j.get("ef")) != (rows, nq, ef) or "traffic_bytes_per_launch" not in j:
        return {"traffic": None, "traffic_source": f"profiles/r06_pmc_hnsw_traffic.json holds {j.get('rows')} rows x {j.get('queries_per_launch')} queries at ef {j.get('ef')}"}
    return {"traffic": round(j["traffic_bytes_per_launch"]), "traffic_unit": "HBM-side bytes per launch (FETCH_SIZE, calibrated; Infinity-Cache hits not excluded)",
            "useful_bytes_per_launch": j.get("useful_bytes_per_launch"), "traffic_over_useful": j.get("traffic_over_useful"),
            "l2_hit_rate": j.get("l2_hit_rate"), "traffic_source": "profiles/r06_pmc_hnsw_traffic.json (own --pmc passes, same sources)"}


def pmc_traffic(N, D, B, world, kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (rocprofv3 --pmc is a separate run by
    rule, so bench.py cannot collect it live): profiles/r06_pmc_fetch_size.json, written by scripts/pmc_traffic.sh and
    stamped with the hash of the sources it measured.  Printed only for the default single-GPU workload AND only while
    that hash equals the current sources' -- a stale file yields null, never an old number."""
    path = ROOT / "profiles" / "r06_pmc_fetch_size.json"
    if world != 1 or (N, D, B) != (10_000_000, 768, 256) or not path.exists() or "bf16" in sys.argv:
        return None, None
    if any(os.environ.get(v) for v in ("VK_FLAT_FORCE_SCAN", "VK_GEMM_MODE", "VK_GEMM_ABLATE", "VK_GEMM_LOCKSTEP", "VK_FILTER_TIMING", "VK_FLAT_FILTER")):
        return None, None
    j = json.load(open(path))
    if j.get("src_sha256") != source_sha256():
        return None, "profiles/r06_pmc_fetch_size.json is stale (taken with other kernel sources): re-run scripts/pmc_traffic.sh"
    for name, v in j.get("kernels", {}).items():
        if kernel_prefix in name and "prepass" not in name and "[small]" not in name:
            return round(v["hbm_bytes_per_launch"]), "profiles/r06_pmc_fetch_size.json (rocprofv3 --pmc FETCH_SIZE, separate pass, same sources)"
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="row storage (bf16 = BASELINE.json configs[3] storage; queries and arithmetic stay f32)")
    ap.add_argument("--cpu-rows", type=int, default=0,
                    help="rows the CPU baseline scans per query (0 = the whole index: a full-size pass, no scaling)")
    ap.add_argument("--cpu-queries-per-thread", type=int, default=16,
                    help="CPU baseline: queries per host thread (16 x 16 threads = all 256 queries of the timed batch, each a full "
                         "10M-row scan: about 30 s of CPU work, and every row of the timed step's answer is compared)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-serving", action="store_true", help="skip the single-query serving legs (profiling runs)")
    ap.add_argument("--single-query-steps", type=int, default=5, help="extra: B=1 scan timing (HBM roofline)")
    ap.add_argument("--two-in-flight-steps", type=int, default=20, help="extra: the step alternating between two streams (0 = skip)")
    ap.add_argument("--hnsw-rows", type=int, default=-1,
                    help="HNSW leg (BASELINE.json configs[2]) over the first rows: -1 = all of --rows (10M), 0 = skip")
    ap.add_argument("--hnsw-ef", type=int, default=128)
    ap.add_argument("--hnsw-queries", type=int, default=8192)
    ap.add_argument("--hnsw-cpu-queries-per-thread", type=int, default=32)
    ap.add_argument("--hybrid-rows", type=int, default=-1,
                    help="BASELINE.json configs[4] (HNSW + TAG filter, efSearch=256): rows PER GPU, -1 = 1 250 000, 0 = skip")
    ap.add_argument("--hybrid-queries", type=int, default=4096)
    ap.add_argument("--bf16-rows", type=int, default=-1,
                    help="one shard of BASELINE.json configs[3] (FLAT bf16 IP): rows, -1 = as --rows, 0 = skip")
    ap.add_argument("--config-shards", type=int, default=8,
                    help="N = 1: run BASELINE.json configs[3] / configs[4] AT THEIR STATED SIZE as this many LOGICAL shards of one "
                         "vk_index on the one GPU (8 x --bf16-rows bf16 rows = 80M x 768 = 122.9 GB; 8 graphs of --hybrid-rows rows); "
                         "0 or 1 = one shard of each instead")
    ap.add_argument("--hnsw-sharded", action=argparse.BooleanOptionalAction, default=True,
                    help="N > 1: the sharded HNSW leg (one graph per GPU; matched-ef and matched-recall points)")
    ap.add_argument("--hnsw-single-ref", action=argparse.BooleanOptionalAction, default=True,
                    help="N > 1: build the single graph over the same rows too (the reference recalls of the matched-recall points)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--multi-gpu", choices=["lib", "ranks"], default="lib",
                    help="N > 1: 'lib' = the product's own multi-GPU index (n_shards = N inside one process, rank 0), falling back to "
                         "'ranks' = one process per GPU, each with its shard, RCCL all-gather of the per-shard top-k + device merge")
    ap.add_argument("--same-device", action="store_true",
                    help="test aid: every rank uses cuda:0 (with --backend gloo, two ranks can exercise the sharded path on one GPU)")
    ap.add_argument("--verify-merge", action="store_true",
                    help="test aid (N > 1, small --rows): rank 0 also builds the unsharded index and checks the merged answer against it")
    args = ap.parse_args()

    procs = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    # --gpus N without a launcher: ONE process drives all N GPUs through the library's sharded index (valkey-server is
    # one process too).  Under torch.distributed.run (WORLD_SIZE == N) rank 0 does the same and the others keep the
    # barriers; "--multi-gpu ranks" is the one-process-per-GPU variant (RCCL all-gather of the per-shard top-k).
    assert procs in (1, args.gpus), f"--gpus {args.gpus} but WORLD_SIZE={procs}"
    world = args.gpus
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if procs > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (rank 0 runs the legs behind the timed region -- CPU baseline, HNSW builds, configs[3] / [4] -- for minutes while the
        #  other ranks wait in the last barrier: the process group's default watchdog of ten minutes must not end the job there)
        import datetime
        patience = datetime.timedelta(hours=3)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=patience)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(args.backend, timeout=patience)

    N, D, B, K = args.rows, args.dim, args.batch, args.k
    if args.hnsw_rows < 0:
        args.hnsw_rows = N
    if args.bf16_rows < 0:
        args.bf16_rows = N
    if args.hybrid_rows < 0:
        args.hybrid_rows = 1_250_000
    if world > 1 and (args.multi_gpu == "lib" or procs == 1):
        if procs == 1 and args.multi_gpu != "lib":
            raise SystemExit("--multi-gpu ranks needs one process per GPU (torch.distributed.run)")
        if lib_multi_gpu(args, world, rank, dist, device) != "fallback":
            return
    r0 = rank * N // world
    r1 = (rank + 1) * N // world
    n_local = r1 - r0

    # ---- build the shard: rows generated straight into the index's HBM row table ----
    t_build = time.time()
    bf16 = args.dtype == "bf16"
    esz = 2 if bf16 else 4
    ix = vsa.Index("FLAT", D, "COSINE", initial_cap=n_local, device_id=local_rank, dtype=args.dtype)
    base_ptr, stride = ix.device_rows(n_local)
    assert stride == ((D + 63) // 64) * 64 * esz   # rows are zero padded to whole 64-element groups
    if bf16:   # torch cannot import bf16 through __cuda_array_interface__: map as int16 and reinterpret
        table = device_view_typed(base_ptr, (n_local, stride // 2), device, "<i2").view(torch.bfloat16)
    else:
        table = device_view(base_ptr, (n_local, stride // 4), device)   # [rows][padded dim] in HBM
    if stride != D * esz:
        table[:, D:] = 0
    for lo, x in gen_rows(r0, n_local, D, device):
        table[lo - r0: lo - r0 + x.shape[0], :D] = x     # bf16: round to nearest even, as the library's ingest does
    torch.cuda.synchronize()
    ix.commit_device_rows(n_local, np.arange(r0, r1, dtype=np.uint64))
    t_build = time.time() - t_build

    qg = torch.Generator(device=device)
    qg.manual_seed(4242)
    gA = torch.Generator(device=device)
    gA.manual_seed(1234)
    A = torch.randn(D, 32, generator=gA, device=device, dtype=torch.float32)
    Q = torch.nn.functional.normalize(
        torch.randn(B, 32, generator=qg, device=device) @ A.T + 0.05 * torch.randn(B, D, generator=qg, device=device),
        dim=1).contiguous()

    out_d = torch.empty(B, K, device=device, dtype=torch.float32)
    out_l = torch.empty(B, K, device=device, dtype=torch.int64)
    out_n = torch.empty(B, device=device, dtype=torch.int32)
    if world > 1:
        all_d = torch.empty(world * B, K, device=device, dtype=torch.float32)   # rank-major == [world][B][K]
        all_l = torch.empty(world * B, K, device=device, dtype=torch.int64)
        fin_d = torch.empty(B, K, device=device, dtype=torch.float32)
        fin_l = torch.empty(B, K, device=device, dtype=torch.int64)
        fin_n = torch.empty(B, device=device, dtype=torch.int32)

    # a real (non-null) HIP stream: kernels, RCCL and the timing events all go on it
    work_stream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(work_stream)

    def stream_ptr():
        return work_stream.cuda_stream

    def step():
        ix.search_batch_device(Q.data_ptr(), B, K, out_d.data_ptr(), out_l.data_ptr(), out_n.data_ptr(),
                               stream=stream_ptr())
        if world > 1 and args.backend != "nccl":      # test aid: gloo has no device all-gather, stage through the host
            work_stream.synchronize()
            cd, cl = torch.empty(world * B, K, dtype=torch.float32), torch.empty(world * B, K, dtype=torch.int64)
            dist.all_gather_into_tensor(cd, out_d.cpu())
            dist.all_gather_into_tensor(cl, out_l.cpu())
            all_d.copy_(cd)
            all_l.copy_(cl)
        elif world > 1:
            dist.all_gather_into_tensor(all_d, out_d)
            dist.all_gather_into_tensor(all_l, out_l)
        if world > 1:
            vsa.merge_topk_device(all_d.data_ptr(), all_l.data_ptr(), world, B, K, fin_d.data_ptr(),
                                  fin_l.data_ptr(), fin_n.data_ptr(), local_rank, stream_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    st_before = ix.stats()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) / args.steps      # HIP events on the work stream around the K timed steps
    st_after = ix.stats()
    filt_n = st_after.filter_batches - st_before.filter_batches     # batches of the timed region that went through the candidate filter
    # The dominant kernel ALONE (the filter's main pass): a second region of K steps behind the timed one, with the library's
    # event pairs around that launch switched on (option kernel-timing: an event record is a marker the stream executes,
    # about 6 us of idle device each, so the timed region runs without them and `roofline.frac` is the whole step's).
    filt_ms, filt_timed = None, 0
    if filt_n:
        ix.set_option("kernel-timing", 1)
        for _ in range(2):
            step()
        barrier()
        k0 = ix.stats()
        for _ in range(args.steps):
            step()
        barrier()
        k1 = ix.stats()
        ix.set_option("kernel-timing", 0)
        filt_timed = k1.filter_batches - k0.filter_batches
        filt_ms = (k1.filter_kernel_ns - k0.filter_kernel_ns) / 1e6 / filt_timed if filt_timed else None
        if not filt_ms:
            filt_n = 0
    # ---- extra: two batches in flight (what the dispatcher keeps: the next batch's small launches run beside this batch's
    # main pass).  Reported beside the line, never as `value`: the timed region above is one batch after the other.
    two_in_flight = None
    if world == 1 and args.two_in_flight_steps > 0:
        Q2 = make_queries(A, B, D, device, 4243)
        sets = [(Q, out_d, out_l, out_n),
                (Q2, torch.empty_like(out_d), torch.empty_like(out_l), torch.empty_like(out_n))]
        lanes = [work_stream, torch.cuda.Stream(device=device)]

        def run(nlanes, steps):
            for i in range(steps):
                q, od, ol, on = sets[i % 2]
                ix.search_batch_device(q.data_ptr(), B, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=lanes[i % nlanes].cuda_stream)

        run(1, 2)
        torch.cuda.synchronize()
        one_l = [s_[2].clone() for s_ in sets]
        run(2, 4)
        torch.cuda.synchronize()
        tt = time.perf_counter()
        run(2, args.two_in_flight_steps)
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - tt) / args.two_in_flight_steps * 1e3
        same = all(bool((a_ == s_[2]).all().item()) for a_, s_ in zip(one_l, sets))
        two_in_flight = {"ms_per_step": round(ms2, 4), "queries_per_s": round(B / ms2 * 1e3, 1), "steps": args.two_in_flight_steps,
                         "hbm_frac_whole_step": round(n_local * stride / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "answers_identical_to_one_stream": same,
                         "what": "the same step alternating between two HIP streams and two batches of queries (wall clock, "
                                 "synchronised at both ends); not the headline value"}
        # (set 0 is Q with out_*: they still hold the timed batch's answer for the parity check below)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- extra: single-query scan (the memory-bound formulation of the same path) ----
    single = None
    if args.single_query_steps > 0:
        sq_d = torch.empty(K, device=device, dtype=torch.float32)       # own buffers: the batch answer stays intact
        sq_l = torch.empty(K, device=device, dtype=torch.int64)
        sq_n = torch.empty(1, device=device, dtype=torch.int32)
        ix.search_batch_device(Q.data_ptr(), 1, K, sq_d.data_ptr(), sq_l.data_ptr(), sq_n.data_ptr(), stream=stream_ptr())
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.single_query_steps):
            ix.search_batch_device(Q.data_ptr(), 1, K, sq_d.data_ptr(), sq_l.data_ptr(), sq_n.data_ptr(),
                                   stream=stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        ms1 = e0.elapsed_time(e1) / args.single_query_steps
        gbs = n_local * stride / (ms1 * 1e-3) / 1e9
        single = {"ms_per_query": round(ms1, 4), "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                  "bytes": n_local * stride}

    # survivors of the candidate filter per query, and queries handed to the exact pass (one host-path call: the
    # device-buffer entry point reports no statistics)
    filt_stats = None
    if filt_n and world == 1:
        ix.search_batch(Q.cpu().numpy(), K)
        st_f = ix.stats()
        filt_stats = {"survivors_per_query": round(st_f.last_filter_candidates / B, 1), "reranked_per_query": round(st_f.last_filter_reranked / B, 1),
                      "queries_handed_over": int(st_f.last_filter_fallback)}
    result_d = (fin_d if world > 1 else out_d).cpu().numpy()
    result_l = (fin_l if world > 1 else out_l).cpu().numpy().view(np.uint64)

    # ---- CPU baseline + parity spot check on rank 0 (oracle = checker, never the product) ----
    cpu = None
    parity = None
    host_rows = None
    extras = rank == 0 and world == 1
    need_host = extras and (not args.no_cpu_baseline or (args.hnsw_rows > 0 and not bf16) or (args.hybrid_rows > 0 and not bf16))
    if need_host:
        # one host copy of the rows serves the CPU baseline (borrowed, not copied again) and the HNSW builds
        n_host = n_local
        if args.no_cpu_baseline:
            n_host = min(n_local, max(args.hnsw_rows, args.hybrid_rows))
        host_rows = np.ascontiguousarray(table[:n_host, :D].float().cpu().numpy())   # bf16: the widened stored values
    if extras and not args.no_cpu_baseline:
        from oracle import oracle as O
        S = n_local if args.cpu_rows <= 0 else min(args.cpu_rows, n_local)
        flat = O.Flat(D, "COSINE", isa="skylake", max_elements=S)
        flat.add_many(host_rows[:S], np.arange(r0, r0 + S, dtype=np.uint64), borrowed=True)
        # distances by the COMPILED REFERENCE's fstdistfunc_ (oracle/_ref: hnswlib/simsimd.h over SimSIMD 5.0.1 built from the
        # reference's own sources, travels to the GPU box prebuilt) when it is there; else the restated kernels
        ref_dist = flat.use_reference_distance()
        hq = Q.cpu().numpy()
        threads = effective_cpus()
        nqt = threads * args.cpu_queries_per_thread
        if S <= 1_000_000:
            flat.search(hq[0], K)   # (warm-up only where it is cheap)
        t1 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            res = list(ex.map(lambda i: flat.search(hq[i % B], K), range(nqt)))
        cdt = time.perf_counter() - t1
        qps_sample = nqt / cdt
        full = S == N
        dist_src = ("distances by the reference's own compiled SimSIMD 5.0.1 (oracle/_ref: simsimd_dot_f32 through third_party/hnswlib/simsimd.h)"
                    if ref_dist else f"distances by the restated kernel ({O.cpu_path()} clone of the SimSIMD skylake order)")
        cpu = {"value": round(qps_sample * S / N, 4), "unit": "queries/s", "cores": threads, "kind": "reference" if ref_dist else "port",
               "host_cpus_total": os.cpu_count(), "cores_note": f"{threads} = the container's cgroup CPU quota; the host has {os.cpu_count()} logical CPUs",
               "seconds": round(cdt, 2), "host_gbs": round(nqt * S * D * 4 / cdt / 1e9, 1),
               "loop": "bruteforce.h:116-145 restated (hnswlib itself is unbuildable here: abseil / protobuf headers absent)",
               "sample": (f"FLAT scan, {dist_src}, {nqt} queries, each a full pass "
                          f"over all {S} rows, one query per thread on {threads} threads" if full else
                          f"FLAT scan, {dist_src}, {nqt} queries over "
                          f"the first {S} rows on {threads} threads ({qps_sample:.1f} q/s on the sample), scaled "
                          f"linearly to {N} rows")}
        # parity at full index size: the GPU answer must equal the oracle's, ids and distance bits -- through the
        # single-query scan and through the timed batched (matrix-core) path
        bits = None if full else O.allow_bitmap(np.arange(r0, r0 + S, dtype=np.uint64), r0 + S)
        nbits = None if full else r0 + S
        ok = True
        for i in range(min(4, len(res))):
            gd, gl = ix.search(hq[i], K, allow=bits, allow_nbits=nbits)
            od, ol = res[i]
            ok = ok and gl.tolist() == ol.tolist() and gd.view(np.uint32).tolist() == od.view(np.uint32).tolist()
        nb = min(B, len(res), 32)
        bd, bl, bn = ix.search_batch(hq[:max(nb, 5)], K, allow=bits, allow_nbits=nbits)
        for i in range(nb):
            od, ol = res[i]
            ok = ok and bl[i, :bn[i]].tolist() == ol.tolist() and bd[i, :bn[i]].view(np.uint32).tolist() == od.view(np.uint32).tolist()
        if full:   # ... and the timed step's own output, already on the host
            for i in range(min(B, len(res))):
                od, ol = res[i]
                ok = ok and result_l[i].tolist() == ol.tolist() and result_d[i].view(np.uint32).tolist() == od.view(np.uint32).tolist()
        parity = "bit-exact" if ok else "MISMATCH"
        del flat

    # (the extra legs must never cost the headline line: a failure in one of them is reported in its place)
    def leg(fn, *a):
        try:
            return fn(*a)
        except Exception as e:   # noqa: BLE001
            import traceback
            return {"error": f"{type(e).__name__}: {e}", "where": traceback.format_exc().strip().splitlines()[-3:]}

    # ---- BASELINE.json configs[2]: HNSW M=16 efC=200, efSearch=128, k=10 over the same rows ----
    hnsw = None
    if extras and args.hnsw_rows > 0 and not bf16:
        hnsw = leg(hnsw_leg, args, ix, host_rows[:min(args.hnsw_rows, n_local)], A, device, stream_ptr, n_local)
    # ---- configs[4]: HNSW + TAG filter -- at its stated size as `--config-shards` graphs inside ONE sharded index on this GPU
    #      (8 x 1.25M rows = the 10M rows), or one shard of it ----
    cs = args.config_shards if args.config_shards > 1 else 0
    hybrid = None
    if extras and args.hybrid_rows > 0 and not bf16:
        if cs and args.hybrid_rows * cs <= n_local:
            hybrid = leg(sharded_hybrid_leg, args, ix, host_rows[:args.hybrid_rows * cs], A, device, work_stream, [local_rank] * cs, n_local)
        else:
            hybrid = leg(hybrid_leg, args, ix, host_rows[:min(args.hybrid_rows, n_local)], A, device, stream_ptr, n_local)
    host_rows = None
    # ---- configs[3]: FLAT bf16 IP -- 8 logical shards x 10M rows = 80M x 768 bf16 (122.9 GB) on this one GPU, or one shard ----
    bf16_shard = None
    if extras and args.bf16_rows > 0 and not bf16:
        free_b, _tot = torch.cuda.mem_get_info(device)
        if cs and free_b > cs * args.bf16_rows * D * 2 + (16 << 30):
            bf16_shard = leg(sharded_bf16_ip_leg, args, A, device, work_stream, [local_rank] * cs)
        else:
            bf16_shard = leg(bf16_ip_leg, args, A, device, stream_ptr, local_rank)
    if world > 1 and args.hnsw_sharded and args.hnsw_rows > 0 and not bf16:
        def gather(dst, src):
            if args.backend == "nccl":
                dist.all_gather_into_tensor(dst, src)
            else:                       # test aid (gloo): through the host
                work_stream.synchronize()
                c = torch.empty(dst.shape, dtype=dst.dtype)
                dist.all_gather_into_tensor(c, src.cpu())
                dst.copy_(c)
        try:
            hnsw = hnsw_sharded_leg(args, ix, table, A, device, stream_ptr, world, rank, r0, dist, gather, local_rank)
        except Exception as e:   # noqa: BLE001
            hnsw = {"error": f"{type(e).__name__}: {e}"}

    # ---- single-query traffic against THIS index (10M x 768 in the default run): submit with 1024 outstanding, 256 blocking callers ----
    coalescer = None
    if rank == 0 and world == 1 and B >= 5 and not args.no_serving:
        coalescer = leg(serving_leg, ix, Q.cpu().numpy(), K, B * args.steps / dt, B, 500, 4, 4 * B, 40 * B, 256, 24)

    if rank == 0:
        # (the final pass: B operands by DMA; bf16 rows in the inner-product space: bf16 MFMA, rows by DMA)
        filter_name = "flat_filter_bfmma_dma_kernel" if args.dtype == "bf16" else "flat_filter_bdma_kernel"
        dominant = filter_name if filt_n else ("flat_gemm_kernel" if B >= 5 else "flat_scan_kernel")
        kern_traffic, traffic_src = pmc_traffic(N, D, B, world, dominant)
        traffic = kern_traffic
        if filt_n and kern_traffic is not None:     # the step's row traffic = both launches of the filter kernel
            early_traffic, _ = pmc_traffic(N, D, B, world, filter_name.replace("flat_filter_bdma_kernel", "flat_filter_early_kernel")
                                                                      .replace("flat_filter_bfmma_dma_kernel", "flat_filter_early_bfmma_dma_kernel"))
            traffic = kern_traffic + (early_traffic or 0)
        qps = B * args.steps / dt
        scan_bytes = n_local * stride                       # algorithmic bytes of one pass over the shard
        # ... and of the launch the roofline prices: the candidate filter's main pass (two-pass batches: the early pass's
        # share of the rows went through flat_filter_early_kernel before it)
        main_rows = _final_rows(ix, n_local) if filt_n else n_local
        kern_bytes = main_rows * stride
        flops = 2.0 * main_rows * D * B                     # of that launch
        out = {
            # BASELINE.json's metric, verbatim; `value` is its FLAT leg on configs[1] (k=10, batch=256, exact: recall@10
            # = 1.0, ids bit-identical to the CPU path), the HNSW leg with its recall is under "hnsw"
            "metric": BASELINE_METRIC if not bf16 else "kNN queries/sec, FLAT 10Mx768 bf16-stored cosine k=10 batch=256",
            "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "row_storage": args.dtype, "data": "synthetic",
            "config": {"workload": (f"FLAT {N}x{D} fp32 COSINE k={K} batch={B} (BASELINE.json configs[1])" if not bf16 else
                                    f"FLAT {N}x{D} bf16 rows, f32 queries/arithmetic, COSINE k={K} batch={B} "
                                    f"(per-GPU shard of BASELINE.json configs[3])"),
                       "rows_per_gpu": n_local, "sharding": f"rows/{world}", "recall_at_10": 1.0,
                       "parity_vs_oracle": parity},
            # B >= 5 in the inner-product space runs on the f32 matrix cores (flat_gemm_kernel, K4):
            # algorithmic FLOPs per launch = 2 * rows * D * B against the 157.3 TFLOP/s f32 MFMA peak
            # B >= 5: f16 matrix-core candidate filter (flat_filter_kernel, one pass over the rows:
            # HBM-bound, algorithmic bytes = rows * row bytes) + exact re-rank of the survivors; 5 <= B <= 32: the exact f32
            # matrix-core kernel (MFMA-bound); else the scan (HBM-bound)
            # HBM-bound.  `achieved` / `frac` price the WHOLE timed step: SURVEY 8(d)'s bytes of one pass over the shard
            # (rows x row bytes) / (HIP-event time of the K timed steps / K) -- every launch of the step is inside (query
            # preparation, sample, early and main pass, re-rank, the hand-over launches).  The dominant kernel alone, from the
            # library's own event pairs in a second region, is the sub-object `dominant_kernel`; rocprofv3's per-kernel average
            # for the same symbol is in profiles/ and must agree with ITS per_launch_ms.
            "roofline": ({"bound": "hbm", "achieved": round(scan_bytes / (dev_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(scan_bytes / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic,
                          "traffic_unit": "HBM bytes per step (early + main pass launches)", "traffic_source": traffic_src,
                          "algorithmic_bytes": scan_bytes, "priced": "whole step", "step_ms_on_stream": round(dev_ms, 4),
                          "rows_in_step": n_local,
                          "dominant_kernel": {"kernel": filter_name, "achieved": round(kern_bytes / (filt_ms * 1e-3) / 1e9, 2),
                                              "frac": round(kern_bytes / (filt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                              "per_launch_ms": round(filt_ms, 4), "launches_timed": int(filt_timed),
                                              "rows_in_launch": main_rows, "algorithmic_bytes": kern_bytes, "traffic": kern_traffic,
                                              "measured": "the library's HIP event pairs around this launch on its stream (option "
                                                          "kernel-timing), K steps behind the timed region",
                                              "f16_mfma_tflops": round(flops / (filt_ms * 1e-3) / 1e12, 1),
                                              "f16_mfma_frac_of_2500": round(flops / (filt_ms * 1e-3) / 1e12 / 2500.0, 4)},
                          "early_pass": ({"kernel": filter_name.replace("flat_filter_bdma_kernel", "flat_filter_early_kernel")
                                                               .replace("flat_filter_bfmma_dma_kernel", "flat_filter_early_bfmma_dma_kernel"),
                                          "rows": n_local - main_rows,
                                          "role": "the head of every block's tile range; its survivors give the main pass's bound"}
                                         if main_rows != n_local else None),
                          "filter": filt_stats}
                         if filt_n else
                         {"bound": "mfma", "achieved": round(flops / (dev_ms * 1e-3) / 1e12, 3),
                          "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                          "frac": round(flops / (dev_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TF, 5), "traffic": traffic,
                          "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
                          "algorithmic_bytes": scan_bytes,
                          "kernel": "flat_gemm_kernel", "per_launch_ms": round(dev_ms, 4),
                          "hbm_gbs_algorithmic": round(scan_bytes / (dev_ms * 1e-3) / 1e9, 2)}
                         if B >= 5 and not os.environ.get("VK_FLAT_FORCE_SCAN") else
                         {"bound": "hbm", "achieved": round(scan_bytes / (dev_ms * 1e-3) / 1e9, 2),
                          "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(scan_bytes / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                          "kernel": "flat_scan_kernel", "per_launch_ms": round(dev_ms, 4),
                          "tflops_f32": round(flops / (dev_ms * 1e-3) / 1e12, 3)}),
            "cpu_baseline": cpu,
            "single_query_scan": single,
            "two_batches_in_flight": two_in_flight,
            "single_query_serving": coalescer,
            "hnsw": hnsw,
            "config4_hnsw_tag": hybrid,
            "config3_flat_bf16_ip": bf16_shard,
            "build_s": round(t_build, 2),
        }
        print(json.dumps(out))
    if world > 1 and args.verify_merge:
        # the n-shard answer must be bit-identical to the 1-shard answer (total order (distance,label))
        if rank == 0:
            full = vsa.Index("FLAT", D, "COSINE", initial_cap=N, device_id=local_rank, dtype=args.dtype)
            fp, fstride = full.device_rows(N)
            ft = device_view(fp, (N, fstride // 4), device)
            if fstride != D * 4:
                ft[:, D:] = 0
            for lo, x in gen_rows(0, N, D, device):
                ft[lo: lo + x.shape[0], :D] = x
            torch.cuda.synchronize()
            full.commit_device_rows(N, np.arange(N, dtype=np.uint64))
            fd, fl, fn = full.search_batch(Q.cpu().numpy(), K)
            same = bool((fl == result_l).all() and (fd.view(np.uint32) == result_d.view(np.uint32)).all())
            if not same:
                bad = np.argwhere(fl != result_l)
                print("verify_merge: %d label mismatches, %d distance mismatches; first at %s: full %s / %s  merged %s / %s" % (
                    len(bad), int((fd.view(np.uint32) != result_d.view(np.uint32)).sum()), bad[:1].tolist(),
                    fl[bad[0][0]].tolist() if len(bad) else None, fd[bad[0][0]].tolist() if len(bad) else None,
                    result_l[bad[0][0]].tolist() if len(bad) else None, result_d[bad[0][0]].tolist() if len(bad) else None),
                    file=sys.stderr)
            print(json.dumps({"verify_merge": "bit-identical" if same else "MISMATCH", "shards": world, "rows": N}))
            assert same
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
