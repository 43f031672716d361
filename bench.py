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

    def point(ef_s, reps=5):
        ms = timed(lambda: h.search_batch_device(Qh.data_ptr(), nq, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), ef=ef_s,
                                                 stream=stream_ptr()), reps)
        gl = ol.cpu().numpy().view(np.uint64).copy()
        _D, _L, _N = h.search_batch(hq[:1024], K, ef=ef_s)        # host path once: fills the work counters
        st = h.stats()
        useful = (st.last_n_eval * (D * 4 + 4) + st.last_n_hops * 132) / 1024.0 * nq
        gbs = useful / (ms * 1e-3) / 1e9
        return gl, {"ef": ef_s, "gpu_qps": round(nq / (ms * 1e-3), 1), "ms_per_batch": round(ms, 3),
                    "recall_at_10": round(recall_of(gl, gt, K), 4),
                    "n_eval_per_query": round(st.last_n_eval / 1024.0, 1), "n_hops_per_query": round(st.last_n_hops / 1024.0, 1),
                    "useful_gbs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                    "frac_of_gather_ceiling": round(gbs / GATHER_CEILING_GBS, 4)}

    gl, head = point(ef)
    # matched recall: the smallest ef of the sweep whose recall@10 reaches 0.95 (SURVEY.md 8e asks for both points)
    sweep, matched = [head], None
    for ef_s in (192, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096):
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
            "build_s": round(build_s, 2), "build_inserts_per_s": round(Nh / build_s, 1), "build": "device-assisted (K9)",
            **head, "single_query_ms": round(lat_ms, 3),
            "roofline": {"bound": "hbm", "achieved": head["useful_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": head["frac_of_hbm_peak"], "kernel": "hnsw_search_hash_kernel",
                         "bytes": "n_eval*(D*4+4) + n_hops*132 per query, counted by the kernel",
                         "gather_ceiling_gbs": GATHER_CEILING_GBS, "gather_ceiling_source": "profiles/r01_gather_ceiling_10Mx768.log"},
            "matched_recall_0.95": matched, "ef_sweep": sweep,
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
    ix = vsa.Index("FLAT", D, "IP", initial_cap=N, device_id=local_rank, dtype="bf16")
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


def lib_multi_gpu(args, world, rank, local_rank, dist, device):
    """--gpus N through the PRODUCT's multi-GPU path: rank 0 holds ONE vk_index with n_shards = N (shard s on HIP device
    s; vk_index_params.shard_devices), built and searched from this one process like valkey-server would; the other
    ranks only keep the launch contract (barriers, max-over-ranks timing).  A step = one
    vk_index_search_batch_device on the sharded index: queries broadcast by peer copy, every shard scans its rows on its
    own device and stream, per-shard top-k lists gathered on device 0, (distance,label) merge.  Returns the JSON dict
    on rank 0 (None elsewhere), or raises on rank 0 before any collective if the sharded index cannot be set up."""
    N, D, B, K = args.rows, args.dim, args.batch, args.k
    bf16 = args.dtype == "bf16"
    esz = 2 if bf16 else 4
    devs = [0] * world if args.same_device else list(range(world))
    out = None
    state = {}
    ok = torch.ones(1, device=device if args.backend == "nccl" else "cpu")
    if rank == 0:
        try:
            if vsa.lib().vk_device_count() < (1 if args.same_device else world):
                raise RuntimeError(f"rank 0 sees {vsa.lib().vk_device_count()} HIP devices, needs {world}")
            t_build = time.time()
            ix = vsa.Index("FLAT", D, "COSINE", initial_cap=N, dtype=args.dtype, shard_devices=devs)
            for s_i, dv in enumerate(devs):
                r0, r1 = s_i * N // world, (s_i + 1) * N // world
                sdev = torch.device("cuda", dv)
                ptr, stride = ix.shard_device_rows(s_i, r1 - r0)
                if bf16:
                    tab = device_view_typed(ptr, (r1 - r0, stride // 2), sdev, "<i2").view(torch.bfloat16)
                else:
                    tab = device_view(ptr, (r1 - r0, stride // 4), sdev)
                if stride != D * esz:
                    tab[:, D:] = 0
                for lo, x in gen_rows(r0, r1 - r0, D, sdev):
                    tab[lo - r0: lo - r0 + x.shape[0], :D] = x
                torch.cuda.synchronize(sdev)
                ix.shard_commit_device_rows(s_i, r1 - r0, np.arange(r0, r1, dtype=np.uint64))
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
            state.update(ix=ix, A=A, Q=Q, od=od, ol=ol, on=on, ws=ws)

            def step():
                ix.search_batch_device(Q.data_ptr(), B, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=ws.cuda_stream)

            for _ in range(max(1, args.warmup)):
                step()
            torch.cuda.synchronize()
            state["step"] = step
        except Exception as e:   # noqa: BLE001
            ok.zero_()
            state["error"] = f"{type(e).__name__}: {e}"
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) == 0.0:
        if rank == 0:
            print(f"bench: library multi-GPU path unavailable ({state.get('error')}); falling back to one process per GPU",
                  file=sys.stderr)
        state.clear()
        torch.cuda.empty_cache()
        return "fallback"

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    if rank == 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            state["step"]()
        e1.record()
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank == 0:
        ix, Q, od, ol = state["ix"], state["Q"], state["od"], state["ol"]
        dev_ms = e0.elapsed_time(e1) / args.steps
        res_d, res_l = od.cpu().numpy(), ol.cpu().numpy().view(np.uint64)
        verify = None
        if args.verify_merge:   # the N-shard answer must be bit-identical to the 1-shard answer
            full = vsa.Index("FLAT", D, "COSINE", initial_cap=N, device_id=0, dtype=args.dtype)
            fp, fstride = full.device_rows(N)
            ft = (device_view_typed(fp, (N, fstride // 2), device, "<i2").view(torch.bfloat16) if bf16
                  else device_view(fp, (N, fstride // 4), device))
            if fstride != D * esz:
                ft[:, D:] = 0
            for lo, x in gen_rows(0, N, D, device):
                ft[lo: lo + x.shape[0], :D] = x
            torch.cuda.synchronize()
            full.commit_device_rows(N, np.arange(N, dtype=np.uint64))
            fd, fl, fn = full.search_batch(Q.cpu().numpy(), K)
            same = bool((fl == res_l).all() and (fd.view(np.uint32) == res_d.view(np.uint32)).all())
            verify = "bit-identical" if same else "MISMATCH"
            del full
        hnsw = None
        if args.hnsw_sharded and args.hnsw_rows > 0 and not bf16:
            # one HNSW graph per shard inside ONE index; recall against the sharded FLAT answer over the same rows
            Nh = min(args.hnsw_rows, N)
            nq = min(args.hnsw_queries, 2048)
            Qh = make_queries(state["A"], nq, D, device, 9090)
            rows = torch.cat([x for _, x in gen_rows(0, Nh, D, device)])[:Nh]
            host_rows = np.ascontiguousarray(rows.cpu().numpy())
            del rows
            t1 = time.perf_counter()
            h = vsa.Index("HNSW", D, "COSINE", initial_cap=Nh, m=16, ef_construction=200, ef_runtime=args.hnsw_ef, shard_devices=devs)
            h.add_batch(host_rows)
            h.flush()
            hb = time.perf_counter() - t1
            hd = torch.empty(nq, K, device=device, dtype=torch.float32)
            hl = torch.empty(nq, K, device=device, dtype=torch.int64)
            hn = torch.empty(nq, device=device, dtype=torch.int32)
            ms = timed(lambda: h.search_batch_device(Qh.data_ptr(), nq, K, hd.data_ptr(), hl.data_ptr(), hn.data_ptr(),
                                                     ef=args.hnsw_ef, stream=state["ws"].cuda_stream), 3)
            from oracle import oracle as O
            bits = O.allow_bitmap(np.arange(Nh, dtype=np.uint64), Nh) if Nh < N else None
            gt = np.empty((nq, K), np.uint64)
            hq = Qh.cpu().numpy()
            for i in range(0, nq, 256):
                _, L, _ = ix.search_batch(hq[i:i + 256], K, allow=bits, allow_nbits=Nh if bits is not None else None)
                gt[i:i + 256] = L
            hnsw = {"shards": world, "rows_per_shard": Nh // world, "rows": Nh, "M": 16, "ef_construction": 200, "ef": args.hnsw_ef,
                    "k": K, "queries_per_batch": nq, "build_s": round(hb, 2), "gpu_qps": round(nq / (ms * 1e-3), 1),
                    "recall_at_10": round(recall_of(hl.cpu().numpy().view(np.uint64), gt, K), 4),
                    "merge": "one graph per shard inside one vk_index; per-shard top-k gathered on device 0, (distance,label) merge"}
        qps = B * args.steps / dt
        n_local = N // world
        stride = ((D + 63) // 64) * 64 * esz
        flops = 2.0 * N * D * B
        out = {"metric": BASELINE_METRIC if not bf16 else "kNN queries/sec, FLAT 10Mx768 bf16-stored cosine k=10 batch=256",
               "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "row_storage": args.dtype, "data": "synthetic",
               "config": {"workload": f"FLAT {N}x{D} {'bf16 rows' if bf16 else 'fp32'} COSINE k={K} batch={B} (BASELINE.json configs[1])",
                          "rows_per_gpu": n_local, "sharding": f"rows/{world}", "recall_at_10": 1.0,
                          "parallelism": f"one vk_index with n_shards={world} in ONE process (rank 0): queries broadcast by peer copy, "
                                         f"per-shard top-k gathered on device 0, (distance,label) merge; ranks 1..{world - 1} idle",
                          "devices": devs, "verify_merge": verify},
               "roofline": {"bound": "mfma", "achieved": round(flops / (dev_ms * 1e-3) / 1e12, 3), "peak": F32_MFMA_PEAK_TF * world,
                            "unit": "TFLOP/s", "frac": round(flops / (dev_ms * 1e-3) / 1e12 / (F32_MFMA_PEAK_TF * world), 5),
                            "traffic": None, "kernel": "flat_gemm_kernel (per shard)", "per_launch_ms": round(dev_ms, 4),
                            "algorithmic_bytes": n_local * stride},
               "cpu_baseline": None, "hnsw": hnsw, "build_s": round(state["build_s"], 2)}
        print(json.dumps(out))
        if verify is not None:
            print(json.dumps({"verify_merge": verify, "shards": world, "rows": N}))
            assert verify == "bit-identical"
    dist.barrier()
    dist.destroy_process_group()
    return out


def native_coalescer_leg(rows=2_000_000, dim=768, threads=256, calls=50):
    """The same experiment without the interpreter in the way: scripts/coalescer_native (C++ against include/vk_index.h,
    built by __graft_entry__.build()) -- `threads` native threads of single-query vk_index_search calls on its own FLAT
    index of `rows` x `dim` (the Python leg above is bound by 64 interpreter threads, not by the library)."""
    import re
    import subprocess
    exe = ROOT / "scripts" / "coalescer_native"
    if not exe.exists():
        return None
    txt = subprocess.run([str(exe), str(rows), str(dim), str(threads), str(calls)], capture_output=True, text=True, timeout=300).stdout
    out = {"workload": f"FLAT {rows}x{dim} cosine k=10, {threads} native threads x {calls} single-query calls", "on": []}
    m = re.search(r"coalescing off, (\d+) callers: (\d+) queries/s", txt)
    if m:
        out["off_qps"], out["off_callers"] = int(m.group(2)), int(m.group(1))
    for m in re.finditer(r"max_wait (\d+) us\), \d+ callers x \d+ calls: (\d+) queries/s, (\d+) device batches, mean batch ([0-9.]+)", txt):
        out["on"].append({"max_wait_us": int(m.group(1)), "qps": int(m.group(2)), "device_batches": int(m.group(3)), "mean_batch": float(m.group(4))})
    return out


def coalescer_leg(ix, hq, K, threads=64, per_thread=4):
    """N1: the reference issues one query per FT.SEARCH from a pool of reader threads (search.cc:886-910).
    `threads` callers each issue `per_thread` single-query vk_index_search calls, first one at a time per
    call (a device pass each), then with vk_index_set_coalescing merging concurrent calls into batches."""
    import threading

    def drive(n_threads):
        out = [None] * (n_threads * per_thread)

        def worker(t):
            for r in range(per_thread):
                i = t * per_thread + r
                out[i] = ix.search_one(hq[i % len(hq)], K)

        ts = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
        t0 = time.perf_counter()
        [t.start() for t in ts]
        [t.join() for t in ts]
        return time.perf_counter() - t0, out

    ix.set_coalescing(0, 0)
    dt0, ref = drive(threads)
    before = ix.stats()
    wait_us = int(os.environ.get("VK_BENCH_COALESCE_WAIT_US", "300"))
    ix.set_coalescing(threads, wait_us)
    dt1, got = drive(threads)
    after = ix.stats()
    ix.set_coalescing(0, 0)
    same = all(a[1].tolist() == b[1].tolist() and a[0].view(np.uint32).tolist() == b[0].view(np.uint32).tolist()
               for a, b in zip(ref, got))
    nq = threads * per_thread
    batches = after.coalesced_batches - before.coalesced_batches
    return {"callers": threads, "queries": nq, "uncoalesced_qps": round(nq / dt0, 1), "coalesced_qps": round(nq / dt1, 1),
            "device_batches": int(batches), "mean_batch": round(nq / max(1, batches), 1), "max_wait_us": wait_us,
            "answers_identical": bool(same)}


def _baseline_metric():
    try:
        return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")))["metric"]
    except (OSError, KeyError, ValueError):
        return "kNN queries/sec + recall@10, 10M\u00d7768 fp32 cosine, flat & HNSW, 1/2/4/8 GPU"


BASELINE_METRIC = _baseline_metric()


def source_sha256():
    """Hash of the kernel and library sources (csrc/*.hip, *.hpp, *.cc): the build id a PMC traffic file is stamped with."""
    import hashlib
    h = hashlib.sha256()
    csrc = ROOT / "valkey-search_amd" / "csrc"
    for f in sorted(list(csrc.glob("*.hip")) + list(csrc.glob("*.hpp")) + list(csrc.glob("*.cc"))):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def pmc_traffic(N, D, B, world, kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (rocprofv3 --pmc is a separate run by
    rule, so bench.py cannot collect it live): profiles/r02_pmc_fetch_size.json, written by scripts/pmc_traffic.sh and
    stamped with the hash of the sources it measured.  Printed only for the default single-GPU workload AND only while
    that hash equals the current sources' -- a stale file yields null, never an old number."""
    path = ROOT / "profiles" / "r02_pmc_fetch_size.json"
    if world != 1 or (N, D, B) != (10_000_000, 768, 256) or not path.exists() or "bf16" in sys.argv:
        return None, None
    if any(os.environ.get(v) for v in ("VK_FLAT_FORCE_SCAN", "VK_GEMM_MODE", "VK_GEMM_ABLATE", "VK_GEMM_LOCKSTEP", "VK_FILTER_TIMING", "VK_FLAT_FILTER")):
        return None, None
    j = json.load(open(path))
    if j.get("src_sha256") != source_sha256():
        return None, "profiles/r02_pmc_fetch_size.json is stale (taken with other kernel sources): re-run scripts/pmc_traffic.sh"
    for name, v in j.get("kernels", {}).items():
        if kernel_prefix in name and "prepass" not in name and "[small]" not in name:
            return round(v["hbm_bytes_per_launch"]), "profiles/r02_pmc_fetch_size.json (rocprofv3 --pmc FETCH_SIZE, separate pass, same sources)"
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
    ap.add_argument("--cpu-queries-per-thread", type=int, default=1,
                    help="CPU baseline: queries per host thread (1 x 16 threads x a 10M-row scan is about 30 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-query-steps", type=int, default=5, help="extra: B=1 scan timing (HBM roofline)")
    ap.add_argument("--hnsw-rows", type=int, default=-1,
                    help="HNSW leg (BASELINE.json configs[2]) over the first rows: -1 = all of --rows (10M), 0 = skip")
    ap.add_argument("--hnsw-ef", type=int, default=128)
    ap.add_argument("--hnsw-queries", type=int, default=8192)
    ap.add_argument("--hnsw-cpu-queries-per-thread", type=int, default=32)
    ap.add_argument("--hybrid-rows", type=int, default=1_250_000,
                    help="one shard of BASELINE.json configs[4] (HNSW + TAG filter, efSearch=256): rows (0 = skip)")
    ap.add_argument("--hybrid-queries", type=int, default=4096)
    ap.add_argument("--bf16-rows", type=int, default=-1,
                    help="one shard of BASELINE.json configs[3] (FLAT bf16 IP): rows, -1 = as --rows, 0 = skip")
    ap.add_argument("--hnsw-sharded", action="store_true",
                    help="N > 1: also run the sharded HNSW leg (one graph per rank; extra collectives after the timed region)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--multi-gpu", choices=["lib", "ranks"], default="lib",
                    help="N > 1: 'lib' = the product's own multi-GPU index (n_shards = N inside one process, rank 0), falling back to "
                         "'ranks' = one process per GPU, each with its shard, RCCL all-gather of the per-shard top-k + device merge")
    ap.add_argument("--same-device", action="store_true",
                    help="test aid: every rank uses cuda:0 (with --backend gloo, two ranks can exercise the sharded path on one GPU)")
    ap.add_argument("--verify-merge", action="store_true",
                    help="test aid (N > 1, small --rows): rank 0 also builds the unsharded index and checks the merged answer against it")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(args.backend)

    N, D, B, K = args.rows, args.dim, args.batch, args.k
    if args.hnsw_rows < 0:
        args.hnsw_rows = N
    if args.bf16_rows < 0:
        args.bf16_rows = N
    if world > 1 and args.multi_gpu == "lib":
        if lib_multi_gpu(args, world, rank, local_rank, dist, device) != "fallback":
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
    dev_ms = ev0.elapsed_time(ev1) / args.steps
    st_after = ix.stats()
    # batches of the timed region that went through the f16 candidate filter, and the device time of that kernel alone
    # (HIP events the library records around its launches, on the stream they run on)
    filt_n = st_after.filter_batches - st_before.filter_batches
    filt_ms = (st_after.filter_kernel_ns - st_before.filter_kernel_ns) / 1e6 / filt_n if filt_n else None
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
        filt_stats = {"survivors_per_query": round(st_f.last_filter_candidates / B, 1), "queries_handed_over": int(st_f.last_filter_fallback)}
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
        cpu = {"value": round(qps_sample * S / N, 4), "unit": "queries/s", "cores": threads, "kind": "port",
               "seconds": round(cdt, 2), "host_gbs": round(nqt * S * D * 4 / cdt / 1e9, 1),
               "sample": (f"oracle FLAT scan ({O.cpu_path()} clone of the SimSIMD skylake order), {nqt} queries, each a full pass "
                          f"over all {S} rows, one query per thread on {threads} threads (the container's CPU quota)" if full else
                          f"oracle FLAT scan ({O.cpu_path()} clone of the SimSIMD skylake order), {nqt} queries over "
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
    # ---- one shard of configs[4]: HNSW + TAG filter ----
    hybrid = None
    if extras and args.hybrid_rows > 0 and not bf16:
        hybrid = leg(hybrid_leg, args, ix, host_rows[:min(args.hybrid_rows, n_local)], A, device, stream_ptr, n_local)
    host_rows = None
    # ---- one shard of configs[3]: FLAT bf16 IP ----
    bf16_shard = None
    if extras and args.bf16_rows > 0 and not bf16:
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

    coalescer = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            coalescer = coalescer_leg(ix, Q.cpu().numpy(), K)
        except Exception as e:   # noqa: BLE001
            coalescer = {"error": f"{type(e).__name__}: {e}"}
        try:
            coalescer["native_callers"] = native_coalescer_leg()
        except Exception as e:   # noqa: BLE001
            coalescer["native_callers"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        dominant = "flat_filter_kernel" if filt_n else ("flat_gemm_kernel" if B >= 5 else "flat_scan_kernel")
        traffic, traffic_src = pmc_traffic(N, D, B, world, dominant)
        qps = B * args.steps / dt
        scan_bytes = n_local * stride                       # algorithmic bytes of one pass over the shard
        flops = 2.0 * n_local * D * B                       # per step per GPU
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
            "roofline": ({"bound": "hbm", "achieved": round(scan_bytes / (filt_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(scan_bytes / (filt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic,
                          "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src, "algorithmic_bytes": scan_bytes,
                          "kernel": "flat_filter_kernel", "per_launch_ms": round(filt_ms, 4), "launches_timed": int(filt_n),
                          "step_ms_on_stream": round(dev_ms, 4),
                          "hbm_frac_of_whole_step": round(scan_bytes / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                          "filter": filt_stats,
                          "f16_mfma_tflops": round(flops / (filt_ms * 1e-3) / 1e12, 1),
                          "f16_mfma_frac_of_2500": round(flops / (filt_ms * 1e-3) / 1e12 / 2500.0, 4)}
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
            "coalescer": coalescer,
            "hnsw": hnsw,
            "config4_shard_hybrid": hybrid,
            "config3_shard_bf16_ip": bf16_shard,
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
