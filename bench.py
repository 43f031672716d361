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


def hnsw_leg(args, flat_ix, table, A, device, stream_ptr):
    """HNSW over the first --hnsw-rows rows: device-assisted bulk build (K9), device search, recall vs the
    exact FLAT answer on the same rows, the CPU oracle searching the SAME graph, and the two hybrid paths
    of BASELINE.json configs[4] (inline filter at 10 % selectivity, pre-filter below 0.1 %)."""
    from oracle import oracle as O
    Nh, D, K, ef = min(args.hnsw_rows, table.shape[0]), args.dim, args.k, args.hnsw_ef
    host_rows = np.ascontiguousarray(table[:Nh, :D].cpu().numpy())
    t0 = time.perf_counter()
    h = vsa.Index("HNSW", D, "COSINE", initial_cap=Nh, m=16, ef_construction=200, ef_runtime=ef)
    h.add_batch(host_rows)
    h.flush()
    build_s = time.perf_counter() - t0
    nq = args.hnsw_queries
    qg = torch.Generator(device=device)
    qg.manual_seed(9090)
    Qh = torch.nn.functional.normalize(torch.randn(nq, 32, generator=qg, device=device) @ A.T +
                                       0.05 * torch.randn(nq, D, generator=qg, device=device), dim=1).contiguous()
    hq = Qh.cpu().numpy()
    # exact ground truth from the FLAT index restricted to the same rows (labels < Nh)
    bits = O.allow_bitmap(np.arange(Nh, dtype=np.uint64), Nh)
    _, gt, _ = flat_ix.search_batch(hq, K, allow=bits, allow_nbits=Nh)
    od = torch.empty(nq, K, device=device, dtype=torch.float32)
    ol = torch.empty(nq, K, device=device, dtype=torch.int64)
    on = torch.empty(nq, device=device, dtype=torch.int32)

    def run():
        h.search_batch_device(Qh.data_ptr(), nq, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), ef=ef, stream=stream_ptr())

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gl = ol.cpu().numpy().view(np.uint64)
    recall = float(np.mean([len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gl, gt)])) / K
    _D, _L, _N = h.search_batch(hq[:1024], K, ef=ef)        # host path once: fills the work counters
    st = h.stats()
    useful = (st.last_n_eval * (D * 4 + 4) + st.last_n_hops * 132) / 1024.0 * nq
    # CPU: the oracle searches the very same graph (SaveIndex chunk stream), one query per thread
    t1 = time.perf_counter()
    o = O.HNSW.from_saved_chunks(h.save(), D, "COSINE", 16, ef_construction=200)
    export_s = time.perf_counter() - t1
    threads = effective_cpus()
    ncq = min(nq, threads * 64)
    o.search(hq[0], K, ef=ef)
    t1 = time.perf_counter()
    # the oracle's visited list is per index object: searches are serialised per object, so give each
    # thread its own view of the graph?  No -- hnswlib hands out one visited list per concurrent search;
    # the oracle is single-threaded by design, so time it on one thread and report that honestly.
    cpu_res = [o.search(hq[i], K, ef=ef) for i in range(min(ncq, 512))]
    cdt = time.perf_counter() - t1
    n_cpu = len(cpu_res)
    cpu_recall = float(np.mean([len(set(r[1].tolist()) & set(gt[i].tolist())) for i, r in enumerate(cpu_res)])) / K
    same = sum(int(cpu_res[i][1].tolist() == gl[i][:len(cpu_res[i][1])].tolist()) for i in range(n_cpu))
    # ---- hybrid: TAG-like filters as allow-bitmaps over labels (planner.cc:21-45 picks the path) ----
    tag_bits = O.allow_bitmap(np.arange(3, Nh, 10, dtype=np.uint64), Nh)           # 10 % of the rows: inline
    d_bits = torch.from_numpy(tag_bits.view(np.int64)).to(device)
    _, gt_f, _ = flat_ix.search_batch(hq[:1024], K, allow=tag_bits, allow_nbits=Nh)

    ef_h = 256   # BASELINE.json configs[4]: efSearch = 256 for the hybrid queries

    def run_f():
        h.search_batch_device(Qh.data_ptr(), nq, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), ef=ef_h,
                              d_allow=d_bits.data_ptr(), allow_nbits=Nh, stream=stream_ptr())

    run_f()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run_f()
    e1.record()
    torch.cuda.synchronize()
    ms_f = e0.elapsed_time(e1) / reps
    glf = ol.cpu().numpy().view(np.uint64)[:1024]
    recall_f = float(np.mean([len(set(a.tolist()) & set(b.tolist())) for a, b in zip(glf, gt_f)])) / K
    # pre-filter (<= 0.001 * N keys): exact kNN over the key list, one call per query like CalcBestMatchingPrefilteredKeys
    keys = np.sort(np.random.default_rng(77).choice(Nh, max(1, Nh // 2000), replace=False)).astype(np.uint64)
    key_bits = O.allow_bitmap(keys, Nh)
    t1 = time.perf_counter()
    pre = [h.search_labels(hq[i], K, keys) for i in range(256)]
    pre_dt = time.perf_counter() - t1
    _, gt_p, ngt = flat_ix.search_batch(hq[:256], K, allow=key_bits, allow_nbits=Nh)
    pre_ok = all(set(pre[i][1].tolist()) == set(gt_p[i, :ngt[i]].tolist()) for i in range(256))
    hybrid = {"inline_filter": {"selectivity": 0.1, "ef": ef_h, "gpu_qps": round(nq / (ms_f * 1e-3), 1), "recall_at_10": round(recall_f, 4)},
              "pre_filter": {"keys": int(len(keys)), "selectivity": round(len(keys) / Nh, 5),
                             "qps_single_caller": round(256 / pre_dt, 1), "exact": bool(pre_ok)}}
    return {"rows": Nh, "M": 16, "ef_construction": 200, "ef": ef, "k": K, "queries_per_batch": nq, "hybrid": hybrid,
            "build_s": round(build_s, 2), "build_inserts_per_s": round(Nh / build_s, 1), "build": "device-assisted (K9)", "host_threads": threads,
            "gpu_qps": round(nq / (ms * 1e-3), 1), "ms_per_batch": round(ms, 3), "recall_at_10": round(recall, 4),
            "n_eval_per_query": round(st.last_n_eval / 1024.0, 1), "n_hops_per_query": round(st.last_n_hops / 1024.0, 1),
            "useful_gbs": round(useful / (ms * 1e-3) / 1e9, 1),
            "cpu": {"kind": "port", "threads": 1, "qps_per_thread": round(n_cpu / cdt, 1), "queries": n_cpu,
                    "recall_at_10": round(cpu_recall, 4), "same_graph": True,
                    "ids_identical_to_gpu": f"{same}/{n_cpu}", "graph_export_s": round(export_s, 2)}}


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
    ix.set_coalescing(threads, 300)
    dt1, got = drive(threads)
    after = ix.stats()
    ix.set_coalescing(0, 0)
    same = all(a[1].tolist() == b[1].tolist() and a[0].view(np.uint32).tolist() == b[0].view(np.uint32).tolist()
               for a, b in zip(ref, got))
    nq = threads * per_thread
    batches = after.coalesced_batches - before.coalesced_batches
    return {"callers": threads, "queries": nq, "uncoalesced_qps": round(nq / dt0, 1), "coalesced_qps": round(nq / dt1, 1),
            "device_batches": int(batches), "mean_batch": round(nq / max(1, batches), 1), "max_wait_us": 300,
            "answers_identical": bool(same)}


def _baseline_metric():
    try:
        return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")))["metric"]
    except (OSError, KeyError, ValueError):
        return "kNN queries/sec + recall@10, 10M\u00d7768 fp32 cosine, flat & HNSW, 1/2/4/8 GPU"


BASELINE_METRIC = _baseline_metric()


def pmc_traffic(N, D, B, world):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE
    is a separate run by rule, so bench.py cannot collect it live): profiles/r01_pmc_fetch_size_k4.json,
    valid for the default single-GPU workload only."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_fetch_size_k4.json")
    if world != 1 or (N, D, B) != (10_000_000, 768, 256) or not os.path.exists(path) or "bf16" in sys.argv:
        return None, None
    if os.environ.get("VK_FLAT_FORCE_SCAN") or os.environ.get("VK_GEMM_MODE") or os.environ.get("VK_GEMM_ABLATE"):
        return None, None
    j = json.load(open(path))
    key = "lockstep_off" if os.environ.get("VK_GEMM_LOCKSTEP") == "0" else "lockstep_on"
    return round(j[key]["hbm_bytes_per_launch"]), "profiles/r01_pmc_fetch_size_k4.json (rocprofv3 --pmc FETCH_SIZE, separate pass)"


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
    ap.add_argument("--cpu-rows", type=int, default=200_000, help="rows of the CPU-baseline sample")
    ap.add_argument("--cpu-queries-per-thread", type=int, default=24,
                    help="CPU-baseline sample: queries per host thread (24 x 16 threads x a 200k-row scan is about 13 s of CPU work, under 1 s of wall time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-query-steps", type=int, default=5, help="extra: B=1 scan timing (HBM roofline)")
    ap.add_argument("--hnsw-rows", type=int, default=200_000, help="extra: HNSW leg over the first rows (0 = skip)")
    ap.add_argument("--hnsw-ef", type=int, default=128)
    ap.add_argument("--hnsw-queries", type=int, default=4096)
    ap.add_argument("--hnsw-sharded", action="store_true",
                    help="N > 1: also run the sharded HNSW leg (one graph per rank; extra collectives after the timed region)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
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
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) / args.steps
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

    result_d = (fin_d if world > 1 else out_d).cpu().numpy()
    result_l = (fin_l if world > 1 else out_l).cpu().numpy().view(np.uint64)

    # ---- CPU baseline + parity spot check on rank 0 (oracle = checker, never the product) ----
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # reported at N=1 only (bench contract)
        from oracle import oracle as O
        S = min(args.cpu_rows, n_local)
        host_rows = np.ascontiguousarray(table[:S, :D].float().cpu().numpy())   # bf16: the widened stored values
        flat = O.Flat(D, "COSINE", isa="skylake", max_elements=S)
        flat.add_many(host_rows, np.arange(r0, r0 + S, dtype=np.uint64), borrowed=True)
        hq = Q.cpu().numpy()
        threads = effective_cpus()
        nqt = threads * args.cpu_queries_per_thread
        flat.search(hq[0], K)  # warm
        t1 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            res = list(ex.map(lambda i: flat.search(hq[i % B], K), range(nqt)))
        cdt = time.perf_counter() - t1
        qps_sample = nqt / cdt
        cpu = {"value": round(qps_sample * S / N, 4), "unit": "queries/s", "cores": threads, "kind": "port",
               "sample": f"oracle FLAT scan ({O.cpu_path()} clone of the SimSIMD skylake order), {nqt} queries over "
                         f"the first {S} rows on {threads} threads ({qps_sample:.1f} q/s on the sample), scaled "
                         f"linearly to {N} rows"}
        # parity at full index size: the GPU answer restricted by a filter to the sampled rows must equal
        # the oracle's answer on those rows, ids and distance bits
        bits = O.allow_bitmap(np.arange(r0, r0 + S, dtype=np.uint64), r0 + S)
        ok = True
        for i in range(4):
            gd, gl = ix.search(hq[i], K, allow=bits, allow_nbits=r0 + S)
            od, ol = res[i] if i < len(res) else flat.search(hq[i], K)
            ok = ok and gl.tolist() == ol.tolist() and gd.view(np.uint32).tolist() == od.view(np.uint32).tolist()
        # ... and the same through the timed path (batched: K4 on the matrix cores)
        nb = min(B, 32)
        bd, bl, bn = ix.search_batch(hq[:nb], K, allow=bits, allow_nbits=r0 + S)
        for i in range(nb):
            od, ol = res[i] if i < len(res) else flat.search(hq[i], K)
            ok = ok and bl[i, :bn[i]].tolist() == ol.tolist() and bd[i, :bn[i]].view(np.uint32).tolist() == od.view(np.uint32).tolist()
        parity = "bit-exact" if ok else "MISMATCH"

    # ---- extra: HNSW (BASELINE.json configs[2] shape: M=16 efC=200, cosine, ef=128, k=10) on rank 0 ----
    hnsw = None
    # (the extra legs must never cost the headline line: a failure in one of them is reported in its place)
    if rank == 0 and world == 1 and args.hnsw_rows > 0 and not bf16:
        try:
            hnsw = hnsw_leg(args, ix, table, A, device, stream_ptr)
        except Exception as e:   # noqa: BLE001
            hnsw = {"error": f"{type(e).__name__}: {e}"}
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

    if rank == 0:
        traffic, traffic_src = pmc_traffic(N, D, B, world)
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
            "roofline": ({"bound": "mfma", "achieved": round(flops / (dev_ms * 1e-3) / 1e12, 3),
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
