"""Where single-query serving loses to the device-batch rate: FLAT 10M x 768 (B=256 batches) and HNSW (ef=128, batches of
8192) driven through the raw C ABI (scripts/serving_probe.cc) and through the adaptor classes (scripts/adaptor_probe.cc)
with different numbers of completer threads / front threads, each line with the dispatcher's own account of where its
runner threads' time went (vk_index_stats.dispatch_*_us).  usage: serving_pipeline_probe.py [flat_rows] [hnsw_rows]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _pkg
vsa = _pkg.vsa
from bench import gen_rows, device_view, make_queries, effective_cpus
dev = torch.device("cuda", 0)
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
NH = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
D, K = 768, 10
readers = effective_cpus()


def line(tag, ix, fn, device_qps):
    s0 = ix.stats()
    r = fn()
    s1 = ix.stats()
    d = {k: getattr(s1, f"dispatch_{k}_us") - getattr(s0, f"dispatch_{k}_us") for k in ("idle", "window", "search", "handout", "completer")}
    print(f"{tag:58s} {r.qps:10.0f} QPS  {r.qps / device_qps:5.3f} of device  mean_batch {r.mean_batch:7.1f}  p50 {r.p50_us / 1e3:7.2f} ms  "
          f"ok {int(r.mismatches == 0 and r.errors == 0)}  wall {r.seconds * 1e3:7.0f} ms | runners: idle {d['idle'] / 1e3:7.0f} window {d['window'] / 1e3:6.0f} "
          f"search {d['search'] / 1e3:7.0f} handout {d['handout'] / 1e3:6.0f} ms; completers {d['completer'] / 1e3:6.0f} ms", flush=True)
    return r


A = None
if NF:
    ix = vsa.Index("FLAT", D, "COSINE", initial_cap=NF)
    p, stride = ix.device_rows(NF)
    t = device_view(p, (NF, stride // 4), dev)
    g = torch.Generator(device=dev); g.manual_seed(1234)
    A = torch.randn(D, 32, generator=g, device=dev, dtype=torch.float32)
    for lo, x in gen_rows(0, NF, D, dev):
        t[lo:lo + x.shape[0], :D] = x
    torch.cuda.synchronize()
    ix.commit_device_rows(NF)
    hq = make_queries(A, 256, D, dev, 4242).cpu().numpy()
    rd, rl, rn = ix.search_batch(hq, K)
    t0 = time.perf_counter()
    for _ in range(10):
        ix.search_batch(hq, K)
    dq = 256 * 10 / (time.perf_counter() - t0)
    print(f"FLAT {NF} x {D}: host-entry batch rate {dq:.0f} QPS (B=256)", flush=True)
    for comp in (4, 0):
        ix.set_option("completer-threads", comp)
        ix.set_coalescing(256, 500)
        vsa.probe_submit(ix, hq, K, 1024, 4, 1024, 0, ref=(rd, rl))
        line(f"FLAT raw submit, 4 producers, completers {comp}", ix, lambda: vsa.probe_submit(ix, hq, K, 10240, 4, 1024, 0, ref=(rd, rl)), dq)
        for fronts in (1, 4):
            vsa.adaptor_probe(ix, hq, K, 1024, readers, 1024, 0, hnsw=False, ref=(rd, rl), fronts=fronts)
            line(f"FLAT adaptor async, {readers} readers, fronts {fronts}, completers {comp}", ix,
                 lambda: vsa.adaptor_probe(ix, hq, K, 10240, readers, 1024, 0, hnsw=False, ref=(rd, rl), fronts=fronts), dq)
    ix.set_option("completer-threads", 6)
    line("FLAT blocking, 256 callers", ix, lambda: vsa.probe_blocking(ix, hq, K, 256, 24, 0, ref=(rd, rl)), dq)
    ix.set_coalescing(0, 0)
    del ix, t
    torch.cuda.empty_cache()

if NH:
    rows = torch.empty(NH, D, device=dev)
    for lo, x in gen_rows(0, NH, D, dev):
        rows[lo:lo + x.shape[0]] = x
    if A is None:
        g = torch.Generator(device=dev); g.manual_seed(1234)
        A = torch.randn(D, 32, generator=g, device=dev, dtype=torch.float32)
    host_rows = rows.cpu().numpy()
    del rows
    t0 = time.perf_counter()
    h = vsa.Index("HNSW", D, "COSINE", initial_cap=NH, m=16, ef_construction=200, ef_runtime=128)
    h.add_batch(host_rows)
    h.flush()
    print(f"HNSW {NH} x {D} built in {time.perf_counter() - t0:.1f} s", flush=True)
    nq = 8192
    hq = make_queries(A, nq, D, dev, 9090).cpu().numpy()
    rd, rl, rn = h.search_batch(hq, K, ef=128)
    t0 = time.perf_counter()
    for _ in range(5):
        h.search_batch(hq, K, ef=128)
    dq = nq * 5 / (time.perf_counter() - t0)
    print(f"HNSW host-entry batch rate {dq:.0f} QPS (8192 queries a call, ef=128)", flush=True)
    for comp in [int(c) for c in __import__("os").environ.get("COMPLETERS", "4,2,0").split(",")]:
        h.set_option("completer-threads", comp)
        h.set_coalescing(nq, 2000)
        vsa.probe_submit(h, hq, K, 4 * nq, 8, 4 * nq, 128, ref=(rd, rl))
        line(f"HNSW raw submit, 8 producers, completers {comp}", h, lambda: vsa.probe_submit(h, hq, K, 16 * nq, 8, 4 * nq, 128, ref=(rd, rl)), dq)
        for fronts in (1, 4):
            for bulk in ("1", "0", "1", "0"):   # completions a piece of a batch at a time / one callback per request, alternating
                __import__("os").environ["VK_PROBE_BULK"] = bulk
                vsa.adaptor_probe(h, hq, K, 4 * nq, readers, 4 * nq, 128, hnsw=True, ref=(rd, rl), fronts=fronts)
                line(f"HNSW adaptor async, fronts {fronts}, completers {comp}, completions {'in bulk' if bulk == '1' else 'one by one'}", h,
                     lambda: vsa.adaptor_probe(h, hq, K, 16 * nq, readers, 4 * nq, 128, hnsw=True, ref=(rd, rl), fronts=fronts), dq)
