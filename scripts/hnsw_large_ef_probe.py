"""10M x 768 HNSW (configs[2]) at the matched-recall end of the sweep (ef = 512, 768): what the visited table in memory costs
and what moves it -- the table's density (option hnsw-hash-per-ef: words per unit of ef; 64 = quarter full, 32 = half full,
16 = three quarters: fewer distinct lines per search, longer probe runs, more queries that outgrow it), the way it is kept
(hnsw-visited-mode 0 / 1 / 2) and the batch size.  ONE graph; answers and work counters compared with the first setting."""
import os, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _pkg
vsa = _pkg.vsa
from bench import gen_rows
dev = torch.device("cuda", 0)
N, D, nq = int(os.environ.get("ROWS", 10_000_000)), 768, int(os.environ.get("NQ", 8192))
efs = [int(e) for e in os.environ.get("EFS", "512,768").split(",")]
g = torch.Generator(device=dev); g.manual_seed(4242)
gA = torch.Generator(device=dev); gA.manual_seed(1234)
A = torch.randn(D, 32, generator=gA, device=dev)
Qd = torch.nn.functional.normalize(torch.randn(nq, 32, generator=g, device=dev) @ A.T + 0.05 * torch.randn(nq, D, generator=g, device=dev), dim=1).contiguous()
Q = Qd.cpu().numpy()
h = vsa.Index("HNSW", D, "COSINE", initial_cap=N, m=16, ef_construction=200, ef_runtime=128)
step = 1_000_000
t = time.time()
for lo in range(0, N, step):
    x = torch.empty(min(step, N - lo), D, device=dev)
    for l2, c in gen_rows(lo, x.shape[0], D, dev):
        x[l2 - lo:l2 - lo + c.shape[0]] = c
    h.add_batch(x.cpu().numpy(), np.arange(lo, lo + x.shape[0], dtype=np.uint64))
h.flush()
print(f"built {N} x {D} in {time.time() - t:.1f} s", flush=True)
od = torch.empty(nq, 10, device=dev, dtype=torch.float32)
ol = torch.empty(nq, 10, device=dev, dtype=torch.int64)
on = torch.empty(nq, device=dev, dtype=torch.int32)
settings = [("default", {}), ("hash-per-ef 32", {"hnsw-hash-per-ef": 32}), ("hash-per-ef 24", {"hnsw-hash-per-ef": 24}), ("hash-per-ef 16", {"hnsw-hash-per-ef": 16}),
            ("mode 2", {"hnsw-visited-mode": 2}), ("mode 2, per-ef 32", {"hnsw-visited-mode": 2, "hnsw-hash-per-ef": 32}),
            ("mode 1", {"hnsw-visited-mode": 1}), ("default again", {})]
settings += [("mode 4 (LDS set forced)", {"hnsw-visited-mode": 4})]
if os.environ.get("SETTINGS"):
    keep = [x.strip() for x in os.environ["SETTINGS"].split(",")]
    settings = [s_ for s_ in settings if s_[0] in keep]
defaults = {"hnsw-hash-per-ef": 64, "hnsw-visited-mode": 3}
ws = torch.cuda.Stream(device=dev)    # (a stream of its own: the library takes the null stream for "the index's own")
for ef in efs:
    ref = None
    for name, opts in settings:
        for k, v in {**defaults, **opts}.items():
            h.set_option(k, v)
        Dh, Lh, Nh = h.search_batch(Q, 10, ef=ef)              # host path: answers + work counters
        st = h.stats()
        useful = st.last_n_eval * (D * 4 + 4) + st.last_n_hops * 132
        cur = (Dh.view(np.uint32).copy(), Lh.copy(), st.last_n_eval, st.last_n_hops)
        same = "-" if ref is None else str(bool((cur[0] == ref[0]).all() and (cur[1] == ref[1]).all() and cur[2:] == ref[2:]))
        if ref is None: ref = cur
        s = ws.cuda_stream
        for _ in range(2):
            h.search_batch_device(Qd.data_ptr(), nq, 10, od.data_ptr(), ol.data_ptr(), on.data_ptr(), ef=ef, stream=s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 4
        e0.record(ws)
        for _ in range(reps):
            h.search_batch_device(Qd.data_ptr(), nq, 10, od.data_ptr(), ol.data_ptr(), on.data_ptr(), ef=ef, stream=s)
        e1.record(ws)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"ef={ef} {name:22s}: {nq/ms*1e3:9.0f} QPS {ms:8.2f} ms, useful {useful/ms/1e9:.3f} TB/s = {useful/ms/1e9/8:.3f} of peak, "
              f"mode {st.last_visited_mode} evals/q {st.last_n_eval/nq:.0f} hops/q {st.last_n_hops/nq:.0f} redo {st.last_frontier_redo}, same: {same}", flush=True)
