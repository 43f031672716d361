"""Probe: HNSW build speed (host threads) and device search throughput/recall at a given size."""
import argparse, os, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _pkg
vsa = _pkg.vsa
from bench import gen_rows

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=200000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--nq", type=int, default=1024)
ap.add_argument("--ef", type=int, default=128)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--tombstone", action="store_true", help="after the first measurement: remove ONE label and measure again (a tombstone sends the search to the HBM-frontier kernel without any filter)")
ap.add_argument("--calibrate", action="store_true", help="one single-query FLAT scan (reads rows x row bytes exactly once: the PMC calibration)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
N, D = args.rows, args.dim
x = torch.empty(N, D, device=dev)
for lo, c in gen_rows(0, N, D, dev):
    x[lo:lo + c.shape[0]] = c
hx = x.cpu().numpy()
g = torch.Generator(device=dev); g.manual_seed(4242)
gA = torch.Generator(device=dev); gA.manual_seed(1234)
A = torch.randn(D, 32, generator=gA, device=dev)
Q = torch.nn.functional.normalize(torch.randn(args.nq, 32, generator=g, device=dev) @ A.T + 0.05 * torch.randn(args.nq, D, generator=g, device=dev), dim=1).cpu().numpy()
t = time.time()
h = vsa.Index("HNSW", D, "COSINE", initial_cap=N, m=16, ef_construction=200, ef_runtime=args.ef, build_threads=args.threads, dtype=args.dtype)
h.add_batch(hx)
h.flush()
tb = time.time() - t
print(f"build {N}x{D}: {tb:.1f}s = {N/tb:.0f} inserts/s, threads={args.threads or os.cpu_count()}, maxlevel={h.stats().max_level}", flush=True)
f = vsa.Index("FLAT", D, "COSINE", initial_cap=N)
f.add_batch(hx)
_, Lf, _ = f.search_batch(Q, 10)
if args.calibrate:
    for _ in range(2):
        f.search(Q[0], 10)
    print(f"calibration: flat_scan_kernel<1, ...> over {N} rows x {D * 4} B", flush=True)
for ef in (args.ef,) + ((256,) if not args.calibrate else ()):
    h.search_batch(Q[:64], 10, ef=ef)
    t = time.time()
    reps = 5
    for _ in range(reps):
        Dh, Lh, Nh = h.search_batch(Q, 10, ef=ef)
    dt = (time.time() - t) / reps
    st = h.stats()
    rec = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(Lh, Lf)) / (10.0 * len(Q))
    useful = (st.last_n_eval * (D * 4 + 4) + st.last_n_hops * 132)
    print(f"ef={ef}: {len(Q)/dt:.0f} QPS ({dt*1e3:.2f} ms per {len(Q)}-batch), recall@10={rec:.4f}, "
          f"n_eval/q={st.last_n_eval/len(Q):.0f} hops/q={st.last_n_hops/len(Q):.0f}, useful {useful/dt/1e9:.0f} GB/s", flush=True)
if args.tombstone:
    assert h.remove(N - 1) == 0
    h.flush()
    for ef in (args.ef, 256):
        h.search_batch(Q[:64], 10, ef=ef)
        t = time.time()
        for _ in range(5):
            Dh, Lh, Nh = h.search_batch(Q, 10, ef=ef)
        dt = (time.time() - t) / 5
        st = h.stats()
        useful = (st.last_n_eval * (D * 4 + 4) + st.last_n_hops * 132)
        print(f"one tombstone (HBM-frontier kernel, no filter) ef={ef}: {len(Q)/dt:.0f} QPS, n_eval/q={st.last_n_eval/len(Q):.0f}, useful {useful/dt/1e9:.0f} GB/s, "
              f"visited mode {st.last_visited_mode}", flush=True)
    # ... and with a 10 % allow-set on top (configs[4]'s shape): the filter's own cost
    rng = np.random.default_rng(5)
    bits = np.zeros((N + 63) // 64, np.uint64)
    allowed = np.flatnonzero(rng.random(N) < 0.1)
    np.bitwise_or.at(bits, allowed >> 6, np.uint64(1) << (allowed & 63).astype(np.uint64))
    for ef in (256,):
        h.search_batch(Q[:64], 10, ef=ef, allow=bits, allow_nbits=N)
        ts = []
        for _ in range(12):
            t = time.time()
            Dh, Lh, Nh = h.search_batch(Q, 10, ef=ef, allow=bits, allow_nbits=N)
            ts.append(time.time() - t)
        dt, best = sorted(ts)[len(ts) // 2], min(ts)
        st = h.stats()
        useful = (st.last_n_eval * (D * 4 + 4) + st.last_n_hops * 132)
        print(f"10 % allow-set ef={ef}: median {len(Q)/dt:.0f} QPS (best {len(Q)/best:.0f}), n_eval/q={st.last_n_eval/len(Q):.0f} hops/q={st.last_n_hops/len(Q):.0f}, "
              f"useful {useful/dt/1e9:.0f} GB/s (best {useful/best/1e9:.0f})", flush=True)
t = time.time()
for i in range(50):
    h.search(Q[i], 10, ef=args.ef)
print(f"single-query latency {1e3*(time.time()-t)/50:.3f} ms")
