"""Host-only probe: scaling of the product's HNSW builder with threads (no GPU needed)."""
import ctypes as C, subprocess, sys, time, threading, os
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "valkey-search_amd" / "csrc"
out = Path("/tmp/libgraphshim.so")
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DVK_PROFILE_LOCKS",
                       "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(ROOT / "tests/helpers/graph_shim.cc"),
                       str(CSRC / "hnsw_graph.cc"), str(CSRC / "host_dist.cc"), "-lpthread", "-o", str(out)])
lib = C.CDLL(str(out))
lib.gs_new.restype = C.c_void_p
lib.gs_new.argtypes = [C.c_uint32, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
lib.gs_add.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
lib.gs_free.argtypes = [C.c_void_p]
N, D = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(0)
A = rng.standard_normal((D, 32)).astype(np.float32)
x = rng.standard_normal((N, 32)).astype(np.float32) @ A.T + 0.05 * rng.standard_normal((N, D)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
x = np.ascontiguousarray(x.astype(np.float32))
for T in [int(t) for t in sys.argv[3:]]:
    g = lib.gs_new(D, 0, N, 16, 200, 100, 0)
    lib.gs_add(g, x[0].ctypes.data, 0)
    nxt = [1]
    lock = threading.Lock()
    def work():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 64
            if i >= N: return
            for j in range(i, min(N, i + 64)):
                lib.gs_add(g, x.ctypes.data + j * D * 4, j)
    t = time.time()
    th = [threading.Thread(target=work) for _ in range(T)]
    [a.start() for a in th]; [a.join() for a in th]
    dt = time.time() - t
    lib.gs_spin_cycles.restype = C.c_uint64; lib.gs_spin_waits.restype = C.c_uint64
    print(f"threads={T}: {N/dt:.0f} inserts/s ({dt:.1f}s) spin waits={lib.gs_spin_waits()} cycles={lib.gs_spin_cycles()/1e9:.2f}G", flush=True)
    lib.gs_free(g)
