"""Generates tests/golden/simsimd_f32.npz from the REAL reference SimSIMD 5.0.1
(oracle/_ref/libsimsimd_ref.so = /root/reference/third_party/simsimd/c/lib.c compiled
with the reference's flags, see oracle/Makefile).  Run only where /root/reference
exists:  python tests/golden/gen_simsimd_golden.py

Fixture = data only: seeded inputs, and for each the f64 bit pattern returned by
simsimd_{dot,l2sq}_f32_{haswell,skylake} plus the f32 bit pattern returned by the
hnswlib bridge functions InnerProductDistanceSimsimd / L2SqrSimsimd
(third_party/hnswlib/simsimd.h:16-34) as dispatched on the generating host.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle as O  # noqa: E402

O.build(ref=True)
R = O.Ref()
rng = np.random.default_rng(20240921)
sizes = [1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 100, 127, 128, 129, 255, 256, 384, 767, 768, 769, 1536]
a_all, b_all, n_all = [], [], []
out = {f"{k}_{isa}": [] for k in ("dot", "l2sq") for isa in ("haswell", "skylake")}
out["ip_bridge"], out["l2_bridge"] = [], []
for n in sizes:
    for kind in range(4):
        if kind == 0:
            a = rng.standard_normal(n); b = rng.standard_normal(n)
        elif kind == 1:   # unit vectors (the COSINE case)
            a = rng.standard_normal(n); a /= np.linalg.norm(a); b = rng.standard_normal(n); b /= np.linalg.norm(b)
        elif kind == 2:   # wide dynamic range, cancellation
            a = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n); b = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n)
        else:             # identical vectors (distance 0 / self match)
            a = rng.standard_normal(n); b = a.copy()
        a = a.astype(np.float32); b = b.astype(np.float32)
        pa = np.zeros(1536, np.float32); pb = np.zeros(1536, np.float32)
        pa[:n] = a; pb[:n] = b
        a_all.append(pa); b_all.append(pb); n_all.append(n)
        for k in ("dot", "l2sq"):
            for isa in ("haswell", "skylake"):
                out[f"{k}_{isa}"].append(np.float64(R.kernel(k, isa, a, b)).view(np.uint64))
        out["ip_bridge"].append(R.distance("IP", a, b).view(np.uint32))
        out["l2_bridge"].append(R.distance("L2", a, b).view(np.uint32))
np.savez_compressed(Path(__file__).with_name("simsimd_f32.npz"), a=np.stack(a_all), b=np.stack(b_all),
                    n=np.array(n_all, np.int64), capabilities=np.uint32(R.capabilities()),
                    **{k: np.array(v) for k, v in out.items()})
print("wrote", len(n_all), "cases; capabilities 0x%x" % R.capabilities())
