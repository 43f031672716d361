"""Run by tests/test_flat_filter_gpu.py in its own process with VKINDEX_LIB = libvkindex_exp.so (the -DVK_EXPERIMENTS
build): the four-fat-waves kernel (VK_FILTER_FAT=1, read per launch in that build) against the wave-specialised kernel."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402  (torch's HIP runtime first, see tests/conftest.py)

if torch.cuda.is_available():
    torch.cuda.init()
import _pkg  # noqa: E402

vsa = _pkg.vsa
assert "libvkindex_exp" in str(vsa.LIB_PATH), vsa.LIB_PATH


def unit(x):
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def same(a, b):
    (ad, al, an), (bd, bl, bn) = a, b
    assert an.tolist() == bn.tolist() and (al == bl).all() and (ad.view(np.uint32) == bd.view(np.uint32)).all()


rng = np.random.default_rng(512)
n, dim = 150_000, 128
centres = rng.standard_normal((60, dim)).astype(np.float32)
x = unit(centres[rng.integers(0, 60, n)] + 0.4 * rng.standard_normal((n, dim)).astype(np.float32))
x[30_000:42_000] = x[5]                                  # a query on 12 000 duplicates: spill chunks in the fat kernel too
Q = unit(centres[rng.integers(0, 60, 256)] + 0.4 * rng.standard_normal((256, dim)).astype(np.float32))
Q[9] = x[5]
small = {"filter-prepass-rows": 1024, "filter-min-rows": 32768}
for dtype in ("f32", "bf16"):
    f = vsa.Index("FLAT", dim, "COSINE", initial_cap=n, dtype=dtype, options=small)
    e = vsa.Index("FLAT", dim, "COSINE", initial_cap=n, dtype=dtype, options={"flat-filter": 0})
    f.add_batch(x)
    e.add_batch(x)
    ref = f.search_batch(Q, 10)
    c_ref = f.stats().last_filter_candidates
    same(ref, e.search_batch(Q, 10))
    os.environ["VK_FILTER_FAT"] = "1"
    got = f.search_batch(Q, 10)
    st = f.stats()
    os.environ.pop("VK_FILTER_FAT")
    same(got, ref)
    assert st.last_filter_fallback == 0
    if dtype == "f32":
        assert st.last_filter_candidates == c_ref
    else:                                                # (bf16 rows: the default kernel multiplies in bf16, this one in f16)
        f.set_option("filter-bf16-mfma", 0)
        f.search_batch(Q, 10)
        assert f.stats().last_filter_candidates == st.last_filter_candidates
print("fat kernel ok")
