"""Pins the oracle's distance kernels (oracle/dist_f32.c):
  * always: against tests/golden/simsimd_f32.npz, recorded from the reference's own
    SimSIMD 5.0.1 (tests/golden/gen_simsimd_golden.py);
  * where oracle/_ref exists (this container / any box it travelled to): live,
    bit for bit, against the compiled reference on fresh random inputs.
Bar: bit-exact (f64 kernel results, f32 bridge results)."""
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).resolve().parent / "golden" / "simsimd_f32.npz"


def test_golden_kernels_bit_exact(oracle):
    g = np.load(GOLD)
    for i, n in enumerate(g["n"]):
        a, b = g["a"][i, :n], g["b"][i, :n]
        for kern, fn in (("dot", oracle.dot), ("l2sq", oracle.l2sq)):
            for isa in ("haswell", "skylake"):
                got = np.float64(fn(a, b, isa)).view(np.uint64)
                want = g[f"{kern}_{isa}"][i]
                if isa == "haswell" and n % 8:
                    # the haswell kernels finish with a scalar f64 tail loop (dot.h:875-877);
                    # the reference TU is built with -ffast-math, which lets the compiler
                    # re-associate those adds, so its last bit is build dependent there
                    assert abs(int(got) - int(want)) <= 1, (kern, isa, int(n), i)
                else:
                    assert got == want, (kern, isa, int(n), i)


def test_golden_bridge_distances_follow_skylake_path(oracle):
    """The fixture host dispatched skylake kernels (capabilities has bit for skylake):
    InnerProductDistanceSimsimd / L2SqrSimsimd == oracle distance in skylake order."""
    g = np.load(GOLD)
    assert int(g["capabilities"]) & (1 << 11), "fixture host did not have the skylake capability"
    for i, n in enumerate(g["n"]):
        a, b = g["a"][i, :n], g["b"][i, :n]
        assert oracle.distance("IP", a, b, "skylake").view(np.uint32) == g["ip_bridge"][i]
        assert oracle.distance("L2", a, b, "skylake").view(np.uint32) == g["l2_bridge"][i]


def test_ip_wrapper_equals_single_f32_subtract(oracle):
    """(float)(1.0 - (double)dot) == 1.0f - dot: double rounding is innocuous for one
    add of two f32 values (53 >= 2*24+2), which the device code relies on."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(200000), rng.standard_normal(50000) * 1e-6,
                        1.0 + rng.standard_normal(50000) * 1e-7]).astype(np.float32)
    via_double = (1.0 - x.astype(np.float64)).astype(np.float32)
    direct = np.float32(1.0) - x
    assert np.array_equal(via_double.view(np.uint32), direct.view(np.uint32))


def test_live_against_compiled_reference(oracle):
    if not oracle.Ref.available():
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    R = oracle.Ref()
    rng = np.random.default_rng(99)
    for n in (1, 5, 16, 100, 128, 768, 771):
        for _ in range(25):
            a = rng.standard_normal(n).astype(np.float32)
            b = rng.standard_normal(n).astype(np.float32)
            for kern, fn in (("dot", oracle.dot), ("l2sq", oracle.l2sq)):
                for isa in ("haswell", "skylake"):
                    r = int(np.float64(R.kernel(kern, isa, a, b)).view(np.uint64))
                    o = int(np.float64(fn(a, b, isa)).view(np.uint64))
                    assert r == o or (isa == "haswell" and n % 8 and abs(r - o) <= 1)
            caps = R.capabilities()
            isa = "skylake" if caps & (1 << 11) else "haswell" if caps & (1 << 10) else "serial"
            if isa != "serial":
                assert R.distance("IP", a, b).view(np.uint32) == oracle.distance("IP", a, b, isa).view(np.uint32)
                assert R.distance("L2", a, b).view(np.uint32) == oracle.distance("L2", a, b, isa).view(np.uint32)
