"""The matrix-core accumulation error the candidate filter's margin ALLOWS, measured on the instruction itself.

flat_filter.hip ("Error bound") allows a chain of v_mfma_f32_32x32x16_f16 / _bf16 over D elements
    |acc - sum_i x_i q_i|  <=  D * 2^-22 * sum_i |x_i q_i|
(4 ulp per accumulated term; the products of two f16 / bf16 values are exact in f32).  tests/helpers/mfma_probe.hip runs
exactly the consumers' chain (first K-step into the constant 0, same operand layout) on operands chosen here to be as
unkind as the formats allow -- exponent spreads across the whole f16 range, sums that cancel to nothing, huge terms
followed by hundreds of tiny ones, subnormal inputs -- and the result is compared with the exact dot product of the very
same 16-bit values in f64.  The measured worst case is printed (and recorded in DESIGN.md); the assertion is the allowance
the margin is built from."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent / "helpers"
LIB = HERE / "libmfmaprobe.so"


def build_probe():
    src = HERE / "mfma_probe.hip"
    if not LIB.exists() or LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", str(src), "-o", str(LIB)])
    return LIB


@pytest.fixture(scope="module")
def probe():
    lib = C.CDLL(str(build_probe()))
    lib.mfma_probe_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    lib.mfma_probe_chain.restype = C.c_int

    def run(a_bits, b_bits, bf16):
        n, _, K = a_bits.shape
        a_bits, b_bits = np.ascontiguousarray(a_bits, np.uint16), np.ascontiguousarray(b_bits, np.uint16)
        out = np.empty((n, 32, 32), np.float32)
        rc = lib.mfma_probe_chain(a_bits.ctypes.data, b_bits.ctypes.data, n, K, 1 if bf16 else 0, out.ctypes.data)
        assert rc == 0, rc
        return out
    return run


def to_bits(x, bf16):
    """f32 values -> the 16-bit patterns the kernels hold (round to nearest even) and the values those patterns mean."""
    x = np.asarray(x, np.float32)
    if bf16:
        u = x.view(np.uint32)
        b = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
        return b, (b.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    h = x.astype(np.float16)
    return h.view(np.uint16), h.astype(np.float64)


def check(run, a, b, bf16, what):
    ab, av = to_bits(a, bf16)
    bb, bv = to_bits(b, bf16)
    got = run(ab, bb, bf16).astype(np.float64)
    exact = np.einsum("nik,njk->nij", av, bv)
    mass = np.einsum("nik,njk->nij", np.abs(av), np.abs(bv))
    K = a.shape[2]
    err = np.abs(got - exact)
    # the f32 result itself is rounded: half an ulp of |exact| is not accumulation error
    slack = np.abs(exact) * 2.0 ** -24
    allow = K * 2.0 ** -22 * mass + slack + 1e-300
    worst = float(np.max((err - slack).clip(0) / (mass * 2.0 ** -24 + 1e-300)))   # in f32 ulps of sum |x_i q_i|
    print(f"{what} ({'bf16' if bf16 else 'f16'}, K={K}): worst error {worst:.3f} ulp of sum|x q| (allowed {4 * K})")
    assert np.isfinite(got).all(), what
    assert (err <= allow).all(), (what, worst)
    return worst


def signs(rng, shape):
    return rng.choice(np.array([-1.0, 1.0], np.float32), shape)


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("K", [64, 768, 1536])
def test_accumulation_error_is_within_the_margins_allowance(probe, bf16, K):
    rng = np.random.default_rng(7 + K + bf16)
    n = 24
    worst = 0.0
    # 1. ordinary data
    a = rng.standard_normal((n, 32, K)).astype(np.float32)
    b = rng.standard_normal((n, 32, K)).astype(np.float32)
    worst = max(worst, check(probe, a, b, bf16, "gaussian"))
    # 2. exponents spread over the format's range (products stay finite in f32: |e_a + e_b| <= 30 for f16, 60 for bf16)
    span = 15 if not bf16 else 30
    a = (signs(rng, (n, 32, K)) * np.exp2(rng.integers(-span + 1, span, (n, 32, K)))).astype(np.float32) * (1 + rng.random((n, 32, K), np.float32))
    b = (signs(rng, (n, 32, K)) * np.exp2(rng.integers(-span + 1, span, (n, 32, K)))).astype(np.float32) * (1 + rng.random((n, 32, K), np.float32))
    if not bf16:
        a, b = np.clip(a, -60000, 60000), np.clip(b, -60000, 60000)
    worst = max(worst, check(probe, a, b, bf16, "exponent spread"))
    # 3. cancellation: every other term undoes the one before it exactly, with small terms scattered in between
    a = rng.standard_normal((n, 32, K)).astype(np.float32) * 1e-3
    b = np.ones((n, 32, K), np.float32)
    big = (1000.0 * (1 + rng.random((n, 32, K // 4), np.float32))).astype(np.float32)
    a[:, :, 0::4] = big
    a[:, :, 2::4] = -big
    worst = max(worst, check(probe, a, b, bf16, "cancellation"))
    # 4. one huge term first, then hundreds of terms far below its ulp (absorbed one by one if the accumulator rounds each)
    a = np.full((n, 32, K), 2.0 ** -10, np.float32) * (1 + rng.random((n, 32, K), np.float32))
    a[:, :, 0] = 2.0 ** 14
    b = np.ones((n, 32, K), np.float32)
    b[:, :, 0] = 2.0 ** 1
    worst = max(worst, check(probe, a, b, bf16, "huge then tiny"))
    #    ... and the other way round: the tiny terms first, the huge one last
    worst = max(worst, check(probe, a[:, :, ::-1], b[:, :, ::-1], bf16, "tiny then huge"))
    # 5. same-sign terms of equal size (the sum's ulp grows under the terms: round-to-nearest vs truncation shows here)
    a = np.full((n, 32, K), 1.0, np.float32) * (1 + 2.0 ** -7 * rng.integers(0, 2, (n, 32, K)))
    b = np.full((n, 32, K), 1.0, np.float32) * (1 + 2.0 ** -7 * rng.integers(0, 2, (n, 32, K)))
    worst = max(worst, check(probe, a, b, bf16, "equal same-sign terms"))
    assert worst <= 4 * K


def test_f16_subnormal_operands_are_multiplied_not_flushed(probe):
    """The margin's absolute term (2^-25 per element) is the ROUNDING of a value into the f16 subnormal grid.  It does not
    cover an instruction that flushes subnormal operands to zero (that would be up to 2^-14 per element): measured here."""
    rng = np.random.default_rng(99)
    n, K = 8, 768
    sub = (rng.integers(1, 1024, (n, 32, K)).astype(np.float32) * np.float32(2.0 ** -24)) * signs(rng, (n, 32, K))
    big = (rng.standard_normal((n, 32, K)).astype(np.float32) * 100).astype(np.float32)
    # subnormal rows against ordinary queries, and the other way round
    for a, b, what in ((sub, big, "subnormal rows"), (big, sub, "subnormal queries"), (sub, sub * np.float32(2.0 ** 20), "both small")):
        ab, av = to_bits(a, False)
        bb, bv = to_bits(b, False)
        assert (np.abs(av) < 2.0 ** -14).all() or (np.abs(bv) < 2.0 ** -14).all()
        got = probe(ab, bb, False).astype(np.float64)
        exact = np.einsum("nik,njk->nij", av, bv)
        mass = np.einsum("nik,njk->nij", np.abs(av), np.abs(bv))
        assert (mass > 0).all()
        err = np.abs(got - exact)
        # a flush would leave got == 0 and err == |exact| ~ sqrt(K) * typical term; the allowance is far below that
        assert (err <= K * 2.0 ** -22 * mass + np.abs(exact) * 2.0 ** -24 + 2.0 ** -149).all(), what
        assert np.count_nonzero(got) > 0.99 * got.size, what
