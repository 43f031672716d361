"""Pins oracle/util.c's restatement of libstdc++ (heap sift order, minstd_rand0,
generate_canonical) against the real libstdc++ via a small C++ trace program.
These are what make a single-threaded oracle build follow hnswlib's exact path
(hnswalg.h:202-208 CompareByFirst heaps, :243-247 getRandomLevel)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent


@pytest.fixture(scope="module")
def trace_bin(tmp_path_factory):
    out = tmp_path_factory.mktemp("stdlib") / "stdlib_trace"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(out), str(HERE / "helpers" / "stdlib_trace.cc")])
    return out


class Pair(C.Structure):
    _fields_ = [("d", C.c_float), ("id", C.c_uint32)]


class Heap(C.Structure):
    _fields_ = [("v", C.POINTER(Pair)), ("n", C.c_size_t), ("cap", C.c_size_t)]


@pytest.mark.parametrize("seed", [1, 7, 12345])
def test_heap_pop_order_matches_libstdcxx(oracle, trace_bin, seed):
    ops = 4000
    want = [int(x) for x in subprocess.check_output([str(trace_bin), "heap", str(seed), str(ops)]).split()]
    lib = oracle.LIB
    lib.vko_heap_push.argtypes = [C.POINTER(Heap), C.c_float, C.c_uint32]
    lib.vko_heap_pop.argtypes = [C.POINTER(Heap)]
    h = Heap()
    lib.vko_heap_init(C.byref(h))
    x, nid, got = seed, 0, []

    def nxt():
        nonlocal x
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        return x >> 8

    for _ in range(ops):
        r = nxt()
        if (r % 3) != 0 or h.n == 0:
            d = float(nxt() % 7) * 0.5
            lib.vko_heap_push(C.byref(h), d, nid)
            nid += 1
        else:
            got.append(h.v[0].id)
            lib.vko_heap_pop(C.byref(h))
    while h.n:
        got.append(h.v[0].id)
        lib.vko_heap_pop(C.byref(h))
    lib.vko_heap_free(C.byref(h))
    assert got == want


class Gen(C.Structure):
    _fields_ = [("x", C.c_uint32)]


@pytest.mark.parametrize("seed,M", [(100, 16), (101, 16), (100, 10), (1, 48)])
def test_minstd_uniform_and_levels(oracle, trace_bin, seed, M):
    n = 3000
    lines = subprocess.check_output([str(trace_bin), "rng", str(seed), str(n), str(M)]).split()
    want_d = [float(x) for x in lines[:n]]
    want_l = [int(x) for x in lines[n:2 * n]]
    want_f = [np.float32(float(x)) for x in lines[2 * n:3 * n]]
    lib = oracle.LIB
    lib.vko_uniform01_double.restype = C.c_double
    lib.vko_uniform01_float.restype = C.c_float
    lib.vko_random_level.restype = C.c_int
    lib.vko_random_level.argtypes = [C.POINTER(Gen), C.c_double]
    g = Gen()
    lib.vko_minstd0_seed(C.byref(g), seed)
    got_d = [lib.vko_uniform01_double(C.byref(g)) for _ in range(n)]
    assert got_d == want_d
    lib.vko_minstd0_seed(C.byref(g), seed)
    mult = 1.0 / np.log(1.0 * M)
    got_l = [lib.vko_random_level(C.byref(g), mult) for _ in range(n)]
    assert got_l == want_l
    lib.vko_minstd0_seed(C.byref(g), seed + 1)
    got_f = [np.float32(lib.vko_uniform01_float(C.byref(g))) for _ in range(n)]
    assert got_f == want_f
