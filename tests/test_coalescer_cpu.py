"""N1 on the CPU: csrc/coalescer.hpp driven by many threads against a fake index (tests/helpers/
coalescer_shim.cc).  Every request must be answered exactly once with its own answer; concurrent
requests of one (k, ef) lane must travel in shared device batches; lanes never mix; an error of a batch
reaches every request that travelled in it."""
import ctypes as C
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "valkey-search_amd" / "csrc"


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = tmp_path_factory.mktemp("co") / "libcoalescershim.so"
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-O2", "-std=c++17", "-fPIC", "-shared",
                           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", str(CSRC), "-I", str(ROOT / "include"),
                           str(ROOT / "tests" / "helpers" / "coalescer_shim.cc"), "-lpthread", "-o", str(out)])
    lib = C.CDLL(str(out))
    lib.coalescer_run.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    return lib


def run(lib, threads, per_thread, max_batch, max_wait_us, lanes=1, failing=0):
    out = (C.c_uint64 * 8)()
    bad = lib.coalescer_run(threads, per_thread, max_batch, max_wait_us, lanes, failing, out)
    return bad, {"calls": out[0], "queries": out[1], "max_batch": out[2], "batches": out[3], "co_queries": out[4]}


def test_every_request_gets_its_own_answer_and_batches_form(shim):
    bad, st = run(shim, threads=48, per_thread=20, max_batch=16, max_wait_us=2000)
    assert bad == 0
    cancelled = sum(1 for i in range(48 * 20) if i % 17 == 5)
    assert st["queries"] == 48 * 20 - cancelled == st["co_queries"]   # each live request served exactly once
    assert st["calls"] == st["batches"] < 48 * 20                # ... and not one device pass per request
    assert 1 < st["max_batch"] <= 16


def test_lanes_by_k_and_ef_do_not_mix(shim):
    # three lanes (k = 1, 2, 3 with ef = 100, 101, 102); the fake index derives its answer from k and ef of
    # the batch it is called with, so a request riding in another lane's batch would be caught
    bad, st = run(shim, threads=30, per_thread=30, max_batch=8, max_wait_us=1000, lanes=3)
    assert bad == 0 and st["queries"] == 900 - sum(1 for i in range(900) if i % 17 == 5) and st["max_batch"] <= 8


def test_a_failing_batch_fails_every_rider_and_nobody_else(shim):
    bad, st = run(shim, threads=24, per_thread=16, max_batch=8, max_wait_us=1000, lanes=4, failing=1)
    live = sum(1 for i in range(24 * 16) if not (i % 17 == 5 and i % 4 != 3))
    assert bad == 0 and st["queries"] == live


def test_single_caller_and_max_batch_one(shim):
    bad, st = run(shim, threads=1, per_thread=10, max_batch=8, max_wait_us=100)
    assert bad == 0 and st["calls"] == 9 and st["max_batch"] == 1          # (request 5 is the cancelled one)
    bad, st = run(shim, threads=8, per_thread=10, max_batch=2, max_wait_us=100)
    assert bad == 0 and st["queries"] == 80 - 5 and st["max_batch"] <= 2


def test_coalescing_switched_off_while_requests_are_queued(shim):
    """vk_index_set_coalescing(ix, 0, ..) under load: queued requests are still drained (a leader always takes at least
    its own), nobody spins forever"""
    bad, st = run(shim, threads=32, per_thread=20, max_batch=16, max_wait_us=2000, failing=2)
    assert bad == 0 and st["queries"] == 640 - sum(1 for i in range(640) if i % 17 == 5)


def test_a_lone_caller_leaves_when_arrivals_have_stopped_and_many_callers_are_woken_by_name(shim):
    """A caller alone in its lane is held for a quarter of the window at most (200 us), not for the window; and a
    thousand waiting callers cost a batch no more than a handful do (each waits on its own condition variable)."""
    import time
    t0 = time.perf_counter()
    bad, st = run(shim, threads=1, per_thread=40, max_batch=64, max_wait_us=50_000)     # 40 calls, 50 ms window each
    dt = time.perf_counter() - t0
    assert bad == 0 and st["max_batch"] == 1
    assert dt < 0.5, dt                                   # 40 x 50 ms would be 2 s
    t0 = time.perf_counter()
    bad, st = run(shim, threads=1000, per_thread=4, max_batch=256, max_wait_us=500)
    dt = time.perf_counter() - t0
    assert bad == 0 and st["queries"] == 4000 - sum(1 for i in range(4000) if i % 17 == 5)
    assert st["batches"] < 400 and dt < 20.0, (st, dt)
