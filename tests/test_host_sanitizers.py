"""Host-side sanitizer runs (-m "not gpu").  The reference builds ASAN and TSAN variants of the module
(/root/reference CMakeLists.txt:26-41, cmake/Modules/valkey_search.cmake:63-74); the product's host side has hand-rolled
spin locks, per-request condition variables, an op log and per-shard enqueue threads, so the translation units that
need no device are built with -fsanitize={address,undefined} and -fsanitize=thread here and driven hard:
  coalescer.hpp        many callers, lanes, failing batches, coalescing switched off under load, and a FILTERED follower
                       cancelled while its batch is on the device and freeing its buffers on return
  hnsw_graph.cc        concurrent add / update / mark_delete from several threads, then a structural check
  row_store.cc         random writer phases of the op log over a host-memory stand-in for the HIP runtime
  sharded_index.cc     the multi-GPU pre-flight: the real fan-out (enqueue thread per device, peer-copy broadcast, gather by peer
                       copies and by RCCL all-gather, merge, device-resident filters on every device) over a model of the HIP
                       runtime with 8 VIRTUAL devices and asynchronous streams (tests/helpers/hip_virtual.cc) and fake shards --
                       16 device lists from 1 to 8 devices, even and uneven; every answer against the exact one, no runtime-model
                       violation; plus a MUTANT (the lanes do not wait for the queries) that TSAN must catch
Any sanitizer report fails the test (halt_on_error / non-zero exit code)."""
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "valkey-search_amd" / "csrc"
HELP = ROOT / "tests" / "helpers"
CXX = "/opt/rocm/lib/llvm/bin/clang++"
COMMON = ["-O1", "-g", "-std=c++17", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
          "-I", str(CSRC), "-I", str(ROOT / "include")]
SAN = {"asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], "tsan": ["-fsanitize=thread"]}
ENV = {"ASAN_OPTIONS": "halt_on_error=1:detect_leaks=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
       # (the one suppression is the reference's own updatePoint-vs-reader overlap, see the file)
       "TSAN_OPTIONS": f"halt_on_error=1:second_deadlock_stack=1:suppressions={HELP / 'tsan_suppressions.txt'}"}


def build(tmp, name, san, sources, extra=()):
    exe = tmp / f"{name}_{san}"
    subprocess.check_call([CXX, *COMMON, *SAN[san], *extra, *[str(s) for s in sources], "-lpthread", "-o", str(exe)])
    return exe


def run(exe, *args, timeout=600):
    p = subprocess.run([str(exe), *map(str, args)], env={**os.environ, **ENV}, capture_output=True, text=True, timeout=timeout)
    report = p.stdout[-2000:] + p.stderr[-6000:]
    assert p.returncode == 0, report
    assert "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, report
    assert "bad=0" in p.stdout, report


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_coalescer_under_sanitizers(tmp_path, san):
    exe = build(tmp_path, "coalescer", san, [HELP / "san_coalescer_main.cc"], extra=["-I", str(HELP)])
    run(exe, 1)


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_host_graph_builder_under_sanitizers(tmp_path, san):
    exe = build(tmp_path, "graph", san, [HELP / "san_graph_main.cc", CSRC / "hnsw_graph.cc", CSRC / "host_dist.cc"],
                )
    run(exe, 6, 3000 if san == "asan" else 2000)


def test_row_store_op_log_under_asan_ubsan(tmp_path):
    exe = build(tmp_path, "rowstore", "asan", [HELP / "san_rowstore_main.cc", HELP / "hip_stub.cc", CSRC / "row_store.cc"])
    run(exe, 60)


SHARDED_SOURCES = [HELP / "san_sharded_main.cc", HELP / "hip_virtual.cc", CSRC / "sharded_index.cc", CSRC / "index_common.cc", CSRC / "filter_set.cc"]
SHARDED_FLAGS = ["-I", str(HELP), "-rdynamic", "-ldl"]   # (-rdynamic: the gather dlopen()s its RCCL entry points from the program itself)
# (process-lifetime device pools -- the filter build lanes, the recycled bitmaps -- are never freed by design: no leak check here)
SHARDED_ENV = {"ASAN_OPTIONS": "halt_on_error=1:detect_leaks=0"}


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_sharded_fan_out_over_virtual_devices(tmp_path, san):
    exe = build(tmp_path, "sharded", san, SHARDED_SOURCES, extra=SHARDED_FLAGS)
    p = subprocess.run([str(exe), "1"], env={**os.environ, **ENV, **SHARDED_ENV}, capture_output=True, text=True, timeout=900)
    report = p.stdout[-2000:] + p.stderr[-6000:]
    assert p.returncode == 0 and "bad=0" in p.stdout and "worlds=16" in p.stdout, report
    assert "Sanitizer" not in p.stderr and "runtime error" not in p.stderr and "VIOLATION" not in p.stderr, report


def test_the_virtual_runtime_catches_a_missing_dependency(tmp_path):
    """The pre-flight has teeth: a copy of sharded_index.cc whose lanes do NOT wait for the `ready` event (queries uploaded on
    the serving stream) must be reported by TSAN as a data race between the upload and a shard's read."""
    src = (CSRC / "sharded_index.cc").read_text()
    line = "    VK_HIP_TRY(hipStreamWaitEvent(l.stream, mc->ready, 0));\n"
    assert src.count(line) == 1
    mutant = tmp_path / "mutant_sharded_index.cc"
    mutant.write_text(src.replace(line, "    /* MUTANT: the lane does not wait for the queries */\n"))
    sources = [mutant if s.name == "sharded_index.cc" else s for s in SHARDED_SOURCES]
    exe = build(tmp_path, "sharded_mutant", "tsan", sources, extra=SHARDED_FLAGS)
    p = subprocess.run([str(exe), "1"], env={**os.environ, "TSAN_OPTIONS": "halt_on_error=1"}, capture_output=True, text=True, timeout=900)
    assert p.returncode != 0 and "ThreadSanitizer: data race" in p.stderr, p.stdout[-1000:] + p.stderr[-3000:]
