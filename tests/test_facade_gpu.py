"""include/vk_algo.h -- the hnswlib-shaped facade of INTEGRATION.md -- driven by a C++ program the way
VectorFlat<float> / VectorHNSW<float> drive their `algo_`; its answers are compared with the oracle."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def parse(line):
    tag, *items = line.split()
    return [(int(a), int(b, 16)) for a, b in (it.split(":") for it in items)]


def test_facade_program_matches_oracle(oracle, tmp_path):
    import _pkg
    vsa = _pkg.vsa
    n, dim, k = 3000, 24, 5
    rng = np.random.default_rng(91)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((4, dim)).astype(np.float32)
    x.tofile(tmp_path / "rows.f32")
    q.tofile(tmp_path / "queries.f32")
    exe = tmp_path / "vk_algo_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", str(ROOT / "include"),
                           str(ROOT / "tests" / "helpers" / "vk_algo_check.cc"), "-o", str(exe),
                           "-L", str(vsa.LIB_PATH.parent), "-lvkindex", f"-Wl,-rpath,{vsa.LIB_PATH.parent}"])
    out = subprocess.run([str(exe), str(tmp_path / "rows.f32"), str(tmp_path / "queries.f32")], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = {" ".join(l.split()[:2]): l for l in out.stdout.splitlines()}
    of = oracle.Flat(dim, "L2", max_elements=n)
    of.add_many(x)
    oh = oracle.HNSW(dim, "L2", max_elements=n, M=16, ef_construction=100)
    oh.add_many(x)
    for name, o in (("FLAT", of), ("HNSW", oh)):
        assert lines[f"{name} count"].split()[2:] == ["3000", "capacity", "3048", "resizes", "2"]
        for i in range(4):
            od, ol = (o.search(q[i], k) if name == "FLAT" else o.search(q[i], k, ef=64))
            got = parse(lines[f"{name} knn{i}"].split(" ", 1)[1].replace(f"knn{i}", "x"))
            assert [g[0] for g in got] == ol.tolist()
            assert [g[1] for g in got] == od.view(np.uint32).tolist()
        even = parse(lines[f"{name} even"].split(" ", 1)[1].replace("even", "x"))
        assert len(even) == k and all(l % 2 == 0 for l, _ in even)
        if name == "HNSW":     # vector_hnsw.cc:336-340: cancelled and no partial results -> CancelledError
            assert "Search operation cancelled due to timeout" in lines[f"{name} cancel"]
        else:                  # VectorFlat::Search has no such branch: the scan just stops (bruteforce.h:125-129)
            assert lines[f"{name} cancel"] == "FLAT cancel no-throw"
        want_count = "2999" if name == "FLAT" else "3000"          # removePoint compacts, markDelete tombstones
        assert lines[f"{name} after-delete"].split()[3] == want_count
        assert parse(lines[f"{name} self1"].split(" ", 1)[1].replace("self1", "x"))[0] == (1, 0)
        pre = parse(lines[f"{name} prefilter"].split(" ", 1)[1].replace("prefilter", "x"))
        assert len(pre) == 3 and {l for l, _ in pre} <= {5, 9, 11, 2999, 1234, 77}
        assert lines[f"{name} knn0-reloaded"].split()[2:] != []     # the reloaded index answers
        # label 0 was deleted before the save: the reloaded answer equals the oracle's after the same delete
    of.remove(0)
    oh.mark_delete(0)
    for name, o in (("FLAT", of), ("HNSW", oh)):
        od, ol = (o.search(q[0], k) if name == "FLAT" else o.search(q[0], k, ef=64))
        got = parse(lines[f"{name} knn0-reloaded"].split(" ", 1)[1].replace("knn0-reloaded", "x"))
        assert [g[0] for g in got] == ol.tolist() and [g[1] for g in got] == od.view(np.uint32).tolist()


def test_vectorbase_adaptor_program_matches_oracle(oracle, tmp_path):
    """include/vk_vector_adaptor.h -- the VectorBase-derived binding (submit + completion, token polled by the caller,
    filter functor materialised, INFO counters, SaveIndex) -- driven through the mocked base class's entry points."""
    import _pkg
    vsa = _pkg.vsa
    n, dim, k = 3000, 24, 5
    rng = np.random.default_rng(92)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((3, dim)).astype(np.float32)
    x.tofile(tmp_path / "rows.f32")
    q.tofile(tmp_path / "queries.f32")
    exe = tmp_path / "adaptor_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", str(ROOT / "include"), "-I", str(ROOT / "tests" / "helpers"),
                           str(ROOT / "tests" / "helpers" / "adaptor_check.cc"), "-o", str(exe),
                           "-L", str(vsa.LIB_PATH.parent), "-lvkindex", "-lpthread", f"-Wl,-rpath,{vsa.LIB_PATH.parent}"])
    out = subprocess.run([str(exe), str(tmp_path / "rows.f32"), str(tmp_path / "queries.f32")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "adaptor ok" in out.stdout, out.stdout + out.stderr
    lines = out.stdout.splitlines()

    def find(prefix):
        return next(l for l in lines if l.startswith(prefix))
    of = oracle.Flat(dim, "L2", max_elements=n)
    of.add_many(x)
    oh = oracle.HNSW(dim, "L2", max_elements=n, M=16, ef_construction=100)
    oh.add_many(x)
    for name, o in (("flat", of), ("hnsw", oh)):
        assert find(f"{name} capacity").split()[2:] == ["1000", "->", "3048", "count", "3000", "max_label", "2999"]
        for i in range(3):
            od, ol = (o.search(q[i], k) if name == "flat" else o.search(q[i], k, ef=64))
            got = parse("x " + find(f"{name} q{i} ").split(" ", 2)[2])
            assert [g[0] for g in got] == ol.tolist() and [g[1] for g in got] == od.view(np.uint32).tolist()
            even = parse("x " + find(f"{name} q{i} even").split(" ", 3)[3])
            assert len(even) == k and all(l % 2 == 0 for l, _ in even)
        # a raised token: HNSW answers CancelledError without partial results (vector_hnsw.cc:327-329), an answer with them;
        # FLAT returns what its scan had (VectorFlat::Search has no such branch)
        assert find(f"{name} cancelled") == f"{name} cancelled: {'ok' if name == 'flat' else 'CancelledError'} / partial ok"
        assert find(f"{name} distance").split()[-3:] == ["00000000", "label", "0"]
        assert "stored row" in find(f"{name} GetValue") and "GetValue(2) null" in find(f"{name} GetValue") and find(f"{name} GetValue").endswith("IsVectorMatch 1")
        info = find(f"{name} info")
        assert "data_type=FLOAT32" in info and f"algorithm={name.upper()}" in info and "gpu_searches=" in info
        assert int(info.split("gpu_searches=")[1].split()[0]) >= 7
        assert find(f"{name} save/load").split()[-3:] == (["2999", "->", "2999"] if name == "flat" else ["3000", "->", "3000"])
        # ---- r05 ----
        for i in range(3):
            # SearchAsync answers what Search answers; a burst of 256 in flight all come back with the same answer
            assert find(f"{name} async q{i} ").split(" ", 3)[3] == find(f"{name} q{i} ").split(" ", 2)[2]
            # filters built from EntriesFetchers (union of two key lists; a list + per-key predicate; the cached handle through
            # SearchAsync) all equal the functor-built filter's answer: the k nearest EVEN ids
            even = find(f"{name} q{i} even").split(" ", 3)[3]
            for tag in ("fetch", "fetchpred", "fetchasync"):
                assert find(f"{name} {tag} q{i} ").split(" ", 3)[3] == even, tag
        assert find(f"{name} burst q0").split(" ", 3)[3] == find(f"{name} q0 ").split(" ", 2)[2]
        assert find(f"{name} burst 256") == f"{name} burst 256 in flight, 256 identical"
        # 1500 even ids allowed by both; first build = 1 built + 1 miss, the repeat is a cache hit with nothing built,
        # after a write phase the entry is stale: built again, one more miss
        assert find(f"{name} filters") == (f"{name} filters allowed 1500 / 1500; built 1 hits 0 misses 1 | cached: built +0 hits +1 | "
                                           "after a write phase: built +1 misses +1")
        # LoadFromRDB: the saved stream through the chunk iterator; same answer as the live index, vectors tracked again
        assert find(f"{name} loadrdb q0").split(" ", 3)[3] == find(f"{name} final q0").split(" ", 3)[3]
        assert find(f"{name} loadrdb count").split()[3:] == (["2999" if name == "flat" else "3000", "max_label", "2999", "GetValue(1)", "stored", "row"])
    # the token watcher relays a raised token (with or without a known deadline) and leaves an unregistered request alone
    assert find("watch raised") == "watch raised 1 1 untouched 1 within 100 ms"


def test_adaptor_loads_hand_assembled_reference_streams(oracle, tmp_path):
    """VectorGpuFlat / VectorGpuHNSW ::LoadFromRDB (vector_flat.cc:100-124, vector_hnsw.cc:135-166, called from
    index_schema.cc:175,200) on streams written out byte by byte in the reference's layout (tests/helpers/streams.py) and
    handed over through the mock SupplementalContentChunkIter: answers equal the oracle's on the same stream, the label
    counter resumes behind the largest loaded label (a loaded index reported max label 0, so every filter
    bitmap was one bit wide)."""
    import sys
    sys.path.insert(0, str(ROOT / "tests" / "helpers"))
    import streams
    import _pkg
    vsa = _pkg.vsa
    exe = tmp_path / "adaptor_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", str(ROOT / "include"), "-I", str(ROOT / "tests" / "helpers"),
                           str(ROOT / "tests" / "helpers" / "adaptor_check.cc"), "-o", str(exe),
                           "-L", str(vsa.LIB_PATH.parent), "-lvkindex", "-lpthread", f"-Wl,-rpath,{vsa.LIB_PATH.parent}"])
    rng = np.random.default_rng(7)
    # HNSW: the three-level ring graph
    rows, labels, l0, levels, upper = streams.ring_graph()
    n, dim, m = rows.shape[0], rows.shape[1], 4
    chunks = streams.hand_hnsw_stream(rows, labels, l0, levels, upper, ep=0, max_level=2, m=m)
    streams.write_chunk_file(tmp_path / "hnsw.chunks", chunks)
    Q = (rows[rng.integers(0, n, 6)] + 0.05 * rng.standard_normal((6, dim))).astype(np.float32)
    Q.tofile(tmp_path / "hq.f32")
    out = subprocess.run([str(exe), "load", "hnsw", str(tmp_path / "hnsw.chunks"), str(dim), str(m), str(tmp_path / "hq.f32"), "6"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "adaptor ok" in out.stdout, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    assert lines[0].split()[2:] == ["count", str(n), "max_label", str(1000 + n - 1), "capacity", str(n + 5)]
    o = oracle.HNSW.from_saved_chunks(chunks, dim, "L2", m, ef_construction=20)
    for i in range(6):
        e_d, e_l = o.search(Q[i], 5, ef=16)
        got = parse("x " + lines[1 + i].split(" ", 3)[3])
        assert [g[0] for g in got] == e_l.tolist() and [g[1] for g in got] == e_d.view(np.uint32).tolist()
    # FLAT: 300 rows under scattered labels
    n, dim = 300, 24
    frows = rng.standard_normal((n, dim)).astype(np.float32)
    flabels = rng.permutation(10 * n)[:n].astype(np.uint64)
    fchunks = streams.hand_flat_stream(frows, flabels, n + 100)
    streams.write_chunk_file(tmp_path / "flat.chunks", fchunks)
    FQ = rng.standard_normal((6, dim)).astype(np.float32)
    FQ.tofile(tmp_path / "fq.f32")
    out = subprocess.run([str(exe), "load", "flat", str(tmp_path / "flat.chunks"), str(dim), "16", str(tmp_path / "fq.f32"), "6"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "adaptor ok" in out.stdout, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    assert lines[0].split()[2:] == ["count", str(n), "max_label", str(int(flabels.max())), "capacity", str(n + 100)]
    of = oracle.Flat(dim, "L2", max_elements=n)
    of.add_many(frows, flabels)
    for i in range(6):
        e_d, e_l = of.search(FQ[i], 5)
        got = parse("x " + lines[1 + i].split(" ", 3)[3])
        assert [g[0] for g in got] == e_l.tolist() and [g[1] for g in got] == e_d.view(np.uint32).tolist()
