"""SaveIndex / LoadIndex streams that break: a writer that refuses a chunk (RDBChunkOutputStream::SaveChunk failing,
vector_flat.cc:203-222, vector_hnsw.cc:281-311) and a reader that runs dry at every position of the stream.  The call fails
with a status, nothing crashes, a failed save leaves the index as it was (the next save writes the same stream), a failed load
leaves no index behind -- for FLAT, HNSW (with tombstones) and the sharded index."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _build(vsa, algo, shards):
    rng = np.random.default_rng(31)
    n, dim = 700, 48
    x = rng.standard_normal((n, dim)).astype(np.float32)
    kw = dict(shard_devices=[0] * shards) if shards else {}
    g = vsa.Index(algo, dim, "L2", initial_cap=n, m=8, ef_construction=40, ef_runtime=40, **kw)
    g.add_batch(x)
    for i in (5, 77, 400):
        g.remove(i)
    g.flush()
    return g, x, dim


def _save_failing_at(vsa, g, fail_at):
    seen = []

    @vsa.WRITE_CHUNK
    def wr(_u, data, n):
        if len(seen) == fail_at:
            return 1
        seen.append(C.string_at(data, n))
        return 0

    rc = vsa.lib().vk_index_save(g._h, wr, None)
    return rc, seen


@pytest.mark.parametrize("algo,shards", [("FLAT", 0), ("HNSW", 0), ("FLAT", 3), ("HNSW", 3)])
def test_a_writer_that_refuses_a_chunk(vsa, algo, shards):
    g, x, dim = _build(vsa, algo, shards)
    good = g.save()
    L = len(good)
    assert L > 10
    for fail_at in sorted({0, 1, 2, 3, L // 3, L // 2, L - 2, L - 1}):
        rc, seen = _save_failing_at(vsa, g, fail_at)
        assert rc != vsa.VK_OK and len(seen) == fail_at, (fail_at, rc)
        assert seen == good[:fail_at]                       # what did go out is the stream's prefix
    assert g.save() == good                                  # ... and the index is as it was
    d, l = g.search(x[9], 3)
    assert l[0] == 9 and d[0] == 0.0


@pytest.mark.parametrize("algo,shards", [("FLAT", 0), ("HNSW", 0), ("FLAT", 3), ("HNSW", 3)])
def test_a_reader_that_runs_dry(vsa, algo, shards):
    g, x, dim = _build(vsa, algo, shards)
    good = g.save()
    L = len(good)
    kw = dict(shard_devices=[0] * shards) if shards else {}
    cuts = sorted(set(range(0, min(L, 12))) | {L // 3, L // 2, L - 3, L - 2, L - 1})
    for cut in cuts:
        with pytest.raises(vsa.VkError) as e:
            vsa.Index.load(good[:cut], algo, dim, "L2", m=8, ef_construction=40, ef_runtime=40, **kw)
        assert e.value.code in (vsa.VK_ERR_INTERNAL, vsa.VK_ERR_INVALID), (cut, e.value)
    # a chunk cut short in the middle (half a header, half an element)
    for j in (0, 1, L // 2, L - 1):
        if len(good[j]) < 2:
            continue
        broken = list(good)
        broken[j] = good[j][: len(good[j]) // 2]
        with pytest.raises(vsa.VkError):
            vsa.Index.load(broken, algo, dim, "L2", m=8, ef_construction=40, ef_runtime=40, **kw)
    # the whole stream still loads, into an index that answers like the one that wrote it
    h = vsa.Index.load(good, algo, dim, "L2", m=8, ef_construction=40, ef_runtime=40, **kw)
    assert h.stats().count == g.stats().count
    for i in (0, 9, 300, 699):
        a, b = g.search(x[i], 5), h.search(x[i], 5)
        assert a[1].tolist() == b[1].tolist() and a[0].view(np.uint32).tolist() == b[0].view(np.uint32).tolist()
