"""LoadIndex validation (hnswalg.h:872-1139 loadCheck) -- the reference's corruption tests
(testing/vector_test.cc:1004-1203) replayed against vk_index_load: every mutation of a valid chunk
stream must be rejected with the reference's message; valid streams round-trip byte for byte."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M, DIM, N = 4, 20, 96          # small M => several layers within a few dozen nodes (mult = 1/ln 4)
U32 = 4
STRIDE = M * U32 + U32          # upper-level list
LINKS0 = 2 * M * U32 + U32      # level-0 list
VEC = DIM * 4
LABEL_OFF = LINKS0 + VEC
ELEM = LABEL_OFF + 8


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


# ---- a minimal protobuf codec for HNSWIndexHeader (third_party/hnswlib/index.proto) ----------------
def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def parse_header(buf):
    f, i = {}, 0
    while i < len(buf):
        key, sh = 0, 0
        while True:
            b = buf[i]; i += 1
            key |= (b & 0x7F) << sh; sh += 7
            if not b & 0x80:
                break
        field, wire = key >> 3, key & 7
        if wire == 0:
            v, sh = 0, 0
            while True:
                b = buf[i]; i += 1
                v |= (b & 0x7F) << sh; sh += 7
                if not b & 0x80:
                    break
            f[field] = v
        elif wire == 1:
            f[field] = struct.unpack("<d", buf[i:i + 8])[0]; i += 8
        else:
            raise AssertionError("unexpected wire type")
    return f


def build_header(f):
    out = bytearray()
    for field in sorted(f):
        v = f[field]
        if isinstance(v, float):
            out += _varint(field << 3 | 1) + struct.pack("<d", v)
        else:
            out += _varint(field << 3) + _varint(v)
    return bytes(out)


# field numbers (index.proto): 1 offset_level_0, 2 max_elements, 3 curr_element_count,
# 4 serialize_size_data_per_element, 7 max_level, 8 enterpoint_node, 9 max_m, 10 max_m_0, 11 m, 12 mult
F_OFF0, F_MAXEL, F_CUR, F_SIZE, F_MAXLEVEL, F_EP, F_MAXM, F_MAXM0, F_M, F_MULT = 1, 2, 3, 4, 7, 8, 9, 10, 11, 12


@pytest.fixture(scope="module")
def golden(vsa):
    """A valid multi-layer stream written by the library itself (its graph equals the oracle's,
    tests/test_hnsw_gpu.py), plus the chunk positions AnalyzeGolden (vector_test.cc:899-925) derives."""
    rng = np.random.default_rng(7)
    x = rng.standard_normal((N, DIM)).astype(np.float32)
    g = vsa.Index("HNSW", DIM, "L2", initial_cap=128, m=M, ef_construction=20, build_threads=1)
    for i in range(N):
        g.add(i, x[i])
    chunks = g.save()
    h = parse_header(chunks[0])
    assert h[F_CUR] == N and h[F_M] == M and h[F_MAXM0] == 2 * M
    max_level = struct.unpack("<i", struct.pack("<I", h[F_MAXLEVEL] & 0xFFFFFFFF))[0]
    assert max_level >= 2, "pick another N/M: the corruption cases need an entry point on level >= 2"
    size_chunk, data_chunk, idx = [], [], 1 + N
    for i in range(N):
        size_chunk.append(idx)
        (lls,) = struct.unpack("<Q", chunks[idx])
        idx += 1
        if lls:
            data_chunk.append(idx); idx += 1
        else:
            data_chunk.append(-1)
    assert idx == len(chunks)
    return {"chunks": chunks, "ep": h[F_EP], "max_level": max_level, "size_chunk": size_chunk,
            "data_chunk": data_chunk, "header": h}


def load(vsa, chunks, **kw):
    return vsa.Index.load(list(chunks), "HNSW", DIM, "L2", initial_cap=128, m=kw.get("m", M), ef_construction=20)


def expect_reject(vsa, chunks, substr):
    with pytest.raises(vsa.VkError) as e:
        load(vsa, chunks)
    assert "HNSW index load validation failed" in e.value.msg
    assert substr in e.value.msg, e.value.msg


def with_header_f(golden, field, value):
    h = dict(golden["header"])
    h[field] = value
    c = list(golden["chunks"])
    c[0] = build_header(h)
    return c


def poke(chunk, off, fmt, v):
    b = bytearray(chunk)
    struct.pack_into(fmt, b, off, v)
    return bytes(b)


# ---- happy path (vector_test.cc:1004-1029) -----------------------------------------------------------
def test_load_validates_empty_and_single(vsa):
    g = vsa.Index("HNSW", DIM, "L2", initial_cap=32, m=M, ef_construction=20)
    chunks = g.save()
    assert len(chunks) == 1
    assert load(vsa, chunks).stats().count == 0
    g.add(0, np.ones(DIM, np.float32))
    chunks = g.save()
    lvl = g.stats().max_level                     # the first draw of minstd_rand0(100) at this M
    assert len(chunks) == 3 + (1 if lvl > 0 else 0)   # header, element, link-list size [, upper lists]
    g2 = load(vsa, chunks)
    assert g2.stats().count == 1 and g2.stats().max_level == lvl


def test_round_trip_identity(vsa, golden):
    g2 = load(vsa, golden["chunks"])
    st = g2.stats()
    assert st.count == N and st.max_level == golden["max_level"] and st.entry_point == golden["ep"]
    assert g2.save() == golden["chunks"]          # save -> load -> save is byte-identical


# ---- header corruption (vector_test.cc:1033-1090) ------------------------------------------------------
def test_reject_header_fields(vsa, golden):
    expect_reject(vsa, with_header_f(golden, F_M, M + 1), "header M does not match")
    expect_reject(vsa, with_header_f(golden, F_MAXM0, 2 * M + 1), "maxM0 does not equal 2*M")
    expect_reject(vsa, with_header_f(golden, F_EP, N), "enterpoint_node is out of range")
    expect_reject(vsa, with_header_f(golden, F_MAXLEVEL, 1000), "max_level exceeds the element count")
    expect_reject(vsa, with_header_f(golden, F_SIZE, golden["header"][F_SIZE] + 4),
                  "serialized element size is inconsistent")
    expect_reject(vsa, with_header_f(golden, F_OFF0, 8), "offset_level_0 must be 0")
    expect_reject(vsa, with_header_f(golden, F_MULT, 0.5), "mult is inconsistent with M")



# ---- level-0 corruption (vector_test.cc:1094-1120) ------------------------------------------------------
def test_reject_level0(vsa, golden):
    c = list(golden["chunks"])
    c[1] = c[1][:-1]
    expect_reject(vsa, c, "level-0 element chunk has the wrong size")
    c = list(golden["chunks"])
    c[1] = poke(c[1], 0, "<H", 2 * M + 1)
    expect_reject(vsa, c, "level-0 neighbor count exceeds 2*M")
    c = list(golden["chunks"])
    c[2] = poke(poke(c[2], 0, "<H", 1), U32, "<I", 9999)
    expect_reject(vsa, c, "level-0 neighbor id out of range")
    c = list(golden["chunks"])
    (label0,) = struct.unpack_from("<Q", c[1], LABEL_OFF)
    c[3] = poke(c[3], LABEL_OFF, "<Q", label0)
    expect_reject(vsa, c, "duplicate live label in index")


# ---- upper-level corruption (vector_test.cc:1123-1188) ----------------------------------------------------
def test_reject_upper_levels(vsa, golden):
    ep, sc, dc = golden["ep"], golden["size_chunk"], golden["data_chunk"]
    c = list(golden["chunks"])
    c[sc[ep]] = c[sc[ep]][:4]
    expect_reject(vsa, c, "link-list size chunk has the wrong size")
    c = list(golden["chunks"])
    c[sc[ep]] = struct.pack("<Q", golden["max_level"] * STRIDE + 1)
    expect_reject(vsa, c, "not a multiple of the stride")
    c = list(golden["chunks"])
    c[sc[ep]] = struct.pack("<Q", (golden["max_level"] + 1) * STRIDE)
    expect_reject(vsa, c, "element level exceeds max_level")
    c = list(golden["chunks"])
    c[dc[ep]] = c[dc[ep]][:-1]
    expect_reject(vsa, c, "upper-level link-list chunk has the wrong")
    c = list(golden["chunks"])
    c[dc[ep]] = poke(c[dc[ep]], 0, "<H", M + 1)
    expect_reject(vsa, c, "upper-level neighbor count exceeds M")
    c = list(golden["chunks"])
    c[dc[ep]] = poke(poke(c[dc[ep]], 0, "<H", 1), U32, "<I", 9999)
    expect_reject(vsa, c, "upper-level neighbor id out of range")
    # a level-2 edge to a node that only exists on level 0
    low = next(i for i in range(N) if dc[i] == -1)
    c = list(golden["chunks"])
    c[dc[ep]] = poke(poke(c[dc[ep]], STRIDE, "<H", 1), STRIDE + U32, "<I", low)
    expect_reject(vsa, c, "neighbor is absent at that level")
    # the entry point demoted below max_level
    c = list(golden["chunks"])
    c[sc[ep]] = struct.pack("<Q", STRIDE)
    c[dc[ep]] = c[dc[ep]][:STRIDE]
    expect_reject(vsa, c, "enterpoint node is not at max_level")


# ---- kill switch (vector_test.cc:1191-1203, valkey_search_options.cc:156-162) -----------------------------
def load_unvalidated(vsa, chunks):
    return vsa.Index.load(list(chunks), "HNSW", DIM, "L2", initial_cap=128, m=M, ef_construction=20, load_skip_validation=True)


def test_validation_disabled_bypasses_the_graph_invariant_checks(vsa, golden):
    """`hnsw-validation-enable no`: a self-loop is rejected by default and loads under the switch (it is not memory-unsafe);
    what the device kernels could not survive -- a neighbour id out of range, a wrong chunk size -- is refused either way."""
    c = list(golden["chunks"])
    c[2] = poke(poke(c[2], 0, "<H", 1), U32, "<I", 1)          # element 1: count = 1, neighbour[0] == itself
    expect_reject(vsa, c, "level-0 self-loop")
    g = load_unvalidated(vsa, c)
    assert g.stats().count == N
    d, l = g.search(np.zeros(DIM, np.float32), 5, ef=32)         # ... and the graph with the self-loop is searchable
    assert len(l) == 5
    ep, dc = golden["ep"], golden["data_chunk"]
    c = list(golden["chunks"])
    c[dc[ep]] = poke(poke(c[dc[ep]], 0, "<H", 1), U32, "<I", ep)   # upper-level self-loop at the entry point
    expect_reject(vsa, c, "upper-level self-loop")
    assert load_unvalidated(vsa, c).stats().count == N
    assert load_unvalidated(vsa, with_header_f(golden, F_MULT, 0.5)).stats().count == N
    c = list(golden["chunks"])
    c[2] = poke(poke(c[2], 0, "<H", 1), U32, "<I", 9999)
    with pytest.raises(vsa.VkError) as e:
        load_unvalidated(vsa, c)
    assert "level-0 neighbor id out of range" in e.value.msg
    c = list(golden["chunks"])
    c[1] = c[1][:-1]
    with pytest.raises(vsa.VkError):
        load_unvalidated(vsa, c)


# ---- an old snapshot's header (vector_test.cc:764-800) ------------------------------------------------------
def test_old_snapshot_header_with_unpadded_offsets_loads(vsa, golden):
    """Snapshots written before the level-0 record was padded to 8 bytes carry offset_data = 4 + 2*M*4 (132 at M = 16)
    and label_offset = offset_data + 8.  LoadIndex recomputes the geometry from M and ignores both (hnswalg.h:921-930);
    so does vk_index_load -- whatever the two fields hold, the graph and the answers are the same."""
    F_LABEL_OFF, F_OFF_DATA = 5, 6
    ref = load(vsa, golden["chunks"])
    q = np.random.default_rng(3).standard_normal(DIM).astype(np.float32)
    rd, rl = ref.search(q, 10, ef=64)
    unpadded = LINKS0                                              # 4 + 2*M*4: not a multiple of 8 for even M
    assert unpadded % 8 != 0
    for off_data, label_off in ((unpadded, unpadded + 8), (0, 0), (12345, 7)):
        h = dict(golden["header"])
        h[F_OFF_DATA], h[F_LABEL_OFF] = off_data, label_off
        c = list(golden["chunks"])
        c[0] = build_header(h)
        g = load(vsa, c)
        assert g.stats().count == N
        d, l = g.search(q, 10, ef=64)
        assert l.tolist() == rl.tolist() and d.view(np.uint32).tolist() == rd.view(np.uint32).tolist()
        assert g.save()[1:] == golden["chunks"][1:]               # everything behind the header byte for byte
        # ... and the header it writes is the reference's own (SaveIndex, hnswalg.h:815-816): offset_data = the size of
        # the serialized level-0 list, label_offset = the in-memory (padded) one
        w = parse_header(g.save()[0])
        assert w[F_OFF_DATA] == LINKS0 and w[F_LABEL_OFF] == ((LINKS0 + 7) & ~7) + 8
    # an empty old snapshot, as in the reference's test
    h = {F_OFF0: 0, F_MAXEL: 16, F_CUR: 0, F_SIZE: unpadded + VEC + 8, F_LABEL_OFF: unpadded + 8, F_OFF_DATA: unpadded,
         F_MAXLEVEL: (1 << 64) - 1, F_EP: 0, F_MAXM: M, F_MAXM0: 2 * M, F_M: M, F_MULT: 1.0 / np.log(M), 13: 20}
    assert load(vsa, [build_header(h)]).stats().count == 0


# ---- streams assembled BY HAND from the documented layout, not from the product's own save ------------------------
def _hand_hnsw_stream(rows, labels, l0_lists, levels, upper_lists, ep, max_level, m):
    """hnswalg.h:808-865: header chunk; per element [count u16 | flags u16 | 2*M neighbour ids | vector | label u64]; then
    per element a u64 size chunk and, if non-zero, `level` lists of [count u16 | flags u16 | M neighbour ids]."""
    n, dim = rows.shape
    sl0, slu = (2 * m + 1) * 4, (m + 1) * 4
    off_data = (sl0 + 7) & ~7
    hdr = {2: n + 5, 3: n, 4: sl0 + dim * 4 + 8, 5: off_data + 8, 6: sl0, 7: max_level & ((1 << 64) - 1), 8: ep, 9: m, 10: 2 * m, 11: m,
           12: 1.0 / np.log(m), 13: 20}
    chunks = [build_header({k: v for k, v in hdr.items() if v != 0})]      # (proto3 does not serialize zero-valued fields)
    for i in range(n):
        rec = bytearray(sl0)
        struct.pack_into("<HH", rec, 0, len(l0_lists[i]), 0)
        for j, e in enumerate(l0_lists[i]):
            struct.pack_into("<I", rec, 4 + 4 * j, e)
        chunks.append(bytes(rec) + rows[i].astype("<f4").tobytes() + struct.pack("<Q", int(labels[i])))
    for i in range(n):
        chunks.append(struct.pack("<Q", levels[i] * slu))
        if levels[i]:
            blk = bytearray(levels[i] * slu)
            for lv in range(levels[i]):
                lst = upper_lists[i][lv]
                struct.pack_into("<HH", blk, lv * slu, len(lst), 0)
                for j, e in enumerate(lst):
                    struct.pack_into("<I", blk, lv * slu + 4 + 4 * j, e)
            chunks.append(bytes(blk))
    return chunks


def test_hand_assembled_hnsw_stream_loads_in_product_and_oracle(vsa, oracle):
    """A three-level graph of 40 nodes written out byte by byte in the test: nodes on a ring at level 0 (each linked to its
    four nearest ring neighbours), every fifth node also at level 1, every twentieth at level 2.  The product and the
    oracle both load it, agree on every answer (ids and distance bits), and the product's own save reproduces the bytes."""
    rng = np.random.default_rng(40)
    n, dim, m = 40, 12, 4
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    rows = np.zeros((n, dim), np.float32)
    rows[:, 0], rows[:, 1] = np.cos(ang), np.sin(ang)
    rows[:, 2:] = 0.01 * rng.standard_normal((n, dim - 2)).astype(np.float32)
    labels = 1000 + np.arange(n)
    l0 = [[(i + d) % n for d in (1, -1, 2, -2)] for i in range(n)]
    levels = [2 if i % 20 == 0 else 1 if i % 5 == 0 else 0 for i in range(n)]
    l1_nodes = [i for i in range(n) if levels[i] >= 1]
    l2_nodes = [i for i in range(n) if levels[i] >= 2]
    upper = {}
    for i in range(n):
        if levels[i] >= 1:
            p = l1_nodes.index(i)
            lists = [[l1_nodes[(p + 1) % len(l1_nodes)], l1_nodes[(p - 1) % len(l1_nodes)]]]
            if levels[i] >= 2:
                lists.append([x for x in l2_nodes if x != i])
            upper[i] = lists
    chunks = _hand_hnsw_stream(rows, labels, l0, levels, upper, ep=0, max_level=2, m=m)
    g = vsa.Index.load(list(chunks), "HNSW", dim, "L2", initial_cap=16, m=m, ef_construction=20)
    st = g.stats()
    assert st.count == n and st.max_level == 2 and st.entry_point == 0
    assert g.save() == chunks
    o = oracle.HNSW.from_saved_chunks(chunks, dim, "L2", m, ef_construction=20)
    Q = rows[rng.integers(0, n, 16)] + 0.05 * rng.standard_normal((16, dim)).astype(np.float32)
    for q in Q:
        for ef in (4, 16, 64):
            d, l = g.search(q, 5, ef=ef)
            e_d, e_l = o.search(q, 5, ef=ef)
            assert l.tolist() == e_l.tolist() and d.view(np.uint32).tolist() == e_d.view(np.uint32).tolist()
    assert min(l.tolist()) >= 1000                               # labels, not slots
    # the same stream with one neighbour id beyond the element count is refused
    bad = list(chunks)
    bad[3] = poke(bad[3], 4, "<I", n)
    with pytest.raises(vsa.VkError) as e:
        vsa.Index.load(bad, "HNSW", dim, "L2", initial_cap=16, m=m, ef_construction=20)
    assert "level-0 neighbor id out of range" in e.value.msg


def test_hand_assembled_flat_stream_loads_in_product_and_oracle(vsa, oracle):
    """bruteforce.h:147-169: header (max_elements, size_per_element, curr_element_count), then per element [vector | label].
    Assembled in the test, loaded by the product, answers equal to the oracle over the same rows; saved back byte for byte."""
    rng = np.random.default_rng(41)
    n, dim = 300, 24
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    labels = rng.permutation(10 * n)[:n].astype(np.uint64)
    hdr = bytes(_varint(1 << 3) + _varint(n + 100) + _varint(2 << 3) + _varint(dim * 4 + 8) + _varint(3 << 3) + _varint(n))
    chunks = [hdr] + [rows[i].astype("<f4").tobytes() + struct.pack("<Q", int(labels[i])) for i in range(n)]
    g = vsa.Index.load(list(chunks), "FLAT", dim, "L2")
    assert g.stats().count == n and g.stats().capacity == n + 100
    assert g.save() == chunks
    o = oracle.Flat(dim, "L2", max_elements=n)
    o.add_many(rows, labels)
    Q = rng.standard_normal((8, dim)).astype(np.float32)
    D, L, Nn = g.search_batch(Q, 10)
    for i in range(8):
        e_d, e_l = o.search(Q[i], 10)
        assert L[i].tolist() == e_l.tolist() and D[i].view(np.uint32).tolist() == e_d.view(np.uint32).tolist()
    # a header whose size_per_element belongs to another dimension is refused with the reference's message
    bad = [bytes(_varint(1 << 3) + _varint(n) + _varint(2 << 3) + _varint(dim * 4 + 12) + _varint(3 << 3) + _varint(n))] + chunks[1:]
    with pytest.raises(vsa.VkError) as e:
        vsa.Index.load(bad, "FLAT", dim, "L2")
    assert "Persisted size_per_element does not match expectation." in e.value.msg
