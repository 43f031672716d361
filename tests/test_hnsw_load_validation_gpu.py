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
