"""Chunk streams in the reference's on-disk layout, ASSEMBLED BY HAND from the documented format -- not produced by the
product's own save (bruteforce.h:147-169; hnswalg.h:808-865; third_party/hnswlib/index.proto).  Test infrastructure."""
import struct

import numpy as np


def varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def build_header(fields):
    out = bytearray()
    for field in sorted(fields):
        v = fields[field]
        if isinstance(v, float):
            out += varint(field << 3 | 1) + struct.pack("<d", v)
        else:
            out += varint(field << 3) + varint(v)
    return bytes(out)


def hand_flat_stream(rows, labels, max_elements):
    """header (1 max_elements, 2 size_per_element, 3 curr_element_count), then per element [vector | label u64]"""
    n, dim = rows.shape
    hdr = bytes(varint(1 << 3) + varint(max_elements) + varint(2 << 3) + varint(dim * 4 + 8) + varint(3 << 3) + varint(n))
    return [hdr] + [rows[i].astype("<f4").tobytes() + struct.pack("<Q", int(labels[i])) for i in range(n)]


def hand_hnsw_stream(rows, labels, l0_lists, levels, upper_lists, ep, max_level, m, max_elements=None):
    """header chunk; per element [count u16 | flags u16 | 2*M neighbour ids | vector | label u64]; then per element a u64
    size chunk and, if non-zero, `level` lists of [count u16 | flags u16 | M neighbour ids]"""
    n, dim = rows.shape
    sl0, slu = (2 * m + 1) * 4, (m + 1) * 4
    off_data = (sl0 + 7) & ~7
    hdr = {2: max_elements or n + 5, 3: n, 4: sl0 + dim * 4 + 8, 5: off_data + 8, 6: sl0, 7: max_level & ((1 << 64) - 1), 8: ep, 9: m,
           10: 2 * m, 11: m, 12: 1.0 / np.log(m), 13: 20}
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


def ring_graph(n=40, dim=12, m=4, seed=40):
    """a three-level graph: nodes on a ring at level 0 (each linked to its four nearest ring neighbours), every fifth node
    also at level 1, every twentieth at level 2; labels 1000.."""
    rng = np.random.default_rng(seed)
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
    return rows, labels, l0, levels, upper


def write_chunk_file(path, chunks):
    with open(path, "wb") as f:
        for c in chunks:
            f.write(struct.pack("<Q", len(c)))
            f.write(c)
