"""Mechanical check of the module-side boundary: abseil / protobuf are absent here, so
include/vk_vector_adaptor.h is compiled against a MOCK of VectorBase (tests/helpers/mock_valkey_search.h), and the
VK_ADAPTOR_IN_TREE branch never meets a compiler in this container.  What CAN be checked without one: that every
declaration the adaptor overrides or calls is declared in the mock exactly as the reference declares it.

This test reads the reference headers where they lie (build container only: it SKIPS on the GPU box, where
/root/reference does not exist), extracts the declarations with a small C++ declaration tokenizer (comments, thread-safety
annotations, `override`, inline bodies and parameter NAMES removed; everything else token for token) and compares:

  * every `virtual` of class VectorBase (src/indexes/vector_base.h:129-282) against the mock's VectorBase;
  * the non-virtual members of VectorBase the adaptor calls;
  * `Create` / `LoadFromRDB` / `Search` of VectorFlat<T> (vector_flat.h:37-63) and VectorHNSW<T> (vector_hnsw.h:36-73) against
    VectorGpuFlat<T> / VectorGpuHNSW<T> / VectorGpu<T>: the reference's parameter list must be a prefix of the adaptor's (the
    adaptor may add trailing parameters WITH defaults), defaults included;
  * BaseFilterFunctor (third_party/hnswlib/hnswlib.h:144-149), EntriesFetcherBase / EntriesFetcherIteratorBase
    (index_base.h:108-121), IndexerType's two vector enumerators.

The ONE deliberate difference is listed in DELIBERATE and asserted to be exactly that."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
pytestmark = pytest.mark.skipif(not (REF / "src" / "indexes" / "vector_base.h").exists(),
                                reason="the reference tree is only present in the build container")

# declaration (normalised) -> why the mock differs
DELIBERATE = {
    "SaveIndexImpl": ("virtual absl::Status SaveIndexImpl ( RDBChunkOutputStream ) const = 0",
                      "virtual absl::Status SaveIndexImpl ( RDBChunkOutputStream & ) const = 0",
                      "the real RDBChunkOutputStream is a move-only wrapper passed by value; the mock's is an abstract "
                      "sink passed by reference (the adaptor spells the parameter VK_CHUNK_OUT_PARAM, by value in tree)"),
}


# ---- a small declaration tokenizer ------------------------------------------------------------------------------------
def _strip(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#[^\n]*", " ", src, flags=re.M)
    # thread-safety annotations: ABSL_XXX or ABSL_XXX(balanced, one level)
    src = re.sub(r"\bABSL_[A-Z_]+\s*(\((?:[^()]|\([^()]*\))*\))?", " ", src)
    return src


def _class_body(src: str, name: str) -> str:
    m = re.search(r"\b(?:class|struct)\s+" + re.escape(name) + r"\b[^;{]*\{", src)
    assert m, "class %s not found" % name
    depth, i = 1, m.end()
    while depth:
        c = src[i]
        depth += (c == "{") - (c == "}")
        i += 1
    return src[m.end():i - 1]


def _statements(body: str):
    """top-level statements of a class body: inline bodies and constructor initialiser lists dropped, access labels removed"""
    out, cur, depth, i = [], [], 0, 0
    while i < len(body):
        c = body[i]
        if c == "{" and depth == 0:          # an inline body: skip it, the declaration ends here
            d, i = 1, i + 1
            while d:
                d += (body[i] == "{") - (body[i] == "}")
                i += 1
            out.append("".join(cur))
            cur = []
            continue
        if c in "(<[":
            depth += c != "<" or _is_template_open(body, i)
        elif c in ")>]":
            depth -= c != ">" or depth > 0 and _is_template_close(cur)
        if c == ";" and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(c)
        i += 1
    res = []
    for s in out:
        s = re.sub(r"\b(public|protected|private)\s*:", " ", s).strip()
        if s:
            res.append(s)
    return res


def _is_template_open(body, i):
    return bool(re.search(r"[A-Za-z_0-9:]\s*$", body[:i]))


def _is_template_close(cur):
    return True


_TOK = re.compile(r"[A-Za-z_][A-Za-z_0-9]*(?:::[A-Za-z_][A-Za-z_0-9]*)*|::|->|&&|[0-9.]+f?|[^\sA-Za-z_0-9]")
_NOT_A_NAME = {"const", "unsigned", "struct", "class", "int", "bool", "char", "float", "double", "size_t", "uint64_t",
               "uint32_t", "void", "long", "short", "auto"}


def _split_params(tokens):
    parts, cur, depth = [], [], 0
    for t in tokens:
        if t in "(<[{":
            depth += 1
        elif t in ")>]}":
            depth -= 1
        if t == "," and depth == 0:
            parts.append(cur)
            cur = []
        else:
            cur.append(t)
    if cur:
        parts.append(cur)
    return parts


def _param(tokens):
    """type [= default] with the parameter's NAME dropped"""
    default = []
    if "=" in tokens:
        k = tokens.index("=")
        tokens, default = tokens[:k], tokens[k:]
    if len(tokens) >= 2 and re.fullmatch(r"[A-Za-z_][A-Za-z_0-9]*", tokens[-1]) and tokens[-1] not in _NOT_A_NAME \
            and tokens[-2] not in ("::", "<", ","):
        tokens = tokens[:-1]
    return " ".join(tokens + default)


def _normalise(decl: str, self_name=None):
    """-> (name, normalised declaration, [normalised parameters])"""
    toks = [t for t in _TOK.findall(decl) if t not in ("override", "final", "inline", "explicit")]
    # hnswlib:: qualification inside namespace hnswlib is optional
    toks = [t[len("hnswlib::"):] if t.startswith("hnswlib::") and self_name == "hnswlib" else t for t in toks]
    depth = 0
    for i, t in enumerate(toks):      # a constructor's initialiser list is not part of its declaration
        depth += (t == "(") - (t == ")")
        if t == ":" and depth == 0:
            toks = toks[:i]
            break
    if "(" not in toks:
        return None
    # the declarator's parameter list: the LAST top-level ( ... ) group that is followed only by cv / ref / = 0 / = default
    # (operator()(...) has two groups; the return type may contain parentheses in templates only)
    depth, groups, start = 0, [], None
    for i, t in enumerate(toks):
        if t == "(":
            if depth == 0:
                start = i
            depth += 1
        elif t == ")":
            depth -= 1
            if depth == 0:
                groups.append((start, i))
    lo, hi = groups[-1]
    head = toks[:lo]
    if head[-1] == ")" and len(groups) >= 2:      # operator()
        name = "operator()"
    else:
        name = ("~" if len(head) >= 2 and head[-2] == "~" else "") + head[-1]
    params = [_param(p) for p in _split_params(toks[lo + 1:hi])]
    tail = toks[hi + 1:]
    norm = " ".join(head + ["("] + [", ".join(p.split(" = ")[0] for p in params)] + [")"] + tail)
    norm = re.sub(r"\(\s+\)", "( )", norm)
    return name, norm, params


def _decls(path, cls, self_name=None):
    body = _class_body(_strip(Path(path).read_text()), cls)
    out = {}
    for s in _statements(body):
        n = _normalise(s, self_name)
        if n:
            out.setdefault(n[0], []).append(n)
    return out


MOCK = ROOT / "tests" / "helpers" / "mock_valkey_search.h"
ADAPTOR = ROOT / "include" / "vk_vector_adaptor.h"


def _virtuals(decls):
    return {name: [d for d in ds if d[1].startswith("virtual ")] for name, ds in decls.items()
            if any(d[1].startswith("virtual ") for d in ds)}


def test_tokenizer_on_known_shapes():
    n = _normalise("virtual absl::StatusOr<std::pair<float, hnswlib::labeltype>>\n ComputeDistanceFromRecordImpl(uint64_t internal_id,\n absl::string_view query) const = 0")
    assert n[0] == "ComputeDistanceFromRecordImpl"
    assert n[1] == "virtual absl::StatusOr < std::pair < float , hnswlib::labeltype > > ComputeDistanceFromRecordImpl ( uint64_t, absl::string_view ) const = 0"
    assert _normalise("virtual bool operator()(hnswlib::labeltype id)", "hnswlib")[1] == "virtual bool operator ( ) ( labeltype )"
    assert _normalise("virtual bool operator()(labeltype)", "hnswlib")[1] == "virtual bool operator ( ) ( labeltype )"
    assert _normalise("char* TrackVector(uint64_t internal_id, char* vector, size_t len)")[2] == ["uint64_t", "char *", "size_t"]
    p = _normalise("X Search(absl::string_view query, std::unique_ptr<hnswlib::BaseFilterFunctor> filter = nullptr, bool e = false)")[2]
    assert p == ["absl::string_view", "std::unique_ptr < hnswlib::BaseFilterFunctor > = nullptr", "bool = false"]


def test_vector_base_virtuals_match_the_mock():
    ref = _virtuals(_decls(REF / "src/indexes/vector_base.h", "VectorBase"))
    mock = _virtuals(_decls(MOCK, "VectorBase"))
    expected = {"GetCapacity", "GetDataTypeSize", "GetMaxInternalLabel", "GetLabelCount", "AddRecordImpl", "RemoveRecordImpl",
                "ModifyRecordImpl", "RespondWithInfoImpl", "ToProtoImpl", "SaveIndexImpl", "GetValueImpl",
                "ComputeDistanceFromRecordImpl", "TrackVector", "IsVectorMatch", "UnTrackVector"}
    assert set(ref) == expected, "the reference's VectorBase grew or lost a virtual: %s" % sorted(set(ref) ^ expected)
    different = {}
    for name, ds in ref.items():
        assert name in mock, "the mock lacks virtual %s" % name
        r = sorted(d[1] for d in ds)
        m = sorted(d[1] for d in mock[name])
        if r != m:
            different[name] = (r, m)
    assert set(different) == set(DELIBERATE), different
    for name, (r, m) in different.items():
        assert (r, m) == ([DELIBERATE[name][0]], [DELIBERATE[name][1]]), (name, r, m)
    # the mock declares no pure virtual the reference does not have (the adaptor would otherwise override nothing in tree)
    assert set(mock) - {"~VectorBase"} <= set(ref), set(mock) - set(ref)


def test_vector_base_members_the_adaptor_calls():
    ref, mock = _decls(REF / "src/indexes/vector_base.h", "VectorBase"), _decls(MOCK, "VectorBase")
    for name in ("GetNormalize", "GetVectorDataSize", "GetKeyDuringSearch", "GetInternalIdDuringSearch"):
        assert [d[1] for d in ref[name]] == [d[1] for d in mock[name]], name
    # the VectorTracker entry LoadIndex calls (vector_base.h `char* TrackVector(uint64_t, char*, size_t) override`)
    three = lambda ds: [d[1] for d in ds if len(d[2]) == 3]
    assert three(ref["TrackVector"]) == three(mock["TrackVector"]) == ["char * TrackVector ( uint64_t, char *, size_t )"]
    # the protected constructor the backends chain to
    assert [d[2] for d in ref["VectorBase"]] == [d[2] for d in mock["VectorBase"]] == \
        [["IndexerType", "int", "data_model::AttributeDataType", "absl::string_view"]]
    # GetInternalIdDuringSearch is PRIVATE upstream: the one access change INTEGRATION.md section 2 asks for
    src = _strip((REF / "src/indexes/vector_base.h").read_text())
    body = _class_body(src, "VectorBase")
    assert body.index("GetInternalIdDuringSearch") > body.rindex("private:")


@pytest.mark.parametrize("ref_header,ref_cls,our_cls", [("src/indexes/vector_flat.h", "VectorFlat", "VectorGpuFlat"),
                                                        ("src/indexes/vector_hnsw.h", "VectorHNSW", "VectorGpuHNSW")])
def test_backend_entry_points_match_the_adaptor(ref_header, ref_cls, our_cls):
    ref = _decls(REF / ref_header, ref_cls)
    ours = _decls(ADAPTOR, our_cls)
    base = _decls(ADAPTOR, "VectorGpu")
    for name in ("Create", "LoadFromRDB"):
        (r,) = ref[name]
        (o,) = ours[name]
        assert o[2][:len(r[2])] == r[2], (name, r[2], o[2])
        assert all(" = " in p for p in o[2][len(r[2]):]), "added parameters of %s need defaults: %s" % (name, o[2])
        ret = lambda d: d[1].split(" %s (" % name)[0].replace(our_cls, ref_cls)
        assert ret(o) == ret(r), (ret(o), ret(r))
    # Search: the blocking form with the reference's parameter list.  VectorFlat's stops after `filter`; the adaptor's one
    # Search (in VectorGpu<T>) carries VectorHNSW's, whose extra parameters have defaults -- so a VectorFlat call site compiles
    (r,) = ref["Search"]
    cands = [o for o in base["Search"] if o[2][:len(r[2])] == r[2]]
    assert len(cands) == 1, (r[2], [o[2] for o in base["Search"]])
    assert all(" = " in p for p in cands[0][2][len(r[2]):])
    assert cands[0][1].split(" Search (")[0] == r[1].split(" Search (")[0] == "absl::StatusOr < std::vector < Neighbor > >"
    # every Impl / tracking virtual the reference backend overrides is overridden by the adaptor too
    over = lambda decls, txt: {n for n in decls if re.search(r"\b%s\s*\([^;{]*\)[^;{]*\boverride\b" % re.escape(n), txt)}
    ref_txt, our_txt = _strip((REF / ref_header).read_text()), _strip(ADAPTOR.read_text())
    ref_over = over(ref, ref_txt) - {"~" + ref_cls, ref_cls}      # (the regex sees `~VectorFlat() override` under both names)
    our_over = over(ours, our_txt) | over(base, our_txt)
    assert ref_over <= our_over, ref_over - our_over


def test_filter_functor_fetchers_and_indexer_type():
    ref = _virtuals(_decls(REF / "third_party/hnswlib/hnswlib.h", "BaseFilterFunctor", "hnswlib"))
    mock = _virtuals(_decls(MOCK, "BaseFilterFunctor", "hnswlib"))
    assert [d[1] for d in ref["operator()"]] == [d[1] for d in mock["operator()"]] == ["virtual bool operator ( ) ( labeltype )"]
    for cls in ("EntriesFetcherIteratorBase", "EntriesFetcherBase"):
        r, m = _virtuals(_decls(REF / "src/indexes/index_base.h", cls)), _virtuals(_decls(MOCK, cls))
        assert {n: [d[1] for d in ds] for n, ds in r.items()} == {n: [d[1] for d in ds] for n, ds in m.items()}, cls
    enum = lambda p: re.search(r"enum\s+class\s+IndexerType\s*\{([^}]*)\}", _strip(Path(p).read_text())).group(1)
    ref_e = [e.strip() for e in enum(REF / "src/indexes/index_base.h").split(",")]
    mock_e = [e.strip() for e in enum(MOCK).split(",")]
    assert ref_e[:2] == mock_e == ["kHNSW", "kFlat"]          # (same enumerators, same values)
    src = _strip((REF / "third_party/hnswlib/hnswlib.h").read_text())
    assert re.search(r"typedef\s+size_t\s+labeltype\s*;", src) and "using labeltype = size_t;" in MOCK.read_text()
    # Neighbor: the two members and the (key, distance) constructor the adaptor's reply uses
    nb = _class_body(_strip((REF / "src/indexes/vector_base.h").read_text()), "Neighbor")
    assert re.search(r"InternedStringPtr\s+external_id\s*;", nb) and re.search(r"float\s+distance\s*;", nb)
    assert re.search(r"Neighbor\s*\(\s*const\s+InternedStringPtr\s*&\s*external_id\s*,\s*float\s+distance\s*\)", nb)
