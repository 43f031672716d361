"""CPU-side checks of the drop-in boundary: libvkindex.so builds for gfx950, loads, and
exports every entry point include/vk_index.h declares.  No compute calls (no GPU here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    v = _pkg.vsa
    if not v.LIB_PATH.exists():
        v.build()
    return v


def declared_functions():
    text = (ROOT / "include" / "vk_index.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|void|char)\s*\*?\s*(vk_[a-z_0-9]+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("vk_index_create", "vk_index_add", "vk_index_remove", "vk_index_search", "vk_index_search_batch",
                 "vk_index_search_batch_device", "vk_index_search_labels", "vk_index_distance", "vk_index_save",
                 "vk_index_load", "vk_merge_topk_device", "vk_last_error", "vk_index_search_submit", "vk_index_set_option",
                 "vk_index_get_option", "vk_index_shard_stats"):
        assert must in names


def test_library_exports_every_declared_symbol(vsa):
    lib = C.CDLL(str(vsa.LIB_PATH))
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_library_reports_the_struct_sizes_of_the_header_it_was_built_from(vsa):
    """vk_abi_struct_size: what a binding checks before its first call (the Python binding and the adaptor classes do)"""
    lib = C.CDLL(str(vsa.LIB_PATH))
    lib.vk_abi_struct_size.argtypes = [C.c_int]
    lib.vk_abi_struct_size.restype = C.c_uint64
    assert lib.vk_abi_struct_size(0) == C.sizeof(vsa.Params) and lib.vk_abi_struct_size(1) == C.sizeof(vsa.Stats)
    assert lib.vk_abi_struct_size(2) == 0


def test_struct_layouts_match_header(vsa, tmp_path):
    """The ctypes mirrors against what a C compiler makes of include/vk_index.h: size and the
    offset of every field (gcc compiles a probe that prints them)."""
    import subprocess
    assert C.sizeof(vsa.Params) == 144
    assert C.sizeof(vsa.Stats) == 440 + 16 * 8
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vk_index.h"', 'int main(void){']
    for cname, mirror in (("vk_index_params", vsa.Params), ("vk_index_stats", vsa.Stats)):
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in mirror._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append('return 0;}')
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", str(ROOT / "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, mirror in (("vk_index_params", vsa.Params), ("vk_index_stats", vsa.Stats)):
        assert int(got[cname]) == C.sizeof(mirror)
        for fname, _ in mirror._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(mirror, fname).offset, fname


def test_fails_loudly_without_device(vsa):
    import os
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    with pytest.raises(vsa.VkError) as e:
        vsa.Index("FLAT", 16)
    assert e.value.code == vsa.VK_ERR_NO_DEVICE


def test_argument_validation_needs_no_device(vsa):
    lib = vsa.lib()
    h = C.c_void_p()
    p = vsa.make_params("FLAT", 0, "L2", 10)
    assert lib.vk_index_create(C.byref(p), C.byref(h)) == vsa.VK_ERR_INVALID
    p = vsa.make_params("FLAT", 16, "L2", 10)
    p.struct_size = 8
    assert lib.vk_index_create(C.byref(p), C.byref(h)) == vsa.VK_ERR_INVALID
    assert b"struct_size" in lib.vk_last_error()
    assert lib.vk_index_add(None, 1, None) == vsa.VK_ERR_INVALID


def test_hnswlib_shaped_facade_compiles_against_the_abi(vsa, tmp_path):
    """include/vk_algo.h (the adaptor INTEGRATION.md describes) and the program that drives it the way
    VectorFlat / VectorHNSW drive `algo_` build with a plain host compiler against libvkindex.so."""
    import subprocess
    exe = tmp_path / "vk_algo_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", str(ROOT / "include"),
                           str(ROOT / "tests" / "helpers" / "vk_algo_check.cc"), "-o", str(exe),
                           "-L", str(vsa.LIB_PATH.parent), "-lvkindex", f"-Wl,-rpath,{vsa.LIB_PATH.parent}"])
    assert exe.exists()


def test_vectorbase_adaptor_compiles_against_the_mocked_interface(vsa, tmp_path):
    """include/vk_vector_adaptor.h: VectorGpuFlat<T> / VectorGpuHNSW<T> DERIVE from VectorBase and override the virtuals
    VectorFlat<T> / VectorHNSW<T> override (vector_base.h:129-282, vector_flat.h:37-63, vector_hnsw.h:36-73).  Compiled
    with -Wall -Wextra -Werror against tests/helpers/mock_valkey_search.h (a mock of the interface, not the module's
    headers): an override that does not match its virtual, or a pure virtual left unimplemented, fails here."""
    import subprocess
    exe = tmp_path / "adaptor_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-Wsuggest-override", "-I", str(ROOT / "include"),
                           "-I", str(ROOT / "tests" / "helpers"), str(ROOT / "tests" / "helpers" / "adaptor_check.cc"), "-o", str(exe),
                           "-L", str(vsa.LIB_PATH.parent), "-lvkindex", "-lpthread", f"-Wl,-rpath,{vsa.LIB_PATH.parent}"])
    assert exe.exists()


def test_binaries_that_embed_the_abi_structs_are_not_older_than_the_header():
    """vk_index_get_stats fills the caller's vk_index_stats: a helper binary compiled against an older, smaller struct gets
    its stack overwritten (seen once: a field added to the header, scripts/libservingprobe.so not rebuilt, `stack smashing
    detected` in the middle of a profiling run).  build() rebuilds them all; this fails loudly when it was not run."""
    hdr = (ROOT / "include" / "vk_index.h").stat().st_mtime
    stale = [str(p.relative_to(ROOT)) for p in (ROOT / "scripts" / "libservingprobe.so", ROOT / "scripts" / "coalescer_native",
                                                 ROOT / "valkey-search_amd" / "libvkhost.so", ROOT / "valkey-search_amd" / "libvkindex.so",
                                                 ROOT / "valkey-search_amd" / "libvkindex_exp.so")
             if p.exists() and p.stat().st_mtime < hdr]
    assert not stale, f"rebuild (python -c 'import __graft_entry__ as g; g.build()'): older than include/vk_index.h: {stale}"
