"""Loads the package directory `valkey-search_amd/` (a hyphen is not importable) under the
module name `valkey_search_amd`.  Used by tests/, bench.py and __graft_entry__.py."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent


def load():
    if "valkey_search_amd" in sys.modules:
        return sys.modules["valkey_search_amd"]
    pkg = ROOT / "valkey-search_amd"
    spec = importlib.util.spec_from_file_location("valkey_search_amd", pkg / "__init__.py",
                                                  submodule_search_locations=[str(pkg)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["valkey_search_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


vsa = load()
