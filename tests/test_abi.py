"""The C-ABI library loads and exports every symbol include/coevo.h declares (no compute: runs without a GPU)."""
import ctypes
import os
import re

from coevonet_amd import lib as L

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "coevo.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(coevo_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from coevonet_amd.build import build
    build()
    dll = ctypes.CDLL(L.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(dll, n), f"{n} declared in include/coevo.h but not exported"
    assert sorted(L.exported_symbols()) == names, "lib.py binds a different set than the header declares"


def test_version_and_layout_constants():
    dll = L.load()
    assert dll.coevo_version() == 103
    # parameter counts of the reference's FCNetwork (139 781 good / 138 757 adversary)
    assert L.fc_param_count(10) == 139781 and L.fc_param_count(8) == 138757
    assert L.fc_slab_stride(10) % 64 == 0 and L.fc_slab_stride(10) >= 139781
    assert L.fc_param_count(9) < 0  # unsupported width is an argument error, not a crash


def test_graft_entry_build_runs():
    """the driver's "does it build" check (__graft_entry__.build): compiles what is stale, builds the oracle's C restatement,
    binds every symbol and compares the library's version with the header's (a hard-coded number there went stale in round 5)"""
    import importlib
    import sys
    sys.path.insert(0, REPO)
    g = importlib.import_module("__graft_entry__")
    g.build()


def test_header_is_plain_c(tmp_path):
    """include/coevo.h is the drop-in boundary: it must compile as C99 (what a cgo / JNI / ctypes-gen binding includes)"""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "coevo.h"\nint main(void) { return coevo_version() ? 0 : 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(repo, "include"),
                    "-c", str(src), "-o", str(tmp_path / "hdr.o")], check=True)
