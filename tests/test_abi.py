"""The C-ABI libraries load and export every symbol include/hwocr.h declares (no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re

from handwritten_ocr_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "hwocr.h")).read()
    return sorted(set(re.findall(r"\b(hwocr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    build.build_all()
    hip = C.CDLL(build.HIP_LIB)
    txt = C.CDLL(build.TEXT_LIB)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(hip, n) or hasattr(txt, n), f"{n} declared in hwocr.h but exported by neither library"


def test_binding_table_matches_header():
    assert sorted(_lib.HIP_SYMBOLS + _lib.TEXT_SYMBOLS) == _declared()


def test_abi_version_and_structs():
    import re

    with open(os.path.join(ROOT, "include", "hwocr.h")) as f:
        declared = int(re.search(r"#define HWOCR_ABI_VERSION (\d+)", f.read()).group(1))
    assert _lib.hip().hwocr_abi_version() == declared == _lib.ABI_VERSION
    # layout contract with include/hwocr.h (LP64): ints, one float, then 8-byte-aligned pointers
    assert C.sizeof(_lib.Vit) == 48 + 13 * 8 and C.sizeof(_lib.VitBlock) == 21 * 8 and C.sizeof(_lib.VitLayout) == 7 * 8 and C.sizeof(_lib.Decoder) == 48 + 7 * 8 and C.sizeof(_lib.DecLayer) == 19 * 8
    assert C.sizeof(_lib.GenState) == 6 * 8 + 8 * 4 + 8 + 2 * 4 and C.sizeof(_lib.Kv) == 32
