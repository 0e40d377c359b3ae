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
    # layout contract with include/hwocr.h: sizes and field offsets as gcc lays the structs out, against the ctypes mirrors
    import subprocess
    import tempfile

    pairs = {"hwocr_vit": _lib.Vit, "hwocr_vit_block": _lib.VitBlock, "hwocr_vit_layout": _lib.VitLayout,
             "hwocr_vit_ws": _lib.VitWs, "hwocr_decoder": _lib.Decoder, "hwocr_dec_layer": _lib.DecLayer,
             "hwocr_gen_state": _lib.GenState, "hwocr_kv": _lib.Kv, "hwocr_dec_ws": _lib.DecWs, "hwocr_w8": _lib.W8,
             "hwocr_rows16_norm": _lib.Rows16Norm, "hwocr_vit_split": _lib.VitSplit}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "hwocr.h"', "int main(void) {"]
    for cname, ct in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "abi.c"), os.path.join(d, "abi")
        with open(src, "w") as f:
            f.write("\n".join(lines))
        subprocess.run(["gcc", "-I" + os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        got = dict(l.split() for l in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, ct in pairs.items():
        assert int(got[cname]) == C.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, f"{cname}.{fname}"


def test_the_shipped_library_reads_one_environment_switch_and_a_gpu_test_walks_it():
    """kernel paths that no test runs must not be selectable from the environment.  The A/B switches of concluded
    experiments compile only into the -DHWOCR_DIAG library (csrc/common.h: HWOCR_DIAG_ENV_INT); the product library keeps the names
    listed here, each of which a `-m gpu` test sets."""
    build.build_all()
    with open(build.HIP_LIB, "rb") as f:
        blob = f.read()
    names = {n.decode() for n in re.findall(rb"HWOCR_[A-Z0-9_]{3,}", blob)}
    names -= {n for n in names if n.startswith(("HWOCR_STATUS_", "HWOCR_EPI_", "HWOCR_ABI_", "HWOCR_SELECT_"))}   # (constants in messages)
    assert names == {"HWOCR_VIT80_KERNEL"}, names
    assert b"getenv" in blob
    gpu_tests = open(os.path.join(ROOT, "tests", "test_ops_gpu.py")).read()
    for n in names:
        assert f'setenv("{n}"' in gpu_tests, f"{n} is read by the library but no GPU test sets it"
