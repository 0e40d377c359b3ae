"""In-tree build of the native libraries (no JIT cache: the .so files travel with the source tree).

  csrc/libhwocr_hip.so   hipcc --offload-arch=gfx950   gemm / attention / elementwise / runtime
  csrc/libhwocr_text.so  g++                            Levenshtein / LCS alignment (host)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
HIP_SOURCES = ["gemm.hip", "gemm256.hip", "gemm256w4.hip", "gemm_stream.hip", "gemm_rows16.hip", "attention.hip", "attention_vit80x.hip", "elementwise.hip", "imagepre.hip",
               "runtime.hip"]
# attention_vit80x.hip: its score MFMAs must write arch VGPRs (the accumulator file is owned by its inline asm, see the file)
EXTRA_FLAGS = {"attention_vit80x.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
HIP_LIB = os.path.join(CSRC, "libhwocr_hip.so")
DIAG_LIB = os.path.join(CSRC, "diag", "libhwocr_hip_diag.so")  # -DHWOCR_DIAG: measurement variants with WRONG results; tools/ only
TEXT_LIB = os.path.join(CSRC, "libhwocr_text.so")


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _run(cmd: list[str]) -> None:
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the page-read engine needs the ROCm toolchain to build")


def build_hip(force: bool = False, diag: bool = False) -> str:
    """diag=True: the library with -DHWOCR_DIAG (the ablation / timeline variants of the 256x256 GEMM that tools/bench_gemm_ablate.py
    and tools/bench_gemm_timeline.py measure; they compute WRONG results by construction) into csrc/diag/ — never what _lib.hip()
    loads unless a tool points it there (use_diag_library)."""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "common.h"), os.path.join(CSRC, "attention_args.h"),
                   os.path.join(CSRC, "diag_src", "gemm256w4_experiments.inc"), os.path.join(INCLUDE, "hwocr.h")]
    target = DIAG_LIB if diag else HIP_LIB
    if not force and _newer(target, deps):
        return target
    objdir = os.path.dirname(target)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        if force or not _newer(o, [s] + deps[len(srcs):]):
            _run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value",
                  *(["-DHWOCR_DIAG"] if diag else []), *EXTRA_FLAGS.get(os.path.basename(s), []), "-I" + INCLUDE, "-c", s, "-o", o])
            if os.path.basename(s) == "attention_vit80x.hip":
                # the unit's inline asm owns the accumulator file: audit what hipcc emitted around it before it can ship
                from . import asmcheck

                r = asmcheck.check()
                if not r["ok"]:
                    os.remove(o)
                    raise RuntimeError(f"attention_vit80x.hip: the emitted code breaks the invariants its inline asm relies on: {r}")
        objs.append(o)
    _run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs)
    return target


def use_diag_library() -> str:
    """For tools/ only: build the -DHWOCR_DIAG library and make it the one _lib.hip() loads in THIS process (call before the first
    _lib.hip())."""
    global HIP_LIB
    HIP_LIB = build_hip(diag=True)
    return HIP_LIB


def build_text(force: bool = False) -> str:
    src = os.path.join(CSRC, "text.cpp")
    if not force and _newer(TEXT_LIB, [src, os.path.join(INCLUDE, "hwocr.h")]):
        return TEXT_LIB
    _run(["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-I" + INCLUDE, src, "-o", TEXT_LIB])
    return TEXT_LIB


def build_all(force: bool = False) -> None:
    build_text(force)
    build_hip(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built", HIP_LIB, TEXT_LIB)
