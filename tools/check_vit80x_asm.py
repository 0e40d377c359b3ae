#!/usr/bin/env python3
"""attention_vit80x.hip keeps its output accumulators in AGPRs a0..a95 that only its inline-asm statements name.  That is sound only
while the register allocator itself stays out of the accumulator file for that kernel.  This script compiles the unit to assembly
(device only, the flags of build.py) and fails if any instruction OUTSIDE the ;;#ASMSTART / ;;#ASMEND inline-asm regions of
attn_vit80x_kernel touches an AGPR, if a compiler VALU instruction writes an operand of an asm MFMA within the two instructions before it (the asm
MFMAs carry no hazard padding of their own), if the kernel spills, or if its AGPR count is not exactly the 96 the asm owns.
Run after every change to that file or to the compiler (tests/test_build_asm.py runs it in the CPU suite)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def check() -> dict:
    from handwritten_ocr_amd import asmcheck

    return asmcheck.check()


if __name__ == "__main__":
    r = check()
    print(r)
    sys.exit(0 if r["ok"] else 1)
