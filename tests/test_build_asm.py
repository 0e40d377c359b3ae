"""The one-wave-per-SIMD tower attention keeps its accumulators in AGPRs that only its inline asm names (attention_vit80x.hip):
audit the code hipcc emits for it - no compiler use of the accumulator file, no spill, no freshly written operand in front of an
asm MFMA.  CPU-side (hipcc cross-compiles gfx950 here); the numerics of the kernel are tests/test_ops_gpu.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_vit80x_asm_owns_its_accumulators():
    import check_vit80x_asm

    r = check_vit80x_asm.check()
    assert r["ok"], r
    assert r["asm_mfma"] >= 24 and r["Occupancy"] == 1
