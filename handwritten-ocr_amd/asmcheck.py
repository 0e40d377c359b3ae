"""Build-time audit of attention_vit80x.hip (run by build.build_hip whenever that unit is recompiled; tools/check_vit80x_asm.py and
tests/test_build_asm.py call the same function).

The kernel keeps its output accumulators in AGPRs a0..a95 that only its inline-asm statements name.  That is sound only while the
register allocator itself stays out of the accumulator file for that kernel.  check() compiles the unit to assembly (device only,
the flags of build.py) and reports not-ok if any instruction OUTSIDE the ;;#ASMSTART / ;;#ASMEND regions of attn_vit80x_kernel touches
an AGPR, if a compiler VALU instruction writes an operand of an asm MFMA within the two instructions before it (the asm MFMAs carry
no hazard padding of their own), if the kernel spills, or if its AGPR count is not exactly the 96 the asm owns.  A library whose
kernel fails this would corrupt the tower's attention output silently, so the build refuses to produce it."""
import os
import re
import subprocess
import tempfile


def check() -> dict:
    from . import build

    src = os.path.join(build.CSRC, "attention_vit80x.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "vit80x.s")
        cmd = [build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
               *build.EXTRA_FLAGS.get("attention_vit80x.hip", []), "-I" + build.INCLUDE, "-I" + build.CSRC, "--cuda-device-only", "-S", src,
               "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if ".type" in l and "attn_vit80x_kernel" in l and "@function" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    in_app, bad, n_app_mfma, n_mfma = False, [], 0, 0
    agpr = re.compile(r"(?<![\w.])a(\d+|\[\d+:\d+\])(?![\w])")
    vreg = re.compile(r"(?<![\w.])v(\d+)(?![\w])|(?<![\w.])v\[(\d+):(\d+)\]")

    def vregs(text):
        out = set()
        for m in vreg.finditer(text):
            if m.group(1) is not None:
                out.add(int(m.group(1)))
            else:
                out.update(range(int(m.group(2)), int(m.group(3)) + 1))
        return out

    recent, fresh = [], []  # the last compiler VALU instructions before an asm MFMA: none may write one of its operands (no pad there)
    for l in lines[start:end]:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_app = True
            continue
        if t.startswith(";;#ASMEND"):
            in_app = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        code = t.split(";")[0]
        if "v_mfma" in code:
            n_mfma += 1
            n_app_mfma += in_app
            if in_app:
                srcs = vregs(code.split(",", 1)[1])
                for prev in recent[-2:]:
                    if prev.startswith("v_") and "v_mfma" not in prev and vregs(prev.split(",")[0]) & srcs:
                        fresh.append(prev + "  ->  " + code.strip())
        if not in_app:
            recent.append(code.strip())
        if not in_app and ("accvgpr" in code or agpr.search(code)):
            bad.append(t)
    meta = {}
    for l in lines[end:end + 120]:
        m = re.match(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy): (\d+)", l)
        if m and m.group(1) not in meta:
            meta[m.group(1)] = int(m.group(2))
    ok = not bad and not fresh and meta.get("NumAgprs") == 96 and meta.get("ScratchSize") == 0 and n_app_mfma > 0
    return {"ok": ok, "compiler_agpr_uses": bad[:10], "operands_written_just_before_an_asm_mfma": fresh[:10], "mfma": n_mfma, "asm_mfma": n_app_mfma, **meta}
