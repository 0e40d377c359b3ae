"""The product path has no CPU fallback and never touches the oracle: without a GPU (this container) every read entry
point raises, a missing device library is reported as such, and no module of the package imports `oracle`."""
import ast
import os

import pytest
import torch

from handwritten_ocr_amd import _lib, engine, tools

PKG = os.path.dirname(os.path.abspath(engine.__file__))

needs_no_gpu = pytest.mark.skipif(torch.cuda.is_available(), reason="checks the behaviour of a host without a GPU")


@needs_no_gpu
def test_engine_and_run_ocr_raise_without_a_gpu(tmp_path, monkeypatch):
    with pytest.raises(_lib.HwocrError, match="no CPU path"):
        engine.ReadEngine(engine.preset("tiny"), {}, max_reads=2, ctx=64)
    from PIL import Image

    p = tmp_path / "page.png"
    Image.new("RGB", (64, 64), "white").save(p)
    monkeypatch.setattr(tools, "_ocr_model", None)
    with pytest.raises(_lib.HwocrError, match="no CPU path"):
        tools.run_ocr(str(p))
    with pytest.raises(_lib.HwocrError):
        tools.run_ocr_batch([str(p)])


def test_missing_device_library_is_an_error_not_a_fallback(monkeypatch):
    from handwritten_ocr_amd import build

    monkeypatch.setattr(_lib, "_hip", None)
    monkeypatch.setattr(build, "HIP_LIB", os.path.join(PKG, "csrc", "does_not_exist.so"))
    with pytest.raises(_lib.HwocrError, match="no CPU fallback"):
        _lib.hip()


def test_package_never_imports_the_oracle():
    for root, _, files in os.walk(PKG):
        for fn in files:
            if not fn.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(root, fn), encoding="utf-8").read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                assert not any(n == "oracle" or n.startswith("oracle.") for n in names), f"{fn} imports the oracle"
