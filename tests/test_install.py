"""`handwritten_ocr_amd.tools.install()` — the binding INTEGRATION.md describes — against a stand-in `ocr_agent` package
built the way the reference is laid out (tools.py defines the names, nodes.py copies five of them at import time,
nodes.py:8-14).  Mode A: patch before `ocr_agent.nodes` is imported; mode B: after."""
import importlib
import sys
import textwrap

import pytest

from handwritten_ocr_amd import tools

NAMES = ("compare_versions", "merge_versions", "preprocess_image", "run_ocr", "unload_ocr_model")


@pytest.fixture
def fake_reference(tmp_path, monkeypatch):
    pkg = tmp_path / "ocr_agent"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "tools.py").write_text(textwrap.dedent("""
        def _orig(*a, **k): return "reference"
        compare_versions = merge_versions = preprocess_image = run_ocr = unload_ocr_model = _orig
        _load_ocr_model = levenshtein = _levenshtein_words = cer = wer = tier1_metrics = normalize_text = _orig
        evaluate = _orig            # a name the drop-in leaves alone
    """))
    (pkg / "nodes.py").write_text("from ocr_agent.tools import compare_versions, merge_versions, preprocess_image, run_ocr, unload_ocr_model\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    for m in [k for k in sys.modules if k == "ocr_agent" or k.startswith("ocr_agent.")]:
        monkeypatch.delitem(sys.modules, m)
    yield
    for m in [k for k in sys.modules if k == "ocr_agent" or k.startswith("ocr_agent.")]:
        sys.modules.pop(m, None)


def test_mode_a_patch_before_nodes_import(fake_reference):
    tools.install()
    ref_tools = importlib.import_module("ocr_agent.tools")
    nodes = importlib.import_module("ocr_agent.nodes")   # copies the already-patched names
    for n in NAMES:
        assert getattr(ref_tools, n) is getattr(tools, n) and getattr(nodes, n) is getattr(tools, n)
    assert ref_tools._load_ocr_model is tools._load_ocr_model and ref_tools.levenshtein is tools.levenshtein
    assert ref_tools.evaluate() == "reference"


def test_mode_b_patch_after_nodes_import(fake_reference):
    nodes = importlib.import_module("ocr_agent.nodes")   # holds the reference's own functions
    assert nodes.run_ocr() == "reference"
    tools.install()
    for n in NAMES:
        assert getattr(nodes, n) is getattr(tools, n)
    assert nodes.merge_versions(["a b", "a b"]) == "a b"
