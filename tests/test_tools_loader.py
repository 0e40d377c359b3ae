"""`tools._load_ocr_model` host logic (no GPU): one process per GPU binds the rank's own device, and a drop-in that
forgets HWOCR_MODEL fails loudly instead of transcribing with random weights."""
import os

import pytest
import torch

from handwritten_ocr_amd import _lib, engine, tokenizer, tools


@pytest.fixture
def fake_engine(monkeypatch):
    made = []

    class FakeEngine:
        def __init__(self, cfg, sd, **kw):
            made.append(kw)
            self.dev = kw.get("device")

        def close(self):
            pass

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(engine, "ReadEngine", FakeEngine)
    monkeypatch.setattr(engine, "random_state_dict", lambda cfg, seed=0, device="cuda": {"device": device})
    monkeypatch.setattr(tools, "_ocr_model", None)
    monkeypatch.setattr(tools, "_ocr_processor", None)
    return made


def test_model_must_be_named(fake_engine, monkeypatch):
    monkeypatch.delenv("HWOCR_MODEL", raising=False)
    with pytest.raises(_lib.HwocrError, match="HWOCR_MODEL is not set"):
        tools._load_ocr_model()


def test_presets_need_an_explicit_opt_in(fake_engine, monkeypatch):
    monkeypatch.setenv("HWOCR_MODEL", "tiny")
    monkeypatch.delenv("HWOCR_ALLOW_RANDOM_INIT", raising=False)
    with pytest.raises(_lib.HwocrError, match="HWOCR_ALLOW_RANDOM_INIT"):
        tools._load_ocr_model()
    assert fake_engine == []


@pytest.mark.parametrize("local", [None, "0", "3"])
def test_engine_is_built_on_the_ranks_device(fake_engine, monkeypatch, capsys, local):
    monkeypatch.setenv("HWOCR_MODEL", "tiny")
    monkeypatch.setenv("HWOCR_ALLOW_RANDOM_INIT", "1")
    if local is None:
        monkeypatch.delenv("LOCAL_RANK", raising=False)
    else:
        monkeypatch.setenv("LOCAL_RANK", local)
    model, proc = tools._load_ocr_model()
    want = f"cuda:{local or 0}"
    assert fake_engine[0]["device"] == want
    assert (f"on {want}..." if local is not None else "on cuda...") in capsys.readouterr().out
    assert tools._load_ocr_model()[0] is model  # singleton (tools.py:683-688)
    assert isinstance(proc.tokenizer, tokenizer.ByteTokenizer)


def test_checkpoint_dir_without_tokenizer_is_refused(fake_engine, monkeypatch, tmp_path):
    (tmp_path / "config.json").write_text("{}")
    monkeypatch.setenv("HWOCR_MODEL", str(tmp_path))
    with pytest.raises(_lib.HwocrError, match="tokenizer.json"):
        tools._load_ocr_model()
