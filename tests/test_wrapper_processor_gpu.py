"""SURVEY.md 8(f) row 2 on the GPU: the OCR-D processor layer (`wrapper/rate.py`, against the stand-ins of tests/ocrd_shim)
-> lattice construction (`wrapper/lattice.py`) -> `Rater.rate` / `Rater.rate_best` -> the HIP engine through the C ABI,
on the toy PAGE documents of tests/test_wrapper_processor.py.  The same workspace is processed twice -- with the CPU test
double (the f64 oracle) and with `HipLM` -- and must give the same chosen texts, the same single surviving alternatives and
confidences within 1e-3 (rate.py:249-326: the path `rate_best` chooses is what `_page_update_from_path` writes back)."""
import importlib
import sys

import pytest

from ocrd_keraslm_amd.lib import Rater
from tests.oracle_engine import OracleLM
from tests.test_wrapper_processor import SHIM, glyph_equivs, make_workspace, model_file  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _rate_module(monkeypatch, engine_factory):
    monkeypatch.syspath_prepend(SHIM)
    for name in [m for m in sys.modules if m == "ocrd" or m.startswith(("ocrd.", "ocrd_"))]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.delitem(sys.modules, "ocrd_keraslm_amd.wrapper.rate", raising=False)
    mod = importlib.import_module("ocrd_keraslm_amd.wrapper.rate")
    if engine_factory is not None:
        monkeypatch.setattr(mod.lib, "Rater", lambda **kw: Rater(engine_factory=engine_factory, **kw))
    return mod


def _run(monkeypatch, tmp_path, model_path, engine_factory, alternative_decoding):
    mod = _rate_module(monkeypatch, engine_factory)
    tmp_path.mkdir(parents=True, exist_ok=True)
    alts = {1: [('b', 0.9), ('h', 0.85)]}
    ws, pages = make_workspace(mod, tmp_path, 3, alts)
    proc = mod.KerasRate(ws, {'model_file': model_path, 'textequiv_level': 'glyph', 'alternative_decoding': alternative_decoding,
                              'beam_width': 4, 'lm_weight': 0.5}, 'OCR-D-IN', 'OCR-D-OUT')
    engine = type(proc.rater.model).__name__
    proc.process_workspace(ws)
    texts, confs = [], []
    for page in pages:
        texts.append(page.get_Page().get_TextRegion()[0].get_TextEquiv()[0].Unicode)
        for tes in glyph_equivs(page):
            assert len(tes) == 1
            confs.append((tes[0].Unicode, tes[0].conf))
    out = [f.ID for f in ws.mets.find_files(fileGrp='OCR-D-OUT')]
    return engine, texts, confs, out


@pytest.mark.parametrize("alternative_decoding", [True, False])
def test_processor_on_the_hip_engine_matches_the_cpu_double(monkeypatch, tmp_path, model_file, alternative_decoding):  # noqa: F811
    ref = _run(monkeypatch, tmp_path / "cpu", model_file, OracleLM, alternative_decoding)
    got = _run(monkeypatch, tmp_path / "gpu", model_file, None, alternative_decoding)       # (None: the Rater's own engine, HipLM)
    assert ref[0] == "OracleLM" and got[0] == "HipLM", (ref[0], got[0])
    assert got[1] == ref[1]                         # page texts: the chosen path
    assert got[3] == ref[3]                         # output files
    assert [u for u, _ in got[2]] == [u for u, _ in ref[2]]
    for (_, a), (_, b) in zip(got[2], ref[2]):
        assert abs(a - b) < 1e-3, (a, b)
