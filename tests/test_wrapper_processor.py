"""The OCR-D processor layer `ocrd_keraslm_amd/wrapper/rate.py` (drop-in for ocrd_keraslm/wrapper/rate.py:64-326) driven
through stand-ins for OCR-D core (tests/ocrd_shim: written from the v3 Processor contract, see its README -- OCR-D itself
is not installed here).  What is checked is the layer's own behaviour: model set-up per mode, scoring every page through
`Rater.rate`, alternative decoding with the traceback carried from page to page (a page is written once the NEXT page has
been searched, the last one through `next_path`), and the OCRD_EXISTING_OUTPUT / OCRD_MISSING_OUTPUT policies."""
import importlib
import logging
import os
import sys

import pytest

from ocrd_keraslm_amd.lib import Rater
from tests.oracle_engine import OracleLM
from tests.test_wrapper_lattice import Page, Pcgts, line, region, word

SHIM = os.path.join(os.path.dirname(__file__), "ocrd_shim")
CHARS = "abcdefghyz- \n"


@pytest.fixture
def rate_module(monkeypatch):
    """wrapper.rate imported against the stand-ins, with the CPU test double as engine"""
    monkeypatch.syspath_prepend(SHIM)
    for name in [m for m in sys.modules if m == "ocrd" or m.startswith(("ocrd.", "ocrd_"))]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.delitem(sys.modules, "ocrd_keraslm_amd.wrapper.rate", raising=False)
    mod = importlib.import_module("ocrd_keraslm_amd.wrapper.rate")
    monkeypatch.setattr(mod.lib, "Rater", lambda **kw: Rater(engine_factory=OracleLM, **kw))
    yield mod
    sys.modules.pop("ocrd_keraslm_amd.wrapper.rate", None)
    for name in [m for m in sys.modules if m == "ocrd" or m.startswith(("ocrd.", "ocrd_"))]:
        sys.modules.pop(name, None)


@pytest.fixture
def model_file(tmp_path):
    r = Rater(engine_factory=OracleLM)
    r.width, r.depth, r.length = 16, 1, 8
    r.stateful = True
    r.mapping = ({c: i + 1 for i, c in enumerate(CHARS)}, {i + 1: c for i, c in enumerate(CHARS)})
    r.voc_size = len(CHARS) + 1
    r.configure()
    r.model.init_weights(seed=3)
    r.status = 2
    path = str(tmp_path / "model.h5")
    r.save(path)
    return path


def make_workspace(rate_module, tmp_path, n_pages, alts=None):
    import ocrd
    from ocrd_models.ocrd_page import OcrdPage

    class ShimPcgts(Pcgts, OcrdPage):
        pass
    ws = ocrd.Workspace(tmp_path)
    pages = []
    for k in range(n_pages):
        page = ShimPcgts(Page([region('r1', [line('l1', [word('w1', 'ab', alts), word('w2', 'c')]),
                                             line('l2', [word('w3', 'de')])])]), id_='OCR-D-IN_%04d' % k)
        f = ocrd.ShimFile('OCR-D-IN_%04d' % k, 'PHYS_%04d' % k, 'OCR-D-IN', local_filename='OCR-D-IN/%04d.xml' % k,
                          mimetype='application/vnd.prima.page+xml', pcgts=page)
        ws.mets.files.append(f)
        pages.append(page)
    return ws, pages


def glyph_equivs(page):
    for reg in page.get_Page().get_TextRegion():
        for ln in reg.get_TextLine():
            for wd in ln.get_Word():
                for g in wd.get_Glyph():
                    yield g.get_TextEquiv()


def test_scoring_mode_rates_every_page(rate_module, model_file, tmp_path):
    ws, pages = make_workspace(rate_module, tmp_path, 2)
    proc = rate_module.KerasRate(ws, {'model_file': model_file, 'textequiv_level': 'glyph', 'alternative_decoding': False,
                                      'beam_width': 4, 'lm_weight': 0.5}, 'OCR-D-IN', 'OCR-D-OUT')
    assert proc.rater.stateful and proc.rater.batch_size == 1      # rate.py:84-85
    assert proc.executable == 'ocrd-keraslm-rate'
    proc.process_workspace(ws)
    out = list(ws.mets.find_files(fileGrp='OCR-D-OUT'))
    assert [f.pageId for f in out] == ['PHYS_0000', 'PHYS_0001']
    for page in pages:
        for tes in glyph_equivs(page):
            assert len(tes) == 1 and 0.0 < tes[0].conf <= 1.0
        assert page.metadata_items == 1


def test_alternative_decoding_carries_the_traceback_across_pages(rate_module, model_file, tmp_path, caplog):
    alts = {1: [('b', 0.9), ('h', 0.85)]}
    ws, pages = make_workspace(rate_module, tmp_path, 3, alts)
    proc = rate_module.KerasRate(ws, {'model_file': model_file, 'textequiv_level': 'glyph', 'alternative_decoding': True,
                                      'beam_width': 4, 'lm_weight': 0.5}, 'OCR-D-IN', 'OCR-D-OUT')
    assert not proc.rater.stateful and proc.rater.incremental       # rate.py:80-83
    searched, written = [], []
    search, finish = proc.process_page_pcgts_stateful, proc._finish
    proc.process_page_pcgts_stateful = lambda pcgts, prev, fid, pid: (searched.append(pid), search(pcgts, prev, fid, pid))[1]
    proc._finish = lambda pending, path, entropy: (written.append((pending.page_id, len(searched))), finish(pending, path, entropy))[1]
    proc.process_workspace(ws)
    # page k is written after page k + 1 has been searched; the last one at the end of the document
    assert searched == ['PHYS_0000', 'PHYS_0001', 'PHYS_0002']
    assert written == [('PHYS_0000', 2), ('PHYS_0001', 3), ('PHYS_0002', 3)]
    out = list(ws.mets.find_files(fileGrp='OCR-D-OUT'))
    assert [f.ID for f in out] == ['OCR-D-OUT_0000', 'OCR-D-OUT_0001', 'OCR-D-OUT_0002']
    for page in pages:
        for tes in glyph_equivs(page):
            assert len(tes) == 1                                     # every non-best alternative is gone
        text = page.get_Page().get_TextRegion()[0].get_TextEquiv()[0].Unicode
        assert text in ('ab c\nde', 'ah c\nde')
        assert page.get_pcGtsId().startswith('OCR-D-OUT_')
    # the context variable comes from the METS identifier (…_1850 -> decade 185)
    assert proc._context() == [185]


def test_existing_output_is_not_overwritten_unless_asked(rate_module, model_file, tmp_path, caplog):
    import ocrd
    from ocrd_utils import config
    ws, pages = make_workspace(rate_module, tmp_path, 2, {1: [('b', 0.9), ('h', 0.85)]})
    ws.mets.files.append(ocrd.ShimFile('OCR-D-OUT_0000', 'PHYS_0000', 'OCR-D-OUT', content='old'))
    proc = rate_module.KerasRate(ws, {'model_file': model_file, 'textequiv_level': 'glyph', 'alternative_decoding': True,
                                      'beam_width': 4, 'lm_weight': 0.5}, 'OCR-D-IN', 'OCR-D-OUT')
    with caplog.at_level(logging.ERROR):
        proc.process_workspace(ws)
    assert any('already exists' in r.getMessage() for r in caplog.records)
    assert next(ws.mets.find_files(ID='OCR-D-OUT_0000')).content == 'old'          # page 0 was skipped, not replaced
    assert next(ws.mets.find_files(ID='OCR-D-OUT_0001')).content != 'old'
    # OVERWRITE: the page is decoded again and replaces the old file
    ws2, _ = make_workspace(rate_module, tmp_path, 2, {1: [('b', 0.9), ('h', 0.85)]})
    ws2.mets.files.append(ocrd.ShimFile('OCR-D-OUT_0000', 'PHYS_0000', 'OCR-D-OUT', content='old'))
    ws2.overwrite_mode = True
    config.OCRD_EXISTING_OUTPUT = 'OVERWRITE'
    try:
        proc2 = rate_module.KerasRate(ws2, dict(proc.parameter), 'OCR-D-IN', 'OCR-D-OUT')
        proc2.process_workspace(ws2)
    finally:
        config.OCRD_EXISTING_OUTPUT = 'ABORT'
    assert next(ws2.mets.find_files(ID='OCR-D-OUT_0000')).content != 'old'


@pytest.mark.parametrize("policy", ["ABORT", "SKIP", "COPY"])
def test_missing_output_policies(rate_module, model_file, tmp_path, policy):
    from ocrd_utils import config
    ws, pages = make_workspace(rate_module, tmp_path, 3, {1: [('b', 0.9), ('h', 0.85)]})
    proc = rate_module.KerasRate(ws, {'model_file': model_file, 'textequiv_level': 'glyph', 'alternative_decoding': True,
                                      'beam_width': 4, 'lm_weight': 0.5}, 'OCR-D-IN', 'OCR-D-OUT')
    search = proc.process_page_pcgts_stateful

    def flaky(pcgts, prev, fid, pid):
        if pid == 'PHYS_0001':
            raise RuntimeError("page cannot be decoded")
        return search(pcgts, prev, fid, pid)
    proc.process_page_pcgts_stateful = flaky
    config.OCRD_MISSING_OUTPUT, config.OCRD_MAX_MISSING_OUTPUTS = policy, 0.9
    try:
        if policy == "ABORT":
            with pytest.raises(RuntimeError):
                proc.process_workspace(ws)
            return
        proc.process_workspace(ws)
    finally:
        config.OCRD_MISSING_OUTPUT, config.OCRD_MAX_MISSING_OUTPUTS = 'ABORT', 0.1
    ids = sorted(f.ID for f in ws.mets.find_files(fileGrp='OCR-D-OUT'))
    if policy == "SKIP":
        assert ids == ['OCR-D-OUT_0000', 'OCR-D-OUT_0002']
    else:      # COPY: the input page is passed through under the output file ID
        assert ids == ['OCR-D-OUT_0000', 'OCR-D-OUT_0001', 'OCR-D-OUT_0002']


def test_tokenisation_problems_are_passed_to_the_lattice(rate_module, model_file, tmp_path):
    """a parent whose text differs from its children's concatenation in white space only (rate.py:599-619)"""
    from ocrd_validators.page_validator import ConsistencyError
    ws, pages = make_workspace(rate_module, tmp_path, 1)
    pages[0].planted_errors = [ConsistencyError('TextLine', 'l1', 'f', actual='ab c', expected='abc'),
                               ConsistencyError('TextRegion', 'r1', 'f', actual='x', expected='y')]
    problems = rate_module.tokenisation_problems('word', pages[0], logging.getLogger('t'))
    assert list(problems) == ['l1']             # the region-level error is another level's business; equal token counts are none
