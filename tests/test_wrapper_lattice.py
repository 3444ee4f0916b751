"""Host logic of the OCR-D processor surface: PAGE hierarchy -> lattice,
white-space edges and tokenisation repair, choice filter, write-back and upward consistency
(reference: ocrd_keraslm/wrapper/rate.py:343-679).  OCR-D core is absent here (and the reference's
wrapper cannot be imported), so the PAGE objects are plain stand-ins exposing the accessor
subset the logic uses and the expectations are written from the reference's rules."""
import types

import numpy as np
import pytest

from ocrd_keraslm_amd.wrapper import lattice as LT
from ocrd_keraslm_amd.wrapper.lattice import PlainTextEquiv as TE


class Elem(object):
    def __init__(self, id_, text=None, children=(), alts=None, kind=None, **attrs):
        self.id = id_
        self.kind = kind
        self.children = list(children)
        self.textequivs = [TE(Unicode=u, conf=c) for u, c in alts] if alts else ([TE(Unicode=text, conf=None)] if text else [])
        self.readingDirection = attrs.get('readingDirection')
        self.textLineOrder = attrs.get('textLineOrder')

    def get_TextEquiv(self):
        return self.textequivs

    def set_TextEquiv(self, tes):
        self.textequivs = list(tes)

    def get_TextRegion(self):
        return [c for c in self.children if c.kind == 'region']

    def get_TextLine(self):
        return [c for c in self.children if c.kind == 'line']

    def get_Word(self):
        return [c for c in self.children if c.kind == 'word']

    def get_Glyph(self):
        return [c for c in self.children if c.kind == 'glyph']

    def get_readingDirection(self):
        return self.readingDirection

    def get_textLineOrder(self):
        return self.textLineOrder


class Page(Elem):
    def __init__(self, regions, relations=None, **attrs):
        Elem.__init__(self, 'page', children=regions, kind='page', **attrs)
        self.relations = relations

    def get_Relations(self):
        return self.relations

    def get_ReadingOrder(self):
        return None

    def get_AllRegions(self, classes=None):
        out = []

        def walk(region):
            for sub in region.get_TextRegion():
                walk(sub)
            out.append(region)
        for region in self.get_TextRegion():
            walk(region)
        return out


class Pcgts(object):
    def __init__(self, page, id_='FILE_0001'):
        self.page, self.id = page, id_

    def get_Page(self):
        return self.page

    def get_pcGtsId(self):
        return self.id


def glyphs(word_id, text, alts=None):
    alts = alts or {}
    return [Elem('%s_g%d' % (word_id, i), kind='glyph', alts=alts.get(i, [(ch, 0.9)])) for i, ch in enumerate(text)]


def word(id_, text, alts=None):
    return Elem(id_, text, glyphs(id_, text, alts), kind='word')


def line(id_, words):
    return Elem(id_, ' '.join(w.textequivs[0].Unicode for w in words), words, kind='line')


def region(id_, lines):
    return Elem(id_, '\n'.join(l.textequivs[0].Unicode for l in lines), lines, kind='region')


def sample_page():
    l1 = line('l1', [word('w1', 'ab'), word('w2', 'c')])
    l2 = line('l2', [word('w3', 'de')])
    r2 = region('r2', [line('l3', [word('w4', 'f')])])
    return Pcgts(Page([region('r1', [l1, l2]), r2]))


def path_string(graph):
    return ''.join(e['alternatives'][0].Unicode for e in LT.lattice_edges(graph, 0))


@pytest.mark.parametrize("level,expected,n_real", [("glyph", "ab c\nde\nf", 6), ("word", "ab c\nde\nf", 4),
                                                   ("line", "ab c\nde\nf", 3), ("region", "ab c\nde\nf", 2)])
def test_linear_graph_text_and_spaces(level, expected, n_real):
    graph, start, end = LT.page_get_linear_graph_at(level, sample_page())
    assert start == 0 and end == graph.number_of_edges()
    assert path_string(graph) == expected
    edges = LT.lattice_edges(graph, 0)
    assert sum(1 for e in edges if e['element'] is not None) == n_real
    # pseudo edges carry exactly one white-space alternative with confidence 1
    for e in edges:
        if e['element'] is None:
            assert len(e['alternatives']) == 1 and e['alternatives'][0].Unicode in (' ', '\n') and e['alternatives'][0].conf == 1.0
    assert graph.graph['level'] == level


def test_first_elements_get_no_space_and_empty_elements_are_skipped():
    empty_line = Elem('l0', None, [], kind='line')
    r1 = Elem('r1', None, [empty_line, line('l1', [word('w1', 'x')])], kind='region')
    graph, _, end = LT.page_get_linear_graph_at('line', Pcgts(Page([r1])))
    # the empty first line adds no edge, but it is no longer "first": the next line is preceded by a newline
    assert path_string(graph) == "\nx" and end == 2


def test_filter_choices():
    tes = [TE('a', '0.9'), TE('b', 0.85), TE('c', 0.5), TE('d', None), TE('e', 0.9)]
    kept = LT.filter_choices(tes)
    assert [t.Unicode for t in kept] == ['a', 'b', 'd']       # top 4 only; c drops 0.4 > 0.1; missing conf = 1.0
    assert all(isinstance(t.conf, float) for t in tes[:4]) and tes[3].conf == 1.0
    assert LT.filter_choices([]) == []


def test_repair_tokenisation_skips_space():
    # the line's own text reads "ab-c" where the words concatenate to "ab-" + "c": no blank must be inserted
    assert LT.repair_tokenisation("ab-c", "ab-", "c")
    assert not LT.repair_tokenisation("ab- c", "ab-", "c")
    assert not LT.repair_tokenisation("xyz", "ab-", "c")
    l1 = line('l1', [word('w1', 'ab-'), word('w2', 'c')])
    page = Pcgts(Page([region('r1', [l1])]))
    problem = types.SimpleNamespace(actual="ab-c", expected="ab- c")
    graph, _, _ = LT.page_get_linear_graph_at('word', page, problems={'l1': problem})
    assert path_string(graph) == "ab-c"
    graph, _, _ = LT.page_get_linear_graph_at('word', page, problems={})
    assert path_string(graph) == "ab- c"


def test_context_from_identifier():
    assert LT.context_from_identifier("http://x/y/author_title_1784") == [179]
    assert LT.context_from_identifier("author_title") == [0]
    assert LT.context_from_identifier(None) == [0]


def test_apply_ratings_combines_scores():
    page = sample_page()
    graph, _, _ = LT.page_get_linear_graph_at('word', page)
    text = [(e['element'], e['alternatives']) for e in LT.lattice_edges(graph, 0)]
    n = len(path_string(graph))
    probs = list(np.linspace(0.2, 0.9, n))
    avg, ppl, _ = LT.apply_ratings(text, probs, 0.5, 'word')
    w1 = page.get_Page().get_TextRegion()[0].get_TextLine()[0].get_Word()[0]
    # word "ab": mean LM probability of its 2 chars, mixed 50:50 with the OCR confidence (absent = 1.0)
    assert abs(w1.get_TextEquiv()[0].conf - (0.5 * (probs[0] + probs[1]) / 2 + 0.5 * 1.0)) < 1e-12
    assert abs(avg - np.mean(probs)) < 1e-12 and abs(ppl - 2 ** np.mean(-np.log2(probs))) < 1e-9


def test_update_from_path_and_higher_levels():
    # glyph alternatives: decoding picks the second choice of one glyph; upper levels must follow
    alts = {1: [('b', 0.9), ('h', 0.85)]}
    w1 = word('w1', 'ab', alts)
    page = Pcgts(Page([region('r1', [line('l1', [w1, word('w2', 'c')])])]))
    graph, start, end = LT.page_get_linear_graph_at('glyph', page)
    edges = LT.lattice_edges(graph, 0)
    assert [len(e['alternatives']) for e in edges] == [1, 2, 1, 1]
    path = []
    for e in edges:
        choice = e['alternatives'][-1]                 # take the last alternative everywhere
        path.append((e['element'], choice, 0.5))
    stats = LT.page_update_from_path('glyph', path, entropy=8.0)
    assert stats is not None and abs(stats[1] - 2 ** (8.0 / 4)) < 1e-12     # "ah c": 4 characters
    g1 = w1.get_Glyph()[1]
    assert [t.Unicode for t in g1.get_TextEquiv()] == ['h'] and g1.get_TextEquiv()[0].conf == 0.5
    LT.page_update_higher_textequiv_levels('glyph', page)
    reg = page.get_Page().get_TextRegion()[0]
    assert w1.get_TextEquiv()[0].Unicode == 'ah'
    assert reg.get_TextLine()[0].get_TextEquiv()[0].Unicode == 'ah c'
    assert reg.get_TextEquiv()[0].Unicode == 'ah c'
    assert abs(w1.get_TextEquiv()[0].conf - 0.5) < 1e-12


def test_higher_levels_respect_direction_order_and_joins():
    join = types.SimpleNamespace(get_type=lambda: 'join',
                                 get_SourceRegionRef=lambda: types.SimpleNamespace(get_regionRef=lambda: 'w2'),
                                 get_TargetRegionRef=lambda: types.SimpleNamespace(get_regionRef=lambda: 'w3'))
    relations = types.SimpleNamespace(get_Relation=lambda: [join])
    l1 = line('l1', [word('w1', 'ab'), word('w2', 'cd-')])
    l2 = line('l2', [word('w3', 'ef')])
    l3 = Elem('l3', 'zy', [word('w4', 'zy')], kind='line', readingDirection=LT.RIGHT_TO_LEFT)
    page = Pcgts(Page([region('r1', [l1, l2, l3])], relations=relations))
    LT.page_update_higher_textequiv_levels('glyph', page)
    reg = page.get_Page().get_TextRegion()[0]
    # hyphenation join: no newline between l1 and l2; right-to-left line: glyphs reversed inside the word
    assert reg.get_TextEquiv()[0].Unicode == 'ab cd-ef\nyz'
    reg.textLineOrder = LT.BOTTOM_TO_TOP
    LT.page_update_higher_textequiv_levels('glyph', page)
    assert reg.get_TextEquiv()[0].Unicode.startswith('yz\n')


def test_lattice_through_rate_best_on_oracle_engine():
    """the whole host chain on the CPU test double: lattice -> Rater.rate_best -> next_path -> write-back"""
    from ocrd_keraslm_amd.lib import Rater
    from tests.oracle_engine import OracleLM
    r = Rater(engine_factory=OracleLM)
    r.width, r.depth, r.length = 16, 1, 8
    r.stateful, r.incremental = False, True
    chars = "abcdefh \n"
    r.mapping = ({c: i + 1 for i, c in enumerate(chars)}, {i + 1: c for i, c in enumerate(chars)})
    r.voc_size = len(chars) + 1
    r.configure()
    r.model.init_weights(seed=3)
    r.status = 2
    alts = {1: [('b', 0.9), ('h', 0.85)]}
    page = Pcgts(Page([region('r1', [line('l1', [word('w1', 'ab', alts), word('w2', 'c')])])]))
    graph, start, end = LT.page_get_linear_graph_at('glyph', page)
    path, entropy, traceback = r.rate_best(graph, start, end, context=[0], lm_weight=0.5, beam_width=4,
                                           beam_clustering_dist=LT.BEAM_CLUSTERING_DIST)
    path, entropy, _ = r.next_path(traceback[0], ([], traceback[1]))
    assert len(path) == 4 and entropy > 0
    LT.page_update_from_path('glyph', path, entropy)
    LT.page_update_higher_textequiv_levels('glyph', page)
    text = page.get_Page().get_TextRegion()[0].get_TextEquiv()[0].Unicode
    assert text in ('ab c', 'ah c')
    for g in page.get_Page().get_TextRegion()[0].get_TextLine()[0].get_Word()[0].get_Glyph():
        assert len(g.get_TextEquiv()) == 1 and 0.0 < g.get_TextEquiv()[0].conf <= 1.0
