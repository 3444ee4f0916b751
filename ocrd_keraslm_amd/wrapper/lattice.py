# -*- coding: utf-8 -*-
"""PAGE hierarchy <-> rating lattice, the host logic either side of `Rater.rate` / `Rater.rate_best`.

Everything here works on duck-typed PAGE objects (`get_Page`, `get_TextRegion`, `get_TextLine`,
`get_Word`, `get_Glyph`, `get_TextEquiv` / `set_TextEquiv`, `.id`; TextEquivs with `.Unicode`,
`.conf`, `set_conf`), so it needs neither `ocrd` nor `ocrd_models` and is unit-tested with plain
stand-ins.  Behaviour follows the reference processor (ocrd_keraslm/wrapper/rate.py):

  lattice construction      rate.py:343-423  (page_get_linear_graph_at)
  white-space edges         rate.py:621-644  (_add_space), 646-660 (_repair_tokenisation)
  choice filter             rate.py:665-679  (_filter_choices)
  result write-back         rate.py:425-440  (_page_update_from_path)
  upward consistency        rate.py:476-597  (page_update_higher_textequiv_levels)
  METS context              rate.py:328-341  (mets_get_context)

The reference's latent bugs are not reproduced (`getLogger` used without import, rate.py:345 etc.).
"""
from __future__ import absolute_import

import logging
from math import ceil

import networkx as nx

CHOICE_THRESHOLD_NUM = 4      # maximum number of choices to try per element (rate.py:42)
CHOICE_THRESHOLD_CONF = 0.1   # maximum score drop from best choice to try per element (rate.py:43)
BEAM_CLUSTERING_ENABLE = True  # prune partial paths by history clustering (rate.py:45)
BEAM_CLUSTERING_DIST = 5       # maximum distance between state vectors of a cluster (rate.py:46)

LEVELS = ('region', 'line', 'word', 'glyph')
# PAGE enumeration literals (ocrd_models.ocrd_page_generateds), compared as strings
BOTTOM_TO_TOP = 'bottom-to-top'
RIGHT_TO_LEFT = 'right-to-left'

_LOG = logging.getLogger('ocrd.processor.KerasRate')


class PlainTextEquiv(object):
    """Stand-in for `ocrd_models.ocrd_page.TextEquivType` where `ocrd_models` is absent
    (white-space pseudo edges never reach the XML; unit tests)."""

    def __init__(self, Unicode=None, conf=None, index=None):
        self.Unicode = Unicode
        self.conf = conf
        self.index = index

    def get_Unicode(self):
        return self.Unicode

    def get_conf(self):
        return self.conf

    def set_conf(self, conf):
        self.conf = conf


def default_textequiv_factory():
    """`TextEquivType` when ocrd_models is importable, else the plain stand-in"""
    try:
        from ocrd_models.ocrd_page import TextEquivType
        return TextEquivType
    except ImportError:
        return PlainTextEquiv


def context_from_identifier(ident):
    """[ceil(year / 10)] from a METS identifier whose last path component ends in `_YEAR`, else [0]
    (rate.py:328-341)"""
    context = [0]
    if ident:
        year = ident.split('/')[-1].split('_')[-1]
        if year.isnumeric():
            context = [ceil(int(year) / 10)]
    return context


def filter_choices(textequivs):
    """At most CHOICE_THRESHOLD_NUM alternatives (given in input order = by confidence), their conf
    attributes made float (absent = 1.0), and only those within CHOICE_THRESHOLD_CONF of the first
    (rate.py:665-679)."""
    if not textequivs:
        return []
    kept = textequivs[:CHOICE_THRESHOLD_NUM]
    for te in kept:
        te.set_conf(float(te.conf) if te.conf else 1.0)
    best = kept[0].conf
    return [te for te in kept if best - te.conf < CHOICE_THRESHOLD_CONF]


def repair_tokenisation(tokenisation, concatenation, next_token, logger=None):
    """Does the parent's own text (`tokenisation`) continue the text concatenated so far directly with
    `next_token`, i.e. without the white space the hierarchy implies (rate.py:646-660)?  The two texts are
    aligned on the longest prefix of `tokenisation` that `concatenation` ends with."""
    longest = min(len(tokenisation), len(concatenation))
    overlap = next((k for k in range(longest, 0, -1) if concatenation.endswith(tokenisation[:k])), 0)
    joined = overlap > 0 and tokenisation.startswith(next_token, overlap)
    if joined:
        (logger or _LOG).warning('Repairing tokenisation between "%s" and "%s"', concatenation[-overlap:], next_token)
    return joined


def lattice_edges(graph, start_node):
    """edge attribute dicts in breadth-first order from `start_node` (rate.py:662-663)"""
    return [graph.edges[a, b] for a, b in nx.bfs_edges(graph, start_node)]


class LatticeBuilder(object):
    """Linear graph of one page at `level`: one edge per element (attrs `element`, `alternatives`)
    plus white-space pseudo edges (`element` None) where the implicit tokenisation needs them:
    newline between regions and lines, blank between words of a line -- unless a detected
    tokenisation problem shows that the parent's text joins the two without white space."""

    def __init__(self, level, problems=None, textequiv_factory=None, logger=None):
        assert level in LEVELS, level
        self.level = level
        self.depth = LEVELS.index(level)
        self.problems = problems or {}
        self.make_textequiv = textequiv_factory or default_textequiv_factory()
        self.logger = logger or _LOG
        self.graph = nx.DiGraph(level=level)
        self.graph.add_node(0)
        self.tip = 0

    # -- edges
    def _edge(self, element, textequivs):
        self.graph.add_node(self.tip + 1)
        self.graph.add_edge(self.tip, self.tip + 1, element=element, alternatives=filter_choices(textequivs))
        self.tip += 1

    def _space(self, char, span_start, problem, textequivs):
        """white-space edge before the next element, skipped when `problem` (the consistency error of
        the enclosing element) says the annotation joins here without it (rate.py:621-644)"""
        if textequivs and textequivs[0].Unicode and problem:
            so_far = ''.join(edge['alternatives'][0].Unicode for edge in lattice_edges(self.graph, span_start))
            if repair_tokenisation(problem.actual, so_far, textequivs[0].Unicode, logger=self.logger):
                return
        self._edge(None, [self.make_textequiv(Unicode=char, conf=1.0)])

    def _element(self, kind, element):
        textequivs = element.get_TextEquiv()
        self.logger.debug("Getting text in %s '%s'", kind, element.id)
        if textequivs:
            self._edge(element, textequivs)
        else:
            self.logger.warning("%s '%s' contains no text results", kind.capitalize(), element.id)

    # -- traversal
    def add_page(self, pcgts):
        """append the page's elements; returns (graph, start_node, end_node)"""
        page_start = self.tip
        regions = pcgts.get_Page().get_TextRegion()
        if not regions:
            self.logger.warning("Page contains no text regions")
        first_region = True
        for region in regions:
            if self.depth == 0:
                if not first_region:
                    self._space('\n', page_start, self.problems.get(pcgts.get_pcGtsId()), region.get_TextEquiv())
                self._element('region', region)
            else:
                self._add_lines(region, first_region)
            first_region = False
        return self.graph, page_start, self.tip

    def _add_lines(self, region, first_region):
        lines = region.get_TextLine()
        if not lines:
            self.logger.warning("Region '%s' contains no text lines", region.id)
        region_start = self.tip
        first_line = True
        for line in lines:
            if self.depth == 1:
                if not (first_line and first_region):
                    self._space('\n', region_start, (not first_line) and self.problems.get(region.id), line.get_TextEquiv())
                self._element('line', line)
            else:
                self._add_words(line, first_line and first_region)
            first_line = False

    def _add_words(self, line, first_line_of_page):
        words = line.get_Word()
        if not words:
            self.logger.warning("Line '%s' contains no words", line.id)
        line_start = self.tip
        first_word = True
        for word in words:
            if not (first_word and first_line_of_page):
                self._space('\n' if first_word else ' ', line_start, (not first_word) and self.problems.get(line.id),
                            word.get_TextEquiv())
            if self.depth == 2:
                self._element('word', word)
            else:
                glyphs = word.get_Glyph()
                if not glyphs:
                    self.logger.warning("Word '%s' contains no glyphs", word.id)
                for glyph in glyphs:
                    self._element('glyph', glyph)
            first_word = False


def page_get_linear_graph_at(level, pcgts, problems=None, textequiv_factory=None, logger=None):
    """(graph, start_node, end_node) for one page (rate.py:343-423); `problems` maps element ids (the
    pcGtsId for the page) to consistency errors with `.actual` (rate.py:599-619)"""
    return LatticeBuilder(level, problems, textequiv_factory, logger).add_page(pcgts)


def page_update_from_path(level, path, entropy, logger=None):
    """Write a decoded path back: every real element keeps only its chosen TextEquiv with the combined
    score as confidence; logs average probability and perplexities (rate.py:425-440).  `path` is a list of
    (element or None for white space, TextEquiv, score)."""
    chosen = [step for step in path if step[0]]
    for element, textequiv, score in chosen:
        element.set_TextEquiv([textequiv])
        textequiv.set_conf(score)
    n_chars = sum(len(textequiv.Unicode) for _element, textequiv, _score in chosen) + (len(path) - len(chosen))
    if not n_chars:
        return None
    bits_per_char = entropy / n_chars
    bits_per_step = entropy / len(path)      # per TextEquiv at `level`, white space included (a character need not be a glyph)
    stats = (2.0 ** -bits_per_char, 2.0 ** bits_per_char, 2.0 ** bits_per_step)
    (logger or _LOG).info("avg: %.3f, char ppl: %.3f, %s ppl: %.3f", stats[0], stats[1], level, stats[2])
    return stats


def apply_ratings(text, confidences, lm_weight, level, logger=None):
    """Non-decoding mode (rate.py:293-326): `text` = [(element, alternatives)] along the lattice,
    `confidences` = one LM probability per character of the concatenated first alternatives.  Keeps
    the first alternative only, sets conf = lm_weight * mean LM probability + (1 - lm_weight) * OCR conf.
    Returns (avg, char perplexity, element perplexity)."""
    from math import log
    logger = logger or _LOG
    i = 0
    for element, textequivs in text:
        textequiv = textequivs[0]
        if element:
            element.set_TextEquiv([textequiv])
        n = len(textequiv.Unicode)
        conf = sum(confidences[i:i + n]) / n
        textequiv.set_conf(conf * lm_weight + textequiv.conf * (1. - lm_weight))
        i += n
    if i != len(confidences):
        logger.critical("Input text length and output scores length are off by %d characters", i - len(confidences))
    if not confidences:
        return None
    avg = sum(confidences) / len(confidences)
    ent = sum(-log(max(p, 1e-99), 2) for p in confidences) / len(confidences)
    stats = (avg, pow(2.0, ent), pow(2.0, ent * len(confidences) / len(text)))
    logger.info("avg: %.3f, char ppl: %.3f, %s ppl: %.3f", stats[0], stats[1], level, stats[2])
    return stats


# ------------------------------------------------------------------ upward consistency
def element_unicode0(element):
    """text of the first TextEquiv, '' if none (rate.py:442-447)"""
    tes = element.get_TextEquiv()
    return (tes[0].Unicode or '') if tes else ''


def element_conf0(element):
    """confidence of the first TextEquiv as float, 1.0 if none (rate.py:449-454)"""
    tes = element.get_TextEquiv()
    return float(tes[0].conf or "1.0") if tes else 1.0


def _is_ordered_group(obj):
    return hasattr(obj, 'get_RegionRefIndexed')


def _is_unordered_group(obj):
    return hasattr(obj, 'get_RegionRef') and callable(getattr(obj, 'get_RegionRef')) and hasattr(obj, 'get_UnorderedGroup') \
        and not _is_ordered_group(obj)


def collect_reading_order(ro, group):
    """region id -> ReadingOrder element, recursively through (un)ordered groups (rate.py:456-474)"""
    members = []
    if _is_ordered_group(group):
        members = group.get_RegionRefIndexed() + group.get_OrderedGroupIndexed() + group.get_UnorderedGroupIndexed()
    elif _is_unordered_group(group):
        members = group.get_RegionRef() + group.get_OrderedGroup() + group.get_UnorderedGroup()
    for member in members:
        ro[member.get_regionRef()] = member
        if _is_ordered_group(member) or _is_unordered_group(member):
            collect_reading_order(ro, member)


def _mean(values):
    values = list(values)
    return sum(values) / len(values) if values else 0


def page_update_higher_textequiv_levels(level, pcgts, overwrite=True, textequiv_factory=None):
    """Make every level above `level` consistent with it (rate.py:476-597): glyphs concatenate into
    words, words join with blanks into lines, lines (and nested regions) with newlines into regions --
    except across `join` relations -- confidences are averaged; words/glyphs are taken in reading
    direction, lines in text line order, nested regions in reading order where one is given."""
    make = textequiv_factory or default_textequiv_factory()
    page = pcgts.get_Page()
    relations = page.get_Relations()
    relations = relations.get_Relation() if relations else []
    joins = [(rel.get_SourceRegionRef().get_regionRef(), rel.get_TargetRegionRef().get_regionRef())
             for rel in relations if rel.get_type() == 'join']
    reading_order = dict()
    ro = page.get_ReadingOrder()
    if ro:
        collect_reading_order(reading_order, ro.get_OrderedGroup() or ro.get_UnorderedGroup())
    if level == 'region':
        return

    def assign(element, text, conf):
        if overwrite or not element.get_TextEquiv():
            element.set_TextEquiv([make(Unicode=text, conf=conf)])

    def direction(*elements):
        for element in elements:
            value = element.get_readingDirection()
            if value:
                return value
        return None

    # depth first: inner regions come before the regions that contain them
    for region in page.get_AllRegions(classes=['Text']):
        subregions = region.get_TextRegion()
        if subregions:
            # (as the reference: only when the first sub-region's entry is itself an ordered group)
            if (all(sub.id in reading_order for sub in subregions) and
                    _is_ordered_group(reading_order[subregions[0].id])):
                subregions = sorted(subregions, key=lambda sub: reading_order[sub.id].index)
            text = element_unicode0(subregions[0])
            for sub, nxt in zip(subregions, subregions[1:]):
                if (sub.id, nxt.id) not in joins:
                    text += '\n'
                text += element_unicode0(nxt)
            assign(region, text, _mean(element_conf0(sub) for sub in subregions))
            continue
        lines = region.get_TextLine()
        if (region.get_textLineOrder() or page.get_textLineOrder()) == BOTTOM_TO_TOP:
            lines = list(reversed(lines))
        if level != 'line':
            for line in lines:
                words = line.get_Word()
                if direction(line, region, page) == RIGHT_TO_LEFT:
                    words = list(reversed(words))
                if level != 'word':
                    for word in words:
                        glyphs = word.get_Glyph()
                        if direction(word, line, region, page) == RIGHT_TO_LEFT:
                            glyphs = list(reversed(glyphs))
                        assign(word, ''.join(element_unicode0(glyph) for glyph in glyphs),
                               _mean(element_conf0(glyph) for glyph in glyphs))
                assign(line, ' '.join(element_unicode0(word) for word in words), _mean(element_conf0(word) for word in words))
        text = ''
        if lines:
            text = element_unicode0(lines[0])
            for line, nxt in zip(lines, lines[1:]):
                words, next_words = line.get_Word(), nxt.get_Word()
                if not (words and next_words and (words[-1].id, next_words[0].id) in joins):
                    text += '\n'
                text += element_unicode0(nxt)
        assign(region, text, _mean(element_conf0(line) for line in lines))
