# -*- coding: utf-8 -*-
"""`KerasRate`: the OCR-D workspace processor around `Rater.rate` / `Rater.rate_best`
(drop-in for ocrd_keraslm/wrapper/rate.py:64-326; same executable, ocrd-tool parameters, outputs).

All PAGE <-> lattice logic lives in `wrapper.lattice` (no OCR-D dependency); this module is the thin
OCR-D v3 `Processor` surface and is imported only when the processor is used.
"""
from __future__ import absolute_import

import os
from collections import defaultdict
from dataclasses import dataclass
from typing import Any, Optional

from ocrd import Processor, Workspace, OcrdPageResult
from ocrd_modelfactory import page_from_file
from ocrd_models.ocrd_page import OcrdPage, TextEquivType, to_xml
from ocrd_utils import MIMETYPE_PAGE, config, make_file_id, pushd_popd
from ocrd_validators.page_validator import ConsistencyError, PageValidator

from .. import lib
from . import lattice

# parent element tag -> level of its children (as ocrd_validators' hierarchy)
_CHILD_LEVEL = {'Page': 'region', 'TextRegion': 'line', 'TextLine': 'word', 'Word': 'glyph', 'Glyph': ''}


@dataclass
class PendingPage:
    """a decoded page whose final path is only known once the next page has been searched"""
    traceback: Any          # (beam, last node) as returned by Rater.rate_best
    pcgts: OcrdPage
    file_id: str
    page_id: str


def tokenisation_problems(level, pcgts, logger):
    """element id -> consistency error where a parent's text differs from its children's concatenation
    in white space only (rate.py:599-619): there the implicit white-space rules must not be applied"""
    report = PageValidator.validate(ocrd_page=pcgts, page_textequiv_consistency='strict')
    problems = {}
    if not report.is_valid:
        logger.warning("Page validation failed: %s", report.to_xml())
        for err in report.errors or []:
            if (isinstance(err, ConsistencyError) and _CHILD_LEVEL[err.tag] == level and err.actual and
                    len(err.actual.split()) != len(err.expected.split())):
                problems[err.ID] = err
    return problems


class KerasRate(Processor):
    max_workers = 1    # one engine, and decoding shares state from page to page

    @property
    def executable(self):
        return 'ocrd-keraslm-rate'

    @property
    def metadata_filename(self) -> str:
        return os.path.join('wrapper', 'ocrd-tool.json')

    def setup(self):
        """load the model in the mode the parameters ask for (rate.py:76-91)"""
        model = self.resolve_resource(self.parameter['model_file'])
        self.rater = lib.Rater(logger=self.logger)
        self.rater.load_config(model)
        if self.parameter['alternative_decoding']:
            self.rater.stateful = False      # no implicit state transfer,
            self.rater.incremental = True    # but explicit state transfer
        elif self.rater.stateful:
            self.rater.batch_size = 1
        self.rater.configure()
        self.rater.load_weights(model)
        self.logger.debug("Loaded model_file '%s'", model)

    # ------------------------------------------------------------------ helpers
    def _lattice(self, pcgts):
        level = self.parameter['textequiv_level']
        problems = tokenisation_problems(level, pcgts, self.logger)
        return lattice.page_get_linear_graph_at(level, pcgts, problems=problems, textequiv_factory=TextEquivType,
                                                logger=self.logger)

    def _context(self):
        return lattice.context_from_identifier(self.workspace.mets.unique_identifier)

    def _finish(self, pending, path, entropy):
        """apply the final path of a decoded page and write it to the workspace"""
        level = self.parameter['textequiv_level']
        lattice.page_update_from_path(level, path, entropy, logger=self.logger)
        lattice.page_update_higher_textequiv_levels(level, pending.pcgts, textequiv_factory=TextEquivType)
        pending.pcgts.set_pcGtsId(pending.file_id)
        self.add_metadata(pending.pcgts)
        self.workspace.add_file(ID=pending.file_id, pageId=pending.page_id, file_grp=self.output_file_grp,
                                local_filename=os.path.join(self.output_file_grp, pending.file_id + '.xml'),
                                mimetype=MIMETYPE_PAGE, content=to_xml(pending.pcgts))

    # ------------------------------------------------------------------ scoring only
    def process_page_pcgts(self, *input_pcgts: Optional[OcrdPage], page_id: Optional[str] = None) -> OcrdPageResult:
        """first alternatives only: one windowed forward over the page text (rate.py:293-326)"""
        pcgts = input_pcgts[0]
        level = self.parameter['textequiv_level']
        self.rater.logger.info("Scoring text in page '%s' at the %s level", pcgts.get_pcGtsId(), level)
        graph, _start, _end = self._lattice(pcgts)
        text = [(edge['element'], edge['alternatives']) for edge in lattice.lattice_edges(graph, 0)]
        textstring = ''.join(textequivs[0].Unicode for _element, textequivs in text)
        self.logger.info("Rating %d elements with a total of %d characters", len(text), len(textstring))
        confidences = self.rater.rate(textstring, self._context())
        lattice.apply_ratings(text, confidences, self.parameter['lm_weight'], level, logger=self.logger)
        return OcrdPageResult(pcgts)

    # ------------------------------------------------------------------ alternative decoding
    def process_workspace(self, workspace: Workspace) -> None:
        """Rate text with the language model, either for scoring or for finding the best path across
        alternatives (rate.py:93-131).  With `alternative_decoding` the pages are decoded in order by
        one beam search whose traceback is carried from page to page; a page is written once the search
        over the next page has settled its path."""
        if not self.parameter['alternative_decoding']:
            return super().process_workspace(workspace)
        self.process_workspace_stateful(workspace)

    def process_page_pcgts_stateful(self, pcgts, prev, file_id, page_id):
        """search one page, continuing `prev`'s traceback; finish `prev` (rate.py:248-291)"""
        level = self.parameter['textequiv_level']
        self.rater.logger.info("Scoring text in page '%s' at the %s level", pcgts.get_pcGtsId(), level)
        graph, start_node, end_node = self._lattice(pcgts)
        self.rater.logger.info("Rating %d elements including its alternatives", end_node - start_node)
        path, entropy, traceback = self.rater.rate_best(
            graph, start_node, end_node,
            start_traceback=prev and prev.traceback,
            context=self._context(),
            lm_weight=self.parameter['lm_weight'],
            beam_width=self.parameter['beam_width'],
            beam_clustering_dist=lattice.BEAM_CLUSTERING_DIST if lattice.BEAM_CLUSTERING_ENABLE else 0)
        if prev:
            self._finish(prev, path, entropy)
        return PendingPage(traceback=traceback, pcgts=pcgts, file_id=file_id, page_id=page_id)

    def process_workspace_stateful(self, workspace: Workspace) -> None:
        """Decode the pages of a workspace in document order with one beam search (rate.py:133-246): a page's final path
        is fixed -- and its file written -- while the NEXT page is searched, the last one when the inputs are exhausted.
        Three separate concerns, each in its own unit: which inputs can be decoded at all (`_decodable_inputs`), what a
        failure on a page means under OCRD_MISSING_OUTPUT / OCRD_MAX_MISSING_OUTPUTS (`_PageTally`), and the search itself."""
        log = self._base_logger
        with pushd_popd(workspace.directory):
            self.workspace = workspace
            self.verify()
            inputs = list(self.input_files)
            tally = _PageTally(len(inputs), log)
            pending = None
            for job in self._decodable_inputs(inputs, log):
                try:
                    pending = self.process_page_pcgts_stateful(job.pcgts, pending, job.output_id, job.page_id)
                except FileExistsError as err:
                    # an output that exists is the caller's decision, not a failure of the page
                    if config.OCRD_EXISTING_OUTPUT == 'ABORT':
                        raise
                    if config.OCRD_EXISTING_OUTPUT == 'OVERWRITE':
                        raise Exception("got %s despite OCRD_EXISTING_OUTPUT==OVERWRITE" % err)
                except KeyboardInterrupt:
                    raise
                except Exception as err:
                    if tally.failure(job.page_id, err) == 'COPY':
                        self._copy_page_file(job.input_file)
                    tally.check_limit(len(inputs))
                else:
                    tally.success()
            if pending:
                # end of the document: nothing follows that could still change the last page's path
                beam, last_node = pending.traceback
                path, entropy, _ = self.rater.next_path(beam, ([], last_node))
                self._finish(pending, path, entropy)
            tally.report()

    def _decodable_inputs(self, inputs, log):
        """the inputs a page search can start from, in order: present locally (downloaded if allowed), PAGE-XML, and with
        an output ID that is free or may be overwritten; everything else is logged and passed over (rate.py:150-183)"""
        for input_file in inputs:
            page_id = input_file.pageId
            log.info("preparing page %s", page_id)
            if self.download:
                try:
                    input_file = self.workspace.download_file(input_file)
                except Exception as err:     # ValueError, FileNotFoundError, HTTP errors
                    log.error(repr(err))
                    log.warning("failed downloading file %s for page %s", input_file, page_id)
            if input_file.local_filename is None:
                log.debug("ignoring missing file for page %s", page_id)
                continue
            log.info("processing page %s", page_id)
            try:
                pcgts = page_from_file(input_file)
                assert isinstance(pcgts, OcrdPage)
            except ValueError as err:
                log.error("non-PAGE input for page %s: %s", page_id, err)
                continue
            same_group = input_file.fileGrp == self.output_file_grp
            output_id = input_file.ID if same_group else make_file_id(input_file, self.output_file_grp)
            taken = next(self.workspace.mets.find_files(ID=output_id), None)
            if taken and config.OCRD_EXISTING_OUTPUT != 'OVERWRITE':
                log.error("A file with ID==%s already exists %s and neither force nor ignore are set", output_id, taken)
                continue
            yield _PageJob(input_file, pcgts, output_id, page_id)


@dataclass
class _PageJob:
    input_file: Any
    pcgts: OcrdPage
    output_id: str
    page_id: str


class _PageTally:
    """what happened to the pages, and what OCRD_MISSING_OUTPUT / OCRD_MAX_MISSING_OUTPUTS make of the failures
    (the accounting of ocrd.Processor.process_workspace_handle_tasks, which the sequential loop cannot use)"""

    WORDING = {'SKIP': "skipped", 'COPY': "fallback-copied"}

    def __init__(self, n_inputs, log):
        self.n_inputs, self.log = n_inputs, log
        self.good, self.bad, self.causes = 0, 0, defaultdict(int)

    def success(self):
        self.good += 1

    def failure(self, page_id, err):
        """log and count a failed page; returns the policy that applies ('SKIP' or 'COPY'); re-raises under 'ABORT'"""
        policy = config.OCRD_MISSING_OUTPUT
        what = str(err) or err.__class__.__name__
        if policy == 'ABORT':
            self.log.error("Failure on page %s: %s", page_id, what)
            raise err
        self.log.exception("Failure on page %s: %s", page_id, what)
        if policy not in self.WORDING:
            raise ValueError("unknown configuration value %s for OCRD_MISSING_OUTPUT" % policy)
        self.causes[err.__class__.__name__] += 1
        self.bad += 1
        return policy

    def _too_many(self, of):
        return 0 < config.OCRD_MAX_MISSING_OUTPUTS < self.bad / of

    def _complaint(self):
        return "too many failures with %s output (%d of %d, %s)" % (
            self.WORDING.get(config.OCRD_MISSING_OUTPUT, "aborted"), self.bad, self.bad + self.good, str(dict(self.causes)))

    def check_limit(self, of):
        """stop early once the share of failed pages among all inputs is beyond repair"""
        if self._too_many(of):
            raise Exception(self._complaint())

    def report(self):
        total = self.good + self.bad
        causes = str(dict(self.causes))
        if self.bad:
            if self._too_many(total):
                raise Exception(self._complaint())
            self.log.warning("%s %d of %d pages due to %s", self.WORDING.get(config.OCRD_MISSING_OUTPUT, "aborted"), self.bad, total, causes)
        self.log.debug("succeeded %d, missed %d of %d pages due to %s", self.good, self.bad, total, causes)
