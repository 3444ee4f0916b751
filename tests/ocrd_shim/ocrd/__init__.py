"""stand-in for `ocrd` (see ../README.md): Processor, Workspace, OcrdPageResult"""
import logging
import os


class OcrdPageResult(object):
    def __init__(self, pcgts, *images):
        self.pcgts = pcgts
        self.images = list(images)


class ShimFile(object):
    def __init__(self, ID, pageId, fileGrp, local_filename=None, mimetype=None, content=None, pcgts=None):
        self.ID, self.pageId, self.fileGrp = ID, pageId, fileGrp
        self.local_filename, self.mimetype, self.content, self.pcgts = local_filename, mimetype, content, pcgts

    def __repr__(self):
        return "<file %s>" % self.ID


class ShimMets(object):
    def __init__(self, unique_identifier):
        self.unique_identifier = unique_identifier
        self.files = []

    def find_files(self, ID=None, fileGrp=None, pageId=None):
        for f in self.files:
            if (ID is None or f.ID == ID) and (fileGrp is None or f.fileGrp == fileGrp) and (pageId is None or f.pageId == pageId):
                yield f


class Workspace(object):
    def __init__(self, directory, unique_identifier="http://example.org/doc_1850"):
        self.directory = str(directory)
        self.mets = ShimMets(unique_identifier)
        self.overwrite_mode = False

    def add_file(self, file_grp, ID=None, pageId=None, local_filename=None, mimetype=None, content=None, **kwargs):
        existing = next(self.mets.find_files(ID=ID), None)
        if existing is not None:
            if not self.overwrite_mode:
                raise FileExistsError("A file with ID==%s already exists" % ID)
            self.mets.files.remove(existing)
        f = ShimFile(ID, pageId, file_grp, local_filename, mimetype, content, kwargs.get('pcgts'))
        self.mets.files.append(f)
        return f

    def download_file(self, f):
        return f


class Processor(object):
    """the slice of ocrd.Processor (v3) that KerasRate relies on"""
    max_workers = 0

    def __init__(self, workspace, parameter=None, input_file_grp=None, output_file_grp=None, download_files=False):
        self.workspace = workspace
        self.parameter = dict(parameter or {})
        self.input_file_grp, self.output_file_grp = input_file_grp, output_file_grp
        self.download = download_files
        self.logger = logging.getLogger('ocrd.processor.' + type(self).__name__)
        self._base_logger = logging.getLogger('ocrd.processor.base')
        self.setup()

    def setup(self):
        pass

    @property
    def input_files(self):
        return list(self.workspace.mets.find_files(fileGrp=self.input_file_grp))

    def resolve_resource(self, name):
        return name

    def add_metadata(self, pcgts):
        pcgts.metadata_items = getattr(pcgts, 'metadata_items', 0) + 1

    def verify(self):
        return True

    def _copy_page_file(self, input_file):
        from ocrd_utils import make_file_id
        self.workspace.add_file(self.output_file_grp, ID=make_file_id(input_file, self.output_file_grp), pageId=input_file.pageId,
                                local_filename=input_file.local_filename, mimetype=input_file.mimetype, content=input_file.content)

    def process_workspace(self, workspace):
        """page-parallel default: every input page through process_page_pcgts, result written to the output group"""
        from ocrd_modelfactory import page_from_file
        from ocrd_models.ocrd_page import to_xml
        from ocrd_utils import MIMETYPE_PAGE, make_file_id
        self.workspace = workspace
        for input_file in self.input_files:
            pcgts = page_from_file(input_file)
            result = self.process_page_pcgts(pcgts, page_id=input_file.pageId)
            file_id = make_file_id(input_file, self.output_file_grp)
            result.pcgts.set_pcGtsId(file_id)
            self.add_metadata(result.pcgts)
            workspace.add_file(self.output_file_grp, ID=file_id, pageId=input_file.pageId,
                               local_filename=os.path.join(self.output_file_grp, file_id + '.xml'), mimetype=MIMETYPE_PAGE,
                               content=to_xml(result.pcgts))
