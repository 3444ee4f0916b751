"""stand-in for `ocrd_validators.page_validator` (see ../README.md): reports whatever the test planted on the page"""


class ConsistencyError(Exception):
    def __init__(self, tag, ID, file_id, actual, expected):
        Exception.__init__(self, "INCONSISTENCY in %s ID '%s' of file '%s'" % (tag, ID, file_id))
        self.tag, self.ID, self.file_id, self.actual, self.expected = tag, ID, file_id, actual, expected


class _Report(object):
    def __init__(self, errors):
        self.errors = list(errors)
        self.is_valid = not self.errors

    def to_xml(self):
        return "<report errors=%d/>" % len(self.errors)


class PageValidator(object):
    @staticmethod
    def validate(ocrd_page=None, page_textequiv_consistency='strict', **kwargs):
        return _Report(getattr(ocrd_page, 'planted_errors', ()))
