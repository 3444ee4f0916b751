"""stand-in for `ocrd_utils` (see README.md)"""
import contextlib
import os

MIMETYPE_PAGE = 'application/vnd.prima.page+xml'


class _Config(object):
    OCRD_EXISTING_OUTPUT = 'ABORT'
    OCRD_MISSING_OUTPUT = 'ABORT'
    OCRD_MAX_MISSING_OUTPUTS = 0.1


config = _Config()


def make_file_id(ocrd_file, output_file_grp):
    return ocrd_file.ID.replace(ocrd_file.fileGrp, output_file_grp) if ocrd_file.fileGrp in ocrd_file.ID \
        else output_file_grp + '_' + ocrd_file.ID


@contextlib.contextmanager
def pushd_popd(newcwd):
    old = os.getcwd()
    os.chdir(newcwd)
    try:
        yield newcwd
    finally:
        os.chdir(old)
