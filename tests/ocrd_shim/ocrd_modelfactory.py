"""stand-in for `ocrd_modelfactory` (see README.md)"""
from ocrd_models.ocrd_page import OcrdPage


def page_from_file(input_file):
    if not isinstance(getattr(input_file, 'pcgts', None), OcrdPage):
        raise ValueError("not a PAGE file: %r" % (input_file,))
    return input_file.pcgts
