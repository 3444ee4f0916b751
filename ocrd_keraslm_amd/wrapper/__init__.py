'''OCR-D conformance: the `ocrd-keraslm-rate` workspace processor and its command line.

  KerasRate          ocrd.Processor for METS / PAGE-XML workspaces   (reference: wrapper/rate.py:64-326)
  ocrd_keraslm_rate  ocrd.cli command-line interface                 (reference: wrapper/cli.py)

Both need OCR-D core (`ocrd`, `ocrd_models`, `ocrd_validators`), which is imported only when they are
first used; `wrapper.lattice` (the PAGE <-> lattice logic) has no such dependency.
'''


def __getattr__(name):
    if name == 'KerasRate':
        from .rate import KerasRate
        return KerasRate
    if name == 'ocrd_keraslm_rate':
        from .cli import ocrd_keraslm_rate
        return ocrd_keraslm_rate
    raise AttributeError(name)
