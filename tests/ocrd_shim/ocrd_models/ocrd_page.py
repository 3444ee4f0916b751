"""stand-in for `ocrd_models.ocrd_page` (see ../README.md)"""
from ocrd_keraslm_amd.wrapper.lattice import PlainTextEquiv as TextEquivType      # (Unicode, conf, index, ...)


class OcrdPage(object):
    """marker base of the stand-in PcGts objects the tests build"""
    def set_pcGtsId(self, id_):
        self.id = id_


def to_xml(pcgts):
    """what gets written: enough to tell pages and their chosen texts apart"""
    def text_of(elem):
        tes = elem.get_TextEquiv()
        return tes[0].Unicode if tes else ''
    return "<PcGts id=%r>%s</PcGts>" % (pcgts.get_pcGtsId(), '|'.join(text_of(r) for r in pcgts.get_Page().get_TextRegion()))
