class Rater(object):
    pass
