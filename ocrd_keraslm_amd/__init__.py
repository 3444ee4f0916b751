"""MI355X-native implementation of the ocrd_keraslm Rater hot path."""
__version__ = "0.1.0"
