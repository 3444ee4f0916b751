'''backend library interface (mirrors ocrd_keraslm/lib/__init__.py:1-7)

Rater - encapsulates LM definition and application
Node - tree data type for beam search
'''
from .rater import Rater
from .node import Node

__all__ = ["Rater", "Node"]
