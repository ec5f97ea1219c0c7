"""Minimal stand-in for the PyTorch-Geometric-1.6 symbols the DAGNN reference uses.

Test infrastructure only (see ../README.md). Semantics: SURVEY.md Appendix B.
"""
from . import typing, utils, data, nn  # noqa: F401

__version__ = "1.6.0-standin"


def is_debug_enabled():
    return False
