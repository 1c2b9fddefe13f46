"""irec -- MI355X-native iREC beam-search coder (host package).

Mirrors the `rec.coding` operator surface of gergely-flamich/relative-entropy-coding for the
sampler='beam_search' path; all arithmetic of the hot path runs in hand-written gfx950 kernels behind the
C ABI of libirec_hip.so (include/irec.h).  There is no CPU fallback: without the built library or without
a GPU the device entry points raise.
"""
from . import _lib  # noqa: F401
from .coding import BeamSearchCoder, Coder, CodingError, GaussianCoder  # noqa: F401
from .engine import Engine, get_engine  # noqa: F401

__all__ = ["BeamSearchCoder", "Coder", "GaussianCoder", "CodingError", "Engine", "get_engine"]
