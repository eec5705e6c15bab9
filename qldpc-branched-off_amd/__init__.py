"""MI355X-native qLDPC min-sum BP decoder + Monte-Carlo syndrome simulator.

Drop-in for the hot path of michelebanfi/qLDPC-branched-off: the modules below keep the reference's
function names, argument order, defaults and return tuples (``src/decoding``, ``src/noise``,
``src/simulation``) and dispatch through ctypes into ``csrc/libqldpc_hip.so`` (hand-written HIP, gfx950).
There is no CPU fallback: without the built library or without a GPU every compute call raises.
"""
from . import _lib  # noqa: F401

__all__ = ["decoding", "noise", "simulation", "codes", "parallel"]
__version__ = "0.1.0"
