"""phanotate_amd — MI355X-native drop-in for PHANOTATE's ORF-graph gene-calling path.

The compute path is libphx.so (hand-written HIP for gfx950 behind the C-ABI in include/phx.h).
Importing this package never falls back to a CPU implementation: `_lib.lib()` raises if the
shared library has not been built, and phx_create fails without a HIP device.
"""
from .api import Annotator, PhxError, Pool, make_params, synth_contig  # noqa: F401
from .pipeline import Pipeline  # noqa: F401

__version__ = "0.4.0"
