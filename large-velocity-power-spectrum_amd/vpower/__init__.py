"""MI355X-native velocity / momentum / kinetic-energy power spectra.

Same import surface as the reference package (vpower/__init__.py:1-2).
"""
from .interp import *      # noqa: F401,F403
from .spctrm import *      # noqa: F401,F403
from . import device, synth  # noqa: F401
