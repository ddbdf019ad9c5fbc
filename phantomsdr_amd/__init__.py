"""phantomsdr_amd — MI355X-native spectrum-distributor DSP core.

One hot path of PhantomSDR, re-built as hand-written HIP for gfx950 behind a C-ABI
(include/psdr.h): windowed 50 %-overlap forward FFT -> per-client slice + small inverse
DFT demodulation -> waterfall power pyramid + int8 quantisation.  See DESIGN.md.

Importing this package requires the built HIP library (phantomsdr_amd/libpsdr_hip.so);
there is no CPU implementation behind it.
"""
from ._lib import PsdrError, load  # noqa: F401
from .core import (AM, FM, LSB, MODES, USB, AudioClient, Context, Group, HipFFT,  # noqa: F401
                   SpectrumEngine, WaterfallClient, derived_params)

load()  # fail loudly at import time if the extension is missing
