"""MI355X-native Lip2Speech hot path (visual encoder -> mel decoder -> post-net).

The arithmetic runs in hand-written HIP kernels for gfx950 behind the C-ABI declared
in ``include/l2s.h`` (``lip2speech_amd/csrc``); this package is the Python host side
that mirrors the reference's ``model.model`` / ``hparams`` interface.  Importing the
package does not load the native library; the first compute call does, and raises if
it is missing - there is no CPU fallback.
"""
__version__ = "0.1.0"
