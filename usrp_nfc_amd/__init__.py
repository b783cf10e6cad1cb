"""MI355X-native ISO-14443A IQ -> bit eavesdrop path (the hot path of giech/usrp_nfc).

Modules with the reference's names and call surface: ``decoder``, ``transition_sink``,
``background`` (+ the constants of ``utilities`` / ``packets``); ``api`` is the Python face of the
C-ABI in include/nfc_amd.h; ``synth`` builds synthetic captures.  All compute is in HIP kernels
(csrc/); importing this package does not need a GPU, creating a context does.
"""
__all__ = ['api', 'synth', 'decoder', 'transition_sink', 'background', 'packets', 'utilities']
