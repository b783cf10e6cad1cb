"""Reference-named module (multiplier.py:11-22): complex carrier times the input stream.

In the reference this is GNU Radio's ``analog.sig_source_c(samp_rate, GR_COS_WAVE, freq, A)`` into
``blocks.multiply_vcc`` -- third-party arithmetic that is not under the reference tree (parity at that boundary is
unpinned).  Here the multiplication is fused into the renderer's pass (csrc/tx.hip.h states the arithmetic);
``apply`` gives the same product for an array already on the host (numpy, for tests and small inputs)."""
import numpy as np


class multiplier(object):
    def __init__(self, samp_rate=4e6, freq=13.56e6, A=1):
        self.samp_rate, self.freq, self.A = float(samp_rate), float(freq), float(A)

    def kwargs(self):
        return dict(carrier=True, freq=self.freq, amp=self.A)

    def carrier(self, n, first_index=0):
        """The carrier the renderer multiplies by: phase of sample k = (k * inc) mod 2^64, top 24 bits -> angle."""
        turns = self.freq / self.samp_rate
        inc = int((turns - np.floor(turns)) * 18446744073709551616.0)
        k = np.arange(first_index, first_index + n, dtype=np.uint64)
        ph = (k * np.uint64(inc))   # wraps mod 2^64
        turn = (ph >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
        ang = 2.0 * np.pi * turn.astype(np.float64)
        return (np.float32(self.A) * (np.cos(ang) + 1j * np.sin(ang))).astype(np.complex64)

    def apply(self, x, first_index=0):
        x = np.asarray(x, np.complex64)
        return (x * self.carrier(len(x), first_index)).astype(np.complex64)
