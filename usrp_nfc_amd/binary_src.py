"""Reference-named module (binary_src.py:15-103): the bit source of the TX chain.

``binary_src(samp_rate, encode, idle_bit, repeat, pause_dur)`` keeps the reference constructor and ``set_bits``.  The
reference's ``work`` fills GNU Radio output buffers one scheduler call at a time; here the queued runs are rendered in
one pass on the GPU (``render`` / ``render_device``): the samples are what the concatenated ``work`` calls produce up to
the point where the queue is empty (the idle fill of the rest of a buffer depends on the scheduler's buffer size and is
left to the caller: ``idle_samples``).  Pause lengths follow the reference under Python 2, where ``pause/2`` and
``pause/div`` are integer divisions (binary_src.py:52,60).
"""
import numpy as np

from . import tx as _tx


class encoder:
    @staticmethod
    def encode_bits(bits):
        return _tx.encode_bits(_tx.NFC_TX_SAME, bits)


_ENCODINGS = dict(manchester=_tx.NFC_TX_MANCHESTER, miller=_tx.NFC_TX_MILLER)


class binary_src(object):
    "Binary source"

    def __init__(self, samp_rate, encode="same", idle_bit=0, repeat=[], pause_dur=25000, device=0):
        self._encoding = _ENCODINGS.get(encode, _tx.NFC_TX_SAME)
        self._samp_rate = float(samp_rate)
        self._bits = []
        self._idle = idle_bit
        self._repeat = list(repeat)
        self._has_finished = True
        self._pause_dur = pause_dur
        self._device = device

    def _encode_pause(self, pause, has_finished):   # binary_src.py:46-58
        if has_finished:
            pause = self._pause_dur
        if pause == 0:
            return [(2, 0)]
        div = 1000
        a = [(self._idle, div)] * int(pause // div)
        r = pause % div
        if r:
            a += [(self._idle, r)]
        return a

    def set_bits(self, bits, has_finished=False, pause=0):   # binary_src.py:60-63
        encoded = self._encode_pause(pause // 2, has_finished)
        self._bits.extend(encoded + _tx.encode_bits(self._encoding, bits) + encoded)
        self._has_finished = has_finished

    def runs(self):
        return _tx.as_runs(self._bits)

    def n_samples(self):
        return _tx.sample_count(self.runs(), self._samp_rate)

    def render(self, carrier=None):
        """The queued runs as complex64 samples (GPU); ``carrier``: a ``multiplier`` to apply in the same pass.
        The queue is consumed, as ``work`` consumes it."""
        r = self.runs()
        self._bits = []
        kw = carrier.kwargs() if carrier is not None else {}
        return _tx.render(r, self._samp_rate, device=self._device, **kw)

    def idle_samples(self, n):
        return np.full(int(n), self._idle, np.complex64)
