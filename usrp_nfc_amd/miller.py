"""Reference-named module (miller.py).

``miller_encoder.encode_bits`` (miller.py:200-233) through the C-ABI.  ``miller_decoder(cpp)`` keeps the reference's
class and its ``process_transition(list of (cur, dur_us))`` (miller.py:13-197): the walk itself is the shared library's
(nfc_host_decode_steps -> csrc/decoder_tables.h: miller_step, the function the GPU's look-up tables are built from), the decoder
state lives in this object across calls, and every symbol goes to ``cpp.append_bit(symbol, READER_TO_TAG)`` as miller.py:191-197
does.  On the GPU path nothing calls this class -- the device decodes (csrc/decode.hip.h); it exists so that the reference's
own host wiring (background.py:8-52) runs against this package's names."""
import ctypes as C

import numpy as np

from . import _lib
from . import tx as _tx
from .packets import PacketType


class _HostDecoder(object):
    _TYPE = None

    def __init__(self, cpp):
        self._cpp = cpp
        self._state = C.c_int32(0)
        self._L = _lib.load()

    def _reset(self):
        self._state = C.c_int32(0)

    def process_transition(self, transitions):
        n = len(transitions)
        if not n:
            return
        cur = np.fromiter((t[0] for t in transitions), np.int8, n)
        dur = np.fromiter((t[1] for t in transitions), np.float64, n)
        out = np.zeros(2 * n + 2, np.uint8)
        got = C.c_size_t(0)
        rc = self._L.nfc_host_decode_steps(self._TYPE, cur.ctypes.data, dur.ctypes.data, n, C.byref(self._state), out.ctypes.data,
                                           out.size, C.byref(got))
        if rc != 0:
            raise ValueError('nfc_host_decode_steps status %d' % rc)
        for sym in out[:got.value].tolist():
            self._cpp.append_bit(sym, self._TYPE)


class miller_decoder(_HostDecoder):
    _TYPE = PacketType.READER_TO_TAG


class miller_encoder:
    @staticmethod
    def encode_bits(bits):
        return _tx.encode_bits(_tx.NFC_TX_MILLER, bits)
