"""Reference-named module (manchester.py).

``manchester_encoder.encode_bits`` (manchester.py:64-79) through the C-ABI.  ``manchester_decoder(cpp)`` keeps the reference's
class and its ``process_transition(list of (cur, dur_us))`` (manchester.py:13-61) over the shared library's walk
(nfc_host_decode_steps -> csrc/decoder_tables.h: manch_step); symbols go to ``cpp.append_bit(symbol, TAG_TO_READER)``
(manchester.py:27-28).  See miller.py for why the class exists beside the GPU decoders."""
from . import tx as _tx
from .miller import _HostDecoder
from .packets import PacketType


class manchester_decoder(_HostDecoder):
    _TYPE = PacketType.TAG_TO_READER


class manchester_encoder:
    @staticmethod
    def encode_bits(bits):
        return _tx.encode_bits(_tx.NFC_TX_MANCHESTER, bits)
