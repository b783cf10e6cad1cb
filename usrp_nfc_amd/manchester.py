"""Reference-named module (manchester.py).  ``manchester_encoder.encode_bits`` (manchester.py:64-79) through the C-ABI; the
Manchester DECODER (manchester.py:13-61) runs on the GPU as look-up tables (csrc/decoder_tables.h, csrc/decode.hip.h)."""
from . import tx as _tx


class manchester_encoder:
    @staticmethod
    def encode_bits(bits):
        return _tx.encode_bits(_tx.NFC_TX_MANCHESTER, bits)
