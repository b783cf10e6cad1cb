"""Reference-named module (miller.py).  ``miller_encoder.encode_bits`` (miller.py:200-233) through the C-ABI; the Modified
Miller DECODER (miller.py:13-197) runs on the GPU as look-up tables (csrc/decoder_tables.h, csrc/decode.hip.h)."""
from . import tx as _tx


class miller_encoder:
    @staticmethod
    def encode_bits(bits):
        return _tx.encode_bits(_tx.NFC_TX_MILLER, bits)
