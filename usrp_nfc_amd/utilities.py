"""The reference's ``utilities`` names that the path and the protocol layer use (utilities.py:7-78): status codes and
pulse lengths as constants, CRC_A through the shared library (csrc/protocol.h: nfc_crc_a), byte <-> bit conversion."""


def _constants(name, doc, **values):
    return type(name, (object,), dict(values, __doc__=doc))


ErrorCode = _constants('ErrorCode', 'in-band decoder status codes (utilities.py:7-14)',
                       NO_ERROR=0, TOO_SHORT=2, TOO_LONG=3, ENCODING=4, INTERNAL=5, WRONG_DUR=6, GENERAL=7)

_BIT_US = 9.44     # one bit period at 106 kbit/s
_PAUSE_US = 3.00   # a Modified-Miller pause
PulseLength = _constants('PulseLength', 'microseconds, with the expressions of utilities.py:17-23 (the doubles must be the same)',
                         FULL=_BIT_US, ZERO=_PAUSE_US, HALF=_BIT_US / 2, ZERO_REM=_BIT_US - _PAUSE_US,
                         ONE_REM=_BIT_US / 2 - _PAUSE_US, ONE_HALF=_BIT_US + _BIT_US / 2)


class CRC:   # utilities.py:26-46; only type A is on this path
    CRC_14443_A = 0x6363

    @staticmethod
    def calculate_crc(data, cktp=0x6363):
        if cktp != CRC.CRC_14443_A:
            raise ValueError('only CRC_A is implemented')
        from .fsm import crc_a
        return crc_a(data)

    @staticmethod
    def check_crc(data, cktp=0x6363):
        return list(data[-2:]) == list(CRC.calculate_crc(data[:-2], cktp))


class Convert:   # utilities.py:49-78
    @staticmethod
    def to_bit_ar(data, parity=False):
        """LSB-first bits of every byte, each followed by its odd-parity bit when asked."""
        out = []
        for byte in data:
            bits = [(byte >> i) & 1 for i in range(8)]
            out += bits + ([1 - (sum(bits) & 1)] if parity else [])
        return out

    @staticmethod
    def to_byte_ar(bits):
        """Whole bytes of an LSB-first bit list (a trailing partial byte is dropped)."""
        return [sum((bits[8 * k + i] & 1) << i for i in range(8)) for k in range(len(bits) // 8)]
