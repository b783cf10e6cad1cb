"""The reference's ``utilities`` names that the path and the protocol layer use (utilities.py:7-78): status codes and
pulse lengths as constants, the ISO 14443 CRCs, byte <-> bit conversion."""


class ErrorCode:
    """in-band decoder status codes (utilities.py:7-14)"""
    NO_ERROR = 0
    TOO_SHORT = 2
    TOO_LONG = 3
    ENCODING = 4
    INTERNAL = 5
    WRONG_DUR = 6
    GENERAL = 7


class PulseLength:
    """microseconds, with the expressions of utilities.py:17-23 (the doubles must come out the same)"""
    FULL = 9.44            # one bit period at 106 kbit/s
    ZERO = 3.00            # a Modified-Miller pause
    HALF = FULL / 2
    ZERO_REM = FULL - ZERO
    ONE_REM = HALF - ZERO
    ONE_HALF = FULL + HALF


class CRC:
    """ISO/IEC 14443-3 CRCs (utilities.py:26-46): the reflected CCITT polynomial (0x8408), preset 0x6363 for type A, preset
    0xFFFF and a complemented result for type B; low byte first."""
    CRC_14443_A = 0x6363
    CRC_14443_B = 0xFFFF

    @staticmethod
    def calculate_crc(data, cktp=0x6363):
        if cktp == CRC.CRC_14443_A:
            from .fsm import crc_a     # the shared library's (csrc/protocol.h: nfc_crc_a), pinned by the reference's traces
            return crc_a(data)
        reg = cktp & 0xFFFF
        for byte in data:
            reg ^= byte & 0xFF
            for _ in range(8):
                reg = (reg >> 1) ^ 0x8408 if reg & 1 else reg >> 1
        if cktp == CRC.CRC_14443_B:
            reg ^= 0xFFFF
        return [reg & 0xFF, (reg >> 8) & 0xFF]

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
