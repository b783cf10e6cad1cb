"""Constants of the reference's utilities module that the hot path uses (utilities.py:7-23).
Values, plus CRC_A / bit conversion of the protocol layer (row f1; the CRC itself is csrc/protocol.h)."""


class ErrorCode:
    NO_ERROR = 0
    TOO_SHORT = 2
    TOO_LONG = 3
    ENCODING = 4
    INTERNAL = 5
    WRONG_DUR = 6
    GENERAL = 7


class PulseLength:
    FULL = 9.44
    ZERO = 3.00
    HALF = FULL / 2
    ZERO_REM = FULL - ZERO
    ONE_REM = HALF - ZERO
    ONE_HALF = FULL + HALF


class CRC:   # utilities.py:26-46
    CRC_14443_A = 0x6363

    @staticmethod
    def calculate_crc(data, cktp=0x6363):
        if cktp != CRC.CRC_14443_A:
            raise ValueError('only CRC_A is implemented')
        from .fsm import crc_a
        return crc_a(data)

    @staticmethod
    def check_crc(data, cktp=0x6363):
        crc = CRC.calculate_crc(data[:-2], cktp)
        return crc[0] == data[-2] and crc[1] == data[-1]


class Convert:   # utilities.py:49-78
    @staticmethod
    def to_bit_ar(data, parity=False):
        ret = []
        for b in data:
            bits = [(b >> i) & 1 for i in range(8)]
            ret.extend(bits)
            if parity:
                ret.append(1 - (sum(bits) & 1))
        return ret

    @staticmethod
    def to_byte_ar(bits):
        n = len(bits) // 8
        return [sum((bits[8 * k + i] & 1) << i for i in range(8)) for k in range(n)]
