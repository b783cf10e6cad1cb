"""Constants of the reference's utilities module that the hot path uses (utilities.py:7-23).
Values only; CRC / byte conversion belong to the protocol layer, which this package does not replace."""


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
