"""Packet type ids of the reference (packets.py:18-30).  The framing itself
(PacketProcessor.append_bit, packets.py:67-79) runs on the GPU; see csrc/decode.hip.h."""


class PacketType:
    TAG_TO_READER, READER_TO_TAG, NUM_TYPES = 0, 1, 2
    _START_BIT = {0: 1, 1: 0}   # the first bit of a frame: Manchester frames open with 1, Modified-Miller frames with 0

    @staticmethod
    def start_bit(t):
        try:
            return PacketType._START_BIT[t]
        except KeyError:
            raise ValueError('Unknown Packet Type', str(t))
