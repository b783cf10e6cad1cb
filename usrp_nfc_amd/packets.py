"""Packet type ids of the reference (packets.py:18-30).  The framing itself
(PacketProcessor.append_bit, packets.py:67-79) runs on the GPU; see csrc/decode.hip.h."""


class PacketType:
    TAG_TO_READER = 0
    READER_TO_TAG = 1
    NUM_TYPES = 2

    @staticmethod
    def start_bit(t):
        if t == PacketType.TAG_TO_READER:
            return 1
        elif t == PacketType.READER_TO_TAG:
            return 0
        raise ValueError('Unknown Packet Type', str(t))
