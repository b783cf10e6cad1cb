"""The reference's ``packets`` names (packets.py:11-98).

On the GPU path the framing (PacketProcessor.append_bit, packets.py:67-79) runs on the device (csrc/decode.hip.h: FrameAgg)
and closed packets reach ``fsm.process_bits`` through ``background``.  The classes here keep the reference's host-side
surface -- ``background.py:8-21`` imports ``PacketType`` and ``CombinedPacketProcessor`` from this module and hands the
latter to the symbol decoders, the emulators use ``get_bytes`` / ``get_bits`` -- so that the reference's own wiring, bit by
bit on the host, also runs against this package's names (tests/test_reference_names.py)."""
from . import utilities


class PacketError:
    NO_ERROR, PARITY_ERROR, CLOSED_ERROR, PARITY_CLOSE_ERROR, TRUNCATED_ERROR = 0, 1, 2, 3, 4


class PacketType:
    TAG_TO_READER, READER_TO_TAG, NUM_TYPES = 0, 1, 2
    _START_BIT = {0: 1, 1: 0}   # the first bit of a frame: Manchester frames open with 1, Modified-Miller frames with 0

    @staticmethod
    def start_bit(t):
        try:
            return PacketType._START_BIT[t]
        except KeyError:
            raise ValueError('Unknown Packet Type', str(t))

    @staticmethod
    def get_bytes(command, extra_bytes=()):
        """header + extra bytes (+ CRC_A when the command carries one): packets.py:32-39."""
        out = list(command.header()) + list(extra_bytes)
        if command.needs_crc():
            out += utilities.CRC.calculate_crc(out)
        return out

    @staticmethod
    def get_bits(command, all_bytes):
        """start bit, then every byte LSB first followed by its odd-parity bit: packets.py:42-53."""
        return [PacketType.start_bit(command.packet_type())] + utilities.Convert.to_bit_ar(all_bytes, parity=True)


class PacketProcessor:
    """Frames one direction's symbol stream (packets.py:57-79): the first symbol equal to the start bit opens a packet and is
    dropped; every other 0 / 1 is appended -- also before a packet has opened; any other symbol closes an open packet."""

    def __init__(self, packet_type):
        self._type = packet_type
        self._start_bit = PacketType.start_bit(packet_type)
        self._started, self._cur = False, []

    def append_bit(self, bit):
        if bit in (0, 1):
            if not self._started and bit == self._start_bit:
                self._started = True
            else:
                self._cur.append(bit)
            return None
        if not self._started:
            return None
        done, self._started, self._cur = self._cur, False, []
        return done


class CombinedPacketProcessor:
    """One PacketProcessor per direction in front of the protocol machine (packets.py:83-98).  `fsm`: any object with
    process_bits(bits, packet_type); default this package's fsm (the reference's is Python 2)."""

    def __init__(self, emulator=None, fsm=None):
        self._packet_processors = [PacketProcessor(t) for t in range(PacketType.NUM_TYPES)]
        if fsm is None:
            from . import fsm as _fsm
            if emulator:   # packets.py:88-90: the emulator's packets come through the machine, and so does what it sends
                fsm = _fsm.fsm(emulator.process_packet)
                if hasattr(emulator, 'set_encoder'):
                    emulator.set_encoder(fsm.process_outgoing)
            else:
                fsm = _fsm.fsm()
        self._fsm = fsm

    def append_bit(self, bit, packet_type):
        done = self._packet_processors[packet_type].append_bit(bit)
        if done:   # empty lists never reach the fsm (packets.py:97)
            self._fsm.process_bits(done, packet_type)
