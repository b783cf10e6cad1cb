#!/usr/bin/env python3
"""Golden vectors for the reference-named host classes (utilities.CRC, packets.PacketType.get_bytes / get_bits,
packets.PacketProcessor), produced by the UNMODIFIED reference modules in the build container (needs /root/reference; the
test-suite only reads the JSON this writes).

    python3 tests/golden/make_names_golden.py      # rewrites tests/golden/fx_names.json
"""
import builtins
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
builtins.xrange = range            # packets.get_bits / utilities.Convert are Python 2 at call time only
import make_golden as mg           # noqa: E402  (imports the reference with gnuradio / fsm stubbed)

UT, PK = mg.UT, mg.PK


class Cmd(object):                 # the three methods get_bytes / get_bits call on a command
    def __init__(self, header, crc, ptype):
        self._h, self._c, self._t = header, crc, ptype

    def header(self):
        return list(self._h)

    def needs_crc(self):
        return self._c

    def packet_type(self):
        return self._t


def main():
    rng = np.random.default_rng(14443)
    crc = []
    for n in [0, 1, 2, 3, 5, 9, 16, 18, 64]:
        data = [int(b) for b in rng.integers(0, 256, n)]
        crc.append({'data': data, 'a': UT.CRC.calculate_crc(list(data)), 'b': UT.CRC.calculate_crc(list(data), UT.CRC.CRC_14443_B)})
    crc.append({'data': [0, 0], 'a': UT.CRC.calculate_crc([0, 0]), 'b': UT.CRC.calculate_crc([0, 0], UT.CRC.CRC_14443_B)})
    crc.append({'data': [0x12, 0x34], 'a': UT.CRC.calculate_crc([0x12, 0x34]), 'b': UT.CRC.calculate_crc([0x12, 0x34], UT.CRC.CRC_14443_B)})
    frames = []
    for header, needs, ptype, extra in [([0x93, 0x70], True, 1, [0x88, 0x04, 0xBE, 0x6F, 0x5D]), ([0x26], False, 1, []),
                                        ([0x04, 0x00], False, 0, []), ([0x30], True, 1, [0x04]), ([0x08], True, 0, [])]:
        c = Cmd(header, needs, ptype)
        by = PK.PacketType.get_bytes(c, list(extra))
        frames.append({'header': header, 'crc': needs, 'type': ptype, 'extra': extra, 'bytes': by, 'bits': PK.PacketType.get_bits(c, by)})
    streams = []
    for ptype in (0, 1):
        for _ in range(6):
            syms = [int(s) for s in rng.choice([0, 1, 0, 1, 0, 1, 2, 3, 4, 6], size=int(rng.integers(5, 120)))]
            pp = PK.PacketProcessor(ptype)
            closed = []
            for s in syms:
                r = pp.append_bit(s)
                closed.append(None if r is None else list(r))
            streams.append({'type': ptype, 'symbols': syms, 'returns': closed})
    json.dump({'crc': crc, 'frames': frames, 'packet_processor': streams,
               'packet_error': {k: getattr(PK.PacketError, k) for k in ('NO_ERROR', 'PARITY_ERROR', 'CLOSED_ERROR', 'PARITY_CLOSE_ERROR',
                                                                         'TRUNCATED_ERROR')}},
              open(os.path.join(HERE, 'fx_names.json'), 'w'), indent=0)
    print('wrote fx_names.json')


if __name__ == '__main__':
    main()
