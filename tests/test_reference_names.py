"""The reference-named host classes of this package -- what the reference's own background.py imports
(``from packets import PacketType, CombinedPacketProcessor``, ``from miller import miller_decoder``,
``from manchester import manchester_decoder``: background.py:8-13) -- against vectors the unmodified reference produced
(tests/golden/make_names_golden.py, make_golden.py).  Host logic only: no GPU."""
import numpy as np
import pytest

from tests.golden_util import Case, load_json, load_npz
from usrp_nfc_amd import manchester, miller, packets, utilities
from usrp_nfc_amd.packets import PacketType

NAMES = load_json('fx_names.json')


class Recorder(object):
    def __init__(self):
        self.packets = []

    def process_bits(self, bits, packet_type):
        self.packets.append((packet_type, list(bits)))


def reference_style_run(transitions, reader, tag):
    """background.run's grouping (background.py:41-52) over one transition list, with this package's classes."""
    def process(group, t):   # background.py:30-35
        if t == PacketType.TAG_TO_READER and tag:
            tag.process_transition(group)
        elif t == PacketType.READER_TO_TAG and reader:
            reader.process_transition(group)
    group, cur = [], PacketType.TAG_TO_READER
    for val, t in transitions:
        if t == cur:
            group.append(val)
        else:
            process(group, cur)
            group, cur = [val], t
    if group:
        process(group, cur)


@pytest.mark.parametrize('chunk', [None, 7, 1])
def test_background_wiring_over_package_names(chunk):
    # the Ultralight transaction: the reference's transition list through miller_decoder / manchester_decoder /
    # CombinedPacketProcessor of THIS package, wired as background.py does -> the reference's 19 packets; also when the list
    # arrives in pieces (decoder and framing state carried across process_transition calls)
    c = Case('fx_ultralight_txn')
    rec = Recorder()
    cpp = packets.CombinedPacketProcessor(fsm=rec)
    reader, tag = miller.miller_decoder(cpp), manchester.manchester_decoder(cpp)
    tr = c.transitions
    pieces = [tr] if chunk is None else [tr[i:i + chunk] for i in range(0, len(tr), chunk)]
    if chunk is None:
        reference_style_run(tr, reader, tag)
    else:
        # (background.run starts every list at TAG_TO_READER; a piece boundary inside a same-type run splits one
        # process_transition call in two, which the decoders must not notice)
        for p in pieces:
            reference_style_run(p, reader, tag)
    assert len(rec.packets) == 19
    assert rec.packets == c.packets


@pytest.mark.parametrize('tagname', ['1', '0p5', '0p25', '0p1', 'frames_0p5', 'frames_0p25'])
def test_decoder_classes_on_decoder_vectors(tagname):
    # random (cur, d) lists with every error branch, durations d * factor in microseconds as transition_sink emits them
    z = load_npz('fx_decoder_vectors.npz')
    factor = {'1': 1.0, '0p5': 0.5, '0p25': 0.25, '0p1': 0.1, 'frames_0p5': 0.5, 'frames_0p25': 0.25}[tagname]
    dm = z['dm_' + tagname] if 'dm_' + tagname in z else z['d_' + tagname]
    dt = z['dt_' + tagname] if 'dt_' + tagname in z else z['d_' + tagname]

    class Tap(object):
        def __init__(self):
            self.sym = {0: [], 1: []}

        def append_bit(self, bit, packet_type):
            self.sym[packet_type].append(bit)

    tap = Tap()
    md, nd = miller.miller_decoder(tap), manchester.manchester_decoder(tap)
    pm = [(int(c), float(d) * factor) for c, d in zip(z['curm_' + tagname], dm)]
    pt = [(int(c), float(d) * factor) for c, d in zip(z['curt_' + tagname], dt)]
    for i in range(0, len(pm), 97):
        md.process_transition(pm[i:i + 97])
    for i in range(0, len(pt), 97):
        nd.process_transition(pt[i:i + 97])
    assert tap.sym[1] == z['symm_' + tagname].tolist()
    assert tap.sym[0] == z['symt_' + tagname].tolist()


def test_report_example_and_reqa_through_miller_decoder():
    fx = load_json('fx_report_miller.json')

    class Tap(object):
        def __init__(self):
            self.sym = []

        def append_bit(self, bit, packet_type):
            assert packet_type == PacketType.READER_TO_TAG
            self.sym.append(bit)

    for key in ('report_example', 'reqa'):
        tap = Tap()
        miller.miller_decoder(tap).process_transition([tuple(p) for p in fx[key]['pulses']])
        assert tap.sym == fx[key]['symbols']


def test_crc_a_and_b():
    for v in NAMES['crc']:
        assert utilities.CRC.calculate_crc(v['data']) == v['a']
        assert utilities.CRC.calculate_crc(v['data'], utilities.CRC.CRC_14443_B) == v['b']
        assert utilities.CRC.check_crc(v['data'] + v['b'], utilities.CRC.CRC_14443_B)
    # ISO/IEC 14443-3 annex B: CRC_B of 00 00 00 is CC C6 (transmitted CC first)
    assert utilities.CRC.calculate_crc([0, 0, 0], utilities.CRC.CRC_14443_B) == [0xCC, 0xC6]
    assert utilities.CRC.CRC_14443_A == 0x6363 and utilities.CRC.CRC_14443_B == 0xFFFF


def test_get_bytes_get_bits():
    class Cmd(object):
        def __init__(self, v):
            self.v = v

        def header(self):
            return list(self.v['header'])

        def needs_crc(self):
            return self.v['crc']

        def packet_type(self):
            return self.v['type']

    for v in NAMES['frames']:
        by = PacketType.get_bytes(Cmd(v), v['extra'])
        assert by == v['bytes']
        assert PacketType.get_bits(Cmd(v), by) == v['bits']


def test_packet_processor_and_errors():
    for v in NAMES['packet_processor']:
        pp = packets.PacketProcessor(v['type'])
        assert [pp.append_bit(s) for s in v['symbols']] == v['returns']
    for k, val in NAMES['packet_error'].items():
        assert getattr(packets.PacketError, k) == val
    with pytest.raises(ValueError):
        PacketType.start_bit(5)
