"""Row f1: closed packets -> bytes -> commands.  Golden: the reference's own trace of the Ultralight
transaction, outputs/ultralight.out (a data file of the reference, copied to tests/golden/), which fsm.process_bits
+ CommandStructure.display printed from these very packets."""
import io
import os

import numpy as np
import pytest

from tests.golden_util import Case
from usrp_nfc_amd import command, fsm, utilities

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'ultralight.out')


def trace_of(packets):
    out = io.StringIO()
    m = fsm.fsm(out=out)
    for t, bits in packets:
        m.process_bits(bits, t)
    return out.getvalue()


def test_crc_a_known_answers():
    # CRC fields printed in the reference's trace
    assert fsm.crc_a([0x93, 0x70, 0x88, 0x04, 0xBE, 0x6F, 0x5D]) == [0xA1, 0x8E]      # SEL1R
    assert fsm.crc_a([0x04]) == [0xDA, 0x17]                                            # SEL1U
    assert utilities.CRC.check_crc([0x50, 0x00, 0x57, 0xCD])                            # HALT (ISO 14443-3 annex B)
    assert not utilities.CRC.check_crc([0x50, 0x00, 0x57, 0xCC])
    assert fsm.crc_a([]) == [0x63, 0x63]


def test_command_table_matches_reference_names():
    names = [c.name() for c in command.CommandType.table()]
    assert names[:6] == ['REQA', 'WUPA', 'ATQAUL', 'ATQA1K', 'ATQA1K', 'ATQADS']      # (the 4K answer is named ATQA1K, command.py:84)
    assert command.CommandType.SEL1R.header() == [0x93, 0x70] and command.CommandType.SEL1R.total_len() == 9
    assert command.CommandType.READT.total_len() == 18 and command.CommandType.HALT.stage() == 10
    assert command.CommandType.ANTI1U.packet_type() == 0 and command.CommandType.REQA.packet_type() == 1


def test_ultralight_trace_from_golden_packets():
    c = Case('fx_ultralight_txn')
    assert len(c.packets) == 19
    # (the file lacks the last of the two blank lines display() prints after the final command)
    assert trace_of(c.packets).rstrip('\n') == open(GOLD).read().rstrip('\n')


def test_frame_end_repair_parity_and_unknown():
    to_bits = utilities.Convert.to_bit_ar
    out = io.StringIO()
    m = fsm.fsm(out=out)
    reqa = [0, 1, 1, 0, 0, 1, 0]                       # 0x26 as a 7-bit short frame
    st = m.process_bits(reqa + [0], 1)                 # 8 bits: the missing ninth is assumed (fsm.py:56-57)
    assert st.name() == 'REQA' and st.header() == [0x26]
    st = m.process_bits(to_bits([0x44, 0x00], parity=True) + [1], 0)   # one extra bit equal to the start bit: dropped
    assert st.name() == 'ATQAUL'
    st = m.process_bits(to_bits([0x93, 0x20], parity=True) + [1], 1)   # extra bit that is NOT the start bit: reported
    assert st.name() == 'ANTI1R' and 'EXTRA ERROR' in out.getvalue()
    bad = to_bits([0x88, 0x04, 0xBE, 0x6F, 0x5D], parity=True)
    bad[8] ^= 1
    assert m.process_bits(bad, 0) is None and 'PARITY ERROR' in out.getvalue()
    st = m.process_bits(to_bits([0x12, 0x34, 0x56], parity=True), 1)
    assert st.name() == 'UNKNOWN' and st.extra() == [0x12, 0x34, 0x56]
    st = m.process_bits(to_bits([0x26], parity=True)[:9] + [0, 1, 0], 1)   # 12 bits: "MANY MORE ERROR", cut to nine
    assert st.name() == 'REQA' and 'MANY MORE ERROR' in out.getvalue()


def test_bcc_and_crc_gate_the_lookup():
    to_bits = utilities.Convert.to_bit_ar
    m = fsm.fsm(out=io.StringIO())
    m.process_bits(to_bits([0x26], parity=True)[:8], 1)
    m.process_bits(to_bits([0x44, 0x00], parity=True), 0)
    m.process_bits(to_bits([0x93, 0x20], parity=True), 1)
    st = m.process_bits(to_bits([0x88, 0x04, 0xBE, 0x6F, 0x5C], parity=True), 0)   # wrong BCC: not ANTI1U, not ANTI1G
    assert st.name() == 'UNKNOWN'
    good = [0x93, 0x70, 0x88, 0x04, 0xBE, 0x6F, 0x5D, 0xA1, 0x8E]
    assert m.process_bits(to_bits(good, parity=True), 1).name() == 'SEL1R'
    good[-1] ^= 0x10
    assert m.process_bits(to_bits(good, parity=True), 1).name() == 'UNKNOWN'


@pytest.mark.gpu
def test_ultralight_iq_to_trace_on_gpu():
    # capture -> GPU decode -> batch protocol layer: the reference's printed trace, end to end
    from usrp_nfc_amd import api
    iq = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'fx_ultralight_iq.npz'))['iq']
    ctx = api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32)
    ctx.push(iq)
    tabs = [ctx.packet_table(t) for t in (0, 1)]
    bits = [ctx.packet_bits(t) for t in (0, 1)]
    table = np.concatenate(tabs)
    table = table[np.argsort(table['idx'], kind='stable')]
    table = table[table['n_bits'] > 0]            # empty packets are discarded (packets.py:97)
    out = io.StringIO()
    frames, data = fsm.fsm(out=out).process_packets(table, bits[0], bits[1])
    assert out.getvalue().rstrip('\n') == open(GOLD).read().rstrip('\n')
    assert len(frames) == 19 and (frames['cmd'] >= 0).all()
    ctx.close()


# ---- row f3: CRYPTO1 ------------------------------------------------------------------------------------------
GOLD_1K = os.path.join(os.path.dirname(__file__), 'golden', '1k_with_enc.out')


def packets_from_trace(path):
    """The packets the framing stage would hand over for the frames of a printed trace (synth.frames_from_trace):
    a short frame arrives as its seven bits and the end bit."""
    from usrp_nfc_amd import synth
    frames, text = synth.frames_from_trace(path)
    return [(d, bits + [0] if len(bits) == 7 else bits) for d, bits in frames], text


def test_classic_1k_trace_with_crypto1():
    packets, text = packets_from_trace(GOLD_1K)
    assert len(packets) == 202
    assert trace_of(packets).rstrip('\n') == text.rstrip('\n')
    assert text.count('AR OK') == 16 and text.count('AT OK') == 16      # sixteen authentications, fifteen of them nested


def test_crypto1_wrong_key_is_noticed():
    packets, _ = packets_from_trace(GOLD_1K)
    out = io.StringIO()
    m = fsm.fsm(out=out)
    m.set_keys([0xA0, 0xA1, 0xA2, 0xA3, 0xA4, 0xA5], [0xFF] * 6)
    for t, bits in packets[:12]:
        m.process_bits(bits, t)
    assert 'AR OK' not in out.getvalue()


@pytest.mark.parametrize('role', [0, 1])   # the emulator is the tag (0: sends TAG_TO_READER frames) / the reader (1)
def test_process_outgoing_puts_the_trace_s_ciphertext_on_the_air(role):
    # fsm.process_outgoing (fsm.py:68-112; the emulator's encoder hook of packets.py:88-90), pinned to the reference's own trace:
    # outputs/1k_with_enc.out carries the on-air ciphertext of every frame of 16 authentications (15 nested).  An emulator in
    # either role that sends the PLAIN frames of that transaction through process_outgoing -- and hears the other side through
    # process_bits -- must put exactly those ciphertext bits on the air, frame after frame.
    from usrp_nfc_amd import synth
    packets, _ = packets_from_trace(GOLD_1K)
    seen = []
    monitor = fsm.fsm(callback=lambda cmd, st: seen.append((cmd, st)), out=io.StringIO())
    plain = []
    for t, bits in packets:   # what a monitor decodes: the command and the plain bytes of every frame
        n0 = len(seen)
        monitor.process_bits(bits, t)
        assert len(seen) == n0 + 1 and seen[-1][0] is not None, 'the monitor must know every frame of the trace'
        plain.append((t, seen[-1][0], seen[-1][1].all_bytes()))
    emu = fsm.fsm(callback=lambda cmd, st: None, out=io.StringIO())
    sent = encrypted = nested = 0
    in_session = halted = False
    for (t, air), (_, cmd, data) in zip(packets, plain):
        if halted:
            # (the capture's reader repeats HALT and REQA in the clear once the session is over.  process_outgoing never ends a
            # session -- it does not run process_command, fsm.py:68-112 -- so an emulated reader would go on encrypting: nothing
            # of the trace's tail can pin it)
            break
        if cmd.name() in ('REQA', 'WUPA', 'HALT'):
            in_session = False
            halted = cmd.name() == 'HALT' and t == role
        if t != role:
            emu.process_bits(air, t)        # heard
            continue
        short = len(air) == 8 and cmd.name() in ('REQA', 'WUPA')
        bits = synth.frame_bits(data, 7) + [0] if short else synth.frame_bits(data)
        out = emu.process_outgoing(bits, cmd)
        if cmd.name() == 'RANDTA' and in_session:
            # A NESTED tag nonce: the reference's emulator hook sends it under the OLD session's keystream (fsm.py:96-97,
            # `old_enc.enc_bits(bits)`), a real card -- the trace -- under the new sector key's while uid ^ nonce is fed in.
            # What process_outgoing leaves behind is the new register either way: the frames that follow (at, the data) must
            # match the trace again, which is what pins it.
            assert out != bits and out != list(air)
            nested += 1
        else:
            assert out == list(air), 'frame %d (%s): not what the trace has on the air' % (sent, cmd.name())
        if cmd.name() == 'RANDTA':
            in_session = True
        sent += 1
        encrypted += int(out != bits)
    assert sent >= 95 and encrypted > 60 and nested == (15 if role == 0 else 0), (sent, encrypted, nested)


def test_process_outgoing_before_a_tag_type_and_with_an_ultralight():
    # fsm.py:101-108: no tag type yet -- an ATQA sets it, the bits pass; fsm.py:71-72: an Ultralight's frames go through process_bits
    # (the callback sees them) and pass unchanged
    from usrp_nfc_amd import synth
    seen = []
    m = fsm.fsm(callback=lambda cmd, st: seen.append(cmd.name() if cmd else None), out=io.StringIO())
    atqa = synth.frame_bits([0x44, 0x00])
    assert m.process_outgoing(atqa, command.CommandType.ATQAUL) == atqa and seen == []
    sel = synth.frame_bits([0x04] + fsm.crc_a([0x04]))
    m.process_bits(synth.frame_bits([0x26], 7) + [0], 1)   # (REQA heard: the machine is at the start of a transaction)
    seen.clear()
    m2 = fsm.fsm(callback=lambda cmd, st: seen.append(cmd.name() if cmd else None), out=io.StringIO())
    assert m2.process_outgoing(atqa, command.CommandType.ATQAUL) == atqa
    out = m2.process_outgoing(sel, command.CommandType.SEL1U)
    assert out == sel and len(seen) == 1   # (through process_bits: the callback saw the frame)


@pytest.mark.gpu
def test_classic_1k_capture_to_trace_on_gpu():
    # BASELINE.json configs[3] in miniature: the MIFARE Classic 1K transaction of outputs/1k_with_enc.out, synthesised
    # at 10 Msps from its on-air bits (ciphertext included), decoded on the GPU with the constructor arguments scaled
    # to the rate (hard part 5), then through the protocol layer with CRYPTO1: the printed trace
    from usrp_nfc_amd import api, synth
    frames, text = synth.frames_from_trace(GOLD_1K)
    m = synth.modulation_profile(frames, rate_msps=10.0, lead_in=15000, tail=2000)
    iq = synth.iq_from_profile(m, seed=5)
    ctx = api.NfcContext(samp_rate=10e6, hi_val=1.1, av_window=10000, max_len=250, input_kind=api.NFC_IN_IQ_F32)
    ctx.push(iq)
    assert ctx.stats().used_sequential == 0
    tabs = [ctx.packet_table(t) for t in (0, 1)]
    bits = [ctx.packet_bits(t) for t in (0, 1)]
    table = np.concatenate(tabs)
    table = table[np.argsort(table['idx'], kind='stable')]
    table = table[table['n_bits'] > 0]
    out = io.StringIO()
    frames_out, _ = fsm.fsm(out=out).process_packets(table, bits[0], bits[1])
    assert len(frames_out) == 202
    assert out.getvalue().rstrip('\n') == text.rstrip('\n')
    ctx.close()
