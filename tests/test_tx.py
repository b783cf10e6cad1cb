"""Row f4: the transmit side.  Encoders and the renderer against vectors the unmodified reference produced
(tests/golden/fx_tx.json), the oracle restatement against the same, and -- on the GPU -- the renderer against the oracle
plus the round trip through the receive path."""
import json
import os

import numpy as np
import pytest

from oracle import tx_oracle as txo
from usrp_nfc_amd import api, synth, tx
from usrp_nfc_amd.binary_src import binary_src, encoder
from usrp_nfc_amd.manchester import manchester_encoder
from usrp_nfc_amd.miller import miller_encoder
from usrp_nfc_amd.multiplier import multiplier

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, 'golden', 'fx_tx.json')))['cases']
ENC = dict(same=encoder, manchester=manchester_encoder, miller=miller_encoder)
ORACLE_ENC = dict(same=txo.same_encode, manchester=txo.manchester_encode, miller=txo.miller_encode)


def want_runs(c):
    return [(l, float.fromhex(d)) for l, d in c['runs']]


def want_samples(c):
    parts = [np.full(n, v, np.int8) for v, n in c['samples_rle']]
    return np.concatenate(parts) if parts else np.zeros(0, np.int8)


def test_oracle_encoders_and_render_match_the_reference():
    for c in CASES:
        runs = ORACLE_ENC[c['encoding']](c['bits'])
        assert runs == want_runs(c), c['encoding']          # durations bit for bit (float.hex)
        got = txo.render(runs, c['samp_rate'])
        assert np.all(got.imag == 0)
        assert np.array_equal(got.real.astype(np.int8), want_samples(c))


def test_encoders_through_the_abi_match_the_reference():
    for c in CASES:
        assert ENC[c['encoding']].encode_bits(c['bits']) == want_runs(c)
        assert tx.sample_count(tx.as_runs(want_runs(c)), c['samp_rate']) == len(want_samples(c))


def test_encode_rejects_non_bits():
    with pytest.raises(RuntimeError):
        tx.encode_bits(tx.NFC_TX_MILLER, [0, 1, 2])


def test_binary_src_queue_and_pauses():
    # set_bits wraps the frame in pauses (binary_src.py:46-63); Python-2 integer division for pause/2 and pause/div
    s = binary_src(2e6, encode='miller', idle_bit=1, pause_dur=2500)
    s.set_bits([0, 1, 1, 0, 0, 1, 0], has_finished=True)
    runs = s.runs()
    assert [(int(l), float(d)) for l, d in zip(runs['level'][:3], runs['dur_us'][:3])] == [(1, 1000.0), (1, 1000.0), (1, 500.0)]
    assert s.n_samples() == 2 * 5000 + len(txo.render(txo.miller_encode([0, 1, 1, 0, 0, 1, 0]), 2e6))
    s2 = binary_src(2e6, encode='manchester')
    s2.set_bits([1, 0], pause=0)
    assert int(s2.runs()['level'][0]) == 2 and s2.n_samples() == len(txo.render(txo.manchester_encode([1, 0]), 2e6))


def test_multiplier_host_carrier_matches_the_oracle():
    m = multiplier(samp_rate=4e6, freq=13.56e6, A=0.7)
    assert np.array_equal(m.carrier(5000, 123), txo.carrier(5000, 4e6, 13.56e6, 0.7, 123))
    assert np.allclose(np.abs(m.carrier(1000)), 0.7, atol=1e-6)


@pytest.mark.gpu
def test_renderer_matches_reference_vectors():
    for c in CASES:
        got = tx.render(tx.as_runs(want_runs(c)), c['samp_rate'])
        assert np.all(got.imag == 0)
        assert np.array_equal(got.real.astype(np.int8), want_samples(c)), (c['encoding'], c['samp_rate'])


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(4))
def test_renderer_long_streams_vs_oracle(seed):
    # many frames with pauses, run lengths from 0 samples up; crosses many tiles, both search paths of the kernel
    rng = np.random.default_rng(seed)
    rate = float(rng.choice([2e6, 4e6, 13.56e6]))
    pulses = []
    for _ in range(300):
        bits = rng.integers(0, 2, int(rng.integers(1, 60))).tolist()
        enc = [txo.miller_encode, txo.manchester_encode, txo.same_encode][int(rng.integers(0, 3))]
        pulses += [(2, 0)] + enc(bits) + [(int(rng.integers(0, 2)), float(rng.integers(0, 3000)))]
    if seed == 3:   # runs shorter than a sample: more than 1024 runs inside one tile
        pulses += [(int(i & 1), 0.3) for i in range(5000)] + [(1, 700.0)]
    want = txo.render(pulses, rate)
    got = tx.render(tx.as_runs(pulses), rate)
    assert np.array_equal(got, want)
    # carrier: stated arithmetic, fp32 sincospi on the device vs float64 cos / sin in the oracle: 2 ulp of the amplitude
    gotc = tx.render(tx.as_runs(pulses), rate, carrier=True, freq=13.56e6, amp=0.5, first_index=1 << 40)
    wantc = want * txo.carrier(len(want), rate, 13.56e6, 0.5, 1 << 40)
    assert np.max(np.abs(gotc - wantc)) <= 1.5e-7


@pytest.mark.gpu
def test_round_trip_tx_to_rx():
    # what the TX side renders, the RX path decodes: reader frames (Miller pauses) through the carrier, |.|^2, thresholds
    frames = [synth.frame_bits([0x26], 7), synth.frame_bits([0x93, 0x20]), synth.frame_bits([0x30, 0x04, 0x26, 0xEE])]
    src = binary_src(2e6, encode='miller', idle_bit=1, pause_dur=600)
    idle = [(1, 1600.0)]
    src._bits = list(idle)
    for b in frames:
        src.set_bits(b, has_finished=True)
    src._bits += idle
    iq = src.render(carrier=multiplier(samp_rate=2e6, freq=13.56e6, A=0.5))
    ctx = api.NfcContext(samp_rate=2e6, hi_val=1.1, input_kind=api.NFC_IN_IQ_F32, reader=True, tag=False)
    ctx.push(np.ascontiguousarray(iq).view(np.float32))
    pk = ctx.packets()
    ctx.close()
    assert [t for t, _ in pk] == [1, 1, 1]
    # the packet is the frame's bits plus the encoder's end-of-frame zero; fsm.process_bits repairs the ending, checks
    # parity and splits the bytes (row f1)
    for (t, bits), want in zip(pk, frames):
        assert bits[:len(want)] == want and len(bits) - len(want) <= 2
    import io
    from usrp_nfc_amd.fsm import fsm
    f = fsm(callback=lambda cmd, st: None, out=io.StringIO())
    got = [f.process_bits(bits, t).all_bytes() for t, bits in pk]
    assert got == [[0x26], [0x93, 0x20], [0x30, 0x04, 0x26, 0xEE]]
