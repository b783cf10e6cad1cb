"""The reference-named Python modules (decoder / transition_sink / background): call surface on
CPU, behaviour on the GPU against the golden vectors and the oracle."""
import inspect

import numpy as np
import pytest

from tests.golden_util import Case


def test_reference_call_surface():
    from usrp_nfc_amd import background, decoder, packets, transition_sink, utilities
    # decoder.py:16
    p = inspect.signature(decoder.decoder.__init__).parameters
    assert list(p)[:8] == ['self', 'src', 'dst', 'repeat', 'reader', 'tag', 'samp_rate', 'emulator']
    assert (p['src'].default, p['dst'].default, p['repeat'].default, p['reader'].default, p['tag'].default,
            p['samp_rate'].default, p['emulator'].default) == ("uhd", None, False, True, True, 2e6, None)
    # transition_sink.py:12
    p = inspect.signature(transition_sink.transition_sink.__init__).parameters
    assert list(p)[:7] == ['self', 'samp_rate', 'callback', 'lo_val', 'hi_val', 'av_window', 'max_len']
    assert (p['lo_val'].default, p['hi_val'].default, p['av_window'].default, p['max_len'].default) == (0.1, 1.1, 2000, 50)
    assert hasattr(transition_sink.transition_sink, 'work')
    # background.py:17,27
    p = inspect.signature(background.background.__init__).parameters
    assert list(p)[:4] == ['self', 'reader', 'tag', 'emulator']
    assert hasattr(background.background, 'append')
    # constants (utilities.py:7-23, packets.py:19-28)
    assert utilities.PulseLength.HALF == 4.72 and utilities.ErrorCode.WRONG_DUR == 6
    assert packets.PacketType.start_bit(0) == 1 and packets.PacketType.start_bit(1) == 0


class _Fsm(object):
    def __init__(self):
        self.got = []

    def process_bits(self, bits, packet_type):
        self.got.append((packet_type, list(bits)))


@pytest.mark.gpu
def test_transition_sink_work_contract_and_chunk_invariance():
    from usrp_nfc_amd.transition_sink import transition_sink
    c = Case('fx_ultralight_txn')
    for call_len, batch in ((8192, 1 << 22), (1000, 5000), (4097, 4097)):
        out = []
        ts = transition_sink(c.params['samp_rate'], out.extend, hi_val=c.params['hi_val'], batch=batch)
        i = 0
        while i < len(c.x):
            i += ts.work([c.x[i:i + call_len]], None)
        ts.stop()
        assert out == c.transitions
        ts.close()


@pytest.mark.gpu
def test_decoder_background_deliver_packets_to_fsm():
    from usrp_nfc_amd.decoder import decoder
    c = Case('fx_ultralight_txn')
    s = np.sqrt(c.x.astype(np.float64)).astype(np.float32)   # a "recording": the path squares it (decoder.py:26-27)
    f = _Fsm()
    d = decoder(src=s, reader=True, tag=True, samp_rate=2e6, fsm=f, batch=10000)
    back = d.run()
    from oracle import c_oracle as co
    o = co.COracle(**c.params)
    o.push_real_sq(s)
    assert f.got == o.packets() == back.packets
    assert len(f.got) == 19
    assert back.symbols[0] == o.symbols(0).tolist() and back.symbols[1] == o.symbols(1).tolist()


@pytest.mark.gpu
def test_decoder_reads_16bit_wav(tmp_path):
    import wave
    from usrp_nfc_amd import synth
    from usrp_nfc_amd.decoder import decoder
    frames = synth.txn_frames()
    m = synth.modulation_profile(frames, gap_us=90.0, depth=0.12)
    env = synth.envelope_f32(synth.iq_from_profile(m, seed=5))          # what usrp_src records (|IQ|^2)
    pcm = np.clip(np.round(env / env.max() * 30000), -32768, 32767).astype('<i2')
    path = str(tmp_path / 'ultralight_synth.wav')
    w = wave.open(path, 'wb')
    w.setnchannels(1); w.setsampwidth(2); w.setframerate(2000000)
    w.writeframes(pcm.tobytes()); w.close()
    f = _Fsm()
    decoder(src=path, reader=True, tag=True, samp_rate=2e6, fsm=f).run()
    from oracle import c_oracle as co
    o = co.COracle(samp_rate=2e6, hi_val=1.09)
    o.push_real_sq((pcm.astype(np.float32) / np.float32(32767.0)).astype(np.float32))   # wavfile_source: sample / 0x7FFF (IEEE division)
    assert f.got == o.packets()
    assert len(f.got) >= 17
    # a source normalised by a plain factor instead; and the non-default constructor arguments reach the one context
    f2 = _Fsm()
    decoder(src=pcm, reader=True, tag=True, samp_rate=2e6, fsm=f2, wav_scale=1.0 / 32768.0, av_window=1000, max_len=40).run()
    o2 = co.COracle(samp_rate=2e6, hi_val=1.09, av_window=1000, max_len=40)
    o2.push_real_sq((pcm.astype(np.float32) * np.float32(1.0 / 32768.0)).astype(np.float32))
    assert f2.got == o2.packets() and len(f2.got) >= 10


@pytest.mark.gpu
def test_live_flush_cadence():
    # work() in 8192-sample scheduler calls: with flush_ms=0 and batch=1 the callback comes once per call (the reference's
    # cadence, transition_sink.py:101); with a time limit the hand-over happens without waiting for `batch` samples
    import time
    from usrp_nfc_amd import synth
    from usrp_nfc_amd.background import background
    from usrp_nfc_amd.transition_sink import transition_sink
    from oracle import c_oracle as co
    x = synth.envelope_f32(synth.workload('all', 200_000))
    o = co.COracle(samp_rate=2e6, hi_val=1.1)
    o.push_env(x)
    for kw, expect_calls in ((dict(batch=1, flush_ms=0), 25), (dict(flush_ms=1.0), None), (dict(flush_calls=5, flush_ms=0), 5)):
        calls = []
        got = []
        f = _Fsm()
        back = background(True, True, fsm=f, keep=0)
        back.transitions = got
        ts = transition_sink(2e6, back.append, hi_val=1.1, **kw)
        orig = ts._callback
        ts._callback = lambda lst, _o=orig: (calls.append(1), _o(lst))[1]   # count the callbacks on their way to background.append
        for i in range(0, len(x), 8192):
            ts.work([x[i:i + 8192]], None)
            if 'flush_ms' in kw and kw['flush_ms']:
                time.sleep(0.0005)
        before_stop = len(calls)
        ts.stop()
        assert got == o.transitions() and f.got == o.packets()
        assert len(back.packets) == 0          # keep=0: nothing retained
        if expect_calls is not None:
            assert len(calls) == expect_calls, (kw, len(calls))
        else:
            assert before_stop >= 3            # handed over on the clock, long before 4 Mi samples were buffered
        ts.close()


@pytest.mark.gpu
def test_decoder_reads_raw_iq_file(tmp_path):
    # a raw interleaved complex64 capture (what a UHD file sink writes) takes the UHD branch: |IQ|^2, hi_val 1.1
    from usrp_nfc_amd import synth
    from usrp_nfc_amd.decoder import decoder
    iq = synth.workload('all', 400_000)
    path = str(tmp_path / 'capture.fc32')
    iq.tofile(path)
    f = _Fsm()
    decoder(src=path, reader=True, tag=True, samp_rate=2e6, fsm=f, batch=150_000).run()
    from oracle import c_oracle as co
    o = co.COracle(samp_rate=2e6, hi_val=1.1)
    o.push_iq(iq)
    assert f.got == o.packets() and len(f.got) > 50



@pytest.mark.gpu
def test_background_decodes_a_foreign_sinks_transitions():
    # the mirror of INTEGRATION.md section B: the REFERENCE's transition_sink (here its line-for-line restatement) feeding THIS
    # package's background through append(): lists it did not produce go through the device's decode and framing stages
    # (nfc_push_edges) and come out as the packets the reference's own background.run would have made (background.py:27-52)
    from oracle import py_oracle as po
    from usrp_nfc_amd import background as bg
    c = Case('fx_ultralight_txn')
    got = []

    class Fsm(object):
        def process_bits(self, bits, ptype):
            got.append((ptype, list(bits)))

    back = bg.background(reader=True, tag=True, fsm=Fsm(), samp_rate=c.params['samp_rate'], max_len=c.params['max_len'])
    ts = po.TransitionSink(c.params['samp_rate'], back.append, lo_val=c.params['lo_val'], hi_val=c.params['hi_val'],
                           av_window=c.params['av_window'], max_len=c.params['max_len'])
    po.drive_sink(ts, c.x, 8192)
    back.close()
    assert got == c.packets and len(got) == 19
    assert list(back.packets) == c.packets


@pytest.mark.gpu
def test_integration_md_hybrid_block_runs():
    # INTEGRATION.md section B's code block, executed as written (with a stand-in for gnuradio.gr and the library's real path):
    # the reference's transition_sink class body replaced by the ctypes calls, fed the Ultralight transaction in scheduler-sized
    # calls; the concatenated callback content must be the reference's transition list
    import os
    import re
    import types
    from usrp_nfc_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', md, re.S)
    code = [b for b in blocks if 'class transition_sink(gr.sync_block)' in b]
    assert len(code) == 1
    src = code[0].replace("C.CDLL('libnfc_amd.so')", 'C.CDLL(%r)' % build.SO)
    gr = types.SimpleNamespace()

    class sync_block(object):
        def __init__(self, name=None, in_sig=None, out_sig=None):
            pass

    gr.sync_block = sync_block
    ns = {'gr': gr}
    exec(compile(src, 'INTEGRATION.md', 'exec'), ns)
    c = Case('fx_ultralight_txn')
    out = []
    sink = ns['transition_sink'](c.params['samp_rate'], out.extend, lo_val=c.params['lo_val'], hi_val=c.params['hi_val'],
                                 av_window=c.params['av_window'], max_len=c.params['max_len'])
    for i in range(0, len(c.x), 8192):
        assert sink.work([c.x[i:i + 8192]], None) == len(c.x[i:i + 8192])
    assert out == c.transitions
