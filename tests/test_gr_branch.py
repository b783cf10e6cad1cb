"""The GNU Radio branch of usrp_nfc_amd.decoder (decoder.py:15-33) executed against stand-ins for ``gnuradio.gr`` /
``gnuradio.blocks`` / ``usrp_src``: GNU Radio is not in this image, so the blocks are recording fakes -- what is checked
is OUR side of the boundary: the class is a ``gr.hier_block2``, the ``connect`` graph and its order, ``hi_val`` 1.09 for a
recording and 1.1 for the UHD source, ``transition_sink`` as a ``gr.sync_block`` with the reference's ``in_sig``, and (on
the GPU) the Ultralight transaction pumped through the stubbed sink's ``work()`` in scheduler-sized calls."""
import importlib
import sys
import types

import numpy as np
import pytest

from tests.golden_util import Case


class _Block(object):
    def __init__(self, kind, *args):
        self.kind, self.args = kind, args

    def __repr__(self):
        return '%s%r' % (self.kind, self.args)


def _install_stubs(log):
    gnuradio = types.ModuleType('gnuradio')
    gr = types.ModuleType('gnuradio.gr')
    blocks = types.ModuleType('gnuradio.blocks')

    class hier_block2(object):
        def __init__(self, name, in_sig, out_sig):
            self.gr_name, self.gr_in, self.gr_out = name, in_sig, out_sig
            self.connections = []

        def connect(self, *chain):
            self.connections.append(chain)
            log.append(('connect',) + tuple(chain))

    class sync_block(object):
        def __init__(self, name=None, in_sig=None, out_sig=None):
            self.gr_name, self.gr_in, self.gr_out = name, in_sig, out_sig

    gr.hier_block2 = hier_block2
    gr.sync_block = sync_block
    gr.io_signature = lambda lo, hi, size: ('io_signature', lo, hi, size)
    blocks.wavfile_source = lambda path, repeat: _Block('wavfile_source', path, repeat)
    blocks.float_to_complex = lambda vlen: _Block('float_to_complex', vlen)
    blocks.complex_to_mag_squared = lambda vlen: _Block('complex_to_mag_squared', vlen)
    gnuradio.gr, gnuradio.blocks = gr, blocks
    usrp_src = types.ModuleType('usrp_src')
    usrp_src.usrp_src = lambda samp_rate, dst: _Block('usrp_src', samp_rate, dst)
    return {'gnuradio': gnuradio, 'gnuradio.gr': gr, 'gnuradio.blocks': blocks, 'usrp_src': usrp_src}


@pytest.fixture
def gr_modules():
    """usrp_nfc_amd.transition_sink / .decoder re-imported with the stand-ins in place; the real modules come back afterwards."""
    log = []
    stubs = _install_stubs(log)
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    import usrp_nfc_amd.decoder as dec_mod
    import usrp_nfc_amd.transition_sink as ts_mod
    try:
        ts_mod = importlib.reload(ts_mod)
        dec_mod = importlib.reload(dec_mod)
        yield types.SimpleNamespace(decoder=dec_mod, transition_sink=ts_mod, log=log, stubs=stubs)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        importlib.reload(ts_mod)
        importlib.reload(dec_mod)


class _FakeContext(object):
    """Stands in for api.NfcContext on the CPU: records what transition_sink asks for."""
    made = []

    def __init__(self, **kw):
        self.kw = kw
        _FakeContext.made.append(self)

    def close(self):
        pass


def _check_wiring(m, d, src_kind, hi_val):
    gr = m.stubs['gnuradio.gr']
    assert isinstance(d, gr.hier_block2) and d.gr_name == 'decoder'
    assert d.gr_in == ('io_signature', 0, 0, 0) and d.gr_out == ('io_signature', 0, 0, 0)      # decoder.py:18-19
    assert isinstance(d._trans, gr.sync_block) and d._trans.gr_name == 'transition_sink'
    assert d._trans.gr_in == [np.float32] and d._trans.gr_out is None                           # transition_sink.py:13-18
    if src_kind == 'wav':
        # decoder.py:25-28: wavfile_source -> float_to_complex -> complex_to_mag_squared, then that -> transition_sink
        assert [c[0] for c in m.log] == ['connect', 'connect']
        chain, last = m.log[0][1:], m.log[1][1:]
        assert [b.kind for b in chain] == ['wavfile_source', 'float_to_complex', 'complex_to_mag_squared']
        assert chain[0].args == ('x.wav', True) and chain[1].args == (1,) and chain[2].args == (1,)
        assert last == (chain[2], d._trans)
    else:
        # decoder.py:21-23: usrp_src(samp_rate, dst) -> transition_sink
        assert [c[0] for c in m.log] == ['connect']
        src, sink = m.log[0][1:]
        assert src.kind == 'usrp_src' and src.args == (4e6, 'out.wav') and sink is d._trans
    return hi_val


def test_gr_branch_wiring_and_hi_val(gr_modules, monkeypatch):
    m = gr_modules
    assert m.decoder._gr is m.stubs['gnuradio.gr']
    _FakeContext.made = []
    monkeypatch.setattr(m.transition_sink.api, 'NfcContext', _FakeContext)
    d = m.decoder.decoder(src='x.wav', repeat=True, reader=True, tag=False, samp_rate=2e6)
    _check_wiring(m, d, 'wav', 1.09)
    kw = _FakeContext.made[-1].kw
    assert kw['hi_val'] == 1.09 and kw['samp_rate'] == 2e6 and kw['reader'] is True and kw['tag'] is False   # decoder.py:29-32
    assert kw['lo_val'] == 0.1 and kw['av_window'] == 2000 and kw['max_len'] == 50                          # transition_sink.py:12
    assert kw['input_kind'] == m.transition_sink.api.NFC_IN_ENV_F32    # what the sink is handed is the float32 envelope
    assert d._back._attached and d._back.reader and not d._back.tag
    del m.log[:]
    d = m.decoder.decoder(src='uhd', dst='out.wav', samp_rate=4e6)
    _check_wiring(m, d, 'uhd', 1.1)
    kw = _FakeContext.made[-1].kw
    assert kw['hi_val'] == 1.1 and kw['samp_rate'] == 4e6 and kw['reader'] is True and kw['tag'] is True      # decoder.py:23


class _Fsm(object):
    def __init__(self):
        self.got = []

    def process_bits(self, bits, packet_type):
        self.got.append((packet_type, list(bits)))


@pytest.mark.gpu
def test_gr_branch_pumps_the_ultralight_transaction(gr_modules):
    m = gr_modules
    c = Case('fx_ultralight_txn')
    d = m.decoder.decoder(src='x.wav', repeat=True, reader=True, tag=True, samp_rate=c.params['samp_rate'])
    _check_wiring(m, d, 'wav', 1.09)
    # the fixture was generated with its own hi_val: a second sink with it, behind the same background, takes the stream
    f = _Fsm()
    d._trans.close()
    back = m.decoder.background(True, True, None, fsm=f)
    sink = m.transition_sink.transition_sink(c.params['samp_rate'], back.append, hi_val=c.params['hi_val'])
    assert isinstance(sink, m.stubs['gnuradio.gr'].sync_block)
    i = 0
    while i < len(c.x):   # the scheduler's calls: work(input_items, output_items) -> items consumed
        i += sink.work([c.x[i:i + 8192]], None)
    assert sink.stop() is True
    assert f.got == c.packets and len(f.got) == 19
    sink.close()
