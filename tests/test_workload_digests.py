"""The oracles and the GPU path against what the UNMODIFIED reference produced for prefixes of the bench workloads that are too
long to commit as vectors: 2 M samples of the three 2 Msps workloads, 3 M samples of the 10 Msps Classic-1K capture -- sizes where
the threshold kernel's time chunks have their production length (tests/golden/make_workload_digests.py wrote
fx_workload_digests.json; tests/digests.py defines what is hashed)."""
import numpy as np
import pytest

from tests import digests
from tests.golden_util import load_json

FX = load_json('fx_workload_digests.json')
KEYS = ('n_transitions', 'n_sym_tag', 'n_sym_reader', 'n_packets', 'transitions', 'sym_tag', 'sym_reader', 'packets')


def _path_kw(kw):
    return dict(samp_rate=kw['samp_rate'], hi_val=kw['hi_val'], av_window=kw.get('av_window', 2000), max_len=kw.get('max_len', 50),
                reader=kw['reader'], tag=kw['tag'])


def _same(got, name):
    want = FX[name]
    for k in KEYS:
        assert got[k] == want[k], '%s: %s differs from the reference (%r vs %r)' % (name, k, got[k], want[k])


@pytest.mark.parametrize('name', digests.PINNED)
def test_c_oracle_reproduces_the_reference_digests(name):
    from oracle import c_oracle as co
    iq, kw = digests.workload_prefix(name)
    assert len(iq) // 2 == FX[name]['samples']
    o = co.COracle(**_path_kw(kw))
    o.push_iq(iq)
    _same(digests.digest_result(o.transitions(), o.symbols(0).tolist(), o.symbols(1).tolist(), o.packets()), name)


@pytest.mark.parametrize('name', ('miller', 'classic1k'))
def test_py_oracle_reproduces_the_reference_digests(name):
    # (the line-for-line Python restatement: two of the four, at a fifth of a second per million samples each is enough)
    from oracle import py_oracle as po
    from usrp_nfc_amd import synth
    iq, kw = digests.workload_prefix(name)
    r = po.run_path(synth.envelope_f32(iq), **_path_kw(kw))
    _same(digests.digest_result(r['transitions'], r['symbols_tag'], r['symbols_reader'], r['packets']), name)


@pytest.mark.gpu
@pytest.mark.parametrize('name', digests.PINNED)
@pytest.mark.parametrize('pushes', (1, 3))
def test_gpu_reproduces_the_reference_digests(name, pushes):
    # the HIP path against the reference itself (not the oracle): one push, and three pushes cut off the step size
    from usrp_nfc_amd import api
    iq, kw = digests.workload_prefix(name)
    n = len(iq) // 2
    cuts = [0, n] if pushes == 1 else [0, n // 3 + 7, 2 * n // 3 + 1001, n]
    tr, s0, s1, pk = [], [], [], []
    with api.NfcContext(input_kind=api.NFC_IN_IQ_F32, **_path_kw(kw)) as ctx:
        for a, b in zip(cuts[:-1], cuts[1:]):
            ctx.push(iq[2 * a:2 * b])
            tr += ctx.transitions()
            s0 += ctx.symbols(0).tolist()
            s1 += ctx.symbols(1).tolist()
            pk += ctx.packets()
            st = ctx.stats()
            assert st.used_sequential == 0
    _same(digests.digest_result(tr, s0, s1, pk), name)
