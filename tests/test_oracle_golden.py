"""The Python restatement (oracle/py_oracle.py) against the reference's own outputs
(golden vectors produced by tests/golden/make_golden.py from the unmodified reference)."""
import numpy as np
import pytest

from oracle import py_oracle as po
from tests.golden_util import PATH_CASES, Case, load_json, load_npz


class _Sink(object):
    def __init__(self):
        self.sym = {0: [], 1: []}

    def append_bit(self, bit, t):
        self.sym[t].append(int(bit))


def test_constants_match_reference_hex():
    c = load_json('fx_report_miller.json')['constants_hex']
    assert po.T_FULL.hex() == c['FULL'] and po.T_ZERO.hex() == c['ZERO'] and po.T_HALF.hex() == c['HALF']
    assert po.T_ZERO_REM.hex() == c['ZERO_REM'] and po.T_ONE_REM.hex() == c['ONE_REM']
    assert po.T_ONE_HALF.hex() == c['ONE_HALF']
    m = po.ManchesterDecoder(_Sink())
    assert (m.lo.hex(), m.mid.hex(), m.hi.hex()) == (c['MAN_LO'], c['MAN_MID'], c['MAN_HI'])
    d = po.MillerDecoder(_Sink())
    assert (d.lo.hex(), d.hi.hex()) == (c['MIL_LO'], c['MIL_HI'])


def test_report_worked_example_and_reqa():
    fx = load_json('fx_report_miller.json')
    s = _Sink()
    d = po.MillerDecoder(s)
    d.process_transition([tuple(p) for p in fx['report_example']['pulses']])
    assert s.sym[1] == fx['report_example']['symbols'] == [0, 1, 0, 1]   # report section 4.3
    assert d.stage() == fx['report_example']['stage']
    s = _Sink()
    po.MillerDecoder(s).process_transition([tuple(p) for p in fx['reqa']['pulses']])
    assert s.sym[1] == fx['reqa']['symbols']
    # report section 3.3: REQA on air = 0 0110010 0
    assert s.sym[1][:9] == [0, 0, 1, 1, 0, 0, 1, 0, 0]


def test_decoder_vectors():
    z = load_npz('fx_decoder_vectors.npz')
    for key, factor in (('1', 1.0), ('0p5', 0.5), ('0p25', 0.25), ('0p1', 0.1)):
        s = _Sink()
        po.MillerDecoder(s).process_transition([(int(c), int(k) * factor) for c, k in zip(z['curm_' + key], z['d_' + key])])
        assert s.sym[1] == z['symm_' + key].tolist()
        s = _Sink()
        po.ManchesterDecoder(s).process_transition([(int(c), int(k) * factor) for c, k in zip(z['curt_' + key], z['d_' + key])])
        assert s.sym[0] == z['symt_' + key].tolist()
    for key, factor in (('frames_0p5', 0.5), ('frames_0p25', 0.25)):
        s = _Sink()
        po.MillerDecoder(s).process_transition([(int(c), int(k) * factor) for c, k in zip(z['curm_' + key], z['dm_' + key])])
        assert s.sym[1] == z['symm_' + key].tolist()
        assert set(s.sym[1]) >= {0, 1}
        s = _Sink()
        po.ManchesterDecoder(s).process_transition([(int(c), int(k) * factor) for c, k in zip(z['curt_' + key], z['dt_' + key])])
        assert s.sym[0] == z['symt_' + key].tolist()


@pytest.mark.parametrize('name', PATH_CASES)
@pytest.mark.parametrize('chunk', [8192, 777])
def test_whole_path_matches_reference(name, chunk):
    c = Case(name)
    r = po.run_path(c.x, chunk=chunk, **c.params)
    assert r['transitions'] == c.transitions
    assert r['symbols_tag'] == c.sym_tag.tolist()
    assert r['symbols_reader'] == c.sym_reader.tolist()
    assert r['packets'] == c.packets


def test_envelope_of_iq_fixture():
    iq = load_npz('fx_ultralight_iq.npz')['iq']
    c = Case('fx_ultralight_txn')
    assert np.array_equal(po.envelope_iq(iq), c.x)


@pytest.mark.parametrize('case', ['inf', 'nan'])
def test_nonfinite_iq_fixture(case):
    # samples that are not finite, given as IQ (fx_nonfinite_iq.npz: the reference ran on the envelope of this IQ)
    iq = load_npz('fx_nonfinite_iq.npz')[case + '_iq']
    c = Case('fx_nonfinite_iq:' + case, prefix=case + '_', file='fx_nonfinite_iq')
    assert np.array_equal(po.envelope_iq(iq), c.x, equal_nan=True) and not np.isfinite(c.x).all()
    r = po.run_path(c.x, **c.params)
    assert r['transitions'] == c.transitions and r['packets'] == c.packets
    o = co.COracle(**c.params)
    o.push_iq(iq)
    assert o.transitions() == c.transitions and o.packets() == c.packets
    assert o.symbols(0).tolist() == c.sym_tag.tolist() and o.symbols(1).tolist() == c.sym_reader.tolist()


def test_low_run_timeout_fixture_holds_every_v():
    # SURVEY 7 hard part 4: v in {-1, 0, 1, 2}; v = -1 only behind a LOW run whose last sample fired a time-out
    c = Case('fx_low_run_timeout')
    assert set(c.tr_v.tolist()) == {-1, 0, 1, 2}


# ---------------------------------------------------------------------------
# the C restatement (oracle/nfc_oracle.c) against the same vectors
# ---------------------------------------------------------------------------
from oracle import c_oracle as co


@pytest.mark.parametrize('name', PATH_CASES)
@pytest.mark.parametrize('chunk', [0, 777])
def test_c_oracle_matches_reference(name, chunk):
    c = Case(name)
    o = co.COracle(**c.params)
    if chunk:
        for i in range(0, len(c.x), chunk):
            o.push_env(c.x[i:i + chunk])
    else:
        o.push_env(c.x)
    assert o.transitions() == c.transitions
    assert o.symbols(0).tolist() == c.sym_tag.tolist()
    assert o.symbols(1).tolist() == c.sym_reader.tolist()
    assert o.packets() == c.packets


def test_c_oracle_iq_envelope():
    iq = load_npz('fx_ultralight_iq.npz')['iq']
    c = Case('fx_ultralight_txn')
    o = co.COracle(**c.params)
    o.push_iq(iq)
    assert o.transitions() == c.transitions
    assert o.packets() == c.packets


def test_c_oracle_equals_python_oracle_on_random_streams():
    rng = np.random.default_rng(5)
    for trial in range(6):
        n = int(rng.integers(2500, 9000))
        base = rng.uniform(0.05, 2.0)
        x = (base * (1 + 0.05 * rng.standard_normal(n))).astype(np.float32)
        # bursts of low / high samples
        for _ in range(int(rng.integers(3, 30))):
            s = int(rng.integers(0, n - 80))
            k = int(rng.integers(1, 80))
            x[s:s + k] *= np.float32(rng.choice([0.0, 0.05, 0.09, 0.11, 1.09, 1.1, 1.12, 1.5]))
        x = np.abs(x)
        kw = dict(samp_rate=float(rng.choice([1e6, 2e6, 4e6, 1e7])), hi_val=float(rng.choice([1.05, 1.09, 1.1])),
                  av_window=int(rng.choice([100, 777, 2000])), max_len=int(rng.choice([7, 30, 50])))
        r = po.run_path(x, chunk=1000, want_trace=True, **kw)
        o = co.COracle(trace=True, **kw)
        o.push_env(x)
        assert o.transitions() == r['transitions']
        assert o.trace().tolist() == r['trace']
        assert o.symbols(0).tolist() == r['symbols_tag']
        assert o.symbols(1).tolist() == r['symbols_reader']
        assert o.packets() == r['packets']
