"""Parity of the HIP path (through the C-ABI) with the reference: golden vectors produced by the
unmodified reference, and the pinned C oracle on seeded inputs.  Bit-exact: integer / index work."""
import numpy as np
import pytest

from oracle import c_oracle as co
from tests.golden_util import PATH_CASES, Case, load_npz
from usrp_nfc_amd import _lib, api, synth

pytestmark = pytest.mark.gpu


def first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return i, a[i], b[i]
    return (n, None, None) if len(a) != len(b) else None


def run_gpu(x, params, kind=api.NFC_IN_ENV_F32, pushes=None, flags=0, chunk_samples=0):
    """Feed x in pieces; concatenate what each push produced."""
    ctx = api.NfcContext(input_kind=kind, flags=flags, chunk_samples=chunk_samples, **params)
    per = 2 if kind == api.NFC_IN_IQ_F32 else 1
    n = len(x) // per
    cuts = [0, n] if pushes is None else pushes
    tr, sym0, sym1, pk, val = [], [], [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        ctx.push(x[a * per:b * per])
        tr += ctx.transitions()
        sym0 += ctx.symbols(0).tolist()
        sym1 += ctx.symbols(1).tolist()
        pk += ctx.packets()
        val += ctx.val().tolist()
    st = ctx.stats()
    ctx.close()
    return dict(transitions=tr, sym_tag=sym0, sym_reader=sym1, packets=pk, val=val, stats=st)


def check_case(c, r):
    d = first_diff(r['transitions'], c.transitions)
    assert d is None, 'transition %s' % (d,)
    assert r['sym_tag'] == c.sym_tag.tolist()
    assert r['sym_reader'] == c.sym_reader.tolist()
    assert r['packets'] == c.packets


@pytest.mark.parametrize('name', PATH_CASES)
def test_golden_single_push(name):
    c = Case(name)
    check_case(c, run_gpu(c.x, c.params))


@pytest.mark.parametrize('name', PATH_CASES)
def test_golden_many_chunks(name):
    # smallest legal time chunk: exercises speculate / verify / look-back on the small fixtures
    c = Case(name)
    check_case(c, run_gpu(c.x, c.params, chunk_samples=256))


@pytest.mark.parametrize('name', PATH_CASES)
@pytest.mark.parametrize('step', [777, 8192])
def test_golden_streamed(name, step):
    # transition_sink.work() chunk invariance: any split of the stream gives the same outputs
    c = Case(name)
    cuts = list(range(0, len(c.x), step)) + [len(c.x)]
    check_case(c, run_gpu(c.x, c.params, pushes=cuts))


@pytest.mark.parametrize('name', PATH_CASES)
def test_golden_sequential_kernel(name):
    c = Case(name)
    check_case(c, run_gpu(c.x, c.params, flags=api.NFC_FLAG_FORCE_SEQUENTIAL))


@pytest.mark.parametrize('name', PATH_CASES)
def test_golden_general_kernel(monkeypatch, name):
    # The fixtures are float32 ENVELOPES (what transition_sink.work receives, transition_sink.py:13-18): since round 4 that input
    # kind takes the workgroup / lean kernels like the others (tests above); NFC_LEAN=0 keeps round 1's general kernel reachable
    # as the A/B -- and it is still what re-runs a chunk the fast kernels give up on.
    monkeypatch.setenv('NFC_LEAN', '0')
    c = Case(name)
    r = run_gpu(c.x, c.params)
    check_case(c, r)
    check_case(c, run_gpu(c.x, c.params, chunk_samples=256))


@pytest.mark.parametrize('seed', range(4))
def test_envelope_input_with_negative_samples(seed):
    # A raw envelope is the caller's: nothing says it is >= 0.  The fast kernels keep "slot not written by this chunk" in a ring
    # value's sign bit and order samples by their raw bits, so a chunk that meets a negative sample -- in the window it starts
    # from (the fill phase stores whatever comes) or among its own samples -- gives up and the general kernel takes it.  The
    # reference classifies a negative sample LOW (lo > ratio, transition_sink.py:62-66).
    rng = np.random.default_rng(900 + seed)
    n = 300_000
    x = (0.3 * (1 + 0.01 * rng.standard_normal(n))).astype(np.float32)
    for _ in range(60):
        s = int(rng.integers(2100, n - 300))
        k = int(rng.integers(1, 120))
        x[s:s + k] *= np.float32(rng.choice([0.0, 0.05, 1.12, 1.5]))
    neg = rng.integers(0, n, 40)
    x[neg] = -np.abs(x[neg]) * np.float32(rng.choice([1.0, 1e-3, 50.0]))
    if seed % 2:
        x[rng.integers(0, 2000, 5)] = np.float32(-0.25)   # ... in the fill phase too: the window then holds negative values
    x[4096 * 7 - 1] = np.float32(-1.0)                    # ... and on a seam of time chunks
    params = dict(hi_val=1.09)
    r = check_vs_oracle(x, params)
    assert r['stats'].n_chunks > 8
    check_vs_oracle(x, params, chunk_samples=256)
    check_vs_oracle(x, params, pushes=[0, 1500, 2000, 2001, 50_000, 123_457, n])


def test_golden_iq_input():
    iq = load_npz('fx_ultralight_iq.npz')['iq']
    c = Case('fx_ultralight_txn')
    check_case(c, run_gpu(iq, c.params, kind=api.NFC_IN_IQ_F32))
    check_case(c, run_gpu(iq, c.params, kind=api.NFC_IN_IQ_F32, chunk_samples=256,
                          pushes=[0, 1001, 2000, 2001, 7001, len(iq) // 2]))


@pytest.mark.parametrize('case', ['inf', 'nan'])
def test_nonfinite_iq_input(case):
    # NaN / Inf samples arriving as IQ (transition_sink.py:58-77 on their envelope; the reference's outputs are the fixture's)
    iq = load_npz('fx_nonfinite_iq.npz')[case + '_iq']
    c = Case('fx_nonfinite_iq:' + case, prefix=case + '_', file='fx_nonfinite_iq')
    check_case(c, run_gpu(iq, c.params, kind=api.NFC_IN_IQ_F32))
    check_case(c, run_gpu(iq, c.params, kind=api.NFC_IN_IQ_F32, chunk_samples=256,
                          pushes=[0, 1001, 2000, 2001, 7001, len(iq) // 4, len(iq) // 2]))


def oracle_run(x, params, kind):
    o = co.COracle(trace=True, **params)
    if kind == api.NFC_IN_I16_SQ:   # (16-bit PCM: sample / 32767 as a float32, then squared -- the WAV branch)
        o.push_real_sq((np.asarray(x).astype(np.float32) / np.float32(32767.0)).astype(np.float32))
        return o
    {api.NFC_IN_ENV_F32: o.push_env, api.NFC_IN_IQ_F32: o.push_iq, api.NFC_IN_REAL_F32_SQ: o.push_real_sq}[kind](x)
    return o


def check_vs_oracle(x, params, kind=api.NFC_IN_ENV_F32, **kw):
    o = oracle_run(x, params, kind)
    r = run_gpu(x, params, kind=kind, **kw)
    L = params.get('av_window', 2000)
    d = first_diff(r['val'][L:], o.trace().tolist())
    assert d is None, 'val %s' % (d,)
    e = o.edges()
    d = first_diff(r['transitions'], o.transitions())
    assert d is None, 'transition %s (of %d)' % (d, len(e))
    assert r['sym_tag'] == o.symbols(0).tolist()
    assert r['sym_reader'] == o.symbols(1).tolist()
    assert r['packets'] == o.packets()
    return r


def test_real_squared_input():
    c = Case('fx_ultralight_txn')
    s = np.sqrt(c.x.astype(np.float64)).astype(np.float32)
    check_vs_oracle(s, c.params, kind=api.NFC_IN_REAL_F32_SQ)


def test_low_run_ending_on_a_timeout():
    # a LOW run of exactly max_len+1 (and 2*max_len+1) samples resets the state at its last sample
    # (transition_sink.py:95-99); a HIGH sample right after it must be classified HIGH, not ignored
    rng = np.random.default_rng(3)
    x = (0.25 * (1 + 0.004 * rng.standard_normal(12000))).astype(np.float32)
    for start, ln in ((3000, 51), (4000, 101), (5000, 50), (6000, 52), (7000, 151), (8000, 1), (9000, 500)):
        x[start:start + ln] = 1e-6
        x[start + ln:start + ln + 3] = 0.25 * 1.3
    for kw in (dict(), dict(chunk_samples=256), dict(pushes=[0, 3050, 3051, 4101, 7151, 12000])):
        check_vs_oracle(x, dict(hi_val=1.1), **kw)


@pytest.mark.parametrize('seed', range(6))
def test_random_streams(seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(6000, 60000))
    base = rng.uniform(0.05, 2.0)
    x = (base * (1 + 0.01 * rng.standard_normal(n))).astype(np.float32)
    for _ in range(int(rng.integers(5, 200))):
        s = int(rng.integers(0, n - 200))
        k = int(rng.integers(1, 200))
        x[s:s + k] *= np.float32(rng.choice([0.0, 0.05, 0.09, 0.1, 0.11, 0.5, 1.09, 1.1, 1.105, 1.12, 1.5, 3.0]))
    x = np.abs(x)
    params = dict(samp_rate=float(rng.choice([2e6, 4e6, 1e7])), hi_val=float(rng.choice([1.05, 1.09, 1.1])),
                  av_window=int(rng.choice([128, 777, 2000])), max_len=int(rng.choice([7, 30, 50])))
    check_vs_oracle(x, params)
    check_vs_oracle(x, params, chunk_samples=256)
    cuts = sorted(set([0, n] + rng.integers(0, n, 5).tolist()))
    check_vs_oracle(x, params, pushes=cuts)


@pytest.mark.parametrize('name,kw', [('miller', dict(tag=False)), ('manchester', dict(reader=False)), ('all', dict())])
def test_synthetic_workloads_2M(name, kw):
    # BASELINE.json configs 2-4 at a size the oracle finishes in a blink; 122 time chunks
    iq = synth.workload(name, 2_000_000)
    r = check_vs_oracle(iq, dict(hi_val=1.1, **kw), kind=api.NFC_IN_IQ_F32)
    st = r['stats']
    assert st.used_sequential == 0
    # the speculative pass must be certified for (nearly) every time chunk: a full second pass means
    # the speculation or the margins are broken, even though the result would still be exact
    assert st.chunks_rerun <= st.n_chunks // 4, 'speculation not certified: %d passes, %d of %d chunks rerun' % (
        st.threshold_passes, st.chunks_rerun, st.n_chunks)
    print('%s: %d passes, %d/%d chunks rerun' % (name, st.threshold_passes, st.chunks_rerun, st.n_chunks))
    assert len(r['packets']) > 100


def _run_submitted(ctx, iq, cuts, depth=2):
    """The stream in batches, each submitted before the batches before it are waited for (nfc_submit_device / nfc_wait):
    `depth` batches in flight."""
    bufs = [api.DeviceBuffer(iq[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
    lens = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
    tr, s0, s1, pk, ahead = [], [], [], [], 0
    nxt = 0
    for k in range(len(bufs)):
        while nxt < len(bufs) and nxt < k + depth:
            ctx.submit_device(bufs[nxt], lens[nxt])
            nxt += 1
        assert ctx.submitted() == nxt - k
        ctx.wait()
        tr += ctx.transitions()
        s0 += ctx.symbols(0).tolist()
        s1 += ctx.symbols(1).tolist()
        pk += ctx.packets()
        ahead += int(ctx.stats().ran_ahead)
    assert ctx.submitted() == 0
    return tr, s0, s1, pk, ahead


@pytest.mark.parametrize('hook,depth', [('', 2), ('', 3), ('redo', 2), ('redo', 3)])
def test_batches_submitted_ahead(monkeypatch, hook, depth):
    # batch k + 1's threshold stage runs beside batch k's edge / decode stages, starting from the LOW bookkeeping, window and
    # sums batch k's own threshold stage left on the device; the result must be the single stream's.  With the test hook every
    # third submitted batch is declared irregular in nfc_wait and goes through the synchronous path again (and the batch behind
    # it is enqueued again from what that leaves).
    if hook:
        monkeypatch.setenv('NFC_DEBUG_REDO_SUBMITTED', '1')
    iq = synth.workload('all', 3_000_000)
    n = len(iq) // 2
    o = oracle_run(iq, dict(hi_val=1.1), api.NFC_IN_IQ_F32)
    cuts = [0, 300_000, 700_000, 1_000_003, 1_400_000, 1_800_001, 2_100_000, 2_400_000, 2_700_000, n]
    # (the hook only exists in the test build of the library, -DNFC_TEST_HOOKS)
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32, lib_path=_lib.hooks_path() if hook else None) as ctx:
        tr, s0, s1, pk, ahead = _run_submitted(ctx, iq, cuts, depth)
        extra = api.DeviceBuffer(iq[:600_000])   # (kept alive: a batch that does not run ahead reads it inside nfc_wait)
        with pytest.raises(api.NfcError):   # nothing else touches the stream while batches are in flight
            ctx.submit_device(extra, 300_000)
            ctx.reset()
        ctx.wait()
        st = ctx.stats()
    d = first_diff(tr, o.transitions())
    assert d is None, 'transition %s' % (d,)
    assert s0 == o.symbols(0).tolist() and s1 == o.symbols(1).tolist()
    assert pk == o.packets()
    # the first batch fills the window and the second follows it synchronously; from then on the stages run ahead
    assert ahead >= (3 if hook else 6), ahead
    assert (st.redone_total > 0) == bool(hook)


def test_batches_submitted_ahead_low_runs_across_batches():
    # LOW runs, time-outs and HIGH samples around the batch boundaries: the "HIGH ignored" bookkeeping the next batch starts
    # from is the one the previous batch's kernels left on the device.  At 800 000 (a synchronous batch hands over to one that
    # runs ahead) and 1 200 000 (one that ran ahead hands over to the next) a LOW run ends shortly before the boundary and
    # HIGH samples straddle it: they are ignored while the LOW sample is within max_len + 1 samples and did not time out.
    # At 1 600 000 the runs go THROUGH the boundary (the lean kernel hands such a first chunk to the general kernel: the
    # batch is then processed again synchronously -- exact either way).
    rng = np.random.default_rng(77)
    n = 5 * 400_000
    x = (0.25 * (1 + 0.003 * rng.standard_normal(n))).astype(np.float32)
    hi = np.float32(0.25 * 1.3)
    x[800_000 - 30:800_000 - 10] = 1e-6
    x[800_000 - 5:800_000 + 25] = hi              # ignored up to 800 000 - 11 + 51, classified after
    x[800_000 + 60:800_000 + 64] = hi
    x[1_200_000 - 52:1_200_000 - 2] = 1e-6        # exactly max_len long: its last sample times out, the state is reset
    x[1_200_000 - 1:1_200_000 + 3] = hi           # ... so these classify
    x[1_200_000 + 100:1_200_000 + 130] = 1e-6
    x[1_600_000 - 200:1_600_000 + 300] = 1e-6     # through the boundary, several time-outs long
    x[1_600_000 + 300:1_600_000 + 303] = hi
    iq = np.zeros(2 * n, np.float32)
    iq[0::2] = np.sqrt(x)
    o = oracle_run(iq, dict(hi_val=1.1), api.NFC_IN_IQ_F32)
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
        tr, s0, s1, pk, ahead = _run_submitted(ctx, iq, [0, 400_000, 800_000, 1_200_000, 1_600_000, n])
    d = first_diff(tr, o.transitions())
    assert d is None, 'transition %s' % (d,)
    assert s0 == o.symbols(0).tolist() and s1 == o.symbols(1).tolist() and pk == o.packets()
    # (the lean kernel gives such first and last chunks up -- LOW and HIGH samples in one step at the allowance's first guess --
    # so these batches end up processed again: what is tested here is that this is noticed and exact)


@pytest.mark.parametrize('name,kw', [('miller', dict(tag=False)), ('manchester', dict(reader=False))])
def test_batches_submitted_ahead_cut_inside_frames(name, kw):
    # back-to-back frames: every cut falls inside one, i.e. within max_len samples of a pause (a live LOW key) or of a loaded
    # half bit; the batches that run ahead start from the LOW bookkeeping the batch before left on the device
    iq = synth.workload(name, 3_200_000)
    n = len(iq) // 2
    rng = np.random.default_rng(5)
    cuts = [0, 300_000] + sorted((300_000 + 290_000 * (k + 1) + int(rng.integers(0, 20_000))) for k in range(8)) + [n]
    o = oracle_run(iq, dict(hi_val=1.1, **kw), api.NFC_IN_IQ_F32)
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32, **kw) as ctx:
        tr, s0, s1, pk, ahead = _run_submitted(ctx, iq, cuts, 3)
        st = ctx.stats()
    d = first_diff(tr, o.transitions())
    assert d is None, 'transition %s' % (d,)
    assert s0 == o.symbols(0).tolist() and s1 == o.symbols(1).tolist() and pk == o.packets()
    assert ahead >= 7 and st.redone_total == 0, (ahead, st.redone_total)


def test_batches_submitted_ahead_mixed_with_batches_that_do_not_qualify():
    # long batches (run ahead), short ones (<= 2^18 samples: the one-launch stage, processed inside nfc_wait), a tiny one and
    # an empty one, three in flight: whatever mixture, the stream is the single stream's
    iq = synth.workload('all', 3_000_000)
    n = len(iq) // 2
    o = oracle_run(iq, dict(hi_val=1.1), api.NFC_IN_IQ_F32)
    sizes = [300_000, 100_000, 500_000, 50_000, 400_000, 400_000, 10, 0, 400_000, 300_017, 7, 280_000]
    cuts = [0]
    for z in sizes:
        cuts.append(cuts[-1] + z)
    cuts.append(n)
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
        bufs = [api.DeviceBuffer(iq[2 * a:2 * b] if b > a else np.zeros(4, np.float32)) for a, b in zip(cuts[:-1], cuts[1:])]
        lens = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
        tr, s0, s1, pk, flags = [], [], [], [], []
        nxt = 0
        for k in range(len(bufs)):
            while nxt < len(bufs) and nxt < k + 3:
                ctx.submit_device(bufs[nxt], lens[nxt])
                nxt += 1
            ctx.wait()
            tr += ctx.transitions()
            s0 += ctx.symbols(0).tolist()
            s1 += ctx.symbols(1).tolist()
            pk += ctx.packets()
            flags.append(int(ctx.stats().ran_ahead))
    d = first_diff(tr, o.transitions())
    assert d is None, 'transition %s' % (d,)
    assert s0 == o.symbols(0).tolist() and s1 == o.symbols(1).tolist() and pk == o.packets()
    assert sum(flags) >= 3 and flags[1] == 0 and flags[3] == 0, flags   # the short ones never run ahead


def test_submit_and_wait_where_nothing_can_run_ahead():
    # an envelope input (the general threshold kernel: no device-side hand-over) through the same two calls
    c = Case('fx_ultralight_txn')
    x = np.tile(c.x, 12)   # 380 000 samples
    o = oracle_run(x, c.params, api.NFC_IN_ENV_F32)
    with api.NfcContext(input_kind=api.NFC_IN_ENV_F32, **c.params) as ctx:
        cut = 300_000
        a, b = api.DeviceBuffer(x[:cut]), api.DeviceBuffer(x[cut:])
        ctx.submit_device(a, cut)
        ctx.submit_device(b, len(x) - cut)
        tr, pk = [], []
        for _ in range(2):
            ctx.wait()
            tr += ctx.transitions()
            pk += ctx.packets()
            assert ctx.stats().ran_ahead == 0
    assert first_diff(tr, o.transitions()) is None and pk == o.packets()


def test_state_set_from_the_host_is_not_run_ahead_of():
    # After nfc_set_state (here: a state taken mid-stream and put back) the device-side LOW bookkeeping is not the stream's:
    # the next submitted batch must take the synchronous path, the ones after it may run ahead again
    iq = synth.workload('miller', 1_600_000)
    o = oracle_run(iq, dict(hi_val=1.1, tag=False), api.NFC_IN_IQ_F32)
    cuts = [0, 400_000, 800_000, 1_200_000, 1_600_000]
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32, tag=False) as ctx:
        bufs = [api.DeviceBuffer(iq[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
        tr, pk, flags = [], [], []
        for k, b in enumerate(bufs):
            if k == 2:
                blob = ctx.state_blob()
                ctx.reset()
                ctx.set_state_blob(blob)
                ctx.state_blob()              # (flushes the pending state to the device: nothing "dirty" is left to notice)
            ctx.submit_device(b, cuts[k + 1] - cuts[k])
            ctx.wait()
            tr += ctx.transitions()
            pk += ctx.packets()
            flags.append(int(ctx.stats().ran_ahead))
    assert first_diff(tr, o.transitions()) is None and pk == o.packets()
    assert flags == [0, 0, 0, 1], flags


@pytest.mark.parametrize('runin', ['512', '2048'])
def test_frames_longer_than_the_run_in_take_the_fall_back(monkeypatch, runin):
    # The speculative decode (decode.hip.h: k_dec_spec) derives a tile's incoming decoder states from a run-in of the edges before
    # it: right wherever a frame gap lies within the run-in.  Frames of 250 bytes (4 500 edges, ISO 14443-4 allows 256-byte frames)
    # back to back leave tile seams with no gap in reach: the check (dec_verify) must say so, the stage must be repeated in the
    # three-launch form, and the result must be the reference's -- for that batch, the following ones (which take the three-launch
    # form straight away), and again once the stream is back to short frames.
    monkeypatch.setenv('NFC_NO_SMALL', '1')
    monkeypatch.setenv('NFC_DEC_RUNIN', runin)
    rng = np.random.default_rng(41)
    long_frames = [(synth.READER, synth.frame_bits(rng.integers(0, 256, 250).tolist(), 0)) for _ in range(6)]
    m_long = synth.modulation_profile(long_frames, rate_msps=2.0, gap_us=60.0, lead_in=3000, tail=400)
    short = synth.workload('miller', 600_000)
    iq_long = synth.iq_from_profile(m_long)
    iq = np.concatenate([iq_long, short[2 * 3000:]])   # one stream: the long frames, then ordinary traffic
    n_long = len(iq_long) // 2
    n = len(iq) // 2
    o = oracle_run(iq, dict(hi_val=1.1), api.NFC_IN_IQ_F32)
    cuts = [0, n_long] + list(range(n_long + 50_000, n, 50_000)) + [n]
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
        tr, s0, s1, pk, resp = [], [], [], [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            ctx.push(iq[2 * a:2 * b])
            tr += ctx.transitions()
            s0 += ctx.symbols(0).tolist()
            s1 += ctx.symbols(1).tolist()
            pk += ctx.packets()
            resp.append(int(ctx.stats().decode_respeculated))
    assert resp[0] == 1, resp            # the fall-back was taken for the batch with the long frames ...
    assert resp[-1] == 1, resp           # ... and only there: the batches behind it took the three-launch form, then speculated again
    assert first_diff(tr, o.transitions()) is None
    assert s0 == o.symbols(0).tolist() and s1 == o.symbols(1).tolist()
    assert pk == o.packets() and sum(len(b) > 2000 for _, b in pk) == 6


def _long_frame_stream(n_frames, seed=43):
    """Ordinary reader traffic, then `n_frames` 250-byte reader frames 60 us apart (no gap within a 512-edge run-in at the tile seams
    inside them), then ordinary traffic again; returns (iq, first sample of the long frames, first sample behind them)."""
    rng = np.random.default_rng(seed)
    long_frames = [(synth.READER, synth.frame_bits(rng.integers(0, 256, 250).tolist(), 0)) for _ in range(n_frames)]
    m_long = synth.modulation_profile(long_frames, rate_msps=2.0, gap_us=60.0, lead_in=3000, tail=400)
    head = synth.workload('miller', 700_000)
    tail = synth.workload('miller', 900_000)
    iq_long = synth.iq_from_profile(m_long)
    iq = np.concatenate([head, iq_long, tail[2 * 3000:]])
    return iq, len(head) // 2, (len(head) + len(iq_long)) // 2


def test_long_frames_in_batches_submitted_ahead_repeat_the_decode_stage_only():
    # A batch that ran ahead and whose speculative decode fails its check (a frame longer than the run-in across a tile seam) keeps its
    # threshold and edge stages: nfc_wait repeats the decode stage alone, in the three-launch form -- the batch is NOT processed again
    # (redone_total stays 0) and the batches submitted behind it are not restarted; the batches that follow take the three-launch
    # form straight away, and speculation comes back once the traffic is ordinary again.
    iq, n0, n1 = _long_frame_stream(40)
    n = len(iq) // 2
    o = oracle_run(iq, dict(hi_val=1.1), api.NFC_IN_IQ_F32)
    step = 330_000
    cuts = [0, n0] + list(range(n0 + step, n, step)) + [n]
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
        tr, s0, s1, pk, ahead = _run_submitted(ctx, iq, cuts, 3)
        st = ctx.stats()
    d = first_diff(tr, o.transitions())
    assert d is None, 'transition %s' % (d,)
    assert s0 == o.symbols(0).tolist() and s1 == o.symbols(1).tolist()
    assert pk == o.packets() and sum(len(b) > 2000 for _, b in pk) == 40
    assert st.decode_respeculated >= 1, st.decode_respeculated     # the check failed at least once ...
    assert st.redone_total == 0, st.redone_total                   # ... and no batch went through the synchronous path again for it
    # (the first batch fills the window and the two submitted with it follow it synchronously; every one behind them ran ahead)
    assert ahead >= len(cuts) - 1 - 3, (ahead, len(cuts))


def test_long_frames_through_push_edges_take_the_fall_back():
    # the same frames as a caller's own transition list (nfc_push_edges: decode and framing alone): the check fails, the stage is
    # repeated in the three-launch form, symbols and packets are the reference's
    iq, n0, n1 = _long_frame_stream(8)
    o = oracle_run(iq, dict(hi_val=1.1), api.NFC_IN_IQ_F32)
    want_tr = o.transitions()
    e = np.zeros(len(want_tr), api.EDGE_DTYPE)
    e['v'] = [v for (v, _), _ in want_tr]
    e['d'] = [int(round(us / 0.5)) for (_, us), _ in want_tr]
    e['t'] = [t for _, t in want_tr]
    e['idx'] = np.arange(len(want_tr), dtype=np.uint64)
    with api.NfcContext(hi_val=1.1, samp_rate=2e6, max_len=50) as ctx:
        s0, s1, pk, resp = [], [], [], []
        third = len(e) // 3
        for a, b in ((0, third), (third, 2 * third), (2 * third, len(e))):
            ctx.push_edges(e[a:b])
            s0 += ctx.symbols(0).tolist()
            s1 += ctx.symbols(1).tolist()
            pk += [bits for _, bits in ctx.packets()]
            resp.append(int(ctx.stats().decode_respeculated))
    assert resp[-1] >= 1, resp
    assert s0 == o.symbols(0).tolist() and s1 == o.symbols(1).tolist()
    assert pk == [bits for _, bits in o.packets()] and sum(len(b) > 2000 for b in pk) == 8


def test_stats_after_wait_belong_to_that_batch():
    # nfc_stats.ring_slots_carried is documented as "of the last batch": for a batch submitted ahead it must come from that batch's
    # own snapshot -- by the time nfc_wait returns, the device's summary has been rewritten by the batch submitted behind it
    iq = synth.workload('miller', 900_000)
    n = len(iq) // 2
    iq = iq.copy()
    iq[2 * 300_000:2 * 600_000] *= np.float32(1e-3)   # the second batch: a loss of signal throughout -- no sample accepted, every slot carried
    cuts = [0, 300_000, 600_000, n]
    want = []
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
        for a, b in zip(cuts[:-1], cuts[1:]):
            ctx.push(iq[2 * a:2 * b])
            want.append(int(ctx.stats().ring_slots_carried))
    assert want[0] == 0 and want[1] == 2000 and want[2] == 0, want
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
        bufs = [api.DeviceBuffer(iq[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
        ctx.push_device(bufs[0], cuts[1])          # (the window fills here: batches can be submitted ahead from the next one on)
        got = [int(ctx.stats().ring_slots_carried)]
        ctx.submit_device(bufs[1], cuts[2] - cuts[1])
        ctx.submit_device(bufs[2], cuts[3] - cuts[2])
        for _ in range(2):
            ctx.wait()
            got.append(int(ctx.stats().ring_slots_carried))
    assert got == want, (got, want)


def test_batches_submitted_ahead_int16_and_misuse():
    # 16-bit PCM (the WAV branch: scaled and squared on the fly) through batches submitted ahead; and what the calls refuse
    iq = synth.workload('all', 2_400_000)
    env = np.sqrt(synth.envelope_f32(iq))
    pcm = np.clip(np.round(env / env.max() * 30000.0), -32768, 32767).astype(np.int16)
    params = dict(hi_val=1.09)
    o = co.COracle(trace=False, **params)
    o.push_i16(pcm) if hasattr(o, 'push_i16') else o.push_real_sq((pcm.astype(np.float32) / np.float32(32767.0)).astype(np.float32))
    cuts = [0, 400_000, 800_000, 1_200_000, 1_600_000, 2_000_000, len(pcm)]
    with api.NfcContext(input_kind=api.NFC_IN_I16_SQ, **params) as ctx:
        bufs = [api.DeviceBuffer(pcm[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        with pytest.raises(api.NfcError):
            ctx.wait()                                  # nothing submitted
        tr, pk, ahead = [], [], 0
        for k, b in enumerate(bufs):
            ctx.submit_device(b, cuts[k + 1] - cuts[k])
            if k == 2:
                with pytest.raises(api.NfcError):
                    ctx.push_device(b, 10)              # a synchronous push beside batches in flight
                with pytest.raises(api.NfcError):
                    ctx.state_blob()
            ctx.wait()
            tr += ctx.transitions()
            pk += ctx.packets()
            ahead += int(ctx.stats().ran_ahead)
        # one at a time nothing runs beside anything, but the path is the same: threshold stage from the device-side state
        assert ahead >= 3
    assert first_diff(tr, o.transitions()) is None and pk == o.packets()


def test_compact_transitions_are_the_records():
    # nfc_read_edges_compact hands out what the device keeps (position, code); nfc_read_edges the records built from it
    import ctypes as C
    c = Case('fx_ultralight_txn')
    with api.NfcContext(input_kind=api.NFC_IN_ENV_F32, **c.params) as ctx:
        half = len(c.x) // 2
        for a, b in ((0, half), (half, len(c.x))):
            ctx.push(c.x[a:b])
            e = ctx.edges()
            pos, code = ctx.edges_compact()
            nd = c.params.get('max_len', 50) + 1
            li = (code & 0x3FFF).astype(np.int64)
            assert (e['idx'] == a + pos.astype(np.uint64)).all()
            assert (e['v'] == li // nd - 1).all() and (e['d'] == li % nd).all() and (e['t'] == (code >> 14).astype(np.int64) - 1).all()
            # a range in the middle, through both readers
            if len(e) > 12:
                sub = np.zeros(7, api.EDGE_DTYPE)
                got = C.c_size_t(0)
                assert ctx.L.nfc_read_edges(ctx.h, 5, sub.ctypes.data, 7, C.byref(got)) == 0 and got.value == 7
                assert (sub == e[5:12]).all()
                p7, c7 = np.zeros(7, np.uint32), np.zeros(7, np.uint16)
                assert ctx.L.nfc_read_edges_compact(ctx.h, 5, p7.ctypes.data, c7.ctypes.data, 7, C.byref(got)) == 0 and got.value == 7
                assert (p7 == pos[5:12]).all() and (c7 == code[5:12]).all()


def test_readers_into_pinned_arrays_and_bit_ranges():
    # round 6: nfc_read_edges_compact copies straight into the caller's arrays when they are pinned (api.PinnedArray: no staging, no
    # second pass on the host) -- the same entries as through the staged path, whole and from the middle; nfc_read_packet_bits unpacks
    # the packed words eight bits at a time -- any (first, count) range equals the slice of the whole array
    import ctypes as C
    iq = synth.workload('all', 1_500_000)
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
        ctx.push(iq)
        pos, code = ctx.edges_compact()
        n = len(pos)
        assert n > 50_000
        pp, pc = api.PinnedArray(n + 100, np.uint32), api.PinnedArray(n + 100, np.uint16)
        pp.array[:] = 0xFFFFFFFF
        pc.array[:] = 0xFFFF
        p2, c2 = ctx.edges_compact(out=(pp.array, pc.array))
        assert len(p2) == n and (p2 == pos).all() and (c2 == code).all()
        assert (pp.array[n:] == 0xFFFFFFFF).all() and (pc.array[n:] == 0xFFFF).all()   # (nothing written past the batch's entries)
        got = C.c_size_t(0)
        assert ctx.L.nfc_read_edges_compact(ctx.h, 12_345, pp.array.ctypes.data, pc.array.ctypes.data, 1000, C.byref(got)) == 0 and got.value == 1000
        assert (pp.array[:1000] == pos[12_345:13_345]).all() and (pc.array[:1000] == code[12_345:13_345]).all()
        with pytest.raises(api.NfcError):
            ctx.edges_compact(out=(pp.array[:10], pc.array[:10]))
        pp.free()
        pc.free()
        rng = np.random.default_rng(5)
        for t in (0, 1):
            bits = ctx.packet_bits(t)
            assert len(bits) > 10_000 and set(np.unique(bits).tolist()) <= {0, 1}
            for _ in range(40):
                first = int(rng.integers(0, len(bits) - 1))
                cnt = int(rng.integers(1, min(3000, len(bits) - first) + 1))
                out = np.full(cnt + 8, 7, np.uint8)
                assert ctx.L.nfc_read_packet_bits(ctx.h, t, first, out.ctypes.data, cnt, C.byref(got)) == 0 and got.value == cnt
                assert (out[:cnt] == bits[first:first + cnt]).all() and (out[cnt:] == 7).all(), (t, first, cnt)
        assert ctx.packets() == oracle_run(iq, dict(hi_val=1.1), api.NFC_IN_IQ_F32).packets()


@pytest.mark.parametrize('dec_spec', ['1', '0'])
@pytest.mark.parametrize('max_len', [1, 7, 31, 32, 50, 62, 63, 64, 200])
def test_edge_stage_dense_and_long_runs(monkeypatch, max_len, dec_spec):
    # The multi-launch edge stage (edges.hip.h) on a short batch: samples that flicker between LOW, accepted and HIGH from one
    # sample to the next (more entries per tile than the writer stages in one round), runs of every length around max_len
    # and its multiples (time-outs inside a word, across words, across tiles and across pushes), for max_len on both
    # sides of the 32 / 63 boundaries where the in-word time-out count changes form.
    monkeypatch.setenv('NFC_NO_SMALL', '1')
    monkeypatch.setenv('NFC_DEC_SPEC', dec_spec)   # both forms of the decode stage behind it: speculative (k_dec_spec) / three launches
    rng = np.random.default_rng(7000 + max_len)
    n = 150_000
    x = (0.25 * (1 + 0.002 * rng.standard_normal(n))).astype(np.float32)
    lv = np.array([0.01, 0.25, 0.25 * 1.3], np.float32)
    for s in (3000, 40_000, 90_000):   # dense stretches: 35 000 samples of flicker
        k = 35_000 if s == 40_000 else 2500
        x[s:s + k] = lv[rng.integers(0, 3, k)]
    pos = 8000
    for ln in [max_len - 1, max_len, max_len + 1, 2 * max_len, 2 * max_len + 1, 3 * max_len + 2, 70, 5 * max_len + 3]:
        for level in (0.01, 0.25 * 1.3):
            if ln > 0:
                x[pos:pos + ln] = level
            pos += ln + int(rng.integers(1, 40))
    x[130_000:131_000] = 0.01   # a long LOW run near the end
    params = dict(hi_val=1.1, av_window=500, max_len=max_len)
    check_vs_oracle(x, params)
    cuts = sorted(set([0, n, 40_000 + 64 * 7 + 13, 57_344, 57_345, 130_500] + rng.integers(600, n, 4).tolist()))
    check_vs_oracle(x, params, pushes=cuts)


@pytest.mark.parametrize('own_prefix_max', ['0', '3'])
def test_prefix_launch_path(monkeypatch, own_prefix_max):
    # Few tiles: every tile's workgroup folds its predecessors' aggregates itself; long batches keep the single-workgroup
    # prefix launches.  The library reads the switch-over at nfc_create: force the long-batch form on a short batch
    # (and a mixed case: the decode stage's tiles exceed the limit, later pushes of 300k samples do not).
    monkeypatch.setenv('NFC_OWN_PREFIX_MAX', own_prefix_max)
    iq = synth.workload('all', 2_300_000)
    check_vs_oracle(iq, dict(hi_val=1.1), kind=api.NFC_IN_IQ_F32, pushes=[0, 1_400_000, 1_700_000, 2_300_000])


@pytest.mark.parametrize('window,max_len,rate', [(2000, 50, 2.0), (4000, 100, 4.0), (10000, 250, 10.0)])
def test_ring_in_global_memory(monkeypatch, window, max_len, rate):
    # Long windows keep a chunk's ring in global memory when the batch is long enough to fill the machine that way
    # (host_threshold.h: threshold_span); NFC_RING=global forces that kernel on batches of test size, for every window
    monkeypatch.setenv('NFC_RING', 'global')
    period = synth.modulation_profile(synth.txn_frames(), rate_msps=rate, lead_in=0, tail=0)
    iq = synth.iq_from_profile(synth.tiled_profile(period, 1_300_000, lead_in=window + 700), seed=11)
    r = check_vs_oracle(iq, dict(samp_rate=rate * 1e6, hi_val=1.1, av_window=window, max_len=max_len), kind=api.NFC_IN_IQ_F32,
                        pushes=[0, 400_001, 1_300_000])
    assert r['stats'].used_sequential == 0


@pytest.mark.parametrize('n', [2_000_001, 2_000_063, 2_000_064, 2_000_191, 1_999_999])
def test_ragged_batch_end(n):
    # a batch that does not end on a 256-sample step: its last step runs with the lanes past the end masked,
    # on the certified fast path -- no chunk may need a second pass for it
    iq = synth.workload('all', n)
    r = check_vs_oracle(iq, dict(hi_val=1.1), kind=api.NFC_IN_IQ_F32)
    st = r['stats']
    assert st.used_sequential == 0
    assert st.chunks_rerun == 0 and st.threshold_passes == 1, (st.threshold_passes, st.chunks_rerun)


def test_stream_that_starts_inside_a_transaction():
    # No idle lead-in: the fill phase stores pause-level samples, and while they sit in the window the reference's own
    # running sum rounds (its order matters).  The sequential kernel replays a prefix of a few windows, the rest of the
    # batch runs on the parallel path -- exact, and not at the one-lane rate
    period = synth.modulation_profile(synth.txn_frames(), rate_msps=2.0, lead_in=0, tail=0)
    iq = synth.iq_from_profile(synth.tiled_profile(period, 1_500_000, lead_in=0), seed=3)
    r = check_vs_oracle(iq, dict(hi_val=1.1), kind=api.NFC_IN_IQ_F32)
    st = r['stats']
    assert st.used_sequential == 1            # the prefix
    assert st.threshold_passes >= 2 and st.n_chunks > 100   # ... and a certified parallel attempt over the rest


@pytest.mark.parametrize('seed', range(8))
def test_random_parameters_many_chunks(seed):
    # wider sweep: windows up to 12 000 samples, max_len up to 300 (LUT rows beyond the LDS-staged limit), batches of
    # several hundred time chunks, streams that start inside modulation, pushes of very different lengths
    rng = np.random.default_rng(900 + seed)
    L = int(rng.choice([256, 500, 2000, 4096, 10000, 12000]))
    mx = int(rng.choice([1, 2, 13, 50, 64, 127, 128, 250, 300]))
    rate = float(rng.choice([2.0, 4.0, 10.0]))
    n = int(rng.integers(3 * L + 1000, 900_000))
    frames = synth.txn_frames()
    period = synth.modulation_profile(frames, rate_msps=rate, lead_in=0, tail=0, depth=float(rng.choice([0.05, 0.08, 0.2])))
    m = synth.tiled_profile(period, n, lead_in=int(rng.choice([0, L // 2, L + 500])))
    iq = synth.iq_from_profile(m, seed=int(rng.integers(1, 1 << 30)), sigma=float(rng.choice([0.0005, 0.002, 0.01])))
    params = dict(samp_rate=rate * 1e6, hi_val=float(rng.choice([1.05, 1.1])), av_window=L, max_len=mx)
    check_vs_oracle(iq, params, kind=api.NFC_IN_IQ_F32)
    cuts = sorted(set([0, n] + rng.integers(0, n, 4).tolist() + [int(rng.integers(0, min(n, 3000)))]))
    check_vs_oracle(iq, params, kind=api.NFC_IN_IQ_F32, pushes=cuts)


@pytest.mark.gpu
@pytest.mark.parametrize('dec_spec', ['1', '0'])
@pytest.mark.parametrize('tagname,rate', [('1', 1e6), ('0p5', 2e6), ('0p25', 4e6), ('0p1', 10e6), ('frames_0p5', 2e6), ('frames_0p25', 4e6)])
def test_decode_kernels_on_decoder_vectors(monkeypatch, tagname, rate, dec_spec):
    # the reference's decoder-only vectors (6 000 random (cur, d) pairs per rate with every error branch, and whole frames)
    # straight into k_dec_reduce / k_dec_apply / k_frame_write through nfc_push_edges -- interleaved runs of both routes, in
    # several calls (decoder and framing state carried on the device); symbols against the reference's, packets against its
    # PacketProcessor restated (oracle)
    from oracle import py_oracle as po
    from usrp_nfc_amd import api
    from tests.golden_util import load_npz
    monkeypatch.setenv('NFC_DEC_SPEC', dec_spec)   # k_dec_spec (every tile's states from a run-in, checked) / k_dec_reduce + k_dec_apply
    z = load_npz('fx_decoder_vectors.npz')
    dm = z['dm_' + tagname] if 'dm_' + tagname in z else z['d_' + tagname]
    dt = z['dt_' + tagname] if 'dt_' + tagname in z else z['d_' + tagname]
    cm, ct = z['curm_' + tagname], z['curt_' + tagname]
    rng = np.random.default_rng(5)
    rows, im, it, idx = [], 0, 0, 0
    while im < len(cm) or it < len(ct):   # runs of 1..40 entries, routes alternating, idle entries sprinkled in
        for (cur, d, route) in ((cm, dm, 1), (ct, dt, 0)):
            pos = im if route == 1 else it
            k = int(rng.integers(1, 41))
            for j in range(pos, min(pos + k, len(cur))):
                rows.append((idx, int(d[j]), int(cur[j]), route, 0))
                idx += 7
            if route == 1:
                im = min(pos + k, len(cur))
            else:
                it = min(pos + k, len(cur))
            if rng.random() < 0.3:
                rows.append((idx, 50, 0, -1, 0))   # an idle heartbeat: dropped by the router (background.py:30-35)
                idx += 7
    edges = np.array(rows, dtype=api.EDGE_DTYPE)
    sink = po.BitSink()
    mil, man = po.MillerDecoder(sink), po.ManchesterDecoder(sink)
    factor = 1e6 / rate
    for r in rows:
        if r[3] == 1:
            mil.process_transition([(r[2], r[1] * factor)])
        elif r[3] == 0:
            man.process_transition([(r[2], r[1] * factor)])
    with api.NfcContext(samp_rate=rate, max_len=50) as ctx:
        sym = {0: [], 1: []}
        packets = []
        cuts = [0, 1, 17, len(edges) // 3, len(edges) // 3 + 4097, len(edges)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            ctx.push_edges(edges[a:b])
            for t in (0, 1):
                sym[t] += ctx.symbols(t).tolist()
            packets += ctx.packets()
    assert sym[1] == z['symm_' + tagname].tolist()
    assert sym[0] == z['symt_' + tagname].tolist()
    assert sym[1] == sink.symbols[1] and sym[0] == sink.symbols[0]
    want = [(t, b) for t, b in sink.packets]
    # packets of the two routes close in stream order of their closing entries
    assert sorted(packets) == sorted(want) and len(want) >= (10 if tagname.startswith("frames") else 1)
    assert [p for p in packets if p[0] == 1] == [p for p in want if p[0] == 1]
    assert [p for p in packets if p[0] == 0] == [p for p in want if p[0] == 0]


@pytest.mark.gpu
def test_window_at_the_upper_bound():
    # av_window 30000 (the largest nfc_create accepts: ring in global memory), max_len 750, 10 Msps
    from oracle import c_oracle as co
    from usrp_nfc_amd import api, synth
    params = dict(samp_rate=10e6, hi_val=1.1, av_window=30000, max_len=750)
    frames = synth.txn_frames()
    m = synth.tiled_profile(synth.modulation_profile(frames, rate_msps=10.0, lead_in=0, tail=0), 1_500_000, lead_in=40000)
    iq = synth.iq_from_profile(m, seed=3)
    o = co.COracle(**params)
    o.push_iq(iq)
    with api.NfcContext(input_kind=api.NFC_IN_IQ_F32, **params) as ctx:
        ctx.push(iq[:2 * 700_001])
        tr = ctx.transitions()
        pk = ctx.packets()
        ctx.push(iq[2 * 700_001:])
        tr += ctx.transitions()
        pk += ctx.packets()
    assert tr == o.transitions() and pk == o.packets() and len(pk) > 10


@pytest.mark.gpu
def test_rejected_launch_is_reported(monkeypatch):
    # a launch the runtime refuses leaves no trace in the stream: without the check behind every launch (launch_check.h) the
    # push would return the previous batch's totals out of the host's mirror
    from usrp_nfc_amd import api, synth
    iq = synth.workload('miller', 300_000)
    with api.NfcContext(samp_rate=2e6, hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
        ctx.push(iq)
        n_good = len(ctx.edges())
    # (the switch is read once, when a context is created: an environment variable set later changes nothing)
    # ... and only by the test build of the library (-DNFC_TEST_HOOKS); the product library does not know the switch
    monkeypatch.setenv_plain('NFC_DEBUG_BAD_LAUNCH', '1')
    with api.NfcContext(samp_rate=2e6, hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
        ctx.push(iq)
        assert len(ctx.edges()) == n_good
    from usrp_nfc_amd import _lib
    with api.NfcContext(samp_rate=2e6, hi_val=1.1, input_kind=api.NFC_IN_IQ_F32, lib_path=_lib.hooks_path()) as ctx:
        monkeypatch.delenv('NFC_DEBUG_BAD_LAUNCH')
        with pytest.raises(api.NfcError) as e:
            ctx.push(iq)
        assert 'kernel launch failed' in str(e.value)
    assert n_good > 1000


def _wg_torture(seed, n=1_500_000):
    """IQ stream for the workgroup threshold kernel: carrier with noise, LOW runs of every length around the multiples of
    max_len (1 .. 700 samples), each followed -- at 0 .. 70 samples from its end -- by a burst of HIGH samples; now and then a
    level step of a few percent, a tag-like stretch of alternating loaded half bits, and runs placed right on the multiples of
    1024 / 4096 samples where rounds and chunks meet."""
    rng = np.random.default_rng(seed)
    m = np.ones(n, np.float32)
    p = 6000
    k = 0
    while p < n - 5000:
        k += 1
        kind = int(rng.integers(0, 10))
        if kind < 6:
            ln = int(rng.choice([1, 2, 7, 49, 50, 51, 52, 99, 100, 101, 102, 150, 151, 152, 201, 255, 256, 257, 400, 511, 513, 700])) if rng.random() < 0.7 else int(rng.integers(1, 700))
            if rng.random() < 0.3:   # put the run's end (or start) on a seam of rounds / chunks
                seam = (p // 1024 + 1) * 1024 if rng.random() < 0.5 else (p // 4096 + 1) * 4096
                p = seam - (ln if rng.random() < 0.5 else 0) + int(rng.integers(-2, 3))
            m[p:p + ln] = 0.0
            gap = int(rng.integers(0, 70))
            hl = int(rng.integers(1, 40))
            m[p + ln + gap:p + ln + gap + hl] = 1.09
            p += ln + gap + hl + int(rng.integers(60, 2500))
        elif kind < 8:   # a stretch of loaded half bits (mag^2 x 1.17)
            for b in range(int(rng.integers(20, 200))):
                if rng.random() < 0.5:
                    m[p:p + 9] = 1.08
                p += 19
            p += int(rng.integers(100, 3000))
        else:            # a level step (the window follows within one length)
            m[p:] *= np.float32(1.0 + rng.uniform(-0.04, 0.04))
            p += int(rng.integers(3000, 9000))
    return synth.iq_from_profile(m, seed=seed, sigma=0.0015)


@pytest.mark.parametrize('flags', ['0', '1'])
@pytest.mark.parametrize('nr', ['4', '8'])
@pytest.mark.parametrize('chunk', [0, 4096 * 3])
def test_workgroup_kernel_low_runs_time_outs_and_seams(monkeypatch, nr, chunk, flags):
    # k_threshold_wg against the C oracle where its reasoning is thinnest (threshold_wg.hip.h): LOW runs longer than max_len whose
    # last sample may have ended on a time-out with HIGH samples in reach of it, runs and HIGH bursts across the seams of steps,
    # rounds and chunks, level steps between supersteps; every step height, the library's chunk length and a short one (many
    # chunk seams: speculation, certification, chunks that start inside a run).  Per-sample val, edges, symbols, packets.
    # flags = 1 (test build, round 6): the form whose waves wait for each other's counters instead of at a round's first barrier.
    monkeypatch.setenv('NFC_WG_NR', nr)
    monkeypatch.setenv('NFC_WG_FLAGS', flags)
    L = {'4': 2000, '6': 2000, '8': 2560}[nr]
    iq = _wg_torture(int(nr) * 10 + (1 if chunk else 0))
    params = dict(hi_val=1.1, av_window=L)
    o = oracle_run(iq, params, api.NFC_IN_IQ_F32)
    r = run_gpu(iq, params, kind=api.NFC_IN_IQ_F32, chunk_samples=(chunk // (256 * int(nr))) * 256 * int(nr))
    d = first_diff(r['val'][L:], o.trace().tolist())
    assert d is None, 'val %s' % (d,)
    d = first_diff(r['transitions'], o.transitions())
    assert d is None, 'transition %s' % (d,)
    assert r['sym_tag'] == o.symbols(0).tolist() and r['sym_reader'] == o.symbols(1).tolist() and r['packets'] == o.packets()
    assert r['stats'].used_sequential == 0 and len(o.transitions()) > 30000


@pytest.mark.parametrize('flags', ['0', '1'])
@pytest.mark.parametrize('bulk', ['1', '0'])
@pytest.mark.parametrize('nr', ['4', '8'])
def test_workgroup_kernel_plane_staging(monkeypatch, bulk, nr, flags):
    # k_threshold_wg keeps the plane words of its regular rounds in LDS (threshold_wg.hip.h): a whole chunk's, stored when the chunk
    # is done (bulk), or a ring of 2 FR rounds stored FR at a time by one wave as whole lines (NFC_WG_BULK=0; batches submitted ahead
    # always).  Chunks of many rounds -- the ring wraps several times, its last flush is a partial one --, a ragged batch end, a
    # stream that becomes stable in the middle of a chunk (rounds that are not regular store directly, between two flushes), and
    # the same stream submitted ahead.  Per-sample val, edges, symbols, packets against the C oracle.
    # (flags = 1: the waves of the counter form may be a round apart -- the ring's flush waits a round longer)
    monkeypatch.setenv('NFC_WG_BULK', bulk)
    monkeypatch.setenv('NFC_WG_NR', nr)
    monkeypatch.setenv('NFC_WG_FLAGS', flags)
    L = {'4': 2000, '8': 2560}[nr]
    rnd = 256 * int(nr)
    iq = _wg_torture(500 + int(nr), 1_000_000 + 12_345)
    params = dict(hi_val=1.1, av_window=L)
    o = oracle_run(iq, params, api.NFC_IN_IQ_F32)
    for chunk in (61 * rnd, 23 * rnd):
        r = run_gpu(iq, params, kind=api.NFC_IN_IQ_F32, chunk_samples=chunk)
        assert r['stats'].chunk_samples == chunk and r['stats'].used_sequential == 0
        d = first_diff(r['val'][L:], o.trace().tolist())
        assert d is None, 'val %s (chunk %d)' % (d, chunk)
        assert first_diff(r['transitions'], o.transitions()) is None
        assert r['sym_tag'] == o.symbols(0).tolist() and r['sym_reader'] == o.symbols(1).tolist() and r['packets'] == o.packets()
    # ... pushed in pieces (the second piece starts in the middle of what was a chunk), then submitted ahead
    r = run_gpu(iq, params, kind=api.NFC_IN_IQ_F32, chunk_samples=61 * rnd, pushes=[0, 300_001, 777_777, len(iq) // 2])
    assert first_diff(r['transitions'], o.transitions()) is None and r['packets'] == o.packets()
    n = len(iq) // 2
    with api.NfcContext(input_kind=api.NFC_IN_IQ_F32, chunk_samples=61 * rnd, **params) as ctx:
        tr, s0, s1, pk, ahead = _run_submitted(ctx, iq, [0, 290_000, 610_000, n], 2)
    assert first_diff(tr, o.transitions()) is None and pk == o.packets()


def test_low_runs_across_chunk_ends_are_measured_not_given_up():
    # k_threshold_wg, a chunk's summary: when the chunk's last LOW sample is in reach of the next chunk and its run may have been
    # longer than max_len, the key of that sample (did the run end on a time-out? transition_sink.py:95-99) is MEASURED in the plane
    # words the chunk has stored -- it used to make the chunk give up, and a chunk that gives up is re-run by one wave (0.65 ms for a
    # chunk of 98 304 samples whenever a cut fell into a loss of signal).  A LOW run across every chunk end: i samples of it before the
    # end (1 .. 139: every residue of max_len = 50, the time-out cases k * 50 + 1 included), 10 .. 300 behind it, a HIGH burst 0 .. 70
    # samples after its end.  Per-sample val, edges, symbols, packets against the C oracle -- and (nearly) no chunk re-run.
    C = 4096 * 3
    nb = 140
    rng = np.random.default_rng(4242)
    m = np.ones(C * (nb + 1), np.float32)
    for i in range(1, nb):
        b = (i + 1) * C
        behind = int(rng.integers(10, 300))
        m[b - i:b + behind] = 0.0
        gap = int(rng.integers(0, 70))
        hl = int(rng.integers(1, 40))
        m[b + behind + gap:b + behind + gap + hl] = 1.09
    iq = synth.iq_from_profile(m, seed=4242, sigma=0.0015)
    params = dict(hi_val=1.1)
    o = oracle_run(iq, params, api.NFC_IN_IQ_F32)
    r = run_gpu(iq, params, kind=api.NFC_IN_IQ_F32, chunk_samples=C)
    d = first_diff(r['val'][2000:], o.trace().tolist())
    assert d is None, 'val %s' % (d,)
    assert first_diff(r['transitions'], o.transitions()) is None
    assert r['sym_tag'] == o.symbols(0).tolist() and r['sym_reader'] == o.symbols(1).tolist() and r['packets'] == o.packets()
    st = r['stats']
    assert st.used_sequential == 0 and st.n_chunks == nb + 1
    # (a run that DID end on a time-out at the chunk's last sample makes the next chunk's speculated key wrong: those few are re-run)
    assert st.chunks_rerun <= 12, 'chunks re-run: %d of %d' % (st.chunks_rerun, st.n_chunks)


@pytest.mark.parametrize('stream', ['torture', 'all'])
def test_chunks_cut_by_dispatch_row(monkeypatch, stream):
    # host_threshold.h, thr_prepare: a batch that is one full wave of resident workgroups (more than three per CU, at most four) is
    # cut into chunks whose length depends on the row of 256 workgroups they are dispatched in -- the first rows longer, the last
    # shorter (threshold.hip.h: chunk_span).  20 M samples: 21 / 20 / 20 / 16 rounds per chunk where the equal cut has 20.
    # The torture stream (its level steps have chunks re-run, and the batch cut again, from that first pass) and the `all` workload
    # (certified as cut: the count of chunks shows which cut it was).  Per-sample val, edges, symbols and packets against the C
    # oracle, with the equal cut (NFC_WG_ROWBAL=0) beside it; then the stream in two pushes (the second batch starts inside what
    # was a chunk, and is cut by row itself).
    n = 20_000_000
    iq = np.concatenate([_wg_torture(900 + i, n // 10) for i in range(10)]) if stream == 'torture' else synth.workload('all', n)
    params = dict(hi_val=1.1)
    o = oracle_run(iq, params, api.NFC_IN_IQ_F32)
    want_val = np.asarray(o.trace())
    chunks = {}
    for rowbal in ('1', '0'):
        monkeypatch.setenv('NFC_WG_ROWBAL', rowbal)
        with api.NfcContext(input_kind=api.NFC_IN_IQ_F32, **params) as ctx:
            ctx.push(iq)
            st = ctx.stats()
            chunks[rowbal] = st.n_chunks
            assert st.used_sequential == 0
            assert np.array_equal(np.asarray(ctx.val())[2000:], want_val), 'val (NFC_WG_ROWBAL=%s)' % rowbal
            assert first_diff(ctx.transitions(), o.transitions()) is None
            assert ctx.symbols(0).tolist() == o.symbols(0).tolist() and ctx.symbols(1).tolist() == o.symbols(1).tolist()
            assert ctx.packets() == o.packets()
    if stream == 'all':
        assert chunks['0'] == (n + 20479) // 20480 and 768 < chunks['1'] <= 1024 and chunks['1'] != chunks['0'], chunks
    monkeypatch.setenv('NFC_WG_ROWBAL', '1')
    r = run_gpu(iq, params, kind=api.NFC_IN_IQ_F32, pushes=[0, 16_000_123, n])
    assert first_diff(r['transitions'], o.transitions()) is None and r['packets'] == o.packets()
    assert r['sym_tag'] == o.symbols(0).tolist() and r['sym_reader'] == o.symbols(1).tolist()


@pytest.mark.parametrize('window,n,rows', [(10000, 30_000_000, 3), (16000, 20_000_000, 2)])
def test_chunks_cut_by_dispatch_row_long_windows(monkeypatch, window, n, rows):
    # the same cut where a long window leaves room for THREE workgroups per CU (10 000 samples: configs[3] / [4], eight rows per step,
    # rounds of 2 048 samples; 30 M samples are 733 chunks of 20 rounds in the equal cut and 21 / 20 / rest by row) or TWO (16 000
    # samples: a 64 KB ring).  Against the C oracle (val, edges, symbols, packets), with the equal cut beside it.
    w = synth.workload('all', n - 21_000)
    iq = np.concatenate([w[:6000]] * 7 + [w])   # (the window's fill on bare carrier: 21 000 samples of the workload's own lead-in in front)
    params = dict(hi_val=1.1, av_window=window, max_len=250, samp_rate=10e6)
    o = oracle_run(iq, params, api.NFC_IN_IQ_F32)
    want_val = np.asarray(o.trace())
    chunks = {}
    for rowbal in ('1', '0'):
        monkeypatch.setenv('NFC_WG_ROWBAL', rowbal)
        with api.NfcContext(input_kind=api.NFC_IN_IQ_F32, **params) as ctx:
            ctx.push(iq)
            st = ctx.stats()
            chunks[rowbal] = st.n_chunks
            assert st.used_sequential == 0 and st.chunk_samples == 40960
            assert np.array_equal(np.asarray(ctx.val())[window:], want_val), 'val (NFC_WG_ROWBAL=%s)' % rowbal
            assert first_diff(ctx.transitions(), o.transitions()) is None
            assert ctx.symbols(0).tolist() == o.symbols(0).tolist() and ctx.symbols(1).tolist() == o.symbols(1).tolist()
            assert ctx.packets() == o.packets()
    assert chunks['0'] == (n + 40959) // 40960 and 256 * (rows - 1) < chunks['1'] <= 256 * rows and chunks['1'] != chunks['0'], chunks


def test_chunking_adapts_to_a_stream_that_needs_reruns(monkeypatch):
    # A stream whose batches need re-runs (level steps behind losses of signal: the chunk with the step gives up, the one behind it
    # cannot be certified) is cut four times finer -- from the batch that found out on: a re-run pass is one wave walking a chunk -- and goes back
    # to the clean stream's chunking after eight batches without a re-run.  Same stream, synchronous pushes and batches submitted
    # ahead mixed: the outputs are the single stream's whatever the chunking.
    monkeypatch.setenv('NFC_WG_PER_CU', '1')   # (a chunk per CU: batches of a test's size are then cut well above the smallest chunk)
    n_b = 2_400_000
    dirty = synth.stress_workload(3 * n_b, depth=0.08, sigma=0.002, every=250_000)
    clean = synth.workload('all', 10 * n_b)
    iq = np.concatenate([dirty, clean])
    n = len(iq) // 2
    o = oracle_run(iq, dict(hi_val=1.1), api.NFC_IN_IQ_F32)
    tr, s0, s1, pk, chunks, reruns = [], [], [], [], [], []
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
        def take():
            nonlocal tr, s0, s1, pk
            tr += ctx.transitions()
            s0 += ctx.symbols(0).tolist()
            s1 += ctx.symbols(1).tolist()
            pk += ctx.packets()
            st = ctx.stats()
            chunks.append(int(st.chunk_samples))
            reruns.append(int(st.chunks_rerun))
        bufs = [api.DeviceBuffer(iq[2 * k * n_b:2 * (k + 1) * n_b]) for k in range(13)]
        for k in range(5):                       # synchronous: three dirty batches, two clean ones
            ctx.push_device(bufs[k], n_b)
            take()
        nxt = 5
        for k in range(5, 13):                   # the rest submitted ahead, two in flight
            while nxt < 13 and nxt < k + 2:
                ctx.submit_device(bufs[nxt], n_b)
                nxt += 1
            ctx.wait()
            take()
    d = first_diff(tr, o.transitions())
    assert d is None, 'transition %s' % (d,)
    assert s0 == o.symbols(0).tolist() and s1 == o.symbols(1).tolist() and pk == o.packets()
    assert reruns[0] > 0 and reruns[1] > 0 and reruns[2] > 0 and sum(reruns[4:]) == 0, reruns
    # (round 4: the FIRST dirty batch is cut finer too -- pass 0's verdict says the stream needs re-runs, and pass 0 is run again on
    # the fine cut instead of re-running long chunks one wave each)
    assert chunks[0] == chunks[1] == chunks[2], chunks
    assert chunks[4] == chunks[1], chunks                                  # ... and still, two clean batches later
    assert chunks[0] * 2 <= chunks[-1], chunks                             # back on the clean stream's cut after eight without a re-run


def test_no_device_allocation_in_the_middle_of_a_stream():
    # A stream whose transition density changes must not grow its buffers in the middle (the hovering stream's second batch
    # cost 4.4 ms instead of 2.5, all of it hipMalloc).  Clean traffic first, then load modulation hovering at the HIGH threshold with
    # five times the noise (twice the entries per sample, every chunk re-run on the fine cut), then clean again -- synchronous pushes
    # and batches submitted ahead: only the stream's FIRST batch of a length may allocate (nfc_stats.device_allocs), outputs exact.
    n_b = 1_500_000
    clean = synth.workload('all', 3 * n_b)
    hover = synth.stress_workload(3 * n_b, every=400_000)
    iq = np.concatenate([clean[:2 * 2 * n_b], hover, clean[2 * 2 * n_b:]])
    n = len(iq) // 2
    assert n == 6 * n_b
    o = oracle_run(iq, dict(hi_val=1.1), api.NFC_IN_IQ_F32)
    tr, s0, s1, pk, allocs, edges = [], [], [], [], [], []
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
        def take():
            nonlocal tr, s0, s1, pk
            e = ctx.transitions()
            tr += e
            edges.append(len(e))
            s0 += ctx.symbols(0).tolist()
            s1 += ctx.symbols(1).tolist()
            pk += ctx.packets()
            allocs.append(int(ctx.stats().device_allocs))
        bufs = [api.DeviceBuffer(iq[2 * k * n_b:2 * (k + 1) * n_b]) for k in range(6)]
        for k in range(4):                       # clean, clean, hovering, hovering: one batch at a time
            ctx.push_device(bufs[k], n_b)
            take()
        ctx.submit_device(bufs[4], n_b)          # hovering (falls back to the synchronous path inside nfc_wait), clean
        ctx.submit_device(bufs[5], n_b)
        for _ in range(2):
            ctx.wait()
            take()
    assert first_diff(tr, o.transitions()) is None
    assert s0 == o.symbols(0).tolist() and s1 == o.symbols(1).tolist() and pk == o.packets()
    assert edges[2] > 1.3 * edges[1], edges      # the hovering batches do hold more transitions than the estimate of the clean ones
    assert allocs[0] > 0, allocs                 # the stream's first batch sizes the buffers ...
    assert allocs[1:4] == [0, 0, 0], allocs      # ... and nothing grows when the stream turns dense, or when its chunks are cut finer
    # (the first batches submitted ahead bring the second set of planes and the snapshots' buffers: once)
    assert allocs[5] == 0, allocs


@pytest.mark.parametrize('kind', ['iq', 'env', 'real', 'i16'])
def test_failed_rounds_are_evaluated_in_place(monkeypatch, kind):
    # Level steps behind losses of signal: the chunk with the step gives up in pass 0 (the window is being overwritten with a new level:
    # no drift allowance covers that), the ones behind it cannot be certified.  Up to a machine-full of failing chunks are re-run by
    # k_threshold_wg<KIND, 4, true> (round 6): a round that fails its check is taken back and evaluated by the four waves the way
    # k_threshold evaluates a step, and the chunk goes on -- the same transitions, symbols and packets, bit for bit, as the oracle's and
    # as with k_threshold re-running everything (NFC_WG_EX=0), whatever the input kind and wherever the pushes cut the stream.
    n = 3_000_000
    iq = synth.stress_workload(n, depth=0.08, sigma=0.002, every=250_000)
    params = dict(hi_val=1.1)
    if kind == 'iq':
        x, k = iq, api.NFC_IN_IQ_F32
    elif kind == 'env':
        x, k = synth.envelope_f32(iq), api.NFC_IN_ENV_F32
    elif kind == 'real':
        x, k = np.sqrt(synth.envelope_f32(iq).astype(np.float64)).astype(np.float32), api.NFC_IN_REAL_F32_SQ
    else:
        x, k = np.clip(np.round(np.sqrt(synth.envelope_f32(iq).astype(np.float64)) * 20000.0), -32768, 32767).astype(np.int16), api.NFC_IN_I16_SQ
    r = check_vs_oracle(x, params, kind=k)
    st = r['stats']
    assert st.used_sequential == 0 and st.chunks_rerun_in_place > 0 and st.chunks_rerun_in_place >= st.chunks_rerun - 2, (st.chunks_rerun, st.chunks_rerun_in_place)
    r1 = check_vs_oracle(x, params, kind=k, pushes=[0, 249_900, 250_200, 251_500, 1_000_001, 2_750_300, n])
    monkeypatch.setenv('NFC_WG_EX', '0')
    r0 = check_vs_oracle(x, params, kind=k)
    assert r0['stats'].chunks_rerun_in_place == 0 and r0['stats'].chunks_rerun > 0
    assert r0['transitions'] == r['transitions'] == r1['transitions'] and r0['packets'] == r['packets'] == r1['packets']


def test_in_place_reruns_with_a_long_window_and_hovering_samples(monkeypatch):
    # ... with the window of a 10 MS/s capture (pass 0 runs eight rows per step there, the re-runs four), and on the capture whose loaded
    # half bits hover AT the HIGH threshold (every other round fails its check)
    n = 1_500_000
    iq = synth.stress_workload(n, depth=0.08, sigma=0.002, every=200_000)
    # (chunks of whole eight-row rounds: a batch of a test's size would otherwise be cut into the window's smallest chunk, which is not)
    r = check_vs_oracle(iq, dict(hi_val=1.1, samp_rate=1e7, av_window=10000, max_len=250), kind=api.NFC_IN_IQ_F32, chunk_samples=22528)
    assert r['stats'].chunks_rerun_in_place > 0
    # ... a window that holds exactly one round (the steps' ring slots just stay disjoint), one that does not divide by anything; losses of
    # signal longer than the two rounds the in-place form looks back for where a LOW run began (it gives the chunk up: k_threshold's)
    short = synth.stress_workload(1_200_000, depth=0.08, sigma=0.002, every=150_000)
    for L in (1024, 1500):
        r = check_vs_oracle(short, dict(hi_val=1.1, av_window=L), kind=api.NFC_IN_IQ_F32)
        assert r['stats'].chunks_rerun_in_place > 0, L
    gone = synth.stress_workload(2_000_000, depth=0.08, sigma=0.002, every=250_000, dropout=5000)
    r = check_vs_oracle(gone, dict(hi_val=1.1), kind=api.NFC_IN_IQ_F32)
    assert r['stats'].chunks_rerun_in_place > 0
    hov = synth.stress_workload(1_000_000)
    monkeypatch.setenv('NFC_WG_EX', '100000')   # (the product sends a batch where EVERY chunk fails to k_threshold: here the in-place form takes them)
    r = check_vs_oracle(hov, dict(hi_val=1.1), kind=api.NFC_IN_IQ_F32)
    assert r['stats'].chunks_rerun_in_place > 0


@pytest.mark.parametrize('where', [1_500_000, 1_500_000 + 2048, 2_000_001])
def test_a_lone_failure_is_rerun_by_the_workgroup_kernel(monkeypatch, where):
    # One level step of +15 % in an otherwise clean stream: the chunk that holds it gives up, the one behind it cannot be certified
    # against a summary that is worth nothing.  Few failures on a clean batch take k_threshold_wg in mode 1 (from the exact state,
    # four waves per chunk) instead of the general kernel's one wave (host_threshold.h: `lone`; round 6) -- same result, bit for bit,
    # as the oracle's and as with the general kernel re-running everything (NFC_WG_RERUN=0).
    iq = synth.workload('miller', 3_000_000).copy()
    iq[2 * where:] *= np.float32(np.sqrt(1.15))
    params = dict(hi_val=1.1, tag=False)
    r = check_vs_oracle(iq, params, kind=api.NFC_IN_IQ_F32)
    st = r['stats']
    assert st.used_sequential == 0 and 1 <= st.chunks_rerun <= 6 and st.threshold_passes <= 4, (st.chunks_rerun, st.threshold_passes)
    monkeypatch.setenv('NFC_WG_RERUN', '0')
    monkeypatch.setenv('NFC_WG_EX', '0')
    r0 = check_vs_oracle(iq, params, kind=api.NFC_IN_IQ_F32)
    assert r0['transitions'] == r['transitions'] and r0['packets'] == r['packets']
    # ... and across pushes that put the step on a batch's first chunk
    check_vs_oracle(iq, params, kind=api.NFC_IN_IQ_F32, pushes=[0, where - 100, where + 50_000, 3_000_000])


@pytest.mark.parametrize('name,kw', [('miller', dict(tag=False)), ('manchester', dict(reader=False)), ('all', dict())])
def test_fused_tail_on_the_workloads(monkeypatch, name, kw):
    # tail.hip.h: edges, decoders and framing in ONE persistent launch -- a tile end to end per workgroup, three decoupled look-backs
    # over self-validating status words.  Built in round 6, measured (1.8 x the five launches it replaces: DESIGN.md 6c) and kept in the
    # test build only (NFC_TAIL=1); these tests keep it exact: val / edges / symbols / packets against the oracle, in one push and
    # across pushes that cut frames, and the symbol arrays read on demand (k_sym_reduce).
    monkeypatch.setenv('NFC_TAIL', '1')
    iq = synth.workload(name, 3_000_000)
    r = check_vs_oracle(iq, dict(hi_val=1.1, **kw), kind=api.NFC_IN_IQ_F32)
    assert r['stats'].tail_fused == 1 and r['stats'].used_sequential == 0
    n = len(iq) // 2
    r = check_vs_oracle(iq, dict(hi_val=1.1, **kw), kind=api.NFC_IN_IQ_F32, pushes=[0, 700_001, 1_400_003, 1_400_003 + 300_000, n])
    assert r['stats'].tail_fused == 1


@pytest.mark.parametrize('max_len', [7, 50, 120])
def test_fused_tail_dense_tiles_and_long_runs(monkeypatch, max_len):
    # per-sample flicker (more entries than a tile stages: the launch flags the batch, the host repeats it with shorter tiles),
    # LOW runs around max_len across tile seams, idle stretches of heartbeats only
    monkeypatch.setenv('NFC_TAIL', '1')
    monkeypatch.setenv('NFC_NO_SMALL', '1')
    rng = np.random.default_rng(40 + max_len)
    n = 900_000
    x = (0.3 * (1 + 0.003 * rng.standard_normal(n))).astype(np.float32)
    x[300_000:340_000:2] *= np.float32(1.3)            # HIGH every other sample: 20 000 entries in 40 000 samples
    x[500_000:500_000 + 3 * 32768] *= np.float32(1.0)  # (idle tiles)
    for k in range(60):
        s = 600_000 + k * 4000 + int(rng.integers(0, 64))
        ln = int(rng.choice([max_len - 1, max_len, max_len + 1, 2 * max_len + 1, 5 * max_len]))
        x[s:s + ln] = 1e-6
    for seam in range(32768, n - 40000, 32768 * 3):    # runs and pulses across the 512-word tile seams
        x[seam - 3:seam + 2] = 1e-6
        x[seam + 40:seam + 43] *= np.float32(1.3)
    params = dict(hi_val=1.1, max_len=max_len)
    r = check_vs_oracle(x, params)
    assert r['stats'].tail_fused == 1
    check_vs_oracle(x, params, pushes=[0, 310_001, 650_000, n])


def test_fused_tail_golden_fixture_in_long_batches(monkeypatch):
    # a reference-generated fixture through the fused launch: the Ultralight transaction tiled to several tiles
    monkeypatch.setenv('NFC_TAIL', '1')
    monkeypatch.setenv('NFC_NO_SMALL', '1')
    c = Case('fx_ultralight_txn')
    check_case(c, run_gpu(c.x, c.params))
    check_case(c, run_gpu(c.x, c.params, pushes=[0, 2500, 9000, len(c.x)]))
