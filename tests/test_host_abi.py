"""CPU-side checks of the product: the C-ABI library loads and exports every symbol of
include/nfc_amd.h, and the host-built decoder LUTs reproduce the reference decoders'
golden vectors (no GPU calls here)."""
import os
import re

import numpy as np
import pytest

from tests.golden_util import load_json, load_npz
from usrp_nfc_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    hdr = open(os.path.join(ROOT, 'include', 'nfc_amd.h')).read()
    declared = set(re.findall(r'\b(nfc_[a-z_0-9]+)\s*\(', hdr)) - {'nfc_ctx'}
    assert declared == set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.nfc_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header():
    import ctypes as C
    assert _lib.EDGE_DTYPE.itemsize == 16
    assert _lib.PACKET_DTYPE.itemsize == 24
    assert C.sizeof(_lib.Params) == 64
    assert C.sizeof(_lib.Counts) == 64


def test_lut_report_example_and_reqa():
    fx = load_json('fx_report_miller.json')
    # report section 4.3: durations in microseconds at 1 Msps are integer sample counts
    p = fx['report_example']['pulses']
    got = api.host_decode_lut(1, [c for c, _ in p], [d for _, d in p], samp_rate=1e6, max_len=50)
    assert got.tolist() == fx['report_example']['symbols'] == [0, 1, 0, 1]


@pytest.mark.parametrize('key,factor', [('1', 1.0), ('0p5', 0.5), ('0p25', 0.25), ('0p1', 0.1)])
def test_lut_random_vectors(key, factor):
    z = load_npz('fx_decoder_vectors.npz')
    rate = 1e6 / factor
    got = api.host_decode_lut(1, z['curm_' + key], z['d_' + key], samp_rate=rate)
    assert got.tolist() == z['symm_' + key].tolist()
    got = api.host_decode_lut(0, z['curt_' + key], z['d_' + key], samp_rate=rate)
    assert got.tolist() == z['symt_' + key].tolist()


@pytest.mark.parametrize('key,factor', [('frames_0p5', 0.5), ('frames_0p25', 0.25)])
def test_lut_frame_vectors(key, factor):
    z = load_npz('fx_decoder_vectors.npz')
    rate = 1e6 / factor
    got = api.host_decode_lut(1, z['curm_' + key], z['dm_' + key], samp_rate=rate)
    assert got.tolist() == z['symm_' + key].tolist()
    got = api.host_decode_lut(0, z['curt_' + key], z['dt_' + key], samp_rate=rate)
    assert got.tolist() == z['symt_' + key].tolist()


@pytest.mark.parametrize('key,factor', [('1', 1.0), ('0p5', 0.5), ('0p25', 0.25), ('0p1', 0.1), ('frames_0p5', 0.5), ('frames_0p25', 0.25)])
def test_miller_quotient_machine_on_the_reference_vectors(key, factor):
    # The speculative decode kernel walks the Miller decoder's QUOTIENT machine (decoder_tables.h: miller_quotient -- states no
    # transition sequence can tell apart are one class, so a state map is 8 bytes): driven sequentially on the host (type 2) it must
    # emit exactly what the reference's decoder emitted on its own vectors, every error branch included.
    z = load_npz('fx_decoder_vectors.npz')
    rate = 1e6 / factor
    d = z['dm_' + key] if 'dm_' + key in z else z['d_' + key]
    got = api.host_decode_lut(2, z['curm_' + key], d, samp_rate=rate)
    assert got.tolist() == z['symm_' + key].tolist()


def test_miller_classes_are_what_the_reference_decoder_cannot_tell_apart():
    import ctypes as C
    from usrp_nfc_amd import _lib
    L = _lib.load()
    for rate, mx in ((2e6, 50), (1e7, 250), (1e6, 50), (4e6, 100)):
        p = api._params(rate, 0.1, 1.1, 2000, mx, True, True, api.NFC_IN_IQ_F32, 0, 0.0, 0, 0)
        q_of, canon, ncls = (C.c_uint8 * 16)(), (C.c_uint8 * 16)(), C.c_int(0)
        assert L.nfc_host_miller_classes(C.byref(p), q_of, canon, C.byref(ncls)) == 0
        q_of, canon = list(q_of), list(canon)
        # state = stage | has_started << 2 | prev << 3 (miller.py:14-29).  Reachable: has_started False only in stage BEGINNING
        # (reset() sets both, miller.py:65-67); `_prev` is read in ONE place, stage BEGINNING of a started frame (miller.py:81):
        # everywhere else the two values of prev are one class
        assert ncls.value == 6
        reachable = [s for s in range(16) if (s & 4) or (s & 3) == 0]
        assert all(q_of[s] != 0xFF for s in reachable) and all(q_of[s] == 0xFF for s in range(16) if s not in reachable)
        for s in reachable:
            stage, started, prev = s & 3, (s >> 2) & 1, s >> 3
            if stage == 0 and started:
                assert canon[s] == s and q_of[s] != q_of[s ^ 8]      # prev matters here, and only here
            else:
                assert canon[s] == (s & 7) and q_of[s] == q_of[s & 7]


def test_no_gpu_means_loud_failure():
    L = _lib.load()
    if L.nfc_device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(api.NfcError):
        api.NfcContext()


def test_no_gpu_means_loud_failure_for_the_tx_renderer():
    # the transmit-side renderer has no CPU path either (the encoders are host code by design: test_tx.py)
    L = _lib.load()
    if L.nfc_device_count() > 0:
        pytest.skip('a GPU is present')
    from usrp_nfc_amd import tx
    runs = tx.as_runs([(1, 9.44), (0, 3.0)])
    assert tx.sample_count(runs, 2e6) == 24
    with pytest.raises(RuntimeError):
        tx.render_device(runs, 2e6, 32, 24)   # (any pointer: the call must fail before touching it)


def test_wavfile_source_normalisation_is_the_ieee_quotient():
    # NFC_IN_I16_SQ, i16_scale 0: s = fl(pcm / 32767) as GNU Radio's wavfile_source computes it.  The kernels get there
    # without a division (threshold.hip.h: i16_to_float); the identity is checked here for all 65536 inputs on the host
    import ctypes as C
    from usrp_nfc_amd import _lib
    L = _lib.load()
    v = np.arange(-32768, 32768, dtype=np.int64)
    want = (v.astype(np.float32) / np.float32(32767.0)).astype(np.float32)
    got = np.array([L.nfc_host_i16_to_float(int(k), 0.0) for k in v], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    sc = np.float32(1.0 / 32768.0)
    got2 = np.array([L.nfc_host_i16_to_float(int(k), float(sc)) for k in v[::257]], np.float32)
    assert np.array_equal(got2, (v[::257].astype(np.float32) * sc).astype(np.float32))


def test_window_beyond_the_upper_bound_is_refused():
    # argument checks come before any device work: this holds without a GPU
    from usrp_nfc_amd import api
    for kw in (dict(av_window=30001), dict(av_window=0), dict(max_len=4001), dict(samp_rate=-1.0)):
        with pytest.raises(api.NfcError) as e:
            api.NfcContext(**kw)
        assert 'av_window' in str(e.value) or 'max_len' in str(e.value) or 'samp_rate' in str(e.value)


def test_ctypes_structures_match_the_header(tmp_path):
    # the sizes the C compiler gives the header's structures against the ctypes mirrors in usrp_nfc_amd/_lib.py, and the ABI version
    # both sides name: a field added on one side only is caught here (nfc_get_stats writes the whole structure into the caller's)
    import ctypes
    import os
    import subprocess
    from usrp_nfc_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / 'sz.c'
    src.write_text('#include <stdio.h>\n#include "nfc_amd.h"\nint main(void) { printf("%d %zu %zu %zu %zu %zu %zu %zu\\n", NFC_AMD_ABI_VERSION, '
                   'sizeof(nfc_params), sizeof(nfc_counts), sizeof(nfc_stats), sizeof(nfc_state_header), sizeof(nfc_edge), sizeof(nfc_packet), '
                   'sizeof(nfc_frame)); return 0; }\n')
    exe = tmp_path / 'sz'
    subprocess.check_call(['gcc', '-I', os.path.join(root, 'include'), str(src), '-o', str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [_lib.ABI_VERSION, ctypes.sizeof(_lib.Params), ctypes.sizeof(_lib.Counts), ctypes.sizeof(_lib.Stats), ctypes.sizeof(_lib.StateHeader),
            _lib.EDGE_DTYPE.itemsize, _lib.PACKET_DTYPE.itemsize, ctypes.sizeof(_lib.Frame)]
    assert got == want, (got, want)
    assert _lib.load().nfc_abi_version() == _lib.ABI_VERSION == 4


def _row_cut(n, C_, rs, cus, rows, factors, max_len=0):
    import ctypes as C
    L = _lib.load()
    f = (C.c_double * 3)(*(list(factors) + [1.0] * 3)[:3])
    out = (C.c_uint32 * 10)()
    rc = L.nfc_plan_row_cut(n, C_, rs, cus, rows, f, max_len, out)
    assert rc >= 0
    return rc, list(out)


def _spans(n, out):
    """chunk_span of csrc/threshold.hip.h, restated: the chunks of the table that begin inside the batch."""
    ln, start, div, nch = out[0:4], out[4:8], out[8], out[9]
    spans = []
    for c in range(nch):
        r = min(c // div, 3)
        a = start[r] + (c - r * div) * ln[r]
        spans.append((a, min(n, a + ln[r])))
    return spans


@pytest.mark.parametrize('rows,factors', [(4, (1.036, 1.015, 0.990)), (3, (1.045, 1.004)), (2, (1.02,))])
def test_row_cut_covers_every_sample_once(rows, factors):
    # csrc/chunk_cut.h (host_threshold.h: the threshold stage's time chunks cut by dispatch row): whatever the batch length, the chunks
    # of the table are whole rounds, lie end to end from sample 0, cover the batch exactly, are at most rows * cus, the first rows'
    # are the longest -- or the table is the equal cut.  The reference's loop is one chunk (transition_sink.py:37-107): this is the
    # geometry of its parallel restatement, and every sample must be classified exactly once.
    rng = np.random.default_rng(7)
    cus = 256
    by_row = 0
    for trial in range(400):
        rs = int(rng.choice([1024, 1536, 2048]))
        n = int(rng.integers(rs * cus, 2_000_000_000)) if trial % 3 else int(rng.integers(1, 4 * cus * rs * 40))
        slots = rows * cus
        C_ = max(4096 // rs * rs if 4096 % rs == 0 else rs * 3, -(-(-(-n // slots)) // rs) * rs)   # as thr_prepare picks it: one wave of slots, whole rounds
        rc, out = _row_cut(n, C_, rs, cus, rows, factors, max_len=int(rng.choice([0, 0, C_ + 8 * rs, C_ + rs])))
        spans = _spans(n, out)
        assert spans and spans[0][0] == 0 and spans[-1][1] == n, (n, C_, rs, out)
        for (a0, a1), (b0, b1) in zip(spans[:-1], spans[1:]):
            assert a1 == b0 and a1 > a0, (n, C_, rs, out)
        assert all((b - a) % rs == 0 for a, b in spans[:-1])
        assert all(l % rs == 0 and l >= rs for l in out[0:4])
        if rc:
            by_row += 1
            assert out[9] <= rows * cus and out[9] > (rows - 1) * cus
            assert out[0] >= out[rows - 1] and out[0] >= C_ and out[rows - 1] <= C_
            assert out[8] == cus
        else:
            assert out[0:4] == [C_] * 4 and out[9] == -(-n // C_)
    assert by_row > 100


def test_row_cut_is_the_bench_cut():
    # configs[1] on an MI355X: 1e8 samples, 256 CUs, four workgroups per CU, rounds of 1 024 samples
    rc, out = _row_cut(100_000_000, 98304, 1024, 256, 4, (1.036, 1.015, 0.990), max_len=101 * 1024)
    assert rc == 1 and out[0:4] == [101376, 99328, 97280, 93184] and out[8] == 256 and out[9] == 1023
    # ... and where the longest chunk's plane words would not fit the LDS any more: the equal cut
    rc, out = _row_cut(100_000_000, 98304, 1024, 256, 4, (1.036, 1.015, 0.990), max_len=98 * 1024)
    assert rc == 0 and out[0:4] == [98304] * 4 and out[9] == 1018
