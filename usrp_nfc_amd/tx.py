"""Transmit side over the C-ABI: encode_bits of the reference's three encoders
(miller.py:200-233, manchester.py:64-79, binary_src.py:17-20) and the device-side renderer that stands where
binary_src.work (binary_src.py:64-103) and multiplier (multiplier.py:18-22) stand in the reference's TX chain."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import NFC_TX_MANCHESTER, NFC_TX_MILLER, NFC_TX_SAME, TX_RUN_DTYPE   # noqa: F401


def _check(rc, what):
    if rc != 0:
        msg = _lib.load().nfc_last_error(None)
        raise RuntimeError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))


def encode_runs(encoding, bits):
    """-> structured array of nfc_tx_run (level, dur_us)."""
    L = _lib.load()
    b = np.ascontiguousarray(np.asarray(bits, dtype=np.uint8))
    cap = 3 * b.size + 8
    out = np.zeros(cap, TX_RUN_DTYPE)
    n = C.c_size_t(0)
    _check(L.nfc_tx_encode(int(encoding), b.ctypes.data, b.size, out.ctypes.data, cap, C.byref(n)), 'nfc_tx_encode')
    return out[:n.value]


def encode_bits(encoding, bits):
    """The reference's return type: a list of (level, microseconds) tuples."""
    r = encode_runs(encoding, bits)
    return [(int(l), float(d)) for l, d in zip(r['level'], r['dur_us'])]


def as_runs(pulses):
    """(level, us) tuples -> nfc_tx_run array."""
    r = np.zeros(len(pulses), TX_RUN_DTYPE)
    if len(pulses):
        r['level'] = [p[0] for p in pulses]
        r['dur_us'] = [p[1] for p in pulses]
    return r


def sample_count(runs, samp_rate):
    L = _lib.load()
    runs = np.ascontiguousarray(runs)
    n = C.c_uint64(0)
    _check(L.nfc_tx_sample_count(runs.ctypes.data, runs.size, float(samp_rate), C.byref(n)), 'nfc_tx_sample_count')
    return int(n.value)


def render_device(runs, samp_rate, dev_ptr, cap_samples, carrier=False, freq=13.56e6, amp=1.0, first_index=0, device=0,
                  timed=False):
    """Render into device memory (complex64).  -> (n_samples, kernel_ms or None)."""
    L = _lib.load()
    runs = np.ascontiguousarray(runs)
    n = C.c_size_t(0)
    ms = C.c_float(0)
    _check(L.nfc_tx_render_device(int(device), runs.ctypes.data, runs.size, float(samp_rate), 1 if carrier else 0, float(freq),
                                  float(amp), int(first_index), dev_ptr, int(cap_samples), C.byref(n),
                                  C.byref(ms) if timed else None), 'nfc_tx_render_device')
    return int(n.value), (float(ms.value) if timed else None)


def render(runs, samp_rate, carrier=False, freq=13.56e6, amp=1.0, first_index=0, device=0):
    """Render on the GPU and bring the samples back: complex64 array."""
    from . import api
    n = sample_count(runs, samp_rate)
    buf = api.DeviceBuffer(np.zeros(0, np.float32), device, nbytes=max(8 * n, 32))
    try:
        got, _ = render_device(runs, samp_rate, buf.ptr, n, carrier, freq, amp, first_index, device)
        return buf.download(8 * got).view(np.complex64)
    finally:
        buf.free()
