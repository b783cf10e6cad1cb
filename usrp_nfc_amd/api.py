"""Python face of the C-ABI: one NfcContext = one stream (include/nfc_amd.h)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (EDGE_DTYPE, NFC_FLAG_FORCE_SEQUENTIAL, NFC_FLAG_NO_EDGES, NFC_IN_ENV_F32, NFC_IN_I16_SQ,
                   NFC_IN_IQ_F32, NFC_IN_REAL_F32_SQ, PACKET_DTYPE)

__all__ = ['NfcContext', 'NfcError', 'DeviceBuffer', 'host_decode_lut', 'NFC_IN_IQ_F32', 'NFC_IN_ENV_F32', 'NFC_IN_REAL_F32_SQ',
           'NFC_IN_I16_SQ', 'NFC_FLAG_FORCE_SEQUENTIAL', 'NFC_FLAG_NO_EDGES']

_KIND_DTYPE = {NFC_IN_IQ_F32: (np.float32, 2), NFC_IN_ENV_F32: (np.float32, 1),
               NFC_IN_REAL_F32_SQ: (np.float32, 1), NFC_IN_I16_SQ: (np.int16, 1)}


class NfcError(RuntimeError):
    pass


def _params(samp_rate, lo_val, hi_val, av_window, max_len, reader, tag, input_kind, device, i16_scale, flags,
            chunk_samples):
    return _lib.Params(float(samp_rate), float(lo_val), float(hi_val), int(av_window), int(max_len), int(bool(reader)),
                       int(bool(tag)), int(input_kind), int(device), float(i16_scale), int(flags), int(chunk_samples), 0)


class NfcContext(object):
    """Arguments mirror transition_sink.transition_sink (transition_sink.py:12) and
    background.background (background.py:17) of the reference."""

    def __init__(self, samp_rate=2e6, lo_val=0.1, hi_val=1.1, av_window=2000, max_len=50, reader=True, tag=True,
                 input_kind=NFC_IN_IQ_F32, device=0, i16_scale=0.0, flags=0, chunk_samples=0, lib_path=None):
        """lib_path: another build of libnfc_amd.so for this context (tests: the build with the test hooks, _lib.hooks_path())."""
        self.L = _lib.load(lib_path)
        self.h = C.c_void_p()
        self.input_kind = input_kind
        self.factor = 1e6 / samp_rate
        self.av_window = int(av_window)
        p = _params(samp_rate, lo_val, hi_val, av_window, max_len, reader, tag, input_kind, device, i16_scale, flags,
                    chunk_samples)
        rc = self.L.nfc_create(C.byref(p), C.byref(self.h))
        if rc != 0:
            raise NfcError('nfc_create: %s (status %d)' % (self.L.nfc_last_error(None).decode(), rc))

    def close(self):
        if getattr(self, 'h', None):
            self.L.nfc_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, what):
        if rc != 0:
            raise NfcError('%s: %s (status %d)' % (what, self.L.nfc_last_error(self.h).decode(), rc))

    # -- input ---------------------------------------------------------------
    def push(self, samples):
        """Host array: float32 IQ interleaved / complex64, float32 envelope or real, or int16 PCM."""
        dt, per = _KIND_DTYPE[self.input_kind]
        a = np.asarray(samples)
        if a.dtype == np.complex64 and per == 2:
            a = a.view(np.float32)
        a = np.ascontiguousarray(a, dtype=dt)
        n = a.size // per
        self._chk(self.L.nfc_push(self.h, a.ctypes.data, n), 'nfc_push')
        return n

    def push_device(self, dev_ptr, n):
        """Device pointer (int), n samples, 16-byte aligned."""
        ptr = dev_ptr.ptr if isinstance(dev_ptr, DeviceBuffer) else C.c_void_p(int(dev_ptr))
        self._chk(self.L.nfc_push_device(self.h, ptr, n), 'nfc_push_device')
        return n

    def submit_device(self, dev_ptr, n):
        """Enqueue a batch and return (nfc_submit_device); at most three in flight.  The buffer stays untouched until its wait()."""
        ptr = dev_ptr.ptr if isinstance(dev_ptr, DeviceBuffer) else C.c_void_p(int(dev_ptr))
        self._chk(self.L.nfc_submit_device(self.h, ptr, n), 'nfc_submit_device')
        return n

    def wait(self):
        """Complete the oldest submitted batch (nfc_wait): its outputs are then read like those of a push."""
        self._chk(self.L.nfc_wait(self.h), 'nfc_wait')

    def submitted(self):
        return int(self.L.nfc_submitted(self.h))

    # -- outputs of the last push -----------------------------------------------
    def push_edges(self, edges):
        """Transitions (EDGE_DTYPE rows: idx, d in samples, v, t) through the decode and framing stages alone (nfc_push_edges) --
        what background.append receives in the reference."""
        e = np.ascontiguousarray(edges, EDGE_DTYPE)
        self._chk(self.L.nfc_push_edges(self.h, e.ctypes.data if len(e) else None, len(e)), 'nfc_push_edges')

    def counts(self):
        c = _lib.Counts()
        self._chk(self.L.nfc_get_counts(self.h, C.byref(c)), 'nfc_get_counts')
        return c

    def stats(self):
        s = _lib.Stats()
        self._chk(self.L.nfc_get_stats(self.h, C.byref(s)), 'nfc_get_stats')
        return s

    def window_converged(self):
        """After a warm-up (prime + overlap): has every window slot taken an accepted sample since the prime?  (nfc_stats'
        ring_slots_carried of the last batch: slots whose value is still the one the batch started from.)"""
        st = self.stats()   # (a batch that needed the sequential prefix is not vouched for: its parallel rest counts from the prefix's end)
        return int(st.ring_slots_carried) == 0 and not int(st.used_sequential)

    def set_timing(self, level):
        """0: no HIP events (default), 1: threshold kernel durations (events attached to the launches), 2: + batch total and per-stage split (stream markers, a few us each)."""
        self._chk(self.L.nfc_set_timing(self.h, int(level)), 'nfc_set_timing')

    def _read(self, fn, total, dtype, *lead):
        out = np.zeros(int(total), dtype)
        got = C.c_size_t(0)
        if total:
            self._chk(fn(self.h, *lead, out.ctypes.data, out.size, C.byref(got)), fn.__name__)
            if got.value != out.size:
                raise NfcError('%s returned %d of %d items' % (fn.__name__, got.value, out.size))
        return out

    def edges(self):
        return self._read(self.L.nfc_read_edges, self.counts().n_edges, EDGE_DTYPE, 0)

    def edges_compact(self, out=None):
        """(pos, code) of the batch's transitions as the device keeps them (nfc_read_edges_compact): batch-local sample
        position, and ((v + 1) * (max_len + 1) + d) | (t + 1) << 14."""
        n = int(self.counts().n_edges)
        if out is not None:   # caller's arrays (pinned ones -- pinned_array -- are written by the copy engine in place)
            if len(out[0]) < n or len(out[1]) < n:
                raise NfcError('edges_compact: the output arrays hold %d entries, the batch has %d' % (min(len(out[0]), len(out[1])), n))
            pos, code = out[0][:n], out[1][:n]
        else:
            pos, code = np.zeros(n, np.uint32), np.zeros(n, np.uint16)
        got = C.c_size_t(0)
        if n:
            self._chk(self.L.nfc_read_edges_compact(self.h, 0, pos.ctypes.data, code.ctypes.data, n, C.byref(got)), 'nfc_read_edges_compact')
            if got.value != n:
                raise NfcError('nfc_read_edges_compact returned %d of %d items' % (got.value, n))
        return pos, code

    def transitions(self):
        """The list transition_sink hands to its callback: [((v, d*factor), t), ...]."""
        e = self.edges()
        f = self.factor
        return [((int(v), int(d) * f), int(t)) for v, d, t in zip(e['v'], e['d'], e['t'])]

    def symbols(self, ptype):
        return self._read(self.L.nfc_read_symbols, self.counts().n_symbols[ptype], np.uint8, ptype, 0)

    def packet_table(self, ptype):
        n = self.counts().n_packets[ptype]
        out = np.zeros(int(n), PACKET_DTYPE)
        got = C.c_size_t(0)
        if n:
            self._chk(self.L.nfc_read_packets(self.h, ptype, out.ctypes.data, out.size, C.byref(got)), 'nfc_read_packets')
        return out

    def packet_bits(self, ptype):
        """The per-type bit array the packet table's bit_off / n_bits index (nfc_read_packet_bits)."""
        tab = self.packet_table(ptype)
        if not len(tab):
            return np.zeros(0, np.uint8)
        bits = np.zeros(int((tab['bit_off'] + tab['n_bits']).max()), np.uint8)
        got = C.c_size_t(0)
        if bits.size:
            self._chk(self.L.nfc_read_packet_bits(self.h, ptype, 0, bits.ctypes.data, bits.size, C.byref(got)), 'nfc_read_packet_bits')
        return bits

    def packets(self):
        """Closed packets of both types in stream order: [(type, [bits]), ...] -- what
        CombinedPacketProcessor hands to fsm.process_bits (packets.py:96-98)."""
        items = []
        for t in (0, 1):
            tab = self.packet_table(t)
            if not len(tab):
                continue
            hi = int((tab['bit_off'] + tab['n_bits']).max())
            bits = np.zeros(hi, np.uint8)
            got = C.c_size_t(0)
            self._chk(self.L.nfc_read_packet_bits(self.h, t, 0, bits.ctypes.data, bits.size, C.byref(got)),
                      'nfc_read_packet_bits')
            for p in tab:
                o = int(p['bit_off'])
                items.append((int(p['idx']), t, bits[o:o + int(p['n_bits'])].tolist()))
        # one edge feeds one decoder, so closing indices of the two types never tie
        items.sort(key=lambda r: r[0])
        return [(t, b) for _, t, b in items]

    def val(self):
        return self._read(self.L.nfc_read_val, self.counts().n_samples, np.int8, 0)

    def reset(self):
        """Start a new stream (state of a fresh context), keeping the device buffers."""
        self._chk(self.L.nfc_reset(self.h), 'nfc_reset')

    def get_state(self):
        """(header, ring float32[av_window], [pending bits type 0, pending bits type 1])."""
        h = _lib.StateHeader()
        self._chk(self.L.nfc_get_state(self.h, C.byref(h), None, 0, None, 0), 'nfc_get_state')
        ring = np.zeros(h.av_window, np.float32)
        p0, p1 = int(h.n_pending_bits[0]), int(h.n_pending_bits[1])
        pend = np.zeros(p0 + p1, np.uint8)
        self._chk(self.L.nfc_get_state(self.h, C.byref(h), ring.ctypes.data, ring.size, pend.ctypes.data, pend.size),
                  'nfc_get_state')
        return h, ring, [pend[:p0].copy(), pend[p0:].copy()]

    def set_state(self, header, ring, pending=None):
        ring = np.ascontiguousarray(ring, np.float32)
        pend = np.zeros(0, np.uint8) if pending is None else np.ascontiguousarray(np.concatenate(pending), np.uint8)
        self._chk(self.L.nfc_set_state(self.h, C.byref(header), ring.ctypes.data, ring.size, pend.ctypes.data, pend.size),
                  'nfc_set_state')

    def state_blob(self):
        """Boundary state as one uint8 vector: header | ring | pending bits (for RCCL exchange)."""
        h, ring, pend = self.get_state()
        return np.concatenate([np.frombuffer(bytes(h), np.uint8), ring.view(np.uint8), pend[0], pend[1]])

    def export_state(self, dev_ptr, cap):
        """Boundary state into device memory (16-byte aligned), asynchronously: [u32 len | 12 B | state_blob() bytes]."""
        got = C.c_size_t(0)
        self._chk(self.L.nfc_export_state(self.h, C.c_void_p(int(dev_ptr)), int(cap), C.byref(got)), 'nfc_export_state')
        return got.value

    def sync(self):
        self._chk(self.L.nfc_sync(self.h), 'nfc_sync')

    def set_stream(self, stream):
        """Run on the caller's HIP stream (an integer hipStream_t, e.g. torch.cuda.current_stream().cuda_stream; 0 / None:
        the context's own again)."""
        h = int(stream or 0)
        if h == getattr(self, '_stream', 0):
            return   # (the switch waits for the stream it leaves: not something to repeat per batch)
        self._chk(self.L.nfc_set_stream(self.h, C.c_void_p(h)), 'nfc_set_stream')
        self._stream = h

    def set_state_blob(self, blob):
        blob = np.ascontiguousarray(blob, np.uint8)
        hs = C.sizeof(_lib.StateHeader)
        h = _lib.StateHeader.from_buffer_copy(blob[:hs].tobytes())
        rb = 4 * h.av_window
        ring = blob[hs:hs + rb].view(np.float32)
        p0, p1 = int(h.n_pending_bits[0]), int(h.n_pending_bits[1])
        pend = blob[hs + rb:hs + rb + p0 + p1]
        self.set_state(h, ring, [pend[:p0], pend[p0:]])

    def prime(self, start_index, level):
        """Speculative start for a time chunk that does not begin the stream: every ring slot at the
        estimated carrier level, idle state machine, decoders reset.  Pushing an overlap region that ends
        where the chunk starts then converges to the true boundary state (multi-GPU)."""
        self._chk(self.L.nfc_prime(self.h, int(start_index), float(np.float32(level))), 'nfc_prime')


class DeviceBuffer(object):
    """Input kept resident in HBM (nfc_device_alloc / nfc_device_upload) for NfcContext.push_device."""

    def __init__(self, host_array, device=0, nbytes=None):
        """nbytes: allocate that much (at least the array's size) -- room for a kernel to write into."""
        self.L = _lib.load()
        self.device = device
        a = np.ascontiguousarray(host_array)
        self.nbytes = max(a.nbytes, int(nbytes or 0))
        self.ptr = C.c_void_p()
        if self.L.nfc_device_alloc(device, self.nbytes, C.byref(self.ptr)) != 0:
            raise NfcError('nfc_device_alloc: %s' % self.L.nfc_last_error(None).decode())
        if a.nbytes and self.L.nfc_device_upload(device, self.ptr, a.ctypes.data, a.nbytes) != 0:
            raise NfcError('nfc_device_upload failed')

    def download(self, nbytes=None):
        """The buffer's first nbytes (default: all) as a uint8 array (nfc_device_download)."""
        n = self.nbytes if nbytes is None else int(nbytes)
        out = np.zeros(n, np.uint8)
        if n and self.L.nfc_device_download(self.device, out.ctypes.data, self.ptr, n) != 0:
            raise NfcError('nfc_device_download failed')
        return out

    def free(self):
        if getattr(self, 'ptr', None):
            self.L.nfc_device_free(self.device, self.ptr)
            self.ptr = None

    __del__ = free


class PinnedArray(object):
    """A numpy array over pinned host memory (nfc_host_alloc_pinned): the copy engine reads and writes it directly."""

    def __init__(self, count, dtype):
        self.L = _lib.load()
        self.ptr = C.c_void_p()
        dt = np.dtype(dtype)
        nbytes = max(16, int(count) * dt.itemsize)
        if self.L.nfc_host_alloc_pinned(nbytes, C.byref(self.ptr)) != 0:
            raise NfcError('nfc_host_alloc_pinned failed')
        self.array = np.frombuffer((C.c_char * nbytes).from_address(self.ptr.value), dtype=dt, count=int(count))

    def free(self):
        if getattr(self, 'ptr', None):
            self.array = None
            self.L.nfc_host_free_pinned(self.ptr)
            self.ptr = None

    __del__ = free


def device_count():
    """HIP devices this process sees (nfc_device_count; 0 without a GPU)."""
    return int(_lib.load().nfc_device_count())


def host_decode_lut(ptype, cur, d, samp_rate=2e6, max_len=50):
    """Drive the decode kernels' duration LUTs sequentially on the host (no GPU)."""
    L = _lib.load()
    p = _params(samp_rate, 0.1, 1.1, 2000, max_len, True, True, NFC_IN_IQ_F32, 0, 0.0, 0, 0)
    cur = np.ascontiguousarray(cur, np.int8)
    d = np.ascontiguousarray(d, np.int32)
    out = np.zeros(2 * len(cur) + 1, np.uint8)
    got = C.c_size_t(0)
    rc = L.nfc_host_decode_lut(C.byref(p), ptype, cur.ctypes.data, d.ctypes.data, len(cur), out.ctypes.data, out.size,
                               C.byref(got))
    if rc != 0:
        raise NfcError('nfc_host_decode_lut status %d' % rc)
    return out[:got.value]
