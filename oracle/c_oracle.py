"""ctypes binding of oracle/nfc_oracle.c (TEST INFRASTRUCTURE ONLY -- see the
header of that file).  Used by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, '_build', 'libnfc_oracle.so')


class Params(C.Structure):
    _fields_ = [('samp_rate', C.c_double), ('lo', C.c_double), ('hi', C.c_double),
                ('av_window', C.c_int32), ('max_len', C.c_int32), ('reader', C.c_int32), ('tag', C.c_int32)]


EDGE_DTYPE = np.dtype([('idx', '<i8'), ('d', '<i4'), ('v', 'i1'), ('t', 'i1'), ('pad', '<i2')])


def build():
    src = os.path.join(HERE, 'nfc_oracle.c')
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', HERE])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(Params), C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        for f in ('orc_push_env', 'orc_push_iq', 'orc_push_real_sq'):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        for f in ('orc_n_edges', 'orc_n_packets', 'orc_n_packet_bits', 'orc_n_trace'):
            getattr(L, f).restype = C.c_size_t
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_n_symbols.restype = C.c_size_t
        L.orc_n_symbols.argtypes = [C.c_void_p, C.c_int]
        L.orc_symbols.restype = C.c_void_p
        L.orc_symbols.argtypes = [C.c_void_p, C.c_int]
        for f in ('orc_edges', 'orc_packet_types', 'orc_packet_lens', 'orc_packet_bits', 'orc_trace'):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_total.restype = C.c_double
        L.orc_total.argtypes = [C.c_void_p]
        L.orc_nseen.restype = C.c_int64
        L.orc_nseen.argtypes = [C.c_void_p]
        L.orc_clear_outputs.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    dt = np.dtype(dtype)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=n).copy()


class COracle(object):
    def __init__(self, samp_rate=2e6, lo_val=0.1, hi_val=1.1, av_window=2000, max_len=50,
                 reader=True, tag=True, trace=False):
        self.L = lib()
        p = Params(samp_rate, lo_val, hi_val, av_window, max_len, int(reader), int(tag))
        self.h = self.L.orc_create(C.byref(p), int(trace))
        self.factor = 1e6 / samp_rate

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    __del__ = close

    def push_env(self, x):
        x = np.ascontiguousarray(x, np.float32)
        self.L.orc_push_env(self.h, x.ctypes.data, x.size)

    def push_iq(self, iq):
        iq = np.ascontiguousarray(iq, np.float32)
        self.L.orc_push_iq(self.h, iq.ctypes.data, iq.size // 2)

    def push_real_sq(self, x):
        x = np.ascontiguousarray(x, np.float32)
        self.L.orc_push_real_sq(self.h, x.ctypes.data, x.size)

    def clear_outputs(self):
        self.L.orc_clear_outputs(self.h)

    def edges(self):
        return _arr(self.L.orc_edges(self.h), self.L.orc_n_edges(self.h), EDGE_DTYPE)

    def transitions(self):
        e = self.edges()
        f = self.factor
        return [((int(v), int(d) * f), int(t)) for v, d, t in zip(e['v'], e['d'], e['t'])]

    def symbols(self, ptype):
        return _arr(self.L.orc_symbols(self.h, ptype), self.L.orc_n_symbols(self.h, ptype), np.uint8)

    def packets(self):
        n = self.L.orc_n_packets(self.h)
        types = _arr(self.L.orc_packet_types(self.h), n, np.int8)
        lens = _arr(self.L.orc_packet_lens(self.h), n, np.int32)
        bits = _arr(self.L.orc_packet_bits(self.h), self.L.orc_n_packet_bits(self.h), np.uint8)
        out, off = [], 0
        for t, k in zip(types, lens):
            out.append((int(t), bits[off:off + k].tolist()))
            off += k
        return out

    def packet_arrays(self):
        n = self.L.orc_n_packets(self.h)
        return (_arr(self.L.orc_packet_types(self.h), n, np.int8), _arr(self.L.orc_packet_lens(self.h), n, np.int32),
                _arr(self.L.orc_packet_bits(self.h), self.L.orc_n_packet_bits(self.h), np.uint8))

    def trace(self):
        return _arr(self.L.orc_trace(self.h), self.L.orc_n_trace(self.h), np.int8)

    def total(self):
        return self.L.orc_total(self.h)
