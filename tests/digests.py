"""SHA-256 digests of a decode result -- what the reference produced for a workload prefix too long to commit as vectors
(tests/golden/fx_workload_digests.json, written by tests/golden/make_workload_digests.py from the UNMODIFIED reference).

The byte strings hashed are canonical and independent of who produced the result: transitions as (v int8, duration float64
little-endian -- the reference's d * factor, so equality is float.hex() equality --, t int8) columns, symbol streams as uint8,
packets as (type int8, length int32 little-endian) columns followed by all bits as uint8."""
import hashlib

import numpy as np


def _h(*arrays):
    m = hashlib.sha256()
    for a in arrays:
        m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()


def digest_result(transitions, sym_tag, sym_reader, packets):
    """transitions: [((v, dur_us), t), ...]; sym_*: symbol lists; packets: [(type, [bits]), ...]."""
    tv = np.array([v for (v, d), t in transitions], np.int8)
    td = np.array([d for (v, d), t in transitions], '<f8')
    tt = np.array([t for (v, d), t in transitions], np.int8)
    return {
        'n_transitions': int(len(transitions)), 'n_sym_tag': int(len(sym_tag)), 'n_sym_reader': int(len(sym_reader)),
        'n_packets': int(len(packets)),
        'transitions': _h(tv, td, tt),
        'sym_tag': _h(np.asarray(sym_tag, np.uint8)),
        'sym_reader': _h(np.asarray(sym_reader, np.uint8)),
        'packets': _h(np.array([t for t, b in packets], np.int8), np.array([len(b) for t, b in packets], '<i4'),
                      np.array([bit for t, b in packets for bit in b], np.uint8)),
    }


# the prefixes that are pinned: name -> (generator, samples, stream parameters)
def workload_prefix(name):
    """-> (interleaved float32 IQ, keyword arguments of the path) of a pinned workload prefix."""
    from usrp_nfc_amd import synth
    if name in ('miller', 'manchester', 'all'):
        kw = dict(samp_rate=2e6, hi_val=1.1, reader=name in ('miller', 'all'), tag=name in ('manchester', 'all'))
        return synth.workload(name, 2_000_000), kw
    if name == 'classic1k':
        import bench
        _, own = bench.make_capture_slice('classic1k', 3_000_000, 0, 1)
        kw = dict(samp_rate=10e6, hi_val=1.1, av_window=10000, max_len=250, reader=True, tag=True)
        return own, kw
    raise ValueError(name)


PINNED = ('miller', 'manchester', 'all', 'classic1k')
