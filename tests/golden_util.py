"""Load the committed golden vectors (tests/golden/*.npz, written by make_golden.py)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

PATH_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'fx_*.npz'))
                    if os.path.basename(p) not in ('fx_decoder_vectors.npz', 'fx_ultralight_iq.npz', 'fx_nonfinite_iq.npz'))


class Case(object):
    def __init__(self, name, prefix='', file=None):
        z = np.load(os.path.join(GOLDEN, (file or name) + '.npz'))
        if prefix:   # several cases in one file (fx_nonfinite_iq.npz): keys carry the case's prefix
            z = {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}
        self.name = name
        self.x = z['x']
        p = z['params']
        self.params = dict(samp_rate=float(p[0]), lo_val=float(p[1]), hi_val=float(p[2]),
                           av_window=int(p[3]), max_len=int(p[4]), reader=bool(p[5]), tag=bool(p[6]))
        self.tr_v = z['tr_v']
        self.tr_us = z['tr_us']
        self.tr_t = z['tr_t']
        self.sym_tag = z['sym_tag']
        self.sym_reader = z['sym_reader']
        self.pk_type = z['pk_type']
        self.pk_len = z['pk_len']
        self.pk_bits = z['pk_bits']

    @property
    def transitions(self):
        return [((int(v), float(d)), int(t)) for v, d, t in zip(self.tr_v, self.tr_us, self.tr_t)]

    @property
    def packets(self):
        out, off = [], 0
        for t, n in zip(self.pk_type, self.pk_len):
            out.append((int(t), [int(b) for b in self.pk_bits[off:off + n]]))
            off += n
        return out


def load_json(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))
