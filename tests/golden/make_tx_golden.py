#!/usr/bin/env python3
"""Golden vectors of the transmit side (row f4), produced by the UNMODIFIED reference: encode_bits of miller_encoder /
manchester_encoder / binary_src.encoder, and the samples binary_src.work writes for them.

Runs only in the build container (needs /root/reference).  ``gnuradio`` and ``fsm`` are stubbed as in make_golden.py.  binary_src's
pause arithmetic uses Python-2 integer division (binary_src.py:52,60); the vectors here use pause=0, which takes no
division, so the reference runs unmodified under Python 3.

    python3 tests/golden/make_tx_golden.py     # rewrites tests/golden/fx_tx.json
"""
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/code'

g = types.ModuleType('gnuradio')
gr = types.ModuleType('gnuradio.gr')


class sync_block(object):
    def __init__(self, name=None, in_sig=None, out_sig=None):
        pass


gr.sync_block = sync_block
bl = types.ModuleType('gnuradio.blocks')
g.gr = gr
g.blocks = bl
sys.modules.update({'gnuradio': g, 'gnuradio.gr': gr, 'gnuradio.blocks': bl})
f = types.ModuleType('fsm')   # packets.py imports it; its Python-2 prints do not parse here, and it is not on this path
f.fsm = type('fsm', (object,), {})
sys.modules['fsm'] = f
sys.path.insert(0, REF)
import binary_src as BS   # noqa: E402
from manchester import manchester_encoder   # noqa: E402
from miller import miller_encoder   # noqa: E402

def rle(x):
    """lossless run-length form of the sample stream: [[level, count], ...]"""
    out = []
    for v in x.tolist():
        if out and out[-1][0] == v:
            out[-1][1] += 1
        else:
            out.append([v, 1])
    return out


rng = np.random.default_rng(0xF4)
cases = []
for i in range(60):
    nb = int(rng.integers(0, 48)) if i else 0
    bits = rng.integers(0, 2, nb).tolist()
    rate = float(rng.choice([2e6, 4e6, 10e6, 13.56e6]))
    enc = ['same', 'manchester', 'miller'][i % 3]
    src = BS.binary_src(rate, encode=enc, idle_bit=int(rng.integers(0, 2)))
    runs = {'same': BS.encoder, 'manchester': manchester_encoder, 'miller': miller_encoder}[enc].encode_bits(bits)
    src.set_bits(bits)          # pause 0: [(2, 0)] + runs + [(2, 0)]
    buf = np.zeros(1 << 16, np.complex64)
    out = []
    for _ in range(4):          # marker, runs, marker, then the queue is empty (idle fill: not recorded)
        if not src._bits or src._index >= len(src._bits):
            break
        n = src.work(None, [buf])
        out.append(buf[:n].copy())
    samples = np.concatenate(out) if out else np.zeros(0, np.complex64)
    assert np.all(samples.imag == 0)
    cases.append(dict(encoding=enc, samp_rate=rate, bits=bits, runs=[[int(l), float(d).hex()] for l, d in runs],
                      samples_rle=rle(samples.real.astype(np.int8))))
json.dump(dict(note='reference encode_bits and binary_src.work outputs; durations as float.hex(); samples run-length coded', cases=cases),
          open(os.path.join(HERE, 'fx_tx.json'), 'w'))
print('wrote fx_tx.json:', len(cases), 'cases,', sum(sum(n for _, n in c['samples_rle']) for c in cases), 'samples')
