#!/usr/bin/env python3
"""Pins the oracle at a size where the GPU path's time chunks have their production length: the UNMODIFIED reference
(transition_sink, miller, manchester, packets -- imported as in make_golden.py) runs over the first 2 M samples of the three
2 Msps bench workloads and over 3 M samples of the 10 Msps Classic-1K capture; only SHA-256 digests and counts of what it
produced are committed (tests/golden/fx_workload_digests.json; tests/digests.py defines the byte strings).

Runs only in the build container (needs /root/reference):  python3 tests/golden/make_workload_digests.py"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as mg   # noqa: E402  (imports the reference)
from tests import digests   # noqa: E402
from usrp_nfc_amd import synth   # noqa: E402


def main():
    out = {}
    for name in digests.PINNED:
        iq, kw = digests.workload_prefix(name)
        x = synth.envelope_f32(iq)
        t0 = time.time()
        tr, sym_tag, sym_rd, pk = mg.run_reference(x, samp_rate=kw['samp_rate'], hi_val=kw['hi_val'], av_window=kw.get('av_window', 2000),
                                                   max_len=kw.get('max_len', 50), reader=kw['reader'], tag=kw['tag'], chunk=8192)
        d = digests.digest_result(tr, sym_tag, sym_rd, pk)
        d['samples'] = int(len(x))
        d['params'] = {k: kw[k] for k in sorted(kw)}
        out[name] = d
        print('%-11s %8d samples in %5.1f s: %7d transitions, %6d + %6d symbols, %5d packets' % (
            name, len(x), time.time() - t0, d['n_transitions'], d['n_sym_tag'], d['n_sym_reader'], d['n_packets']))
    with open(os.path.join(HERE, 'fx_workload_digests.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write('\n')


if __name__ == '__main__':
    main()
