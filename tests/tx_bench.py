#!/usr/bin/env python3
"""Row f4 measurement (kept under tests/: it uses the oracle as its checker): k_tx_render on a 1e8-sample TX stream (reader frames, Miller), with and without the carrier.
Algorithmic bytes: 8 B per complex64 sample written (the run table is a few hundred KB).  Prints one JSON line."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tx_oracle as txo   # the checker (a sample of the output is compared)
from usrp_nfc_amd import api, synth, tx

n_target = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
rate = 2e6
frames = [synth.frame_bits([0x26], 7), synth.frame_bits([0x93, 0x20]), synth.frame_bits([0x30, 0x04, 0x26, 0xEE])]
period = []
for b in frames:
    period += [(1, 150.0)] + tx.encode_bits(tx.NFC_TX_MILLER, b)   # the product's encoder builds the stream
per_samples = tx.sample_count(tx.as_runs(period), rate)
reps = n_target // per_samples
pulses = period * reps
runs = tx.as_runs(pulses)
n = tx.sample_count(runs, rate)
buf = api.DeviceBuffer(np.zeros(0, np.float32), 0, nbytes=8 * n + 64)
res = {}
for carrier in (False, True):
    ms = []
    for it in range(12):
        got, t = tx.render_device(runs, rate, buf.ptr, n, carrier=carrier, amp=0.5, timed=True)
        if it >= 2:
            ms.append(t)
    head = buf.download(8 * 200000).view(np.complex64)
    want = txo.render(pulses[:len(period) * (200000 // per_samples + 2)], rate)[:200000]
    if carrier:
        want = want * txo.carrier(len(want), rate, 13.56e6, 0.5)
        ok = bool(np.max(np.abs(head - want)) <= 1.5e-7)
    else:
        ok = bool(np.array_equal(head, want))
    t = float(np.mean(ms))
    res['carrier' if carrier else 'levels'] = dict(ms=t, GBps=8.0 * n / t / 1e6, frac_of_8TBps=8.0 * n / t / 1e6 / 8000.0, head_matches_oracle=ok)
print(json.dumps(dict(kernel='k_tx_render', samples=n, runs=int(runs.size), **res)))
