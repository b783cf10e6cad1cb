"""Which part of a speculated boundary state differs from the true one?  (classic1k at 10 Msps, miniature)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usrp_nfc_amd import api, sharding, synth, _lib
n_per = 700_000
params = dict(samp_rate=10e6, hi_val=1.1, av_window=10000, max_len=250)
gold = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', '1k_with_enc.out')
frames, _ = synth.frames_from_trace(gold)
m = synth.tiled_profile(synth.modulation_profile(frames, rate_msps=10.0, lead_in=0, tail=0), 3 * n_per)
m[:15000] = 1.0
iq = synth.iq_from_profile(m, seed=11)
ov_n = int(sys.argv[1]) if len(sys.argv) > 1 else sharding.shard_overlap(10e6, 10000)
a = api.NfcContext(input_kind=api.NFC_IN_IQ_F32, **params)
a.push(iq[:2 * n_per])
ha, ra, pa = a.get_state()
b = api.NfcContext(input_kind=api.NFC_IN_IQ_F32, **params)
lo = n_per
ov = iq[2 * (lo - ov_n):2 * lo]
b.prime(lo - ov_n, sharding.carrier_level(synth.envelope_f32(ov[:2 * 4096])))
b.push(ov)
hb, rb, pb = b.get_state()
for name, _ in _lib.StateHeader._fields_:
    va, vb = getattr(ha, name), getattr(hb, name)
    va = list(va) if hasattr(va, '__len__') else va
    vb = list(vb) if hasattr(vb, '__len__') else vb
    print('%-16s true %-28s spec %-28s %s' % (name, va, vb, '' if va == vb else '<-- differs'))
d = np.nonzero(ra != rb)[0]
print('ring slots differing: %d of %d' % (len(d), len(ra)), d[:10], (ra[d[:5]], rb[d[:5]]) if len(d) else '')
print('pending equal', [np.array_equal(x, y) for x, y in zip(pa, pb)])
