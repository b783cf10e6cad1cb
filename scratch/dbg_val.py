"""Per-sample classification of one golden fixture, GPU vs the C oracle: where do they first differ?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usrp_nfc_amd import api
from oracle import c_oracle as co
name = sys.argv[1] if len(sys.argv) > 1 else 'fx_ultralight_iq'
d = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', name + '.npz'), allow_pickle=True)
print(list(d.keys()))
iq = d['iq'].astype(np.float32) if 'iq' in d else None
x = iq if iq is not None else d['x'].astype(np.float32)
kind = api.NFC_IN_IQ_F32 if iq is not None else api.NFC_IN_ENV_F32
kw = dict(samp_rate=float(d['samp_rate']) if 'samp_rate' in d else 2e6, hi_val=float(d['hi_val']) if 'hi_val' in d else 1.1)
with api.NfcContext(input_kind=kind, device=0, **kw) as ctx:
    ctx.push(x)
    st = ctx.stats()
    print('chunks', st.n_chunks, 'C', st.chunk_samples, 'passes', st.threshold_passes, 'rerun', st.chunks_rerun)
    gv = ctx.val()
o = co.COracle(trace=True, **kw)
if iq is not None:
    o.push_iq(x)
else:
    o.push_env(x)
ov = np.concatenate([np.zeros(len(gv) - len(o.trace()), np.int8), o.trace()])   # (the tap starts at the first stable sample)
diff = np.nonzero(gv != ov)[0]
print('n', len(gv), 'differing samples', len(diff), 'first', diff[:20])
if len(diff):
    i = diff[0]
    print('gpu', gv[max(0, i - 8):i + 24].tolist())
    print('ref', ov[max(0, i - 8):i + 24].tolist())
    print('step', i // 256, 'row', (i % 256) // 64, 'lane', i % 64, 'chunk', i // st.chunk_samples, 'step in chunk', (i % st.chunk_samples) // 256)
if len(diff):
    s0 = (diff[0] // 256) * 256
    sym = {-1: 'L', 0: '.', 1: 'H'}
    for s in (s0 - 256, s0, s0 + 256):
        for r in range(4):
            a = ''.join(sym[int(v)] for v in gv[s + 64 * r:s + 64 * r + 64])
            b = ''.join(sym[int(v)] for v in ov[s + 64 * r:s + 64 * r + 64])
            print('step %d row %d gpu %s' % (s // 256, r, a))
            print('            ref %s%s' % (b, '' if a == b else '   <--'))
