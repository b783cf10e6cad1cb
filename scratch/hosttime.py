import sys, time; sys.path.insert(0,'.')
import numpy as np
from usrp_nfc_amd import api, synth
import bench
ov, own = bench.make_capture_slice('miller', 100_000_000, 0, 1)
d_own = api.DeviceBuffer(own, 0)
ctx = api.NfcContext(samp_rate=2e6, hi_val=1.1, input_kind=api.NFC_IN_IQ_F32, device=0, **bench.decoder_flags('miller'))
n = 100_000_000
for _ in range(5):
    ctx.reset(); ctx.push_device(d_own, n)
T = {'reset':0,'push':0,'stats':0}
K = 20
t00 = time.perf_counter()
for _ in range(K):
    t0 = time.perf_counter(); ctx.reset(); t1 = time.perf_counter(); ctx.push_device(d_own, n); t2 = time.perf_counter(); st = ctx.stats(); t3 = time.perf_counter()
    T['reset'] += t1-t0; T['push'] += t2-t1; T['stats'] += t3-t2
tot = time.perf_counter() - t00
print({k: round(v/K*1e6,1) for k,v in T.items()}, 'total/step us', round(tot/K*1e6,1), 'device ms', st.ms_total)
