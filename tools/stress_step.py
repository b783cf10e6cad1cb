import sys, time
sys.path.insert(0, '.')
import numpy as np
from usrp_nfc_amd import api, synth
import bench
nm = sys.argv[1] if len(sys.argv) > 1 else "stress_hover"
n = 100_000_000
kw = {'stress_dropouts': dict(depth=0.08, sigma=0.002, step=1.0), 'stress_dropouts_steps': dict(depth=0.08, sigma=0.002), 'stress_hover': dict()}[nm]
iq = synth.stress_workload(n, **kw)
buf = api.DeviceBuffer(iq)
with api.NfcContext(input_kind=api.NFC_IN_IQ_F32, **bench.stream_params('all'), **bench.decoder_flags('all')) as ctx:
    for k in range(5):
        ctx.reset(); ctx.sync()
        t0 = time.perf_counter()
        ctx.push_device(buf, n); ctx.sync()
        print(k, (time.perf_counter() - t0) * 1e3, 'ms', ctx.stats().threshold_passes, ctx.stats().chunks_rerun)
