import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usrp_nfc_amd import api, synth
w = sys.argv[1] if len(sys.argv) > 1 else 'miller'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 2_000_000
timing = int(sys.argv[3]) if len(sys.argv) > 3 else 0
iq = synth.workload(w, n)
flags = dict(reader=w in ('miller', 'all'), tag=w in ('manchester', 'all'))
d = api.DeviceBuffer(iq, 0)
with api.NfcContext(samp_rate=2e6, hi_val=1.1, input_kind=api.NFC_IN_IQ_F32, device=0, **flags) as ctx:
    for k in range(4):
        ctx.set_timing(timing if k >= 2 else 0)
        ctx.push_device(d, n)
        st = ctx.stats()
        print('push', k, 'chunks', st.n_chunks, 'C', st.chunk_samples, 'passes', st.threshold_passes, 'rerun', st.chunks_rerun, 'seq', st.used_sequential,
              'kernel ms', [round(st.ms_threshold_kernel[i], 4) for i in range(st.n_threshold_timed)])
