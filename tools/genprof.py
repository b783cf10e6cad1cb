"""Where a wave of the general threshold kernel (k_threshold: the re-run kernel) spends its chunk, on the stress captures.

    NFC_HIPCC_EXTRA=-DNFC_GEN_PROF python usrp_nfc_amd/build.py -f
    python tools/genprof.py hover|dropsteps [0|1]      # second argument: NFC_CHUNK_ADAPT
    python usrp_nfc_amd/build.py -f
"""
import sys, os, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from usrp_nfc_amd import api, synth, _lib
kind = sys.argv[1] if len(sys.argv) > 1 else 'hover'
if len(sys.argv) > 2: os.environ['NFC_CHUNK_ADAPT'] = sys.argv[2]
n = 100_000_000
iq = synth.stress_workload(n, depth=0.08, sigma=0.002) if kind == 'dropsteps' else synth.stress_workload(n)
L = _lib.load()
L.nfc_debug_gen_prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * (8192 * 8))()
buf = api.DeviceBuffer(iq)
with api.NfcContext(samp_rate=2e6, hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
    for _ in range(3):
        ctx.reset(); ctx.push_device(buf, n)
    L.nfc_debug_gen_prof(out, 1)
    ctx.reset(); ctx.push_device(buf, n)
    L.nfc_debug_gen_prof(out, 0)
    st = ctx.stats()
    print('passes %d rerun %d of %d chunks (%d with failed rounds evaluated in place)' % (st.threshold_passes, st.chunks_rerun, st.n_chunks, st.chunks_rerun_in_place))
a = np.frombuffer(out, np.uint64).reshape(8192, 8).astype(np.int64)
v = a[a[:, 7] != 0]
print('chunks that ran the general kernel (last evaluation): %d' % len(v))
if not len(v):   # (round 6: up to a machine-full of failing chunks take k_threshold_wg<KIND, 4, true> -- k_threshold did not run)
    print('k_threshold ran no chunk of this batch (%d of its re-runs took the workgroup kernel with failed rounds evaluated in place)' % st.chunks_rerun_in_place)
    sys.exit(0)
print('mean ticks: incoming %.0f  loop %.0f (exact steps %.0f, of which the sum %.0f)  summary %.0f ; exact steps %.1f of %.0f' % (
    v[:, 0].mean(), v[:, 1].mean(), v[:, 2].mean(), v[:, 3].mean(), v[:, 5].mean(), v[:, 4].mean(), v[:, 6].mean()))
tot = v[:, 0] + v[:, 1] + v[:, 5]
k = tot.argmax()
print('slowest: total %d ticks: incoming %d loop %d exact %d sum %d exact steps %d of %d' % (tot[k], v[k, 0], v[k, 1], v[k, 2], v[k, 3], v[k, 4], v[k, 6]))
