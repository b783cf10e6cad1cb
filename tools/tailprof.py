"""Where the workgroups of the edge and decode kernels spend their time: s_memtime stamps per phase (a profiling build).

    NFC_HIPCC_EXTRA=-DNFC_TAIL_PROF python usrp_nfc_amd/build.py -f     # stamps compiled in (scan.hip.h: TP_MARK)
    python tools/tailprof.py miller 1e8                                 # on the GPU box
    python usrp_nfc_amd/build.py -f                                     # back to the product build

Prints, per kernel, the mean s_memtime ticks (shader clock) thread 0 of a workgroup spent in each phase, on bench.py's capture."""
import sys, os, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from usrp_nfc_amd import api, synth, _lib
wl = sys.argv[1] if len(sys.argv) > 1 else 'miller'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
import bench
_, iq = bench.make_capture_slice(wl, n, 0, 1)
L = _lib.load()
L.nfc_debug_tail_prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * (4 * 4096 * 8))()
fl = {'miller': dict(decode_tag=False), 'manchester': dict(decode_reader=False)}.get(wl, {})
with api.NfcContext(input_kind=api.NFC_IN_IQ_F32, **bench.stream_params(wl), **bench.decoder_flags(wl)) as ctx:
    buf = api.DeviceBuffer(iq)
    for _ in range(3):
        ctx.reset(); ctx.push_device(buf, n)
    L.nfc_debug_tail_prof(out, 1)
    reps = 1
    for _ in range(reps):
        ctx.reset(); ctx.push_device(buf, n)
    L.nfc_debug_tail_prof(out, 0)
    ctx.set_timing(2)
    ctx.reset(); ctx.push_device(buf, n)
    st = ctx.stats()
    print('stage ms: threshold %.4f edges %.4f decode %.4f total %.4f' % (st.ms_threshold, st.ms_edges, st.ms_decode, st.ms_total))
    print('edges', ctx.counts().n_edges)
names = ['k_write_edges', 'k_dec_spec (k_dec_reduce with NFC_DEC_SPEC=0)', 'k_dec_apply (NFC_DEC_SPEC=0)', 'k_concat (k_frame_write with NFC_DEC_SPEC=0)']
a = np.frombuffer(out, np.uint64).reshape(4, 4096, 8).astype(np.int64)
for k, nm in enumerate(names):
    v = a[k]
    v = v[v[:, 0] != 0]
    if not len(v):
        continue
    nst = int((v[0] != 0).sum())
    d = np.diff(v[:, :nst], axis=1)
    print('%-44s %5d WG; mean ticks per phase:' % (nm, len(v)), ' '.join('%8.0f' % x for x in d.mean(0)), ' total mean %.0f max %.0f' % (d.sum(1).mean(), d.sum(1).max()))
