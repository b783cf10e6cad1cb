"""Where the workgroups of the fused tail (tail.hip.h: k_tail) spend their time: clock ticks per phase, summed over a workgroup's tiles.

    python usrp_nfc_amd/build.py --variant scratch/r6/tailprof.so -DNFC_TAIL_PROF
    NFC_AMD_LIB=scratch/r6/tailprof.so python tools/tailprof6.py miller 1e8          # on the GPU box
"""
import sys, os, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from usrp_nfc_amd import api, _lib
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'miller'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
_, iq = bench.make_capture_slice(wl, n, 0, 1)
L = _lib.load()
L.nfc_debug_tail_prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * (4 * 4096 * 8))()
with api.NfcContext(input_kind=api.NFC_IN_IQ_F32, **bench.stream_params(wl), **bench.decoder_flags(wl)) as ctx:
    buf = api.DeviceBuffer(iq)
    for _ in range(3):
        ctx.reset(); ctx.push_device(buf, n)
    L.nfc_debug_tail_prof(out, 1)
    L.nfc_debug_lb_tries(1)
    ctx.reset(); ctx.push_device(buf, n)
    L.nfc_debug_tail_prof(out, 0)
    tries = L.nfc_debug_lb_tries(0)
    ctx.set_timing(2)
    ctx.reset(); ctx.push_device(buf, n)
    st = ctx.stats()
    print('stage ms: threshold %.4f edges %.4f decode %.4f total %.4f' % (st.ms_threshold, st.ms_edges, st.ms_decode, st.ms_total))
    print('edges', ctx.counts().n_edges, ' look-back polls repeated (one batch):', tries)
a = np.frombuffer(out, np.uint64).reshape(4, 4096, 8).astype(np.int64)
v = np.concatenate([a[0], a[1]], axis=1)
v = v[v[:, 14] != 0]
names = ['E1 words+scans', 'LB edges', 'E2 walk', 'store entries', 'D compose+scan', 'LB maps', 'D walk', 'F1 scan+out', 'LB frame', 'F2 bits', 'tiles', '-', '-', '-', 'life', 'ticket']
print('%d workgroups; ticks summed over a workgroup\'s tiles (mean / max over workgroups), tiles per workgroup mean %.2f max %d' % (len(v), v[:, 10].mean(), v[:, 10].max()))
for k, nm in enumerate(names):
    if nm != '-':
        print('  %-16s mean %9.0f  max %9.0f  share of life %5.1f %%' % (nm, v[:, k].mean(), v[:, k].max(), 100.0 * v[:, k].sum() / max(1, v[:, 14].sum())))
