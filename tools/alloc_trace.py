#!/usr/bin/env python3
"""Which device buffers a stream (re)allocates, batch by batch (hooks build, NFC_TRACE_ALLOC): clean, clean, hovering, hovering."""
import os, sys
os.environ['NFC_TRACE_ALLOC'] = '1'
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from usrp_nfc_amd import _lib, api, synth
n_b = 1_500_000
clean = synth.workload('all', 3 * n_b)
hover = synth.stress_workload(3 * n_b, every=400_000)
iq = np.concatenate([clean[:2 * 2 * n_b], hover, clean[2 * 2 * n_b:]])
with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32, lib_path=_lib.hooks_path()) as ctx:
    bufs = [api.DeviceBuffer(iq[2 * k * n_b:2 * (k + 1) * n_b]) for k in range(6)]
    for k in range(5):
        sys.stderr.write('--- batch %d\n' % k)
        ctx.push_device(bufs[k], n_b)
        st = ctx.stats()
        sys.stderr.write('    allocs %d edges %d passes %d reruns %d chunks %d\n' % (st.device_allocs, ctx.counts().n_edges, st.threshold_passes, st.chunks_rerun, st.n_chunks))
