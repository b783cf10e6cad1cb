"""One-off soak of nfc_submit_device / nfc_wait: a long stream in batches of random length, three in flight, against the oracle."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import c_oracle as co
from usrp_nfc_amd import api, synth

n = 40_000_000
iq = synth.workload('all', n)
o = co.COracle(trace=False, hi_val=1.1)
o.push_iq(iq)
want_tr, want_pk = o.transitions(), o.packets()
rng = np.random.default_rng(9)
for trial in range(3):
    cuts = [0]
    while cuts[-1] < n:
        z = int(rng.choice([rng.integers(270_000, 700_000), rng.integers(1, 200_000), rng.integers(1_000_000, 3_000_000)], p=[0.7, 0.1, 0.2]))
        cuts.append(min(n, (cuts[-1] + z + 1) // 2 * 2))
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
        big = api.DeviceBuffer(iq)
        tr, pk, ahead = [], [], 0
        nxt = 0
        nb = len(cuts) - 1
        depth = 2 + trial % 2
        t0 = time.perf_counter()
        for k in range(nb):
            while nxt < nb and nxt < k + depth:
                a, b = cuts[nxt], cuts[nxt + 1]
                ctx.submit_device(big.ptr.value + 8 * a, b - a)
                nxt += 1
            ctx.wait()
            tr += ctx.transitions()
            pk += ctx.packets()
            ahead += int(ctx.stats().ran_ahead)
        dt = time.perf_counter() - t0
        st = ctx.stats()
    ok = tr == want_tr and pk == want_pk
    print('trial %d: %d batches, depth %d, %d ran ahead, %d processed again, %.1f s, %s' % (trial, nb, depth, ahead, st.redone_total, dt, 'EXACT' if ok else 'DIFFERENT'))
    assert ok
