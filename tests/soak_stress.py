"""Soak: a stream that goes in and out of the regime that needs re-runs (stress captures spliced between clean ones), random batch
lengths, synchronous pushes and submitted batches mixed, against the C oracle."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import c_oracle as co
from usrp_nfc_amd import api, synth
seg = 8_000_000
parts = [synth.workload('all', seg), synth.stress_workload(seg, depth=0.08, sigma=0.002, every=400_000), synth.workload('miller', seg),
         synth.stress_workload(seg, every=500_000), synth.workload('manchester', seg), synth.stress_workload(seg, depth=0.08, sigma=0.002, step=1.0, every=300_000)]
iq = np.concatenate(parts)
n = len(iq) // 2
o = co.COracle(trace=False, hi_val=1.1)
o.push_iq(iq)
want_tr, want_pk = o.transitions(), o.packets()
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 21)
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    cuts = [0]
    while cuts[-1] < n:
        z = int(rng.choice([rng.integers(270_000, 900_000), rng.integers(1, 200_000), rng.integers(1_000_000, 4_000_000)], p=[0.6, 0.1, 0.3]))
        cuts.append(min(n, (cuts[-1] + z + 1) // 2 * 2))
    with api.NfcContext(hi_val=1.1, input_kind=api.NFC_IN_IQ_F32) as ctx:
        big = api.DeviceBuffer(iq)
        tr, pk, ahead, rer, rex, fine = [], [], 0, 0, 0, 0
        nb = len(cuts) - 1
        k = 0
        while k < nb:
            if rng.random() < 0.3:     # a synchronous push
                a, b = cuts[k], cuts[k + 1]
                ctx.push_device(big.ptr.value + 8 * a, b - a)
                tr += ctx.transitions(); pk += ctx.packets()
                st = ctx.stats(); rer += int(st.chunks_rerun); rex += int(st.chunks_rerun_in_place)
                k += 1
            else:                       # a run of submitted batches, up to three in flight
                run = min(nb - k, int(rng.integers(2, 7)))
                nxt = k
                for j in range(k, k + run):
                    while nxt < k + run and nxt < j + 3:
                        a, b = cuts[nxt], cuts[nxt + 1]
                        ctx.submit_device(big.ptr.value + 8 * a, b - a)
                        nxt += 1
                    ctx.wait()
                    tr += ctx.transitions(); pk += ctx.packets()
                    st = ctx.stats(); ahead += int(st.ran_ahead); rer += int(st.chunks_rerun); rex += int(st.chunks_rerun_in_place)
                k += run
        st = ctx.stats()
    ok = tr == want_tr and pk == want_pk
    print('trial %d: %d batches, %d ran ahead, %d processed again, %d chunks re-run (%d of them with failed rounds evaluated in place), %s' % (trial, nb, ahead, st.redone_total, rer, rex, 'EXACT' if ok else 'DIFFERENT'), flush=True)
    assert ok
