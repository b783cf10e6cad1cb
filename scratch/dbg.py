import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from golden_util import Case
from usrp_nfc_amd import api
import numpy as np
c = Case('fx_stress_dropout')
for rep in range(4):
    ctx = api.NfcContext(input_kind=api.NFC_IN_ENV_F32, **c.params)
    ctx.push(c.x)
    e = ctx.edges()
    s1 = ctx.symbols(1); s0 = ctx.symbols(0)
    print(len(e), len(s0), len(s1), len(c.sym_tag), len(c.sym_reader))
    m = min(len(s1), len(c.sym_reader))
    d = np.nonzero(s1[:m] != c.sym_reader[:m])[0]
    print(' diffs', d[:6])
    ctx.close()
sel = e['t'] == 1
print(np.c_[np.nonzero(sel)[0], e['v'][sel], e['d'][sel]][:60].T)
print(s1[:50]); print(c.sym_reader[:50])
