"""CPU restatement of the reference's transmit side (row f4) -- TEST INFRASTRUCTURE, like everything under oracle/:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

  miller_encode / manchester_encode / same_encode   miller.py:200-233, manchester.py:64-79, binary_src.py:17-20
  render                                            binary_src.work, binary_src.py:64-103 (the runs it writes out)
  carrier                                           the arithmetic csrc/tx.hip.h states for multiplier.py:18-22
                                                    (GNU Radio's own NCO is third-party: parity unpinned there)
Pinned by tests/golden/fx_tx.json, which the unmodified reference encoders and binary_src.work produced
(tests/golden/make_tx_golden.py).
"""
import numpy as np

FULL = 9.44
ZERO = 3.00
HALF = FULL / 2
ZERO_REM = FULL - ZERO
ONE_REM = HALF - ZERO


def same_encode(bits):   # binary_src.py:17-20
    return [(b, FULL) for b in bits]


def manchester_encode(bits):   # manchester.py:64-79
    durs = [(1, HALF), (0, HALF)]
    last = 0
    for bit in bits:
        if bit == last:
            durs[-1] = (bit, FULL)
            last = 1 - last
            durs.append((last, HALF))
        else:
            durs.append((1 - last, HALF))
            durs.append((last, HALF))
    return durs


def miller_encode(bits):   # miller.py:200-233
    one = [(1, HALF), (0, ZERO), (1, ONE_REM)]
    zero0 = [(0, ZERO), (1, ZERO_REM)]
    zero1 = [(1, FULL)]
    durs = list(zero0)
    last_bit = 0
    for bit in list(bits) + [0]:
        cur = one
        if bit == 0:
            cur = zero0 if last_bit == 0 else zero1
        last_bit = bit
        lp, ld = durs[-1]
        if cur[0][0] == lp:
            durs[-1] = (lp, cur[0][1] + ld)
            durs.extend(cur[1:])
        else:
            durs.extend(cur)
    return durs


def render(pulses, samp_rate):
    """binary_src.work over a queue that is not refilled: int(dur * mult) samples per run (binary_src.py:83), the
    marker level 2 produces nothing.  complex64."""
    mult = samp_rate / 1e6
    parts = [np.full(int(d * mult), float(l), np.complex64) for l, d in pulses if l != 2]
    return np.concatenate(parts) if parts else np.zeros(0, np.complex64)


def carrier(n, samp_rate, freq, amp, first_index=0):
    turns = freq / samp_rate
    inc = int((turns - np.floor(turns)) * 18446744073709551616.0)
    k = np.arange(first_index, first_index + n, dtype=np.uint64)
    turn = ((k * np.uint64(inc)) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    ang = 2.0 * np.pi * turn.astype(np.float64)
    return (np.float32(amp) * (np.cos(ang) + 1j * np.sin(ang))).astype(np.complex64)
