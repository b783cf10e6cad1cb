"""CPU restatement (pure Python) of the giech/usrp_nfc ISO-14443A eavesdrop hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``usrp_nfc_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

Parity status: PINNED.  ``tests/golden/make_golden.py`` drives the unmodified
reference modules (``transition_sink``, ``miller``, ``manchester``, ``packets``
imported from /root/reference/code with a stub ``gnuradio``/``fsm``) and the
committed fixtures under ``tests/golden/`` hold their outputs;
``tests/test_oracle_golden.py`` checks this file against every fixture.  The
GNU Radio blocks (``complex_to_mag_squared``, ``wavfile_source``) are third
party and absent: the envelope formula is restated from its published
definition and is *unpinned* at that boundary.

Each function cites the reference file:line it follows (paths relative to
/root/reference/code).
"""

import numpy as np

# ---------------------------------------------------------------------------
# constants -- utilities.py:7-23
# ---------------------------------------------------------------------------
ERR_NONE, ERR_TOO_SHORT, ERR_TOO_LONG, ERR_ENCODING, ERR_INTERNAL, ERR_WRONG_DUR, ERR_GENERAL = 0, 2, 3, 4, 5, 6, 7

T_FULL = 9.44                 # utilities.py:18
T_ZERO = 3.00                 # utilities.py:19
T_HALF = T_FULL / 2           # utilities.py:20
T_ZERO_REM = T_FULL - T_ZERO  # utilities.py:21
T_ONE_REM = T_HALF - T_ZERO   # utilities.py:22
T_ONE_HALF = T_FULL + T_HALF  # utilities.py:23

TAG_TO_READER, READER_TO_TAG = 0, 1   # packets.py:19-20


def start_bit_of(ptype):
    """packets.py:24-30"""
    if ptype == TAG_TO_READER:
        return 1
    if ptype == READER_TO_TAG:
        return 0
    raise ValueError('Unknown Packet Type', str(ptype))


# ---------------------------------------------------------------------------
# envelope -- gnuradio.blocks.complex_to_mag_squared, call sites
# decoder.py:26-28 and usrp_src.py:31.  fp32, one rounding per product and one
# for the sum, no FMA.
# ---------------------------------------------------------------------------
def envelope_iq(iq):
    """iq: float32 array of interleaved I,Q (len 2N) or complex64 (len N)."""
    a = np.asarray(iq)
    if a.dtype == np.complex64:
        a = a.view(np.float32)
    a = a.astype(np.float32, copy=False)
    i = a[0::2]
    q = a[1::2]
    return (i * i).astype(np.float32) + (q * q).astype(np.float32)


def envelope_real(x):
    """WAV branch, decoder.py:25-28: float_to_complex with Q unconnected (=0)
    then mag squared, i.e. fl(x*x)."""
    x = np.asarray(x, dtype=np.float32)
    return (x * x).astype(np.float32)


# ---------------------------------------------------------------------------
# transition_sink -- transition_sink.py:10-125
# ---------------------------------------------------------------------------
class TransitionSink(object):
    """Restates transition_sink.transition_sink.

    ``work(samples)`` has the GNU Radio sync-block contract used by the
    reference: it returns how many items were consumed (the fill phase may
    consume fewer than offered, transition_sink.py:116-125) and calls
    ``callback(list)`` once per stable-phase call (transition_sink.py:101).
    ``trace`` (optional list) receives the per-sample classification val
    (-1/0/+1) -- a debugging tap that the reference does not have.
    """

    def __init__(self, samp_rate, callback, lo_val=0.1, hi_val=1.1, av_window=2000, max_len=50, trace=None):
        # transition_sink.py:20-34
        self.max_len = max_len
        self.factor = 1e6 / samp_rate
        self.dur = 1
        self.last_bit = 0
        self.index = 0
        self.filled = 0
        self.length = av_window
        self.ring = [0] * av_window
        self.total = 0
        self.state = 0
        self.lo = lo_val
        self.hi = hi_val
        self.callback = callback
        self.stable = False
        self.trace = trace

    def work(self, samples):
        vals = np.asarray(samples, dtype=np.float32).tolist()   # :39 / :110
        if not self.stable:
            return self._fill(vals)
        return self._stable(vals)

    def _fill(self, vals):
        # transition_sink.py:109-125
        need = self.length - self.filled
        can = min(len(vals), need)
        self.ring[self.filled:self.filled + can] = vals[0:can]
        self.filled += can
        if can == need:
            self.total = sum(self.ring)                # :122 (sequential fp64; Python <= 3.11)
            self.dur = self.length % self.max_len      # :123
            self.stable = True                         # :124
        return can

    def _stable(self, vals):
        # transition_sink.py:37-107
        ring, length = self.ring, self.length
        index, state, total = self.index, self.state, self.total
        lo, hi, dur, last_bit = self.lo, self.hi, self.dur, self.last_bit
        mx, factor = self.max_len, self.factor
        out = []
        trace = self.trace
        for bit in vals:
            prev = ring[index]
            prev_state = state
            if total == 0:                             # :59-63
                ratio = 1 if bit == 0 else hi + 0.1
            else:
                ratio = bit * length / total           # :65
            if lo > ratio:                             # :67-70
                val, cur, state = -1, prev, 2
            elif state != 2 and ratio > hi:            # :71-74
                val, cur, state = 1, prev, 1
            else:                                      # :75-77
                val, cur = 0, bit
            ring[index] = cur                          # :80
            index = (index + 1) % length               # :81
            total += (cur - prev)                      # :82
            if trace is not None:
                trace.append(val)
            if val == last_bit:                        # :84-85
                dur += 1
            else:                                      # :86-92
                d = mx if prev_state == 0 else dur
                v = last_bit + 1 if state == 2 else last_bit
                out.append(((v, d * factor), state - 1))
                dur = 1
                last_bit = val
            if dur > mx:                               # :95-99
                v = last_bit + 1 if state == 2 else last_bit
                out.append(((v, mx * factor), state - 1))
                dur = 1
                state = 0
        self.callback(out)                             # :101
        self.index, self.state, self.total = index, state, total
        self.dur, self.last_bit = dur, last_bit
        return len(vals)


def drive_sink(sink, x, chunk=8192):
    """Feed ``x`` to ``sink.work`` the way the GNU Radio scheduler would:
    re-offer what a call did not consume (Appendix A of SURVEY.md)."""
    x = np.asarray(x, dtype=np.float32)
    i = 0
    n = len(x)
    if callable(chunk):
        nxt = chunk
    else:
        nxt = lambda: chunk
    while i < n:
        k = max(1, int(nxt()))
        i += sink.work(x[i:i + k])


# ---------------------------------------------------------------------------
# background router -- background.py:30-52 (run synchronously, no thread)
# ---------------------------------------------------------------------------
class Router(object):
    def __init__(self, reader_dec=None, tag_dec=None):
        self.reader_dec = reader_dec      # background.py:20
        self.tag_dec = tag_dec            # background.py:21

    def _dispatch(self, group, t):
        # background.py:30-35
        if t == TAG_TO_READER and self.tag_dec:
            self.tag_dec.process_transition(group)
        elif t == READER_TO_TAG and self.reader_dec:
            self.reader_dec.process_transition(group)

    def append(self, transitions):
        # background.py:41-52
        group = []
        cur = TAG_TO_READER
        for val, t in transitions:
            if t == cur:
                group.append(val)
            else:
                self._dispatch(group, cur)
                group = [val]
                cur = t
        if group:
            self._dispatch(group, cur)


# ---------------------------------------------------------------------------
# Modified Miller decoder -- miller.py:13-197
# ---------------------------------------------------------------------------
ST_BEGIN, ST_ZS0, ST_OS0, ST_OS1 = 0, 1, 2, 3     # miller.py:14-17


class MillerDecoder(object):
    def __init__(self, sink):
        self.sink = sink                       # object with append_bit(bit, type)
        self.prev = 0                          # :22
        self.tol = 1.5                         # :25
        self.lo = T_ZERO - self.tol            # :26
        self.hi = 2 * T_FULL                   # :27
        self._reset()

    # :31-41
    def stage(self):
        if self.dur0 == 0:
            return ST_BEGIN
        if self.cur_type == 0:
            return ST_ZS0
        return ST_OS0 if self.dur1 == 0 else ST_OS1

    # :43-59
    def set_stage(self, st):
        if st == ST_BEGIN:
            self.dur0, self.dur1, self.cur_type = 0, 0, 0
        elif st == ST_ZS0:
            self.dur0, self.cur_type = T_ZERO, 0
        elif st == ST_OS0:
            self.dur0, self.cur_type = T_HALF, 1
        elif st == ST_OS1:
            self.dur0, self.dur1, self.cur_type = T_HALF, T_ZERO, 1
        else:
            raise ValueError('Unknown Stage', str(st))

    def close(self, dur, av):                  # :62-63
        return abs(dur - av) <= self.tol

    def _reset(self):                          # :65-67
        self.started = False
        self.set_stage(ST_BEGIN)

    def _emit(self, sym):                      # :150-151
        self.sink.append_bit(sym, READER_TO_TAG)

    def _begin(self, cur, dur):                # :73-96
        r = []
        if cur == 0:
            if self.close(dur, T_ZERO):
                self.set_stage(ST_ZS0)
                self.started = True
            else:
                r.append(ERR_TOO_LONG)
        elif self.started:
            bit = 0
            if self.prev == 0:
                bit = ERR_ENCODING
            if self.close(dur, T_HALF):
                self.set_stage(ST_OS0)
            elif self.close(dur, T_FULL):
                r.append(bit)
            elif self.close(dur, T_ONE_HALF):
                r.append(bit)
                self.set_stage(ST_OS0)
            else:
                r.append(ERR_WRONG_DUR)
        return r

    def _zs0(self, cur, dur):                  # :98-112
        r = []
        if cur == 0:
            r.append(ERR_ENCODING)
        elif self.close(dur, T_ZERO_REM):
            self.set_stage(ST_BEGIN)
            r.append(0)
        elif self.close(dur, T_ZERO_REM + T_HALF):
            self.set_stage(ST_OS0)
            r.append(0)
        else:
            r.append(ERR_WRONG_DUR)
        return r

    def _os0(self, cur, dur):                  # :114-122
        r = []
        if cur != 0:
            r.append(ERR_ENCODING)
        elif not self.close(dur, T_ZERO):
            r.append(ERR_WRONG_DUR)
        else:
            self.set_stage(ST_OS1)
        return r

    def _os1(self, cur, dur):                  # :124-148
        r = []
        if cur != 1:
            r.append(ERR_ENCODING)
        elif self.close(dur, T_ONE_REM):
            r.append(1)
            self.set_stage(ST_BEGIN)
        else:
            r.append(1)
            self.set_stage(ST_BEGIN)
            dur -= T_ONE_REM
            if self.close(dur, T_FULL):
                r.append(0)
            elif self.close(dur, T_HALF):
                self.set_stage(ST_OS0)
            elif self.close(dur, T_ONE_HALF):
                r.append(0)
                self.set_stage(ST_OS0)
            else:
                r.append(ERR_WRONG_DUR)
        return r

    def process_transition(self, transitions):     # :153-197
        for cur, dur in transitions:
            if cur == 0 and abs(dur - T_ZERO) < T_ZERO / 2:     # :157-158
                dur = T_ZERO
            err = ERR_NONE
            st = self.stage()
            if (dur < self.lo or dur > self.hi) and (st == ST_ZS0 or st == ST_OS1):   # :165-167
                self._emit(self.cur_type)
                err = ERR_TOO_LONG
            elif dur < self.lo:
                err = ERR_TOO_SHORT
            elif dur > self.hi:
                err = ERR_TOO_LONG
            if err != ERR_NONE:                     # :173-176
                self._emit(err)
                self._reset()
                continue
            if st == ST_BEGIN:
                rets = self._begin(cur, dur)
            elif st == ST_ZS0:
                rets = self._zs0(cur, dur)
            elif st == ST_OS0:
                rets = self._os0(cur, dur)
            else:
                rets = self._os1(cur, dur)
            for s in rets:                          # :191-197
                self._emit(s)
                if s > 1:
                    self._reset()
                    self.prev = 0
                else:
                    self.prev = s


# ---------------------------------------------------------------------------
# Manchester decoder -- manchester.py:13-61
# ---------------------------------------------------------------------------
class ManchesterDecoder(object):
    def __init__(self, sink):
        self.sink = sink
        self.lo = T_HALF - 1          # :18
        self.mid = T_HALF + 1         # :19
        self.hi = 2 * T_HALF + 1      # :20
        self.prev_set = False         # :24
        self.prev = 0                 # :25

    def _emit(self, sym):             # :27-28
        self.sink.append_bit(sym, TAG_TO_READER)

    def process_transition(self, transitions):      # :30-61
        for cur, dur in transitions:
            err = ERR_NONE
            if dur < self.lo:
                err = ERR_TOO_SHORT
            elif dur > self.hi:
                err = ERR_TOO_LONG
            if err != ERR_NONE:                     # :40-43
                self.prev_set = False
                self.prev = 0
                self._emit(err)
                continue
            dual = dur > self.mid                   # :44
            prev = self.prev
            if self.prev_set:                       # :48-54
                if prev == cur or (prev != 0 and prev != 1):
                    self._emit(ERR_INTERNAL)
                    continue
                self._emit(int(prev))
                self.prev_set = dual
            else:                                   # :55-59
                if dual:
                    self._emit(ERR_ENCODING)
                    continue
                self.prev_set = True
            self.prev = cur                         # :61


# ---------------------------------------------------------------------------
# packet framing -- packets.py:57-98
# ---------------------------------------------------------------------------
class PacketProcessor(object):
    def __init__(self, ptype):
        self.ptype = ptype
        self.start_bit = start_bit_of(ptype)    # :60
        self.started = False
        self.cur = []

    def append_bit(self, bit):                  # :67-79
        if bit != 0 and bit != 1:
            if self.started:
                done = self.cur
                self.started = False
                self.cur = []
                return done
        else:
            if not self.started and bit == self.start_bit:
                self.started = True
            else:
                self.cur.append(bit)
        return None


class BitSink(object):
    """Stands where CombinedPacketProcessor stands (packets.py:83-98): records
    the raw symbol stream per type (what the decoders hand to ``append_bit``)
    and the closed packets in arrival order (what ``fsm.process_bits`` would
    receive, packets.py:96-98: empty lists are dropped)."""

    def __init__(self):
        self.procs = [PacketProcessor(TAG_TO_READER), PacketProcessor(READER_TO_TAG)]
        self.symbols = [[], []]
        self.packets = []

    def append_bit(self, bit, ptype):
        self.symbols[ptype].append(bit)
        done = self.procs[ptype].append_bit(bit)
        if done:
            self.packets.append((ptype, done))


# ---------------------------------------------------------------------------
# whole path
# ---------------------------------------------------------------------------
def run_path(x, samp_rate=2e6, hi_val=1.1, lo_val=0.1, av_window=2000, max_len=50,
             reader=True, tag=True, chunk=8192, want_trace=False):
    """Envelope samples -> transitions, symbol streams, packets.

    Wiring follows decoder.py:31-33 + background.py:17-21."""
    sink = BitSink()
    router = Router(MillerDecoder(sink) if reader else None,
                    ManchesterDecoder(sink) if tag else None)
    transitions = []

    def cb(lst):
        transitions.extend(lst)
        router.append(lst)

    trace = [] if want_trace else None
    ts = TransitionSink(samp_rate, cb, lo_val=lo_val, hi_val=hi_val,
                        av_window=av_window, max_len=max_len, trace=trace)
    drive_sink(ts, x, chunk)
    res = {
        'transitions': transitions,
        'symbols_tag': sink.symbols[TAG_TO_READER],
        'symbols_reader': sink.symbols[READER_TO_TAG],
        'packets': sink.packets,
    }
    if want_trace:
        res['trace'] = trace
    return res
