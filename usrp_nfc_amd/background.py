"""Drop-in for the reference's background module (background.py:17-52).

The reference's ``background`` owns the Miller / Manchester decoders and the
CombinedPacketProcessor and is fed lists of transitions through ``append``.  Here
those three stages run on the GPU inside the same context as the threshold stage,
so ``background`` is the host-side sink of their results: every closed packet is
handed, in stream order, to ``fsm.process_bits(bits, packet_type)`` exactly as
packets.py:96-98 does.  There is no busy-spinning thread; delivery is synchronous
with ``transition_sink.work`` / ``flush``.

``background(reader, tag, emulator)`` keeps the reference signature.  ``fsm`` may
be passed explicitly (any object with ``process_bits``); by default the reference's
own ``fsm`` module is used when it is importable (it is Python 2), else this package's
``fsm`` (row f1: it prints the reference's command trace); packets are also kept in
``self.packets``.
"""
from .packets import PacketType


class _Recorder(object):
    def process_bits(self, bits, packet_type):
        pass


class background(object):
    def __init__(self, reader=False, tag=False, emulator=None, fsm=None, keep=None, samp_rate=2e6, max_len=50, device=0):
        """keep: how many packets / symbols of the stream stay in ``self.packets`` / ``self.symbols`` -- None: everything (offline
        decodes, tests), 0: nothing (the reference keeps nothing: a long live capture must not grow without bound), N: the last N.
        samp_rate, max_len: only for transition lists that did NOT come from this package's transition_sink (``append`` below):
        their durations are microseconds, the device decoders count samples (the reference's defaults, transition_sink.py:12)."""
        import collections
        self._attached = False       # this package's transition_sink feeds this object (it decodes on the device and delivers itself)
        self._foreign = None         # the context that decodes lists handed to append() by anybody else
        self._foreign_args = dict(samp_rate=float(samp_rate), max_len=int(max_len), device=int(device))
        self._n_foreign = 0
        self.reader = bool(reader)   # Modified-Miller decoder present (background.py:20)
        self.tag = bool(tag)         # Manchester decoder present      (background.py:21)
        self._keep = keep
        mk = (lambda: []) if keep is None else (lambda: collections.deque(maxlen=int(keep)))
        self.packets = mk()          # (packet_type, [bits]) in stream order
        self.symbols = {PacketType.TAG_TO_READER: mk(), PacketType.READER_TO_TAG: mk()}
        self.transitions = None      # set to a list to also keep the raw transitions
        if fsm is None:
            try:                      # packets.py:88-92
                import fsm as _ref_fsm
                if emulator:
                    fsm = _ref_fsm.fsm(emulator.process_packet)
                    emulator.set_encoder(fsm.process_outgoing)
                else:
                    fsm = _ref_fsm.fsm()
            except Exception:
                # the reference's fsm is Python 2: use this package's (rows f1 / f3: bytes, parity, CRC, commands, CRYPTO1
                # sessions; it prints the same trace) -- with the emulator's encoder hook set as packets.py:88-90 does
                from . import fsm as _own_fsm
                if emulator:
                    fsm = _own_fsm.fsm(emulator.process_packet)
                    if hasattr(emulator, 'set_encoder'):
                        emulator.set_encoder(fsm.process_outgoing)
                else:
                    fsm = _own_fsm.fsm()
        self._fsm = fsm

    # -- reference surface ---------------------------------------------------------
    def append(self, transitions):
        """background.py:27-28.  Fed by this package's ``transition_sink`` the decoders have already consumed these transitions on
        the device (the list is kept only when ``self.transitions`` is a list).  A list from ANY OTHER producer -- the reference's
        own ``transition_sink``, INTEGRATION.md's hybrid -- is what the reference's ``background.run`` would work through
        (background.py:37-52): it goes through the device's decode and framing stages alone (``nfc_push_edges``), synchronously,
        and the packets are delivered as usual.  Entries are ``((v, d * factor), t)`` with ``factor = 1e6 / samp_rate``
        (transition_sink.py:89-90): ``d = round(us / factor)`` is exact for every duration the sink can produce."""
        if self._attached or not transitions:
            if self.transitions is not None:
                self.transitions.extend(transitions)
            return
        import numpy as np
        from . import api
        factor = 1e6 / self._foreign_args['samp_rate']
        # A sink running at another rate or max_len than this object was constructed with would be decoded with the wrong
        # durations and time-outs: its durations are whole samples of ITS rate and at most ITS max_len, so check both.
        us = np.asarray([u for (_, u), _ in transitions], np.float64)
        d = np.rint(us / factor)
        if np.any(np.abs(us / factor - d) > 1e-6 * np.maximum(1.0, d)) or np.any(d < 0) or np.any(d > self._foreign_args['max_len']):
            raise ValueError('transition durations are not whole samples within max_len=%d at samp_rate=%g: construct background(..., '
                             'samp_rate=, max_len=) with the values of the transition_sink that produced them'
                             % (self._foreign_args['max_len'], self._foreign_args['samp_rate']))
        if self.transitions is not None:   # (kept only once the list has been accepted: a rejected call leaves nothing behind)
            self.transitions.extend(transitions)
        if self._foreign is None:
            a = self._foreign_args
            self._foreign = api.NfcContext(samp_rate=a['samp_rate'], max_len=a['max_len'], reader=self.reader, tag=self.tag, device=a['device'])
        e = np.zeros(len(transitions), api.EDGE_DTYPE)
        e['v'] = [v for (v, _), _ in transitions]
        e['d'] = d.astype(np.int64)
        e['t'] = [t for _, t in transitions]
        e['idx'] = np.arange(self._n_foreign, self._n_foreign + len(transitions), dtype=np.uint64)   # (only labels the packets)
        self._n_foreign += len(transitions)
        self._foreign.push_edges(e)
        self._deliver(self._foreign)

    def close(self):
        if getattr(self, '_foreign', None) is not None:
            self._foreign.close()
            self._foreign = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):   # (the device context of foreign lists must not outlive a caller who forgot close())
        try:
            self.close()
        except Exception:
            pass

    # -- GPU delivery (called by transition_sink after each batch) ---------------------
    def _deliver(self, ctx):
        if self._keep != 0:
            for t in (PacketType.TAG_TO_READER, PacketType.READER_TO_TAG):
                self.symbols[t].extend(ctx.symbols(t).tolist())
        for ptype, bits in ctx.packets():
            self.packets.append((ptype, bits))
            self._fsm.process_bits(bits, ptype)
