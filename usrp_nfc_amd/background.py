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
    def __init__(self, reader=False, tag=False, emulator=None, fsm=None, keep=None):
        """keep: how many packets / symbols of the stream stay in ``self.packets`` / ``self.symbols`` -- None: everything (offline
        decodes, tests), 0: nothing (the reference keeps nothing: a long live capture must not grow without bound), N: the last N."""
        import collections
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
                # sessions; it prints the same trace; the outgoing half for an emulator -- process_outgoing -- is out of scope)
                from . import fsm as _own_fsm
                fsm = _own_fsm.fsm(emulator.process_packet) if emulator else _own_fsm.fsm()
        self._fsm = fsm

    # -- reference surface ---------------------------------------------------------
    def append(self, transitions):
        """background.py:27-28.  With the GPU path the decoders have already consumed these
        transitions on the device; the list is kept only when ``self.transitions`` is a list."""
        if self.transitions is not None:
            self.transitions.extend(transitions)

    # -- GPU delivery (called by transition_sink after each batch) ---------------------
    def _deliver(self, ctx):
        if self._keep != 0:
            for t in (PacketType.TAG_TO_READER, PacketType.READER_TO_TAG):
                self.symbols[t].extend(ctx.symbols(t).tolist())
        for ptype, bits in ctx.packets():
            self.packets.append((ptype, bits))
            self._fsm.process_bits(bits, ptype)
