"""Drop-in for the reference's transition_sink module (transition_sink.py:10-125).

``transition_sink(samp_rate, callback, lo_val, hi_val, av_window, max_len)`` keeps the
reference constructor; ``work(input_items, output_items)`` keeps the GNU Radio sync-block
contract (returns the number of items consumed, calls ``callback(list)`` with
``((v, d*factor), t)`` entries in stream order).  The per-sample Python loop is replaced by
the HIP path behind the C-ABI (include/nfc_amd.h).

The reference calls ``callback`` once per ``work()`` (transition_sink.py:101).  Here samples are buffered
and handed to the GPU when ANY of these holds: ``batch`` samples are buffered (throughput: a launch per
8192-sample scheduler call would waste the device), ``flush_ms`` milliseconds have passed since the last
hand-over (latency of a live capture: default 50 ms, i.e. 100 000 samples at 2 Msps -- short batches
take the one-launch edge / decode path and cost ~0.1 ms), or ``flush_calls`` calls of ``work`` have been
buffered.  ``flush_ms=0`` / ``batch=1`` gives the reference's cadence exactly: one callback per
``work()``.  ``flush()`` pushes what is left -- GNU Radio calls ``stop()`` at end of stream, which does
that.  The concatenation of everything handed to ``callback`` is identical to the reference's for any
chunking of the input (verified against the reference's chunk invariance, tests/test_shims.py).

When ``callback`` is ``background.append`` of this package's ``background``, the Miller /
Manchester decoders and the packet framing run on the GPU in the same context and their
results are delivered to that ``background`` after every batch.
"""
import time

import numpy

from . import api

try:  # import-guarded GNU Radio integration (not installed in the build image)
    from gnuradio import gr as _gr
    _Base = _gr.sync_block
except Exception:  # pragma: no cover
    _gr = None

    class _Base(object):
        def __init__(self, name=None, in_sig=None, out_sig=None):
            pass


class transition_sink(_Base):
    "Transition sink"

    def __init__(self, samp_rate, callback, lo_val=0.1, hi_val=1.1, av_window=2000, max_len=50,
                 batch=1 << 22, device=0, input_kind=api.NFC_IN_ENV_F32, i16_scale=0.0, flush_ms=50.0, flush_calls=0):
        _Base.__init__(self, name="transition_sink", in_sig=[numpy.float32 if input_kind != api.NFC_IN_I16_SQ else numpy.int16], out_sig=None)
        self._callback = callback
        self._batch = max(1, int(batch))
        self._flush_s = float(flush_ms) * 1e-3 if flush_ms else 0.0
        self._flush_calls = int(flush_calls)
        self._buf = []
        self._nbuf = 0
        self._ncalls = 0
        self._t_last = time.monotonic()
        self._dtype = numpy.int16 if input_kind == api.NFC_IN_I16_SQ else numpy.float32
        back = getattr(callback, '__self__', None)
        self._back = back if hasattr(back, '_deliver') else None
        if self._back is not None:
            self._back._attached = True   # (its append() then only records: the packets come from this sink's own context)
        reader = bool(self._back.reader) if self._back else False
        tag = bool(self._back.tag) if self._back else False
        self._want_list = self._back is None or self._back.transitions is not None
        self._ctx = api.NfcContext(samp_rate=samp_rate, lo_val=lo_val, hi_val=hi_val, av_window=av_window,
                                   max_len=max_len, reader=reader, tag=tag, input_kind=input_kind, device=device,
                                   i16_scale=i16_scale)

    # GNU Radio gateway contract (transition_sink.py:37-39, 107, 125)
    def work(self, input_items, output_items):
        a = numpy.asarray(input_items[0], dtype=self._dtype)
        if a.size:
            self._buf.append(a.copy())
            self._nbuf += a.size
            self._ncalls += 1
            if (self._nbuf >= self._batch or (self._flush_calls and self._ncalls >= self._flush_calls)
                    or (self._flush_s and time.monotonic() - self._t_last >= self._flush_s)):
                self.flush()
        elif self._flush_s == 0.0 and self._batch == 1:
            self._callback([])   # the reference calls back on an empty call too (transition_sink.py:101)
        return int(a.size)

    def flush(self):
        self._t_last = time.monotonic()
        self._ncalls = 0
        if not self._nbuf:
            return
        x = self._buf[0] if len(self._buf) == 1 else numpy.concatenate(self._buf)
        self._buf, self._nbuf = [], 0
        self.push_now(x)

    def push_now(self, x):
        """One batch through the GPU path right away (decoder.run feeds whole pieces of a recording this way)."""
        self._ctx.push(x)
        # one callback per batch, even if empty (transition_sink.py:101)
        self._callback(self._ctx.transitions() if self._want_list else [])
        if self._back is not None:
            self._back._deliver(self._ctx)

    def stop(self):  # GNU Radio calls this when the flowgraph ends
        self.flush()
        return True

    def close(self):
        self.flush()
        self._ctx.close()
