"""Drop-in for the reference's transition_sink module (transition_sink.py:10-125).

``transition_sink(samp_rate, callback, lo_val, hi_val, av_window, max_len)`` keeps the
reference constructor; ``work(input_items, output_items)`` keeps the GNU Radio sync-block
contract (returns the number of items consumed, calls ``callback(list)`` with
``((v, d*factor), t)`` entries in stream order).  The per-sample Python loop is replaced by
the HIP path behind the C-ABI (include/nfc_amd.h).

Samples are buffered and handed to the GPU in batches of ``batch`` samples (a GPU launch per
8192-sample scheduler call would waste the device); ``flush()`` pushes what is left -- GNU
Radio calls ``stop()`` at end of stream, which does that.  The concatenation of everything
handed to ``callback`` is identical to the reference's for any chunking of the input
(verified against the reference's chunk invariance, tests/test_shims.py).

When ``callback`` is ``background.append`` of this package's ``background``, the Miller /
Manchester decoders and the packet framing run on the GPU in the same context and their
results are delivered to that ``background`` after every batch.
"""
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
                 batch=1 << 22, device=0, input_kind=api.NFC_IN_ENV_F32):
        _Base.__init__(self, name="transition_sink", in_sig=[numpy.float32], out_sig=None)
        self._callback = callback
        self._batch = int(batch)
        self._buf = []
        self._nbuf = 0
        back = getattr(callback, '__self__', None)
        self._back = back if hasattr(back, '_deliver') else None
        reader = bool(self._back.reader) if self._back else False
        tag = bool(self._back.tag) if self._back else False
        self._want_list = self._back is None or self._back.transitions is not None
        self._ctx = api.NfcContext(samp_rate=samp_rate, lo_val=lo_val, hi_val=hi_val, av_window=av_window,
                                   max_len=max_len, reader=reader, tag=tag, input_kind=input_kind, device=device)

    # GNU Radio gateway contract (transition_sink.py:37-39, 107, 125)
    def work(self, input_items, output_items):
        a = numpy.asarray(input_items[0], dtype=numpy.float32)
        if a.size:
            self._buf.append(a.copy())
            self._nbuf += a.size
            if self._nbuf >= self._batch:
                self.flush()
        return int(a.size)

    def flush(self):
        if not self._nbuf:
            return
        x = self._buf[0] if len(self._buf) == 1 else numpy.concatenate(self._buf)
        self._buf, self._nbuf = [], 0
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
