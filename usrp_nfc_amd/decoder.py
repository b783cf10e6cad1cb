"""Drop-in for the reference's decoder module (decoder.py:15-33).

``decoder(src, dst, repeat, reader, tag, samp_rate, emulator)`` keeps the reference signature and
wiring: source -> envelope -> ``transition_sink(samp_rate, background.append, hi_val)`` with
``hi_val`` 1.1 for the UHD source (decoder.py:23) and 1.09 for a recording (decoder.py:29).

* With GNU Radio importable it is a ``gr.hier_block2`` built from the same blocks as the reference
  (``usrp_src`` / ``wavfile_source -> float_to_complex -> complex_to_mag_squared``), so
  ``usrp_nfc.py`` can ``self.connect(decoder(...))`` unchanged.
* Without GNU Radio (this image) ``src`` may be a 16-bit mono WAV path, a raw float32 file, a raw complex64 IQ
  file (``.fc32`` / ``.cfile`` / ``.iq``), or a numpy
  array, and ``run()`` streams it through the GPU path offline: a real recording is squared like the
  reference's WAV branch (float_to_complex with Q = 0, then |.|^2), complex64 / interleaved IQ takes
  the UHD branch's |IQ|^2.
"""
import wave

import numpy

from . import api
from .background import background
from .transition_sink import transition_sink

try:  # (GNU Radio is not in the build image: tests/test_gr_branch.py runs this branch against stand-ins)
    from gnuradio import blocks as _blocks
    from gnuradio import gr as _gr
except Exception:
    _gr = None
    _blocks = None


def _load_source(src, wav_scale):
    """-> (array, input_kind, i16_scale)"""
    if isinstance(src, numpy.ndarray):
        if src.dtype == numpy.complex64:
            return src.view(numpy.float32), api.NFC_IN_IQ_F32, 0.0
        if src.dtype == numpy.int16:
            return src, api.NFC_IN_I16_SQ, wav_scale
        return numpy.ascontiguousarray(src, numpy.float32), api.NFC_IN_REAL_F32_SQ, 0.0
    if str(src).lower().endswith('.wav'):
        w = wave.open(src, 'rb')
        try:
            if w.getnchannels() != 1 or w.getsampwidth() != 2:
                raise ValueError('expected a 16-bit mono WAV (usrp_src.py:36 writes those)')
            pcm = numpy.frombuffer(w.readframes(w.getnframes()), dtype='<i2')
        finally:
            w.close()
        return pcm, api.NFC_IN_I16_SQ, wav_scale
    if str(src).lower().endswith(('.fc32', '.cfile', '.iq', '.c64')):   # raw interleaved complex64, as a UHD / file sink writes
        return numpy.fromfile(src, dtype=numpy.float32), api.NFC_IN_IQ_F32, 0.0
    return numpy.fromfile(src, dtype=numpy.float32), api.NFC_IN_REAL_F32_SQ, 0.0


if _gr is not None:

    class decoder(_gr.hier_block2):
        def __init__(self, src="uhd", dst=None, repeat=False, reader=True, tag=True, samp_rate=2e6, emulator=None):
            _gr.hier_block2.__init__(self, "decoder", _gr.io_signature(0, 0, 0), _gr.io_signature(0, 0, 0))
            if src == "uhd":
                import usrp_src   # the reference's own UHD source block (usrp_src.py)
                self._src = usrp_src.usrp_src(samp_rate=samp_rate, dst=dst)
                hi_val = 1.1
            else:
                self._wav = _blocks.wavfile_source(src, repeat)
                self._r2c = _blocks.float_to_complex(1)
                self._src = _blocks.complex_to_mag_squared(1)
                self.connect(self._wav, self._r2c, self._src)
                hi_val = 1.09
            self._back = background(reader, tag, emulator)
            self._trans = transition_sink(samp_rate, self._back.append, hi_val=hi_val)
            self.connect(self._src, self._trans)

else:

    class decoder(object):
        def __init__(self, src="uhd", dst=None, repeat=False, reader=True, tag=True, samp_rate=2e6, emulator=None,
                     wav_scale=0.0, fsm=None, batch=1 << 22, device=0, lo_val=0.1, av_window=2000, max_len=50, keep=None):
            """wav_scale: int16 PCM -> float.  0 (default): GNU Radio's wavfile_source normalisation, sample / 32767 (what the
            reference's WAV branch feeds the path, decoder.py:25; third party, unpinned: nfc_amd.h); > 0: sample * wav_scale.
            lo_val / av_window / max_len: transition_sink's keyword arguments (transition_sink.py:12), e.g. scaled with the rate."""
            if isinstance(src, str) and src == "uhd":
                raise RuntimeError('the UHD source needs GNU Radio + UHD; pass a recording or an array')
            data, kind, scale = _load_source(src, wav_scale)
            self._data = data
            self._per = 2 if kind == api.NFC_IN_IQ_F32 else 1
            hi_val = 1.1 if kind == api.NFC_IN_IQ_F32 else 1.09   # decoder.py:23 / :29
            self._back = background(reader, tag, emulator, fsm=fsm, keep=keep)
            self._trans = transition_sink(samp_rate, self._back.append, lo_val=lo_val, hi_val=hi_val, av_window=av_window, max_len=max_len,
                                          batch=batch, device=device, input_kind=kind, i16_scale=scale if kind == api.NFC_IN_I16_SQ else 0.0)
            self._batch = int(batch)

        def run(self):
            """Stream the source through the path (what ``tb.run()`` does in usrp_nfc.py:170)."""
            step = self._batch * self._per
            for i in range(0, len(self._data), step):
                self._trans.push_now(self._data[i:i + step])
            return self._back

        @property
        def packets(self):
            return self._back.packets
