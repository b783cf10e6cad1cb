"""fsm.process_bits of the reference (fsm.py:218-238) -- "next" rows f1 and f3 of SURVEY.md section 8.

`fsm(callback)` keeps the reference's constructor and `process_bits(bits, packet_type)` entry point, so
`background` / `CombinedPacketProcessor` hand packets to it unchanged; frame repair, parity, CRC_A, command lookup
and UID tracking run in the shared library (csrc/protocol.h).  `process_packets` is the batch form for the
packet tables a GPU batch produces.  MIFARE Classic sessions are decrypted (CRYPTO1: cipher.py, lfsr.py) with the
keys of set_keys (default FF..FF), nested authentications included; as in the reference, the ciphertext of every
frame of a session is printed before its decoded command."""
import ctypes as C
import sys

import numpy as np

from . import _lib
from .command import CommandStructure, CommandType

NFC_CMD_UNKNOWN, NFC_CMD_PARITY_ERROR = -1, -2
FRAME_EXTRA_ERROR, FRAME_MANY_MORE_ERROR, FRAME_UID_MISMATCH, FRAME_ENCRYPTED = 1, 2, 4, 8
FRAME_AR_OK, FRAME_AR_ERROR, FRAME_AT_OK, FRAME_AT_ERROR = 16, 32, 64, 128
FRAME_DTYPE = np.dtype([('cmd', '<i4'), ('type', '<i4'), ('byte_off', '<u4'), ('n_bytes', '<u2'), ('n_header', '<u2'),
                        ('n_extra', '<u2'), ('n_crc', '<u2'), ('flags', '<u4'), ('n_enc', '<u2'), ('pad', '<u2')])


def _messages(flags, out):
    if flags & FRAME_EXTRA_ERROR:
        out.write('EXTRA ERROR\n')
    if flags & FRAME_MANY_MORE_ERROR:
        out.write('MANY MORE ERROR\n')


def _structure(cmd_index, n_header, n_extra, data):
    cmd = CommandType.by_index(cmd_index)
    if cmd is None:
        return None, CommandStructure('UNKNOWN', [], data)
    return cmd, CommandStructure(cmd.name(), data[:n_header], data[n_header:n_header + n_extra], data[n_header + n_extra:])


class fsm(object):
    def __init__(self, callback=None, out=None):
        self.L = _lib.load()
        self._h = C.c_void_p()
        if self.L.nfc_fsm_create(C.byref(self._h)) != 0:
            raise RuntimeError('nfc_fsm_create failed')
        self._out = out or sys.stdout
        self._callback = callback if callback else self._display

    def __del__(self):
        if getattr(self, '_h', None):
            self.L.nfc_fsm_destroy(self._h)
            self._h = None

    def reset(self):
        self.L.nfc_fsm_reset(self._h)

    def _display(self, cmd, struct):
        struct.display(self._out)

    def set_keys(self, key_a=(0xFF,) * 6, key_b=(0xFF,) * 6):
        """Sector keys of a MIFARE Classic tag (fsm.set_keys, fsm.py:157-160)."""
        a = np.ascontiguousarray(key_a, np.uint8)
        b = np.ascontiguousarray(key_b, np.uint8)
        if a.size != 6 or b.size != 6 or self.L.nfc_fsm_set_keys(self._h, a.ctypes.data, b.ctypes.data) != 0:
            raise ValueError('keys are six bytes each')

    def _dispatch(self, f, data, enc=()):
        flags = int(f['flags'])
        _messages(flags, self._out)
        if flags & FRAME_ENCRYPTED:   # fsm._print_enc (fsm.py:113-131): what was on the air, '!' where the parity bit equals the data parity
            self._out.write(''.join('0x%02X%s ' % (int(e) & 0xFF, '!' if int(e) & 0x100 else '') for e in enc) + '\n')
        if int(f['cmd']) == NFC_CMD_PARITY_ERROR:
            self._out.write('PARITY ERROR\n')
            return None
        if flags & FRAME_UID_MISMATCH:
            self._out.write('MISMATCH BETWEEN READER-TAG UID\n')
        for bit, msg in ((FRAME_AR_OK, 'AR OK'), (FRAME_AR_ERROR, 'ERROR WITH AR'), (FRAME_AT_OK, 'AT OK'), (FRAME_AT_ERROR, 'ERROR WITH AT')):
            if flags & bit:
                self._out.write(msg + '\n')
        cmd, st = _structure(int(f['cmd']), int(f['n_header']), int(f['n_extra']), data)
        self._callback(cmd, st)
        return st

    def process_bits(self, bits, packet_type):
        """One closed packet (fsm.py:218).  Returns the CommandStructure, or None on a parity error."""
        b = np.ascontiguousarray(bits, np.uint8)
        frame = np.zeros(1, FRAME_DTYPE)
        data = np.zeros(b.size // 9 + 1, np.uint8)
        enc = np.zeros(b.size // 9 + 1, np.uint16)
        rc = self.L.nfc_fsm_process(self._h, b.ctypes.data, b.size, int(packet_type), frame.ctypes.data_as(C.POINTER(_lib.Frame)),
                                    data.ctypes.data, data.size, enc.ctypes.data)
        if rc != 0:
            raise ValueError('nfc_fsm_process status %d' % rc)
        f = frame[0]
        return self._dispatch(f, data[:int(f['n_bytes'])].tolist(), enc[:int(f['n_enc'])])

    def process_outgoing(self, bits, cmd):
        """A frame an emulator is about to send (fsm.process_outgoing, fsm.py:68-112; the encoder hook of packets.py:88-90):
        `bits` with parity as the encoders take them, `cmd` the CommandType it is.  Returns the bits to put on the air --
        encrypted while a MIFARE Classic session is up.  With an Ultralight tag the frame goes through process_bits, as in the
        reference ("update state": the callback sees it)."""
        b = np.ascontiguousarray(bits, np.uint8)
        out = np.zeros(max(1, b.size), np.uint8)
        rc = self.L.nfc_fsm_process_outgoing(self._h, b.ctypes.data if b.size else None, b.size, int(cmd.index), out.ctypes.data)
        if rc == 1:
            self.process_bits(bits, cmd.packet_type())
            return list(bits)
        if rc != 0:
            raise ValueError('nfc_fsm_process_outgoing status %d' % rc)
        return [int(v) for v in out[:b.size]]

    def process_packets(self, table, bits0, bits1, dispatch=True):
        """A batch: `table` rows of nfc_packet (both types, in stream order) over the per-type bit arrays.
        Returns (frames, bytes) as numpy arrays; the callback sees every command in order (dispatch=False: only the
        arrays -- a caller that consumes the frame table itself)."""
        t = np.ascontiguousarray(table)
        b0 = np.ascontiguousarray(bits0, np.uint8)
        b1 = np.ascontiguousarray(bits1, np.uint8)
        frames = np.zeros(len(t), FRAME_DTYPE)
        data = np.zeros(int(t['n_bits'].sum()) // 9 + len(t) + 1, np.uint8)
        enc = np.zeros(data.size, np.uint16)
        used = C.c_size_t(0)
        rc = self.L.nfc_fsm_process_packets(self._h, t.ctypes.data, len(t), b0.ctypes.data if b0.size else None,
                                            b1.ctypes.data if b1.size else None, frames.ctypes.data, data.ctypes.data, data.size,
                                            C.byref(used), enc.ctypes.data)
        if rc != 0:
            raise ValueError('nfc_fsm_process_packets status %d' % rc)
        for f in (frames if dispatch else ()):
            o = int(f['byte_off'])
            self._dispatch(f, data[o:o + int(f['n_bytes'])].tolist(), enc[o:o + int(f['n_enc'])])
        return frames, data[:used.value]


def crc_a(data):
    """ISO 14443-3 type A CRC as [low, high] (utilities.CRC.calculate_crc, utilities.py:30-41)."""
    L = _lib.load()
    b = np.ascontiguousarray(data, np.uint8)
    out = np.zeros(2, np.uint8)
    if L.nfc_crc_a(b.ctypes.data if b.size else None, b.size, out.ctypes.data) != 0:
        raise ValueError('nfc_crc_a failed')
    return [int(out[0]), int(out[1])]
