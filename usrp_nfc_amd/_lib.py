"""ctypes loader for libnfc_amd.so -- the C-ABI of include/nfc_amd.h.

There is no CPU implementation behind this package: if the shared library is
missing it is built with hipcc (usrp_nfc_amd/build.py); if that fails, or no GPU
is usable when a context is created, the caller gets an exception."""
import ctypes as C
import os

import numpy as np

from . import build as _build

NFC_IN_IQ_F32, NFC_IN_ENV_F32, NFC_IN_REAL_F32_SQ, NFC_IN_I16_SQ = 0, 1, 2, 3
NFC_FLAG_FORCE_SEQUENTIAL, NFC_FLAG_NO_EDGES = 1, 2
ABI_VERSION = 4   # NFC_AMD_ABI_VERSION of the header these structures mirror


class Params(C.Structure):
    _fields_ = [('samp_rate', C.c_double), ('lo_val', C.c_double), ('hi_val', C.c_double),
                ('av_window', C.c_int32), ('max_len', C.c_int32), ('enable_reader', C.c_int32),
                ('enable_tag', C.c_int32), ('input_kind', C.c_int32), ('device', C.c_int32),
                ('i16_scale', C.c_float), ('flags', C.c_uint32), ('chunk_samples', C.c_int32),
                ('reserved', C.c_int32)]


class Counts(C.Structure):
    _fields_ = [('n_samples', C.c_uint64), ('n_edges', C.c_uint64), ('n_symbols', C.c_uint64 * 2),
                ('n_packets', C.c_uint64 * 2), ('n_packet_bits', C.c_uint64 * 2)]


class Stats(C.Structure):
    _fields_ = [('ms_total', C.c_double), ('ms_threshold', C.c_double), ('ms_edges', C.c_double),
                ('ms_decode', C.c_double), ('threshold_passes', C.c_uint32), ('chunks_rerun', C.c_uint32),
                ('used_sequential', C.c_uint32), ('n_chunks', C.c_uint32), ('bytes_in', C.c_uint64),
                ('ms_threshold_kernel', C.c_double * 6), ('n_threshold_timed', C.c_uint32), ('chunk_samples', C.c_uint32),
                ('ran_ahead', C.c_uint32), ('redone_total', C.c_uint32), ('ring_slots_carried', C.c_uint32), ('decode_respeculated', C.c_uint32),
                ('device_allocs', C.c_uint32), ('tail_fused', C.c_uint32), ('chunks_rerun_in_place', C.c_uint32)]


class Frame(C.Structure):   # nfc_frame
    _fields_ = [('cmd', C.c_int32), ('type', C.c_int32), ('byte_off', C.c_uint32), ('n_bytes', C.c_uint16),
                ('n_header', C.c_uint16), ('n_extra', C.c_uint16), ('n_crc', C.c_uint16), ('flags', C.c_uint32),
                ('n_enc', C.c_uint16), ('pad', C.c_uint16)]


class CommandInfo(C.Structure):   # nfc_command_info
    _fields_ = [('name', C.c_char * 8), ('stage', C.c_int32), ('type', C.c_int32), ('crc', C.c_int32),
                ('n_header', C.c_int32), ('n_extra', C.c_int32), ('xor_check', C.c_int32), ('header', C.c_uint8 * 2),
                ('pad', C.c_uint8 * 2)]


class StateHeader(C.Structure):
    _fields_ = [('n_seen', C.c_uint64), ('ss', C.c_double), ('last_low', C.c_int64), ('filled', C.c_int32),
                ('stable', C.c_int32), ('cur_state', C.c_int32), ('last_bit', C.c_int32), ('dur', C.c_int32),
                ('miller_state', C.c_int32), ('manch_state', C.c_int32), ('pkt_started', C.c_int32 * 2),
                ('n_pending_bits', C.c_uint32 * 2), ('av_window', C.c_int32), ('reserved', C.c_int32)]


TX_RUN_DTYPE = np.dtype([('level', '<i4'), ('pad', '<i4'), ('dur_us', '<f8')])   # nfc_tx_run
NFC_TX_SAME, NFC_TX_MANCHESTER, NFC_TX_MILLER = 0, 1, 2
EDGE_DTYPE = np.dtype([('idx', '<u8'), ('d', '<i4'), ('v', 'i1'), ('t', 'i1'), ('pad', '<i2')])
PACKET_DTYPE = np.dtype([('idx', '<u8'), ('bit_off', '<u8'), ('n_bits', '<u4'), ('type', '<i4')])

# every symbol include/nfc_amd.h declares
SYMBOLS = ['nfc_abi_version', 'nfc_device_count', 'nfc_create', 'nfc_destroy', 'nfc_last_error', 'nfc_push',
           'nfc_push_device', 'nfc_submit_device', 'nfc_wait', 'nfc_submitted', 'nfc_push_edges', 'nfc_sync', 'nfc_set_stream', 'nfc_get_counts', 'nfc_read_edges', 'nfc_read_edges_compact', 'nfc_read_symbols', 'nfc_read_packets',
           'nfc_read_packet_bits', 'nfc_read_val', 'nfc_get_state', 'nfc_set_state', 'nfc_reset', 'nfc_prime', 'nfc_export_state', 'nfc_get_stats', 'nfc_set_timing',
           'nfc_device_alloc', 'nfc_device_free', 'nfc_device_upload', 'nfc_device_download', 'nfc_stream_create', 'nfc_stream_destroy',
           'nfc_stream_sync', 'nfc_device_download_async', 'nfc_device_fill', 'nfc_host_alloc_pinned', 'nfc_host_free_pinned', 'nfc_host_decode_lut', 'nfc_host_miller_classes', 'nfc_host_decode_steps', 'nfc_host_i16_to_float', 'nfc_plan_row_cut',
           'nfc_fsm_create', 'nfc_fsm_destroy', 'nfc_fsm_reset', 'nfc_fsm_process', 'nfc_fsm_process_packets', 'nfc_fsm_process_outgoing', 'nfc_fsm_set_keys',
           'nfc_command_count', 'nfc_command_get', 'nfc_crc_a', 'nfc_tx_encode', 'nfc_tx_sample_count', 'nfc_tx_render_device']

_libs = {}


def lib_path():
    return _build.SO


def hooks_path():
    """The test build (-DNFC_TEST_HOOKS), built on demand: the NFC_DEBUG_* / NFC_TRACE switches only exist there."""
    return _build.build(hooks=True)


def load(path=None):
    """Load (building if needed) the shared library and declare its prototypes.  path: another build of the same ABI (the
    test build, a kernel experiment); default: NFC_AMD_LIB, else the in-tree product library."""
    path = path or os.environ.get('NFC_AMD_LIB') or _build.build()
    if path in _libs:
        return _libs[path]
    L = C.CDLL(path)
    vp, sz, psz = C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)
    L.nfc_abi_version.restype = C.c_int
    if L.nfc_abi_version() != ABI_VERSION:   # (the ctypes structures below are written for exactly this header)
        raise RuntimeError('%s has ABI version %d, this package binds version %d of include/nfc_amd.h' % (path, L.nfc_abi_version(), ABI_VERSION))
    L.nfc_device_count.restype = C.c_int
    L.nfc_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.nfc_destroy.argtypes = [vp]
    L.nfc_destroy.restype = None
    L.nfc_last_error.argtypes = [vp]
    L.nfc_last_error.restype = C.c_char_p
    L.nfc_push.argtypes = [vp, vp, sz]
    L.nfc_push_device.argtypes = [vp, vp, sz]
    L.nfc_submit_device.argtypes = [vp, vp, sz]
    L.nfc_wait.argtypes = [vp]
    L.nfc_submitted.argtypes = [vp]
    L.nfc_sync.argtypes = [vp]
    L.nfc_push_edges.argtypes = [vp, vp, sz]
    L.nfc_set_stream.argtypes = [vp, vp]
    L.nfc_get_counts.argtypes = [vp, C.POINTER(Counts)]
    L.nfc_read_edges.argtypes = [vp, sz, vp, sz, psz]
    L.nfc_read_edges_compact.argtypes = [vp, sz, vp, vp, sz, psz]
    L.nfc_read_symbols.argtypes = [vp, C.c_int, sz, vp, sz, psz]
    L.nfc_read_packets.argtypes = [vp, C.c_int, vp, sz, psz]
    L.nfc_read_packet_bits.argtypes = [vp, C.c_int, sz, vp, sz, psz]
    L.nfc_read_val.argtypes = [vp, sz, vp, sz, psz]
    L.nfc_get_state.argtypes = [vp, C.POINTER(StateHeader), vp, sz, vp, sz]
    L.nfc_set_state.argtypes = [vp, C.POINTER(StateHeader), vp, sz, vp, sz]
    L.nfc_reset.argtypes = [vp]
    L.nfc_export_state.argtypes = [vp, vp, sz, psz]
    L.nfc_sync.argtypes = [vp]
    L.nfc_prime.argtypes = [vp, C.c_uint64, C.c_float]
    L.nfc_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.nfc_set_timing.argtypes = [vp, C.c_int]
    L.nfc_device_alloc.argtypes = [C.c_int, sz, C.POINTER(vp)]
    L.nfc_device_free.argtypes = [C.c_int, vp]
    L.nfc_device_upload.argtypes = [C.c_int, vp, vp, sz]
    L.nfc_device_download.argtypes = [C.c_int, vp, vp, sz]
    L.nfc_stream_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.nfc_stream_destroy.argtypes = [C.c_int, vp]
    L.nfc_stream_sync.argtypes = [C.c_int, vp]
    L.nfc_device_download_async.argtypes = [C.c_int, vp, vp, sz, vp]
    L.nfc_device_fill.argtypes = [C.c_int, vp, C.c_int, sz]
    L.nfc_host_alloc_pinned.argtypes = [sz, C.POINTER(vp)]
    L.nfc_host_free_pinned.argtypes = [vp]
    L.nfc_host_decode_lut.argtypes = [C.POINTER(Params), C.c_int, vp, vp, sz, vp, sz, psz]
    L.nfc_host_miller_classes.argtypes = [C.POINTER(Params), vp, vp, C.POINTER(C.c_int)]
    L.nfc_host_decode_steps.argtypes = [C.c_int, vp, vp, sz, C.POINTER(C.c_int32), vp, sz, psz]
    L.nfc_host_i16_to_float.argtypes = [C.c_int16, C.c_float]
    L.nfc_host_i16_to_float.restype = C.c_float
    L.nfc_plan_row_cut.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.c_uint32, C.POINTER(C.c_uint32)]
    L.nfc_plan_row_cut.restype = C.c_int
    L.nfc_fsm_create.argtypes = [C.POINTER(vp)]
    L.nfc_fsm_destroy.argtypes = [vp]
    L.nfc_fsm_destroy.restype = None
    L.nfc_fsm_reset.argtypes = [vp]
    L.nfc_fsm_process.argtypes = [vp, vp, sz, C.c_int, C.POINTER(Frame), vp, sz, vp]
    L.nfc_fsm_process_packets.argtypes = [vp, vp, sz, vp, vp, vp, vp, sz, psz, vp]
    L.nfc_fsm_process_outgoing.argtypes = [vp, vp, sz, C.c_int, vp]
    L.nfc_fsm_set_keys.argtypes = [vp, vp, vp]
    L.nfc_command_count.restype = C.c_int
    L.nfc_command_get.argtypes = [C.c_int, C.POINTER(CommandInfo)]
    L.nfc_crc_a.argtypes = [vp, sz, vp]
    L.nfc_tx_encode.argtypes = [C.c_int, vp, sz, vp, sz, psz]
    L.nfc_tx_sample_count.argtypes = [vp, sz, C.c_double, C.POINTER(C.c_uint64)]
    L.nfc_tx_render_device.argtypes = [C.c_int, vp, sz, C.c_double, C.c_int, C.c_double, C.c_float, C.c_uint64, vp, sz, psz,
                                       C.POINTER(C.c_float)]
    for name in SYMBOLS:
        getattr(L, name)
    _libs[path] = L
    return L
