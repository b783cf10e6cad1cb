"""Command table of the reference's protocol layer (command.py:11-118, 201-265) over the host C++ of
csrc/protocol.h -- "next" row f1 of SURVEY.md section 8: the names, accessors and the printed form are the
reference's (a `Command` per table row, `CommandType.REQA` ..., `CommandStructure.display()`), the table and
the lookup themselves live in the shared library (nfc_command_get, nfc_fsm_process)."""
import ctypes as C
import sys

from . import _lib
from .packets import PacketType   # noqa: F401  (re-exported as the reference's command module does)


class TagType:   # command.py:70-74
    ULTRALIGHT = 0
    CLASSIC1K = 1
    CLASSIC4K = 2
    DESFIRE = 3


class Command(object):
    """One row of the table; read-only view with the reference's accessor names (command.py:11-42)."""

    def __init__(self, index, info):
        self.index = index
        self._name = info.name.decode()
        self._stage = int(info.stage)
        self._init_bytes = [int(info.header[i]) for i in range(info.n_header)]
        self._packet_type = int(info.type)
        self._crc = bool(info.crc)
        self._num_extra_bytes = int(info.n_extra)
        self._xor = int(info.xor_check)

    def stage(self):
        return self._stage

    def name(self):
        return self._name

    def packet_type(self):
        return self._packet_type

    def needs_crc(self):
        return self._crc

    def num_extra_bytes(self):
        return self._num_extra_bytes

    def header(self):
        return list(self._init_bytes)

    def total_len(self):
        return len(self._init_bytes) + self._num_extra_bytes + (2 if self._crc else 0)

    def __repr__(self):
        return 'Command(%s)' % self._name


def _load_table():
    L = _lib.load()
    out = []
    for i in range(L.nfc_command_count()):
        info = _lib.CommandInfo()
        if L.nfc_command_get(i, C.byref(info)) != 0:
            raise RuntimeError('nfc_command_get(%d) failed' % i)
        out.append(Command(i, info))
    return out


class _CommandTypeMeta(type):
    """CommandType.REQA etc. resolve lazily (the shared library is only needed when the table is first used)."""
    _table = None
    _attr = ['REQA', 'WUPA', 'ATQAUL', 'ATQA1K', 'ATQA4K', 'ATQADS', 'ANTI1R', 'ANTI1U', 'ANTI1G', 'SEL1R', 'SEL1U', 'SEL1K',
             'ANTI2R', 'ANTI2T', 'AUTHA', 'AUTHB', 'RANDTA', 'RANDRB', 'RANDTB', 'SEL2R', 'SEL2T', 'READR', 'READT', 'HALT',
             'WRITE', 'COMPW1', 'COMPW2']   # table order of csrc/protocol.h (attribute names: command.py:78-118)

    def table(cls):
        if _CommandTypeMeta._table is None:
            _CommandTypeMeta._table = _load_table()
            assert len(_CommandTypeMeta._table) == len(_CommandTypeMeta._attr)
        return _CommandTypeMeta._table

    def __getattr__(cls, name):
        if name in _CommandTypeMeta._attr:
            return cls.table()[_CommandTypeMeta._attr.index(name)]
        raise AttributeError(name)


class CommandType(_CommandTypeMeta('CommandTypeBase', (object,), {})):
    @staticmethod
    def by_index(i):
        return CommandType.table()[i] if i >= 0 else None


class CommandStructure(object):   # command.py:204-265
    def __init__(self, name, header, extra=(), crc=()):
        self._name = name
        self._header = list(header)
        self._extra = list(extra)
        self._crc = list(crc)

    def name(self):
        return self._name

    def header(self):
        return self._header

    def set_extra(self, extra):
        self._extra = list(extra)

    def extra(self):
        return self._extra

    def crc(self):
        return self._crc

    def all_bytes(self):
        return self._header + self._extra + self._crc

    def text(self):
        """What display() prints (command.py:232-247), as the traces under the reference's outputs/ show it."""
        def line(tag, bs):
            return '%s: %s \n' % (tag, ' '.join('0x%02X' % b for b in bs))
        s = 'COMMAND: %s\n' % self._name
        if self._header:
            s += line('HEADER', self._header)
        if self._extra:
            s += line('EXTRA', self._extra)
        if self._crc:
            s += line('CRC', self._crc)
        return s + '\n\n'

    def display(self, out=None):
        (out or sys.stdout).write(self.text())
