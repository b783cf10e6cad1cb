import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    # /dev/kfd is what the ROCm runtime opens; absent in the build container.
    return os.path.exists('/dev/kfd')


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def monkeypatch(monkeypatch):
    """The product library reads no environment variable; the switches that select a kernel form, misbehave on purpose or talk
    exist in the TEST build only (usrp_nfc_amd/libnfc_amd_hooks.so, -DNFC_TEST_HOOKS).  A test that sets one gets that build."""
    plain = monkeypatch.setenv

    def setenv(name, value, prepend=None):
        if name.startswith('NFC_') and name != 'NFC_AMD_LIB' and not name.startswith(('NFC_BENCH', 'NFC_TEST')):
            from usrp_nfc_amd import _lib
            plain('NFC_AMD_LIB', _lib.hooks_path())
        return plain(name, value, prepend)

    monkeypatch.setenv = setenv
    monkeypatch.setenv_plain = plain   # (a test that wants the PRODUCT library to see a switch -- and ignore it)
    return monkeypatch
