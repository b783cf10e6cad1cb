"""Build libnfc_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SO = os.path.join(HERE, 'libnfc_amd.so')
SO_HOOKS = os.path.join(HERE, 'libnfc_amd_hooks.so')   # the same sources with -DNFC_TEST_HOOKS: test hooks and diagnostics (README.md)
SOURCES = ['nfc_amd.hip']
DEPS = ['nfc_amd.hip', 'host_context.h', 'host_threshold.h', 'host_stages.h', 'host_submit.h', 'chunk_cut.h', 'launch_check.h', 'threshold.hip.h', 'threshold_lean.hip.h', 'threshold_wg.hip.h', 'edges.hip.h', 'decode.hip.h', 'scan.hip.h', 'small.hip.h', 'tail.hip.h', 'tx.hip.h', 'decoder_tables.h', 'protocol.h',
        os.path.join('..', '..', 'include', 'nfc_amd.h')]

FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
         '-ffp-contract=off', '-fno-fast-math',      # envelope and fp64 sums must round like the reference
         '-Wall', '-Wno-unused-function']


def stale(so=SO):
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False, hooks=False):
    """The product library; hooks=True: the test build (NFC_DEBUG_* / NFC_TRACE switches compiled in), which only tests and
    the profiling helpers under tools/ load (NFC_AMD_LIB or NfcContext(lib_path=...))."""
    so = SO_HOOKS if hooks else SO
    if not force and not stale(so):
        return so
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    extra = ['-DNFC_TEST_HOOKS'] if hooks else []
    cmd = [hipcc] + FLAGS + extra + os.environ.get('NFC_HIPCC_EXTRA', '').split() + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', so]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return so


def device_isa(out_path):
    """The device-side assembly of the same sources with the same flags (what tools/audit_lean_isa.py reads)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    flags = [f for f in FLAGS if f not in ('-fPIC', '-shared')]
    subprocess.check_call([hipcc] + flags + ['-S', '--cuda-device-only', os.path.join(CSRC, SOURCES[0]), '-o', out_path],
                          stderr=subprocess.DEVNULL)
    return out_path


def build_variant(out, extra):
    """An experiment: the same sources with extra compiler flags into another file (loaded through NFC_AMD_LIB)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    subprocess.check_call([hipcc] + FLAGS + list(extra) + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', out])
    return out


if __name__ == '__main__':
    if '--variant' in sys.argv:   # python build.py --variant out.so -DNFC_X=1 ...
        i = sys.argv.index('--variant')
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
        sys.exit(0)
    build(force='-f' in sys.argv, verbose=True)
    if '--hooks' in sys.argv:
        build(force='-f' in sys.argv, verbose=True, hooks=True)
