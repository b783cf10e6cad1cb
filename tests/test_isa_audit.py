"""k_threshold_lean keeps the samples of the steps ahead in accumulator registers it names literally
(csrc/threshold_lean.hip.h).  That is sound only while the compiler itself never touches that register file: no vector
spills, no scratch, no access to a[..] outside the kernel's own asm statements.  tools/audit_lean_isa.py reads the
device assembly of the shipped sources (same flags as the build) and says so; a compiler or source change that breaks
the assumption fails here, on the CPU, before any GPU sees it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lean_kernel_isa_audit(tmp_path):
    from usrp_nfc_amd import build
    asm = build.device_isa(str(tmp_path / 'nfc_amd.s'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'audit_lean_isa.py'), asm], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if 'k_threshold_lean' in l]
    assert len(lines) >= 6 and all(l.endswith('0 findings') for l in lines), r.stdout   # every instantiation was looked at
    wg = [l for l in r.stdout.splitlines() if 'k_threshold_wg' in l]
    assert len(wg) == 10 and all(l.endswith('0 findings') for l in wg), r.stdout        # four rows per step for the four input kinds, eight for IQ and the envelope, the four re-run forms
    assert len(lines) == 8, r.stdout                                                    # the lean kernel: four kinds x two block widths
