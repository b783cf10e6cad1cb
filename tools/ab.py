#!/usr/bin/env python3
"""One A/B driver for the bench (replaces ab.sh, ab_args.sh, ab_c1k.sh, thr_ab.sh, step_ab.sh, variants.sh, wgab.sh, wgsweep.sh).

    tools/ab.py env   <workload> [--rounds N] [--bench "args"] "VAR=val VAR2=val" "VAR=..." ...   environment settings, same library
    tools/ab.py args  <workload> [--rounds N] "bench args" "bench args" ...                         bench arguments
    tools/ab.py lib   <workload> [--rounds N] [--bench "args"] lib1.so lib2.so ...                  builds of the library (NFC_AMD_LIB), alternating
    tools/ab.py kstat <workload> [--rounds N] [--bench "args"] lib1.so lib2.so ...                  ... per-kernel time per step under rocprofv3 (tools/kstats.sh)
    tools/ab.py stress <stress_hover|stress_dropouts_steps|stress_dropouts> "VAR=val ..." ...       a stress capture's steps (tools/stress_step.py)

Every variant is one `python3 bench.py --workload W --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras` (classic1k: 6 / 2
steps, parity kept); one line per run: ms per step, the threshold kernel's event-timed launch, roofline.frac, launches per step, the cut.
Variants ALTERNATE within a round (a box's clocks drift over a call): compare neighbours, not columns."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(workload, extra_args, env_add, label):
    c1k = workload == 'classic1k'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', workload, '--no-cpu-baseline']
    cmd += ['--steps', '6', '--warmup', '2', '--no-extras'] if c1k else ['--steps', '40', '--warmup', '10', '--no-parity', '--no-extras']
    cmd += extra_args
    env = dict(os.environ)
    env.update(env_add)
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env)
    for l in r.stdout.splitlines():
        if l.startswith('{'):
            d = json.loads(l)
            rf = d['roofline']
            par = d.get('parity', {}).get('edges_equal')
            print('%-34s step %.4f ms  launch %.4f ms  frac %.3f  launches/step %s  chunks %d x %d%s' % (
                label[:34], d['ms_per_step'], rf['avg_launch_ms'], rf['frac'], rf['launches_per_step'], d['config']['time_chunks'],
                d['config']['time_chunk_samples'], '' if par is None else '  parity %s' % par))
            sys.stdout.flush()
            return
    print('%-34s FAILED rc %d: %s' % (label[:34], r.returncode, (r.stderr or r.stdout)[-400:].replace('\n', ' | ')))
    sys.stdout.flush()


def parse_env(s):
    return dict(kv.split('=', 1) for kv in s.split() if '=' in kv)


def main():
    a = sys.argv[1:]
    if len(a) < 3:
        sys.exit(__doc__)
    mode, workload, rest = a[0], a[1], a[2:]
    rounds, bench_args = 1, []
    while rest and rest[0] in ('--rounds', '--bench'):
        if rest[0] == '--rounds':
            rounds = int(rest[1])
        else:
            bench_args = rest[1].split()
        rest = rest[2:]
    if mode == 'stress':
        for v in rest:
            print('== %s' % v)
            sys.stdout.flush()
            sys.path.insert(0, ROOT)
            from usrp_nfc_amd import _lib
            env = dict(os.environ, NFC_AMD_LIB=_lib.hooks_path())   # (the switches exist in the test build only)
            env.update(parse_env(v))
            subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'stress_step.py'), workload], cwd=ROOT, env=env)
        return
    for _ in range(rounds):
        for v in rest:
            if mode == 'env':   # (the switches exist in the test build only: usrp_nfc_amd/libnfc_amd_hooks.so)
                sys.path.insert(0, ROOT)
                from usrp_nfc_amd import _lib
                bench(workload, bench_args, dict({'NFC_AMD_LIB': _lib.hooks_path()}, **parse_env(v)), v)
            elif mode == 'args':
                bench(workload, v.split(), {}, v)
            elif mode == 'lib':
                bench(workload, bench_args, {'NFC_AMD_LIB': v}, os.path.basename(v))
            elif mode == 'kstat':
                print('== %s' % v)
                sys.stdout.flush()
                env = dict(os.environ, NFC_AMD_LIB=v, KSTATS_ARGS=' '.join(bench_args))
                r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'kstats.sh'), workload, 'ab'], capture_output=True, text=True, cwd=ROOT, env=env)
                for l in r.stdout.splitlines():
                    if not any(k in l for k in ('rocclr', 'set_state', 'RCCL', 'HIP ver', 'ROCm', 'Hostname', 'Librccl')):
                        print(l)
            else:
                sys.exit(__doc__)


if __name__ == '__main__':
    main()
