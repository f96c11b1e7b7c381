"""bench.py's launch path on CPU: `--gpus 2` without a launcher environment must start two ranks
itself (torch.distributed.run child job, gloo here) and print ONE JSON line with n_gpus == 2; with a
launcher environment whose world size disagrees with --gpus it must fail, never downgrade silently
(ref segmentation/dist_train.sh:8-9 is the launch this replaces)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    return env


def test_bench_gpus2_spawns_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--mock-step'], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['rccl_ranks'] == 2 and out['config']['parallelism'] == 'dp2'


def test_bench_refuses_world_mismatch():
    env = _env()
    env.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                        '--mock-step'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '--gpus 2' in r.stderr
