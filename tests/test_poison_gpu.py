"""No kernel may read device memory nobody wrote: the whole path again with every new work buffer pre-filled with 0xA5 (SQ_POISON=1, hip/ctx.h)
instead of the zeros a fresh allocation usually holds — what a second process on the same GPU, or a long-running one, hands out."""
import json, os, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_results_do_not_depend_on_what_fresh_device_memory_holds(built):
    env = dict(os.environ, SQ_POISON="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "poison_check.py")], capture_output=True, text=True, env=env, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("POISON_RESULT ")]
    assert r.returncode == 0 and lines, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads(lines[-1][len("POISON_RESULT "):])
    assert all(res.values()), res
