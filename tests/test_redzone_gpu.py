"""Every device allocation of the library behind a checker (fast-livo2_amd/csrc/dev_alloc.hpp): a whole frame sequence must leave every guard intact
(LIVO2_REDZONE=1) and must not read or write a single byte outside its allocations (LIVO2_REDZONE=2: each allocation ends at unmapped address space,
3: starts at it).  VERDICT r03 "weak" #5: the library used to pad ensure() by 8 KB instead of proving it stays inside its buffers."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode):
    env = dict(os.environ, LIVO2_REDZONE=str(mode))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "redzone_frame.py")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "REDZONE DONE" in r.stdout
    return r.stdout


def test_frames_leave_every_guard_intact_and_the_checker_sees_a_stray_store():
    out = _run(1)
    assert "clean" in out and "REDZONE poke +0 reported" in out


def test_frames_stay_inside_fenced_allocations_end():
    _run(2)


def test_frames_stay_inside_fenced_allocations_start():
    _run(3)


def test_poke_is_refused_without_the_debug_allocator(ctx, livo2):
    assert ctx.redzone_check() == (0, 0)
    assert ctx.lib.livo2_debug_redzone_poke(ctx.h, 0) != 0
