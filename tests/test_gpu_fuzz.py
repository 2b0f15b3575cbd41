"""Randomised ME parity sweep (tools/me_fuzz.py): random sizes incl. partial SBs, five content kinds (smooth / sub-pel motion,
noise, flat, blocky = tie-heavy), every preset, both list counts, all temporal layers -- HIP == oracle, bit-exact."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [101, 202])
def test_me_random_sweep(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "me_fuzz.py"), "80", str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_lf_random_sweep():
    """tools/lf_fuzz.py: random sizes, masks, levels, sharpness 0..7, noisy and smooth (flat-filter) content"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lf_fuzz.py"), "60", "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_tq_rd_random_sweep():
    """tools/tq_fuzz.py: random plane sizes, quantiser steps from tiny (CAT6, full blocks) to huge (empty blocks), extreme
    residuals, inter / intra mixes -- recon, qcoeff, dqcoeff, eob, distortion pair and bits all equal the oracle's"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tq_fuzz.py"), "40", "9"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_mc_random_sweep():
    """tools/mc_fuzz.py: random sizes, motion-vector ranges up to far outside the picture, both use_subpel modes"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mc_fuzz.py"), "60", "4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
