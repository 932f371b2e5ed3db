"""-m gpu: the C++ module harness (tests/harness/harness.cpp) -- the counterpart of the reference's
test_blur_aware_tracker_modules, driving the SLAM::VO free functions with hipMalloc'd pointers."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_cxx_module_harness(mbavo):
    exe = os.path.join(HERE, "harness", "harness_bin")
    if not os.path.exists(exe):
        subprocess.run(["bash", os.path.join(HERE, "harness", "build.sh")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:]
    assert "HARNESS PASSED" in r.stdout


def test_device_solvers_against_host(mbavo):
    """One-wave Jacobi SVD / LDL^T of the device-side LM (lm_solvers.h) against host_math.cpp on random systems,
    n = 12 ... 78, full rank and rank deficient; tolerance 1e-8 relative on the solution."""
    exe = os.path.join(HERE, "harness", "solver_check_bin")
    if not os.path.exists(exe):
        subprocess.run(["bash", os.path.join(HERE, "harness", "build.sh")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0 and "SOLVER CHECK PASSED" in r.stdout, r.stdout[-2000:]


def test_cheap_division_is_the_ieee_division(mbavo):
    """pixel_math.h reciprocal() / quotient() (the runtime's fp64 division sequence minus range scaling and fix-up)
    give the bits of the compiler's IEEE division on 4M operand pairs drawn from the ranges they are used on and far
    beyond; tri_decode equals the row search for every packed index at ten matrix sizes."""
    exe = os.path.join(HERE, "harness", "div_check_bin")
    if not os.path.exists(exe):
        subprocess.run(["bash", os.path.join(HERE, "harness", "build.sh")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "DIV CHECK PASSED" in r.stdout, r.stdout[-2000:]
