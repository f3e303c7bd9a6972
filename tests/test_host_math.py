"""Host-side field arithmetic of the C ABI's CPU end (window combination, normalisation,
challenge inverses): compiled for the CPU and checked against itself, no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_inverse_matches_fermat(tmp_path):
    exe = tmp_path / "host_inverse_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__host__=", "-D__device__=",
                           "-I", os.path.join(ROOT, "bulletproofs_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "host_inverse_check.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


def test_host_window_combine_matches_xyzz_chain(tmp_path):
    """the Jacobian / plain-residue window combination against the XYZZ chain, both curves"""
    exe = tmp_path / "host_combine_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__host__=", "-D__device__=",
                           "-I", os.path.join(ROOT, "bulletproofs_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "host_combine_check.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
