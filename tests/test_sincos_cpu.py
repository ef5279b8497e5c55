"""CPU check of the device's restricted-range sincos (physicl_amd/csrc/pcl_sincos.h): the very same source text
the HIP kernels compile is built with g++ and compared with long-double libm on the grid of angles the scatter
step can produce (u * 2*pi and u * pi, u = k / 2^53), around every multiple of pi/2 and at both ends of its
range.  Bar: < 1 ulp for sin and for cos (the reference leaves sin/cos to the OpenCL device's library; the parity
contract on scattered velocities is 4 ulp of c)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sincos_2pi_is_within_one_ulp_of_long_double_libm(tmp_path):
    exe = str(tmp_path / "sincos_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "physicl_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "sincos_check.cpp"), "-o", exe])
    out = subprocess.check_output([exe, "3000000"]).decode().split()
    n, max_sin, max_cos = int(out[0]), float(out[1]), float(out[2])
    assert n > 6_000_000
    assert max_sin < 1.0 and max_cos < 1.0, out
