#!/usr/bin/env python3
"""Compiles the REFERENCE's own device kernels - /root/reference/kernels/ocl_program.cl, where it lies and as it
stands - to gfx950 code objects under oracle/_ref/ (untracked, travels to the GPU box like the other built files).

Test infrastructure only (tests/_ocl_ref.py launches the kernels through the HIP module API and
tests/golden/make_ocl_golden.py turns their outputs into fixtures); nothing of the product links, loads or reads this.
Build container only: /root/reference does not exist on the GPU box.

The reference has no build step for this file: its host hands the text to the OpenCL driver at run time with the
configuration as -D macros (Evolutionary_Strategy_OpenCL.hpp:89-105, formatted by snprintf at :245-262).  Here the
image's clang (ROCm 7.2, OpenCL C 1.2, the AMD device libraries of /opt/rocm/amdgcn/bitcode) does what that driver
would, with the same macro list.  Two flavours per configuration:
  asrun  the macro values exactly as the reference's snprintf writes them: "%f" keeps six decimals, so the kernels see
         FFT_ONE_OVER_SIZE = 0.000977 for N = 1024 (1/1024 = 0.0009765625: +4.5e-4 relative), ROOT_TWO_OVER_PI = 0.797885,
         ONE_OVER_ALPHA = 0.714286 ...
  exact  the same expressions printed with nine significant digits (what the host computed before it formatted them)
Neither flavour passes -cl-fast-relaxed-math (the comment at :244 mentions it; the build call at :265 passes only the macros).
"""
import math
import os
import struct
import subprocess
import sys

REF_CL = "/root/reference/kernels/ocl_program.cl"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
CLANG = "/opt/rocm/lib/llvm/bin/clang"

# (tag, WRKGRPSIZE, D, log2 N, parents, offspring): small cases the CPU oracle finishes in seconds
CONFIGS = [
    ("2op_n1024_p512_wg32", 32, 4, 10, 128, 384),
    ("2op_n4096_p256_wg64", 64, 4, 12, 64, 192),
    ("3op_n1024_p256_wg32", 32, 6, 10, 64, 192),
    ("triple_n1024_p256_wg32", 32, 12, 10, 64, 192),
    # BASELINE configs[1]: the free-running loop of tests/test_gpu_statistical.py
    ("2op_n1024_p1024_wg32", 32, 4, 10, 256, 768),
    # BASELINE configs[2] at the reference's shipped workgroup size (parameters.json:31): only timed (tests/ref_kernels_time.py)
    ("2op_n1024_p65536_wg32", 32, 4, 10, 16384, 49152),
]


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


def macros(wg, d, log2n, parents, offspring, flavour):
    """The sixteen -D macros of Evolutionary_Strategy_OpenCL.hpp:89-105 in the reference's order, computed as its
    constructors do (Evolutionary_Strategy.hpp:611-627 for the ES constants, :296-317 for the FFT sizes)."""
    n = 1 << log2n
    p = parents + offspring
    alpha = f32(1.4)
    one_over_alpha = f32(1.0 / alpha)
    root_two_over_pi = f32(math.sqrt(f32(2.0 / f32(math.pi))))
    beta_scale = f32(1.0 / d)
    beta = f32(math.sqrt(beta_scale))
    fft_one_over_size = f32(1.0 / n)
    # Evolutionary_Strategy.hpp:308-317: a float accumulator over the double window, times 1/N, reciprocal in float
    acc = f32(0.0)
    for i in range(n):
        acc = f32(acc + (1.0 - math.cos(float(i) * (fft_one_over_size - 1) * 2.0 * math.pi)))
    fft_one_over_window_factor = f32(1.0 / f32(acc * fft_one_over_size))
    fmt = (lambda v: "%f" % v) if flavour == "asrun" else (lambda v: "%.9g" % v + ("" if "e" in "%.9g" % v or "." in "%.9g" % v else ".0"))
    return [
        ("WRKGRPSIZE", str(wg)), ("NUM_DIMENSIONS", str(d)), ("AUDIO_WAVE_FORM_SIZE", str(n)),
        ("POPULATION_COUNT", str(p)), ("POPULATION_SIZE", str(p * d)), ("NUM_WGS_FOR_PARENTS", str(parents // wg)),
        ("ALPHA", fmt(alpha)), ("ONE_OVER_ALPHA", fmt(one_over_alpha)), ("ROOT_TWO_OVER_PI", fmt(root_two_over_pi)),
        ("BETA_SCALE", fmt(beta_scale)), ("BETA", fmt(beta)), ("FFT_ONE_OVER_SIZE", fmt(fft_one_over_size)),
        ("FFT_ONE_OVER_WINDOW_FACTOR", fmt(fft_one_over_window_factor)),
        ("FFT_OUT_SIZE", str(n + 8)), ("FFT_HALF_SIZE", str(n // 2)), ("WAVETABLE_SIZE", "32768"),
    ]


def main():
    if not os.path.exists(REF_CL):
        print("build_ref_ocl: %s is not here (GPU box?): nothing to do" % REF_CL)
        return 0
    os.makedirs(OUT, exist_ok=True)
    for tag, wg, d, log2n, parents, offspring in CONFIGS:
        for flavour in ("asrun", "exact"):
            out = os.path.join(OUT, "ocl_%s_%s.co" % (tag, flavour))
            if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(REF_CL), os.path.getmtime(__file__)):
                continue
            cmd = [CLANG, "-x", "cl", "-cl-std=CL1.2", "-Xclang", "-finclude-default-header", "-target", "amdgcn-amd-amdhsa",
                   "-mcpu=gfx950", "--rocm-path=/opt/rocm", "-O2", "-w"]
            for k, v in macros(wg, d, log2n, parents, offspring, flavour):
                cmd += ["-D", "%s=%s" % (k, v)]
            cmd += [REF_CL, "-o", out]
            subprocess.check_call(cmd)
            print("built", os.path.relpath(out, os.path.dirname(HERE)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
