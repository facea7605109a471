#!/usr/bin/env python3
"""Timing experiments with parts of a kernel's work removed (gr_dtl_amd/csrc/dtl_ablate.h lists the variants).

  tools/ablate.py build 41 42 43        here (hipcc cross-compiles): gr_dtl_amd/csrc/_obj/ablate_<k>.so = the library with
                                        the variant's source compiled -DDTL_ABLATE=<k> (results wrong by construction)
  tools/ablate.py run "<probe command>" base 41 42 ...
                                        on the GPU box: the probe once per variant, the library chosen through DTL_HIP_LIB
                                        (gr_dtl_amd/_lib.py) -- the shipped libdtl_hip.so is NEVER replaced; "base" = shipped

The probe prints whatever it measures (tools/kernel_split.py: per-kernel device times of one receive step from HIP
events; tools/ldpc_probe.py: the decoder alone)."""
import os
import shlex
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gr_dtl_amd", "csrc")
OBJ = os.path.join(CSRC, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def source_of(k):
    return "ldpc_bp.hip" if 20 <= k < 30 else "rx_kernels.hip"


def build(kinds):
    subprocess.run(["bash", os.path.join(CSRC, "build.sh")], check=True, stdout=subprocess.DEVNULL)
    for k in kinds:
        src = source_of(k)
        obj = "/tmp/%s_ablate_%d.o" % (src, k)
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-result",
                        "-DDTL_ABLATE=%d" % k, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj], check=True)
        others = [os.path.join(OBJ, o) for o in sorted(os.listdir(OBJ)) if o.endswith(".o") and o != src + ".o"]
        out = os.path.join(OBJ, "ablate_%d.so" % k)
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out, obj] + others, check=True)
        print("built", out)


def run(probe, variants):
    for v in variants:
        env = dict(os.environ)
        if v != "base":
            lib = os.path.join(OBJ, "ablate_%s.so" % v)
            if not os.path.exists(lib):
                print("%-6s (not built: %s)" % (v, lib))
                continue
            env["DTL_HIP_LIB"] = lib
        else:
            env.pop("DTL_HIP_LIB", None)
        out = subprocess.run(shlex.split(probe), env=env, cwd=ROOT, capture_output=True, text=True)
        lines = [l for l in out.stdout.strip().split("\n") if l]
        print("%-6s %s" % (v, ("\n       ".join(lines)) if out.returncode == 0 else "FAILED: " + out.stderr[-300:]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "build":
        build([int(v) for v in sys.argv[2:]])
    elif len(sys.argv) >= 4 and sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3:])
    else:
        sys.exit(__doc__)
