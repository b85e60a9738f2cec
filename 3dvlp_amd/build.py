"""Build recipe for libvlp3d_hip.so: plain hipcc, gfx950 only, no torch headers.

    python 3dvlp_amd/build.py [--force]

-ffp-contract=off: every fused multiply-add in the kernels is explicit (bit-exact index parity).
The .so is built IN-TREE (3dvlp_amd/csrc/libvlp3d_hip.so) so that it travels to the GPU box.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libvlp3d_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"] + \
    os.environ.get("VLP3D_EXTRA_HIPCC_FLAGS", "").split()  # experiments: e.g. -DVLP3D_GATHER_PREFETCH=1


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "vlp3d.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        procs.append((src, subprocess.Popen([HIPCC] + FLAGS + ["-c", src, "-o", obj])))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
