"""TEST INFRASTRUCTURE: compile the kernels of packnet-sfm_amd/csrc for the HOST with tests/emu/hipemu.h.

The resulting tests/emu/libpnsfm_emu.so exports the same C ABI as the product library but executes every
kernel on CPU fibers (see hipemu.h).  It lets `pytest -m "not gpu"` check kernel index math / tiling / MFMA
fragment mapping against the oracle in a container without a GPU.  It is never loaded by the product path
(packnet_sfm.hip._lib only accepts a library whose pnsfm_build_target() is "gfx950").
"""
import fcntl
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.abspath(os.path.join(HERE, "..", "..", "packnet-sfm_amd", "csrc"))
SOURCES = ["api.hip", "conv2d.hip", "conv2d_wgrad2.hip", "conv2d_wgrad3.hip", "conv2d_wgrad4.hip", "groupnorm.hip", "pack3d.hip", "elementwise.hip", "invdepth.hip", "loss.hip", "supervised.hip", "augment.hip", "nrs.hip", "sparse.hip", "calib.hip"]
LIB = os.path.join(HERE, "libpnsfm_emu.so")


def _fresh(deps):
    return os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps)


def build_emu(force=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))      # conv2d_bx3.h holds the dominant kernel
    deps = srcs + headers + [os.path.abspath(__file__), os.path.join(HERE, "hipemu.h"),
                             os.path.join(HERE, "..", "..", "include", "pnsfm.h")]
    if not force and _fresh(deps):
        return LIB
    # one builder at a time: the CPU suite runs on several pytest-xdist workers (tests/conftest.py), each of which gets here
    with open(os.path.join(HERE, ".build_emu.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and _fresh(deps):      # another worker built it while this one waited
            return LIB
        cxx = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
        objs, procs = [], []
        for s in srcs:
            o = os.path.join(HERE, os.path.basename(s)[:-4] + ".emu.o")
            objs.append(o)
            cmd = [cxx, "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-DPNSFM_EMU", "-I", HERE, "-I", CSRC,
                   "-Wno-unused-value", "-c", s, "-o", o]
            procs.append((cmd, subprocess.Popen(cmd)))
        for cmd, p in procs:
            if p.wait() != 0:
                raise RuntimeError("emu compile failed: " + " ".join(cmd))
        tmp = LIB + ".tmp%d" % os.getpid()
        subprocess.check_call([cxx, "-shared", "-fPIC", "-o", tmp] + objs)
        os.replace(tmp, LIB)                # a process that has the old file mapped keeps it
    return LIB


if __name__ == "__main__":
    print(build_emu(force=True))
