"""Build libpnsfm_hip.so (gfx950) from the .hip sources in this directory.

    python packnet-sfm_amd/csrc/build.py            # hipcc --offload-arch=gfx950, in-tree .so

hipcc cross-compiles without a GPU.  The .so stays in-tree (git-ignored, but it travels to the GPU box
with the gpurun snapshot).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "conv2d.hip", "conv2d_wgrad2.hip", "conv2d_wgrad3.hip", "conv2d_wgrad4.hip", "groupnorm.hip", "pack3d.hip", "elementwise.hip", "invdepth.hip", "loss.hip", "supervised.hip", "augment.hip", "nrs.hip", "sparse.hip", "calib.hip"]
LIB = os.path.join(HERE, "libpnsfm_hip.so")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=True):
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    # every header a source includes: editing any of them (conv2d_bx3.h holds the dominant kernel) must trigger a rebuild
    headers = sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h"))
    deps = srcs + headers + [os.path.join(HERE, "..", "..", "include", "pnsfm.h")]
    if not force and not _stale(LIB, deps):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _stale(o, [s] + deps[len(srcs):]):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
                   "-Wno-unused-result", "-Wno-unused-value", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


SEQ_SRC = os.path.join(HERE, "seq", "pnsfm_seq.cpp")
SEQ_LIB = os.path.abspath(os.path.join(HERE, "..", "packnet_sfm", "hip", "_pnsfm_seq.so"))


def build_seq(force=False, verbose=True):
    """The block sequencer (csrc/seq/pnsfm_seq.cpp): a host-only torch C++ extension (g++, no device code) that holds the bodies of
    the hot autograd nodes; it reaches the kernels through function pointers into libpnsfm_hip.so handed over at import time
    (packnet_sfm/hip/_seq.py), so it links against torch only.  In-tree, next to the Python it serves."""
    deps = [SEQ_SRC, os.path.join(HERE, "..", "..", "include", "pnsfm.h")]
    if not force and not _stale(SEQ_LIB, deps):
        return SEQ_LIB
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    libdirs = ce.library_paths()
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-DTORCH_EXTENSION_NAME=_pnsfm_seq",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           SEQ_SRC, "-o", SEQ_LIB, "-I" + sysconfig.get_paths()["include"]]
    cmd += ["-I" + i for i in ce.include_paths()] + ["-L" + d for d in libdirs]
    cmd += ["-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python"] + ["-Wl,-rpath," + d for d in libdirs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return SEQ_LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv))
    print(build_seq(force="--force" in sys.argv))
