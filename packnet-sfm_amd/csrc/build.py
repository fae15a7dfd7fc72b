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


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv))
