"""Build libimpulse_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

Usage: python build.py [--force]
The library is built in-tree (csrc/libimpulse_hip.so) so it travels with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libimpulse_hip.so")
SOURCES = ["impulse_hip.hip", "minphase.hip", "curves.hip", "comm.hip"]
HEADERS = ["conv_kernels.hip.h", "fft_regs.hip.h", "ir_kernels.hip.h", "decay_kernels.hip.h",
           "slice_kernels.hip.h", "slice_host.hip.inc", "fft64.hip.h", "internal.h",
           os.path.join("..", "..", "include", "impulse_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS + [os.path.join("..", "build.py")]:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_variant(name, defines):
    """Diagnostic builds (tools/): csrc/libimpulse_hip_<name>.so with extra -D flags."""
    out = os.path.join(CSRC, f"libimpulse_hip_{name}.so")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value"]
    cmd += [f"-D{d}" for d in defines] + SOURCES + ["-ldl", "-o", out]
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError(f"hipcc failed building {out}")
    return out


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-value"] + SOURCES + ["-ldl", "-o", LIB + ".tmp"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building libimpulse_hip.so")
    if verbose:
        sys.stderr.write(res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
        sys.exit(0)
    print(build_library(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
