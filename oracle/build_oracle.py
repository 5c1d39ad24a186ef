"""Build recipe for the parity oracle (TEST INFRASTRUCTURE ONLY - nothing here ships).

  oracle/liboracle.so        gcc build of the C restatements in oracle/*.c
  oracle/_ref/MLCodec_extensions_cpp.so
                              the REFERENCE rANS coder compiled from its own sources where they
                              lie (/root/reference/src/cpp/py_rans/*.cpp, pybind11 module) - only
                              when /root/reference is present (this container). Outputs go to
                              oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun).

Usage: python oracle/build_oracle.py [--force]
"""
import hashlib
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_RANS = "/root/reference/src/cpp/py_rans"
C_SOURCES = ["rans_oracle.c", "nn_oracle.c"]


def _digest(sources):
    h = hashlib.sha256()
    for s in sorted(sources):
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, sources):
    """Content-hash staleness (file times do not survive the copy to the GPU box)."""
    stamp = target + ".manifest"
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != _digest(sources)


def _mark(target, sources):
    with open(target + ".manifest", "w") as f:
        f.write(_digest(sources))


def build_liboracle(force=False):
    srcs = [os.path.join(HERE, s) for s in C_SOURCES if os.path.exists(os.path.join(HERE, s))]
    out = os.path.join(HERE, "liboracle.so")
    deps = srcs + [os.path.join(HERE, "wsilu_table.h")]
    if force or _stale(out, deps):
        # -ffp-contract=off: the oracle spells out every fused multiply-add with fmaf(); the
        # compiler must not invent or remove any (bit-exact arithmetic specification).
        cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", "-Wextra", "-o", out] + srcs + ["-lm"]
        subprocess.check_call(cmd)
        _mark(out, deps)
    return out


def build_ref(force=False):
    """Compile the reference coder itself; returns the path or None if the tree is absent."""
    out_dir = os.path.join(HERE, "_ref")
    out = os.path.join(out_dir, "MLCodec_extensions_cpp.so")
    if not os.path.isdir(REF_RANS):
        return out if os.path.exists(out) else None
    srcs = [os.path.join(REF_RANS, f) for f in ("rans.cpp", "py_rans.cpp", "bind.cpp")]
    if force or _stale(out, srcs):
        import pybind11
        os.makedirs(out_dir, exist_ok=True)
        cmd = ["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra",
               "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"],
               "-o", out] + srcs + ["-lpthread"]
        subprocess.check_call(cmd)
        _mark(out, srcs)
    return out


def main():
    force = "--force" in sys.argv
    print("liboracle:", build_liboracle(force))
    print("_ref     :", build_ref(force))


if __name__ == "__main__":
    main()
