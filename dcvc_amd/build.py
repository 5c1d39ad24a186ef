"""In-tree build of libdcvc_amd.so (HIP kernels for gfx950 + host codec + C ABI).

hipcc compiles every translation unit under dcvc_amd/csrc (``.hip`` = device + host,
``.cpp`` = host only) into objects under dcvc_amd/csrc/_obj and links
dcvc_amd/libdcvc_amd.so. No GPU is needed to build (hipcc cross-compiles gfx950).
Incremental: an object is rebuilt when its source or any header is newer.

Usage: python -m dcvc_amd.build [--force] [--verbose]
"""
import concurrent.futures
import glob
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
ARCH = "gfx950"
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "libdcvc_amd.so")

COMMON = [
    "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-inline-asm",
    # arithmetic policy: no fast-math, no implicit contraction - every fma is spelled fmaf()
    # so that the CPU oracle can reproduce the device arithmetic bit for bit.
    "-ffp-contract=off", "-fno-fast-math",
    # SLP packing of scalar fp32 math into v_pk_* costs more v_mov / s_nop than it saves on gfx950
    "-fno-slp-vectorize",
    "-I", os.path.join(ROOT, "include"), "-I", CSRC,
] + os.environ.get("DCVC_EXTRA_DEFS", "").split()      # tuning experiments: extra -D flags
HIP_FLAGS = ["--offload-arch=" + ARCH, "-munsafe-fp-atomics"]


def _sources():
    srcs = []
    for pat in ("*.cpp", "*.hip", "*/*.cpp", "*/*.hip"):
        srcs += glob.glob(os.path.join(CSRC, pat))
    return sorted(s for s in srcs if os.sep + "_obj" not in s and os.sep + "cli" + os.sep not in s)


def _headers():
    hs = glob.glob(os.path.join(ROOT, "include", "*.h"))
    for pat in ("*.h", "*/*.h", "*.hpp", "*/*.hpp"):
        hs += glob.glob(os.path.join(CSRC, pat))
    return hs


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _host_cxx():
    """Compiler for the host-only translation units (.cpp): the clang++ that belongs to the hipcc in use - its sibling
    ../lib/llvm/bin/clang++, or $ROCM_PATH - so that a ROCm outside /opt/rocm builds too. DCVC_HOST_CXX overrides (NOT
    the generic $CXX: a conda or distro g++ exported there would silently replace the compiler); g++ is the last resort
    and gets the GNU flag set (_host_flags)."""
    cands = [os.environ.get("DCVC_HOST_CXX")]
    hipcc = shutil.which(_hipcc()) or _hipcc()
    if os.path.isabs(hipcc):
        root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        cands += [os.path.join(root, "lib", "llvm", "bin", "clang++"), os.path.join(root, "llvm", "bin", "clang++")]
    if os.environ.get("ROCM_PATH"):
        cands.append(os.path.join(os.environ["ROCM_PATH"], "lib", "llvm", "bin", "clang++"))
    cands += ["/opt/rocm/lib/llvm/bin/clang++", "g++"]
    for cand in cands:
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand) and shutil.which(cand)):
            return cand
    return "g++"


_CLANG_ONLY = ("-fno-slp-vectorize", "-Wno-inline-asm")


def _is_gnu(cxx):
    try:
        out = subprocess.run([cxx, "--version"], capture_output=True, text=True, timeout=20).stdout.lower()
    except (OSError, subprocess.SubprocessError):
        return False
    return "clang" not in out and ("g++" in out or "gcc" in out or "free software foundation" in out)


def _host_flags(cxx):
    """COMMON for a host compiler: a GNU compiler rejects the clang-only switches ('unrecognized command-line option')"""
    if _is_gnu(cxx):
        return [f for f in COMMON if f not in _CLANG_ONLY] + ["-fno-tree-slp-vectorize"]
    return list(COMMON)


def _compile(src, obj, verbose):
    # ``.hip`` = anything that includes the HIP runtime (kernels and the host codec);
    # ``.cpp`` = plain host C++ (the rANS coder), still compiled by hipcc's clang.
    if src.endswith(".hip"):
        cmd = [_hipcc(), "-c"] + COMMON + ["-x", "hip"] + HIP_FLAGS
    else:
        # host-only C++ straight through clang++ (the hipcc wrapper would compile a .cpp as HIP, device pass included:
        # x86 target attributes and builtins - the AVX-512 path of the rANS decoder - do not exist there)
        cxx = _host_cxx()
        cmd = [cxx, "-c"] + _host_flags(cxx)
    cmd += ["-o", obj, src]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("compile failed: %s\n%s\n%s" % (src, res.stdout, res.stderr))
    if verbose and res.stderr.strip():
        print(res.stderr)
    return obj


def _digest(paths):
    # flags with the checkout location factored out: the tree is copied to another path on the GPU box
    # (the host compiler is part of the digest: a library built with another one is not "current")
    h = hashlib.sha256(" ".join(COMMON + HIP_FLAGS + [os.path.basename(_host_cxx())]).replace(ROOT, "<root>").encode())
    for p in sorted(paths):
        h.update(os.path.relpath(p, ROOT).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Build libdcvc_amd.so unless it is already current. "Current" is decided by a content hash
    of every source/header (file times do not survive the copy to the GPU box)."""
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    manifest = LIB + ".manifest"
    digest = _digest(srcs + _headers() + [CLI_SRC])
    if not force and os.path.exists(LIB) and os.path.exists(manifest) and os.path.exists(CLI_BIN):
        with open(manifest) as f:
            if f.read().strip() == digest:
                return LIB
    # several ranks of one node may get here at once (bench.py --gpus N on a box whose library is stale): one builds,
    # the others wait for the lock and find the library current
    import fcntl
    lock = open(LIB + ".lock", "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        return _build_locked(force, verbose, srcs, manifest, digest)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(force, verbose, srcs, manifest, digest):
    if not force and os.path.exists(LIB) and os.path.exists(manifest) and os.path.exists(CLI_BIN):
        with open(manifest) as f:
            if f.read().strip() == digest:
                return LIB
    hdr_digest = _digest(_headers())
    jobs = []
    objs = []
    for s in srcs:
        rel = os.path.relpath(s, CSRC).replace(os.sep, "__")
        o = os.path.join(OBJ, rel + ".o")
        objs.append(o)
        stamp = o + ".stamp"
        want = hdr_digest + _digest([s])
        have = open(stamp).read() if os.path.exists(stamp) else ""
        if force or not os.path.exists(o) or have != want:
            jobs.append((s, o, stamp, want))
    # objects (and stamps) of sources that no longer exist: not linked, but they would travel with every snapshot of the tree
    keep = set(objs) | {o + ".stamp" for o in objs}
    for name in os.listdir(OBJ):
        path = os.path.join(OBJ, name)
        if path not in keep and (name.endswith(".o") or name.endswith(".stamp")):
            os.remove(path)
    if jobs:
        def run(job):
            _compile(job[0], job[1], verbose)
            with open(job[2], "w") as f:
                f.write(job[3])
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs + \
          ["-lpthread", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (res.stdout, res.stderr))
    with open(manifest, "w") as f:
        f.write(digest)
    build_cli(verbose)
    return LIB


CLI_SRC = os.path.join(CSRC, "cli", "dcvc_cli.hip")
CLI_BIN = os.path.join(PKG, "bin", "dcvc")


def build_cli(verbose=False):
    """The standalone encoder / decoder (dcvc_amd/bin/dcvc): host code on top of the C ABI, linked
    against libdcvc_amd.so next to it."""
    os.makedirs(os.path.dirname(CLI_BIN), exist_ok=True)
    cmd = [_hipcc()] + COMMON + ["-x", "hip"] + HIP_FLAGS + [CLI_SRC, "-x", "none", "-o", CLI_BIN, "-L", PKG, "-ldcvc_amd",
                                                       "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed (dcvc tool):\n%s\n%s" % (res.stdout, res.stderr))
    return CLI_BIN


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
