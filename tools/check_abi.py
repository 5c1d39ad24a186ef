"""Checks that libdcvc_amd.so exports every function declared in include/*.h."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DECL = re.compile(r"\b(dcvc_[a-z0-9_]+)\s*\(")


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        with open(h) as f:
            text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
        names.update(_DECL.findall(text))
    return sorted(names)


def missing_symbols():
    from dcvc_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    return [n for n in declared_symbols() if not hasattr(lib, n)]


if __name__ == "__main__":
    print("declared:", len(declared_symbols()), "missing:", missing_symbols())
