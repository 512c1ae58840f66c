"""ctypes binding of libunet_hip.so.  The prototype table is parsed from include/unet_hip.h so the
header is the single source of truth for the C ABI.  There is NO fallback: if the library is missing
or a call fails, a RuntimeError is raised (the product path never routes through a CPU path)."""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
HEADER = os.path.join(ROOT, "include", "unet_hip.h")
# UH_LIB_PATH points the binding at another build of the same C ABI (A/B kernel experiments)
LIB_PATH = os.environ.get("UH_LIB_PATH") or os.path.join(PKG_DIR, "libunet_hip.so")

UH_F32, UH_BF16, UH_F32X3 = 0, 1, 2
UH_WFRAG, UH_WFRAG_D = 0x100, 0x200      # fragment-major filter packs (include/unet_hip.h)

_CTYPES = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "double": ctypes.c_double,
    "size_t": ctypes.c_size_t, "uh_stream": ctypes.c_void_p,
}
_RET = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[str]]]:
    """Return {symbol: (return type, [parameter types])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|int|size_t)\s+(uh_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append("ptr")
                else:
                    types.append(a.rsplit(" ", 1)[0].strip())
        protos[name] = (ret, types)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = parse_header()

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
        dll = ctypes.CDLL(LIB_PATH)
        for name, (ret, types) in self.protos.items():
            fn = getattr(dll, name)          # AttributeError here = header/library mismatch
            fn.restype = _RET[ret]
            fn.argtypes = [ctypes.c_void_p if t == "ptr" else _CTYPES[t] for t in types]
        self._dll = dll
        return dll

    def call(self, name: str, *args):
        dll = self.load()
        rc = getattr(dll, name)(*args)
        if rc != 0:
            msg = dll.uh_last_error().decode("utf-8", "replace")
            raise RuntimeError(f"{name} failed ({rc}): {msg}")

    def query(self, name: str, *args):
        return getattr(self.load(), name)(*args)


LIB = _Lib()
