"""ctypes binding of libovla_hip.so (the C-ABI declared in include/ovla.h).

The argument structs and prototypes are derived from the header text at import time, so the Python side cannot
drift from the C side.  There is NO fallback: if the shared library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes
import os
import re
from pathlib import Path

_PKG_DIR = Path(__file__).resolve().parent
_REPO_ROOT = _PKG_DIR.parent
HEADER_PATH = _REPO_ROOT / "include" / "ovla.h"
LIB_PATH = _PKG_DIR / os.environ.get("OVLA_LIB_NAME", "libovla_hip.so")   # tools/gemm_ablate.py loads libovla_hip_ablate.so

_CTYPE = {
    "void*": ctypes.c_void_p,
    "void**": ctypes.POINTER(ctypes.c_void_p),
    "void": None,
    "float*": ctypes.c_void_p,
    "int32_t*": ctypes.c_void_p,
    "int64_t*": ctypes.c_void_p,
    "uint8_t*": ctypes.c_void_p,
    "uint32_t*": ctypes.c_void_p,
    "uint32_t": ctypes.c_uint32,
    "double*": ctypes.c_void_p,
    "char*": ctypes.c_char_p,
    "int64_t": ctypes.c_int64,
    "int32_t": ctypes.c_int32,
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
}


def _strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _norm_type(t: str) -> str:
    t = t.replace("const", " ").strip()
    t = re.sub(r"\s+", " ", t)
    return t.replace(" *", "*").replace("* ", "*")


def parse_header(path: Path = HEADER_PATH):
    """Returns (structs: {name: [(field, ctype)]}, functions: {name: (restype, [argtypes])})."""
    text = _strip_comments(path.read_text())
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        body, name = m.group(1), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            dm = re.match(r"^((?:const\s+)?\w+\s*\**)\s*(.*)$", decl)
            base = _norm_type(dm.group(1))
            for var in dm.group(2).split(","):
                var = var.strip()
                ctype_name = base
                while var.startswith("*"):
                    ctype_name += "*"
                    var = var[1:].strip()
                ct = _CTYPE[ctype_name]
                am = re.match(r"^(\w+)\s*\[\s*(\d+)\s*\]$", var)      # fixed-size array member, e.g. `float mean[6]`
                if am:
                    var, ct = am.group(1), ct * int(am.group(2))
                fields.append((var, ct))
        structs[name] = fields
    text_wo_structs = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    functions = {}
    for m in re.finditer(r"((?:const\s+)?\w+\s*\*?)\s+(ovla_\w+)\s*\(([^)]*)\)\s*;", text_wo_structs):
        ret, name, args = _norm_type(m.group(1)), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                am = re.match(r"^((?:const\s+)?\w+\s*\**)\s*\w*$", a)
                t = _norm_type(am.group(1))
                if t.endswith("*") and t[:-1] in structs:
                    argtypes.append(("struct", t[:-1]))
                else:
                    argtypes.append(("plain", t))
        functions[name] = (ret, argtypes)
    return structs, functions


STRUCT_FIELDS, FUNCTIONS = parse_header()
STRUCTS = {}
for _name, _fields in STRUCT_FIELDS.items():
    STRUCTS[_name] = type(_name, (ctypes.Structure,), {"_fields_": _fields})

_lib = None


def source_hash() -> str:
    """The hash csrc/build.sh compiles into the library: sha256 over csrc/*.hip + csrc/*.h (C-locale name order), build.sh, ovla.h."""
    import hashlib

    csrc = _PKG_DIR / "csrc"
    names = sorted([f.name for f in csrc.glob("*.hip")] + [f.name for f in csrc.glob("*.h")])
    h = hashlib.sha256()
    for f in [csrc / n for n in names] + [csrc / "build.sh", HEADER_PATH]:
        h.update(f.read_bytes())
    return h.hexdigest()[:32]


def lib() -> ctypes.CDLL:
    """Loads libovla_hip.so; raises if it has not been built (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or {_PKG_DIR / 'csrc' / 'build.sh'}). There is no fallback path."
        )
    # PyTorch-ROCm ships its own libamdhip64; it must be the first HIP runtime mapped into the process, otherwise torch
    # later finds "No HIP GPUs".  Importing torch first makes our library bind to the runtime torch uses (same soname).
    import torch  # noqa: F401

    handle = ctypes.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
    for name, (ret, argtypes) in FUNCTIONS.items():
        fn = getattr(handle, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = _CTYPE[ret] if ret in _CTYPE else ctypes.c_int
        fn.argtypes = [ctypes.POINTER(STRUCTS[t]) if kind == "struct" else _CTYPE[t] for kind, t in argtypes]
    if handle.ovla_abi_version() != 1:
        raise RuntimeError("libovla_hip.so ABI version mismatch")
    built, want = handle.ovla_build_hash().decode(), source_hash()
    if built != want:
        raise RuntimeError(f"{LIB_PATH.name} is STALE: it was built from sources with hash {built}, the sources beside it hash to {want}. "
                           f"Rebuild with {_PKG_DIR / 'csrc' / 'build.sh'} (there is no fallback path).")
    _lib = handle
    return handle


class OvlaError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ovla_last_error()
        raise OvlaError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def call(name: str, args, stream: int) -> None:
    """Calls `int name(const args*, void* stream)` and raises on a non-zero return."""
    check(getattr(lib(), name)(ctypes.byref(args), ctypes.c_void_p(stream)), name)
