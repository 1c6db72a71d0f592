"""The layout of `espm_mu_state` as include/espm_mu.h declares it: the one parser the ctypes binding (espm_amd/_lib.py),
the integration notes and the tests share - the header is the single source of the layout.

    python -m espm_amd._abi            # the ctypes field list of the header, as Python source (INTEGRATION.md's block)
    python -m espm_amd._abi --c-names  # the field names, for espm_mu_state_layout() in csrc/mu_api.hip
"""
from __future__ import annotations

import ctypes as C
import os
import re

# The repository's include/espm_mu.h (the one hand-kept copy); an installed package carries a copy of it next to the library
# (espm_amd/include/espm_mu.h, placed there by __graft_entry__.build() and listed in pyproject.toml's package-data).
_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO_HEADER = os.path.join(os.path.dirname(_HERE), "include", "espm_mu.h")
HEADER = _REPO_HEADER if os.path.exists(_REPO_HEADER) else os.path.join(_HERE, "include", "espm_mu.h")

_SCALARS = {"int32_t": C.c_int32, "uint32_t": C.c_uint32, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "float": C.c_float,
            "double": C.c_double, "int": C.c_int, "size_t": C.c_size_t}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def parse_defines(text):
    """#define NAME <integer expression of earlier defines> -> {NAME: int}; everything else is skipped."""
    out = {}
    for m in re.finditer(r"^[ \t]*#[ \t]*define[ \t]+(\w+)[ \t]+(.+?)[ \t]*$", _strip_comments(text), flags=re.M):
        name, expr = m.group(1), m.group(2)
        if not re.fullmatch(r"[\w\s()+\-*/<>]+", expr):
            continue
        try:
            out[name] = int(eval(expr, {"__builtins__": {}}, dict(out)))
        except Exception:
            pass
    return out


def parse_struct(text, name="espm_mu_state"):
    """[(field, ctypes type)] in declaration order.  Pointers of any kind are c_void_p; `T* f[2]` is c_void_p * 2."""
    body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), _strip_comments(text), flags=re.S)
    if not body:
        raise ValueError(f"struct {name} not found")
    fields = []
    for decl in body.group(1).split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.fullmatch(r"(const )?(\w+)( ?\*)? ?(.+)", decl)
        if not m:
            raise ValueError(f"cannot parse declaration {decl!r}")
        base, is_ptr = m.group(2), bool(m.group(3))
        for item in m.group(4).split(","):
            item = item.strip()
            ptr = is_ptr
            if item.startswith("*"):
                ptr, item = True, item[1:].strip()
            arr = re.fullmatch(r"(\w+)\[(\d+)\]", item)
            fname, count = (arr.group(1), int(arr.group(2))) if arr else (item, 0)
            if not re.fullmatch(r"\w+", fname):
                raise ValueError(f"cannot parse declarator {item!r} in {decl!r}")
            if ptr:
                ct = C.c_void_p
            elif base in _SCALARS:
                ct = _SCALARS[base]
            else:
                raise ValueError(f"unknown type {base!r} in {decl!r}")
            fields.append((fname, ct * count if count else ct))
    return fields


def header_text(path=HEADER):
    with open(path) as f:
        return f.read()


def layout_string(struct_type):
    """The same text espm_mu_state_layout() returns, from a ctypes Structure."""
    return "".join(f"{n}:{getattr(struct_type, n).offset}:{getattr(struct_type, n).size};" for n, _ in struct_type._fields_)


def ctypes_source(fields):
    names = {C.c_int: "c_int", C.c_size_t: "c_size_t", C.c_void_p: "c_void_p", C.c_int32: "c_int32", C.c_uint32: "c_uint32",
             C.c_int64: "c_int64", C.c_uint64: "c_uint64", C.c_float: "c_float", C.c_double: "c_double"}
    lines = ["class MUState(ctypes.Structure):", "    _fields_ = ["]
    for n, t in fields:
        if hasattr(t, "_length_"):
            lines.append(f'        ("{n}", ctypes.{names[t._type_]} * {t._length_}),')
        else:
            lines.append(f'        ("{n}", ctypes.{names[t]}),')
    lines.append("    ]")
    return "\n".join(lines)


if __name__ == "__main__":
    import sys
    fs = parse_struct(header_text())
    if "--c-names" in sys.argv:
        print(" ".join(f"F({n})" for n, _ in fs))
    else:
        print(ctypes_source(fs))
