#!/usr/bin/env python3
"""Rewrites the generated `espm_mu_state` block of INTEGRATION.md from include/espm_mu.h (espm_amd/_abi.py)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "# --- BEGIN GENERATED espm_mu_state ---\n", "# --- END GENERATED espm_mu_state ---\n"


def _abi():   # (loaded by path: importing the package would need the built library)
    spec = importlib.util.spec_from_file_location("_espm_abi", os.path.join(ROOT, "espm_amd", "_abi.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def generated_block():
    abi = _abi()
    return abi.ctypes_source(abi.parse_struct(abi.header_text())) + "\n"


def render(text):
    a, b = text.index(BEGIN) + len(BEGIN), text.index(END)
    return text[:a] + generated_block() + text[b:]


if __name__ == "__main__":
    path = os.path.join(ROOT, "INTEGRATION.md")
    old = open(path).read()
    new = render(old)
    if "--check" in sys.argv:
        sys.exit(0 if new == old else 1)
    open(path, "w").write(new)
