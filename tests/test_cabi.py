"""The C-ABI library loads and exports every symbol include/seal3d_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared():
    hdr = open(os.path.join(REPO, "include", "seal3d_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(s3d_[A-Za-z0-9_]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    import s3d_hip
    assert sorted(s3d_hip.EXPORTS) == _declared()


def test_library_exports_every_declared_symbol():
    import s3d_hip
    if not os.path.exists(s3d_hip.LIB_PATH):
        s3d_hip.build()
    lib = ctypes.CDLL(s3d_hip.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, f"not exported: {missing}"
    lib.s3d_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.s3d_version()


def test_product_path_has_no_cpu_fallback():
    """A CPU tensor handed to the product backend must raise, never silently compute."""
    import torch
    import s3d_hip
    if not os.path.exists(s3d_hip.LIB_PATH):
        s3d_hip.build()
    c = torch.zeros(4, 3, dtype=torch.int32)
    out = torch.zeros(4, dtype=torch.int32)
    with pytest.raises(RuntimeError):
        s3d_hip.RaymarchingBackend.morton3D(c, 4, out)


def test_product_sources_do_not_reference_oracle():
    pkg = os.path.join(REPO, "seal-3d_amd")
    bad = []
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "s3o_" in txt:
                    bad.append(os.path.join(root, f))
    assert not bad, f"product files touching the oracle: {bad}"


def test_ctypes_call_sites_pass_as_many_arguments_as_the_header_declares():
    """ctypes does not check arity: a call site that drifts from include/seal3d_hip.h would corrupt the stack silently.
    Static check: every `lib().s3d_*(...)` call in the binding passes exactly the declared number of arguments."""
    import ast
    import re
    header = open(os.path.join(REPO, "include", "seal3d_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = {}
    for m in re.finditer(r"\b(s3d_\w+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S):
        args = m.group(2).strip()
        declared[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    src = open(os.path.join(REPO, "seal-3d_amd", "s3d_hip", "__init__.py")).read()
    starred_width = {"_live": 2, "_mid_fwd_args": 4, "_mid_bwd_args": 3, "_shadow2": 2, "_shadow1": 1}  # helpers that expand to several C arguments
    seen = set()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith("s3d_") \
                and isinstance(node.func.value, ast.Call):
            n = 0
            for a in node.args:
                if isinstance(a, ast.Starred):
                    assert isinstance(a.value, ast.Call) and a.value.func.id in starred_width, ast.dump(a)
                    n += starred_width[a.value.func.id]
                else:
                    n += 1
            name = node.func.attr
            assert name in declared, name
            assert n == declared[name], f"{name}: call passes {n} arguments, header declares {declared[name]}"
            seen.add(name)
    assert len(seen) >= 40, len(seen)
