"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/primesm_hip.h declares; the host mirrors fail loudly without a GPU
(there is no CPU fallback in the product)."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as G
    G.build()
    from primestereomatch_amd import capi
    return capi


def test_header_symbols_all_exported(built):
    hdr = open(os.path.join(ROOT, "include", "primesm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(psm_[a-z0-9_]+)\s*\(", hdr))
    bound = {name for name, _, _ in built.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    lib = built.load()
    for name in declared:
        assert hasattr(lib, name)
    out = subprocess.run(["nm", "-D", "--defined-only", built.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (psm_[a-z0-9_]+)", out))
    assert declared <= exported


def test_code_object_is_gfx950(built):
    blob = open(built.LIB_PATH, "rb").read()
    assert b"hipv4-amdgcn-amd-amdhsa--gfx950" in blob          # the fat binary's bundle entry
    assert b"amdhsa--gfx942" not in blob and b"amdhsa--gfx90a" not in blob   # gfx950 only


def test_no_gpu_fails_loudly(built):
    if built.device_count() > 0:
        pytest.skip("a GPU is present")
    import primestereomatch_amd as P
    l = np.zeros((16, 16, 3), np.uint8)
    with pytest.raises(built.PsmError):
        P.DispEst(l, l, 8)
    assert "device" in built.last_error(None).lower()


def test_missing_library_fails_loudly(built, tmp_path):
    with pytest.raises(built.PsmError):
        built.load(str(tmp_path / "nope.so"))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under primestereomatch_amd/ may reference it."""
    pkg = os.path.join(ROOT, "primestereomatch_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                for pat in (r'#\s*include\s*[<"][^>"]*oracle', r"^\s*(from|import)\s+oracle", r"libpsm_oracle",
                            r"\bpsmo_[a-z0-9_]+\s*\(", r"dlopen\([^)]*oracle"):
                    assert not re.search(pat, txt, flags=re.M), (fn, pat)


def test_host_demo_builds_and_reports_no_device(built):
    demo = os.path.join(ROOT, "primestereomatch_amd", "lib", "psm_demo")
    assert os.path.exists(demo)
    if built.device_count() > 0:
        pytest.skip("a GPU is present")
    raw = np.zeros((16, 16, 3), np.uint8).tobytes()
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        for n in ("l.raw", "r.raw"):
            open(os.path.join(td, n), "wb").write(raw)
        env = dict(os.environ, PRIMESM_HIP_LIB=built.LIB_PATH)
        p = subprocess.run([demo, td + "/l.raw", td + "/r.raw", "16", "16", "8", td + "/o"], env=env,
                           capture_output=True, text=True)
        assert p.returncode == 3 and "no HIP device" in p.stderr


def test_shard_bounds_cover():
    from shard_model import shard_bounds
    for D in (1, 5, 64, 255, 256):
        for G in (1, 2, 3, 8):
            b = shard_bounds(D, G)
            assert b[0][0] == 0 and b[-1][1] == D
            assert all(b[i][1] == b[i + 1][0] for i in range(G - 1))
