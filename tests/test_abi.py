"""CPU-only checks of the C-ABI boundary: the HIP library builds/loads here (hipcc cross-compiles
for gfx950 without a GPU) and exports every symbol include/raht.h declares. No compute calls."""
import ctypes
import os
import re

import pytest

from .conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "raht.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(raht_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def L():
    import raht_3dgs_codec_amd as R
    from raht_3dgs_codec_amd import _lib
    if not os.path.exists(R.SO_PATH):
        R.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported(L):
    names = _declared()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_python_binding_covers_the_header(L):
    from raht_3dgs_codec_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()


def test_version_and_error_string(L):
    assert L.raht_version() == 300
    assert isinstance(L.raht_last_error(), bytes)
    # argument validation happens before any HIP call, so it is safe without a GPU
    out = ctypes.c_void_p()
    rc = L.raht_plan_create(None, 1, 10, (ctypes.c_double * 3)(0, 0, 0), 1024.0, 10, None, ctypes.byref(out))
    assert rc == -1 and b"NULL" in L.raht_last_error()
    dummy = ctypes.c_void_p(16)
    rc = L.raht_plan_create(dummy, 1, 10, (ctypes.c_double * 3)(0, 0, 0), 1024.0, 22, None, ctypes.byref(out))
    assert rc == -1 and b"depth" in L.raht_last_error()
    rc = L.raht_plan_create(dummy, 1, 0, (ctypes.c_double * 3)(0, 0, 0), 1024.0, 10, None, ctypes.byref(out))
    assert rc == -1


def test_no_cpu_fallback():
    """The operator mirror refuses CPU tensors instead of silently computing on the host."""
    import torch
    import raht_3dgs_codec_amd as R
    V = torch.zeros((4, 3), dtype=torch.float64)
    with pytest.raises(RuntimeError, match="no CPU path"):
        R.raht_fn["RAHT_param"](V, torch.zeros(3, dtype=torch.float64), 2, 1)
    with pytest.raises(RuntimeError):
        R.raht_fn["RAHT"](torch.zeros((4, 2)), [torch.zeros(2, dtype=torch.int64)], None, None)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import, link or load it."""
    pkg = os.path.join(ROOT, "raht-3dgs-codec_amd")
    bad = re.compile(r"(import\s+oracle|from\s+oracle|raht_oracle|libraht_oracle|orc_[a-z_]+\s*\()")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f)).read()
                assert not bad.search(txt), (dp, f)


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py measures the HIP path or nothing: on a host without a GPU it stops with a clear message
    (no CPU number is ever printed as the metric)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs an MI355X" in (r.stderr + r.stdout)
    assert '"metric"' not in r.stdout
