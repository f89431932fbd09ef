"""The C-ABI library loads on a machine without a GPU and exports every symbol the header declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "specengine.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sa_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    for s in ["sa_init", "sa_rfft_peak_f32", "sa_rfft_peak_f64", "sa_rfft_peak_f32_dev", "sa_fft_mag_f64",
              "sa_fft_c2c_f64", "sa_synth_f64", "sa_compute_f64", "sa_last_error"]:
        assert s in syms


def test_library_exports_every_declared_symbol():
    from fft_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the library first (make -C fft_amd/csrc)"
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (sa_[a-z0-9_]+)", out))
    declared = set(header_symbols())
    assert declared <= exported, declared - exported
    assert exported <= declared, f"exported but not declared in the header: {exported - declared}"


def test_ctypes_table_matches_header_and_loads():
    from fft_amd import _lib
    assert set(_lib.PROTOTYPES) == set(header_symbols())
    lib = _lib.load()                                  # dlopen + prototype binding, no GPU needed
    assert lib.sa_version().startswith(b"specengine")
    assert isinstance(lib.sa_device_count(), int)


def test_no_cpu_fallback_without_device():
    """Without a GPU the compute entry points must fail loudly (SA_ENODEV -> RuntimeError), never compute."""
    import numpy as np
    from fft_amd import _lib
    import fft_amd
    if _lib.load().sa_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fft_amd.rfft_peak(np.zeros((1, 4096), np.float32))
    with pytest.raises(RuntimeError):
        fft_amd.fft_mag(np.zeros(1500))


def test_product_package_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fft_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn), encoding="utf-8").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt, fn
