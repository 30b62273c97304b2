"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/ferromic_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""

import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    if not os.path.exists(os.path.join(ROOT, "ferromic_amd", "lib", "libferromic_hip.so")):
        ge.build()
    from ferromic_amd import _abi

    return _abi.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ferromic_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fmh_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    from ferromic_amd import _abi

    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/ferromic_hip.h but not exported"
        assert name in _abi.SYMBOLS, f"{name} has no ctypes prototype in ferromic_amd/_abi.py"
    assert sorted(_abi.SYMBOLS) == names
    assert lib.fmh_abi_version() == 2


def test_struct_layouts_match_header(lib):
    from ferromic_amd import _abi

    assert C.sizeof(_abi.PopTotals) == 32
    assert C.sizeof(_abi.HudsonTotals) == 11 * 8 + 2 * 32
    assert C.sizeof(_abi.HudsonSites) == 8 * 8
    assert C.sizeof(_abi.WcTotals) == 29 * 8 * 3 + 8


def test_totals_pack_roundtrip(lib):
    """Host-only entry points work without a GPU (they move no data to a device)."""
    from ferromic_amd import _abi

    t = _abi.HudsonTotals()
    t.numerator_sum, t.denominator_sum, t.site_dxy_sum = 1.5, 2.5, 0.25
    t.dxy_uncallable_sites, t.sites_with_components = 7, 11
    t.pop[0].pi_sum, t.pop[1].segregating_sites, t.pop[0].haplotype_capacity = 3.25, 5, 40
    f = (C.c_double * _abi.HUDSON_PACK_F64)()
    u = (C.c_uint64 * _abi.HUDSON_PACK_U64)()
    assert lib.fmh_hudson_totals_pack(C.byref(t), f, u) == 0
    # emulate an all-reduce(sum) over 2 ranks
    f2 = (C.c_double * _abi.HUDSON_PACK_F64)(*[2 * x for x in f])
    u2 = (C.c_uint64 * _abi.HUDSON_PACK_U64)(*[2 * x for x in u])
    out = _abi.HudsonTotals()
    assert lib.fmh_hudson_totals_unpack(C.byref(out), f2, u2) == 0
    assert out.numerator_sum == 3.0 and out.dxy_uncallable_sites == 14
    assert out.pop[0].pi_sum == 6.5 and out.pop[1].segregating_sites == 10
    assert out.pop[0].haplotype_capacity == 40  # capacities are per-rank constants, not sums


def test_no_gpu_fails_loudly(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path cannot be exercised")
    import numpy as np

    from ferromic_amd import _abi, device

    with pytest.raises(_abi.NoDeviceError):
        _abi.device_count()
    with pytest.raises(_abi.NoDeviceError):
        device.DeviceMatrix.from_host(np.zeros(8, np.uint8), None, 2, 2, 2, 0)


def test_no_dot4_result_is_read_too_early():
    """gfx950: a v_dot4 result read by a DPP op (the row reductions) needs 3 wait states; the compiler once left 2 and
    the sweep lost counts.  The device assembly of the current sources must be free of that pattern."""
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "scan_dot4_hazard.py")
    res = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert " 0 hazard(s)" in res.stdout
