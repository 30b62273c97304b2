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
    assert lib.fmh_abi_version() == 3


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


def test_no_kernel_spills_to_scratch(lib):
    """Every kernel of the shipped library, from the gfx950 code objects' metadata: no scratch memory (.private_segment_fixed_size)
    and no spilled VGPRs.  Round 1 shipped eleven 8-group multi-allelic instantiations that spilled up to 464 VGPRs; no parity
    test notices that."""
    import subprocess
    import sys

    tool = os.path.join(ROOT, "tools", "kernel_resources.py")
    res = subprocess.run([sys.executable, tool, "--spills-only"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    m = re.search(r"(\d+) kernels, 0 with scratch", res.stdout)
    assert m and int(m.group(1)) >= 150, res.stdout[-500:]


def test_comm_and_packing_entry_points_without_a_gpu(lib):
    """The packing halves are host arithmetic (no device); the communicator itself refuses loudly without a GPU."""
    import torch

    from ferromic_amd import _abi

    t = _abi.WcTotals()
    t.sum_a[0], t.sum_b[2], t.informative_sites[1], t.sites_attempted = 1.5, 2.5, 7, 100
    f, u = (C.c_double * 8)(), (C.c_uint64 * 5)()
    assert lib.fmh_wc_totals_pack(C.byref(t), 3, f, u) == 0
    assert list(f) == [1.5, 0, 0, 0, 0, 0, 2.5, 0] and list(u) == [0, 7, 0, 0, 100]
    back = _abi.WcTotals()
    assert lib.fmh_wc_totals_unpack(C.byref(back), 3, f, u) == 0
    assert back.sum_a[0] == 1.5 and back.sum_b[2] == 2.5 and back.informative_sites[1] == 7 and back.sites_attempted == 100
    assert lib.fmh_wc_totals_pack(C.byref(t), 9, f, u) == _abi.FMH_ERR_INVALID
    pt = (_abi.PopTotals * 2)()
    pt[0].pi_sum, pt[1].segregating_sites, pt[0].haplotype_capacity = 0.75, 9, 30
    pf, pu = (C.c_double * 2)(), (C.c_uint64 * 7)()
    assert lib.fmh_pop_totals_pack(pt, 2, pf, pu) == 0 and pu[6] == 1
    doubled = (C.c_uint64 * 7)(*[2 * x for x in pu])
    out = (_abi.PopTotals * 2)()
    assert lib.fmh_pop_totals_unpack(out, 2, (C.c_double * 2)(*[2 * x for x in pf]), doubled) == 0
    assert out[0].pi_sum == 1.5 and out[1].segregating_sites == 18 and out[0].haplotype_capacity == 30
    if not torch.cuda.is_available():
        buf = (C.c_char * 128)()
        assert lib.fmh_comm_get_unique_id(buf) == _abi.FMH_ERR_NO_DEVICE
        h = C.c_void_p()
        assert lib.fmh_comm_init_all((C.c_int * 2)(0, 0), 2, (C.c_void_p * 2)()) != 0


def test_options_are_set_through_the_abi_not_the_environment(lib):
    """fmh_set_option / fmh_get_option (no GPU needed): integers and the listed words parse, NULL restores what the process started with,
    unknown keys and garbage are refused, and the FMH_* environment is read once - a later os.environ change is NOT seen, a child
    process started with the variable set sees it as its initial value."""
    import subprocess
    import sys

    from ferromic_amd import _abi

    assert _abi.get_option("FMH_DEFER_TILES") == 0 and _abi.get_option("FMH_MASK_MODE") == -1 and _abi.get_option("FMH_LAYOUT") == 0
    _abi.set_option("FMH_LAYOUT", "bytes")
    assert _abi.get_option("FMH_LAYOUT") == 1
    _abi.set_option("FMH_LAYOUT", "packed")
    assert _abi.get_option("FMH_LAYOUT") == 0
    with _abi.options(FMH_DEFER_TILES=7, FMH_GRID_BLOCKS=3):
        assert _abi.get_option("FMH_DEFER_TILES") == 7 and _abi.get_option("FMH_GRID_BLOCKS") == 3
    assert _abi.get_option("FMH_DEFER_TILES") == 0 and _abi.get_option("FMH_GRID_BLOCKS") == 0
    assert lib.fmh_set_option(b"FMH_NO_SUCH_SWITCH", b"1") == _abi.FMH_ERR_INVALID and b"unknown option" in lib.fmh_last_error()
    assert lib.fmh_set_option(b"FMH_DEFER_TILES", b"deep") == _abi.FMH_ERR_INVALID
    assert lib.fmh_set_option(None, b"1") == _abi.FMH_ERR_INVALID
    os.environ["FMH_GRID_PER_CU"] = "5"  # after the snapshot: ignored
    try:
        assert _abi.get_option("FMH_GRID_PER_CU") == 0
        code = ("from ferromic_amd import _abi; print(_abi.get_option('FMH_GRID_PER_CU'), _abi.get_option('FMH_LAYOUT'), _abi.get_option('FMH_COMM_TRANSPORT'));"
                "_abi.set_option('FMH_GRID_PER_CU', 2); _abi.set_option('FMH_GRID_PER_CU', None); print(_abi.get_option('FMH_GRID_PER_CU'))")
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT,
                             env=dict(os.environ, FMH_LAYOUT="bytes", FMH_COMM_TRANSPORT="host"))
        assert res.returncode == 0, res.stderr[-2000:]
        assert res.stdout.split() == ["5", "1", "1", "5"], res.stdout
    finally:
        del os.environ["FMH_GRID_PER_CU"]


def test_no_getenv_on_the_launch_path():
    """The kernel-routing sources read no environment variable: every switch is an option (abi_internal.hpp Options)."""
    csrc = os.path.join(ROOT, "ferromic_amd", "csrc")
    for name in ("sweep_launch.inc", "sweep_mfma.hip", "pairwise.hip", "upload.hip", "abi_internal.hpp"):
        with open(os.path.join(csrc, name)) as fh:
            assert "getenv" not in fh.read(), name
    with open(os.path.join(csrc, "abi.hip")) as fh:
        text = fh.read()
    assert text.count("getenv(") == 1, "abi.hip reads the environment in options() only"
