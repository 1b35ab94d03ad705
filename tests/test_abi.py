"""CPU: the C-ABI library loads, exports every symbol include/rssf.h declares, and the ctypes table covers them all.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rssf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rssf_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "rssf_winattn_fwd" in syms and "rssf_winattn_bwd" in syms and "rssf_conv_gather" in syms
    assert len(syms) >= 20


def test_library_exports_every_declared_symbol():
    from representationlearning_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail("librssf.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_ctypes_table_matches_header():
    from representationlearning_amd import _lib
    assert sorted(_lib.SIGNATURES.keys()) == declared_symbols()
    lib = _lib.load()
    assert lib.rssf_arch() == b"gfx950"
    assert lib.rssf_version().decode().count(".") == 2
    assert lib.rssf_conv_tile_n(32) == 32 and lib.rssf_conv_tile_n(64) == 64 and lib.rssf_conv_tile_n(480) == 128


def test_param_struct_layout_matches_header():
    """Field order of the ctypes mirror == field order in rssf.h (the struct is passed by pointer)."""
    from representationlearning_amd import _lib
    text = open(os.path.join(ROOT, "include", "rssf.h")).read()
    body = re.search(r"typedef struct \{(.*?)\} rssf_winattn_fwd_params;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"(\w+)\s*(?=[,;])", body)
    assert [n for n, _ in _lib.WinAttnFwdParams._fields_] == names


def test_library_reads_no_environment_variable():
    """rssf.h: kernel selection is a function of the call's arguments.  The library neither imports getenv nor mentions it in its sources
    (the explicit knob is RSSF_CONV_GENERIC in the dtype argument of the convolution entry points)."""
    import glob
    import subprocess
    from representationlearning_amd import _lib
    csrc = os.path.join(ROOT, "representationlearning_amd", "csrc")
    hits = [os.path.basename(f) for pat in ("*.hip", "*.cpp", "*.h") for f in glob.glob(os.path.join(csrc, pat)) if "getenv" in open(f).read()]
    assert not hits, hits
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und
    assert _lib.CONV_GENERIC == 0x100 and "#define RSSF_CONV_GENERIC 0x100" in open(os.path.join(ROOT, "include", "rssf.h")).read()
    # the loss scratch the Python side allocates is the header's (the ABI had changed from 6 to 192 floats per sample silently)
    assert "#define RSSF_LOSS_ACC_ELEMS %d\n" % _lib.LOSS_ACC_ELEMS in open(os.path.join(ROOT, "include", "rssf.h")).read()
