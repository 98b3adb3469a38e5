"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/htj2k_amd.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes
import os
import re
import subprocess

import pytest

import ffmpeg_ht_amd as m

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "htj2k_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(htj2k_[a-z0-9_]+)\s*\(", text)) - {"htj2k_log_fn"})


def test_library_is_built_and_exports_header_symbols():
    lib = m.load_library()
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert sorted(m.EXPORTS) == sorted(set(m.EXPORTS))
    for n in m.EXPORTS:
        assert n in names, n + " bound in Python but not declared in the header"


def test_no_oracle_linkage_in_product():
    """the product library must not contain or depend on anything from oracle/"""
    out = subprocess.check_output(["nm", "-D", "--defined-only", m.LIB_PATH]).decode()
    assert "orc_" not in out
    ldd = subprocess.check_output(["ldd", m.LIB_PATH]).decode()
    assert "oracle" not in ldd and "vecgen" not in ldd


def test_product_sources_do_not_reference_oracle():
    pkg = os.path.join(ROOT, "ffmpeg-ht_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".c", ".h", ".hpp", ".hip", ".py")):
                text = open(os.path.join(dp, f)).read()
                assert "#include \"../../oracle" not in text and "import oracle" not in text, f


def test_block_descriptor_layout_matches_c():
    # struct J2kBlock is 32 bytes; the Python mirror used by the unit tests must agree
    assert ctypes.sizeof(m.BlockDesc) == 32
    assert m.BlockDesc.f_step.offset == 24 and m.BlockDesc.i_step.offset == 28


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_open_fails_loudly_without_gpu():
    with pytest.raises(m.Htj2kError) as e:
        m.Decoder()
    assert e.value.code == -38          # HTJ2K_ERR_ENOSYS: no fallback path exists
