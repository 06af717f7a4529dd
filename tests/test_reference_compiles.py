"""The drop-in boundary, checked by the compiler: the reference's OWN KvazaarFilter / OpenHEVCFilter translation
units -- /root/reference/src/media/processing/kvazaarfilter.cpp and openhevcfilter.cpp, compiled where they lie, nothing copied -- type-check
against THIS repository's include/kvazaar.h and include/openHevcWrapper.h.  Qt 6 and uvgRTP are absent from the build image, so their few
classes the two files touch are declared by tests/qtshim/ (test infrastructure, declarations only); every kvz_api member, kvz_config /
kvz_picture / kvz_data_chunk field, enum constant and libOpenHevc* signature the files use comes from include/.

Skipped where /root/reference does not exist (the GPU box)."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
UNITS = ["media/processing/kvazaarfilter.cpp", "media/processing/openhevcfilter.cpp"]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("g++") is None, reason="the reference tree is not mounted here")


def _syntax_only(unit, include_dir):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-I", include_dir, "-I", os.path.join(ROOT, "tests", "qtshim"),
           "-I", REF, "-I", os.path.join(REF, "media", "processing"), "-I", os.path.join(REF, "media"), os.path.join(REF, unit)]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("unit", UNITS)
def test_reference_filter_compiles_against_our_headers(unit):
    r = _syntax_only(unit, os.path.join(ROOT, "include"))
    assert r.returncode == 0, r.stderr[-4000:]


@pytest.mark.parametrize("unit,header,needle", [
    ("media/processing/kvazaarfilter.cpp", "kvazaar.h", "target_bitrate"),            # a kvz_config field uvgComm writes directly (kvazaarfilter.cpp:223)
    ("media/processing/openhevcfilter.cpp", "openHevcWrapper.h", "libOpenHevcGetPictureInfo"),   # openhevcfilter.cpp:199
])
def test_the_check_notices_a_missing_declaration(unit, header, needle):
    """the compile test is a real check: with the named field / function renamed in a scratch copy of OUR header the reference's file no longer compiles"""
    with tempfile.TemporaryDirectory() as tmp:
        for name in os.listdir(os.path.join(ROOT, "include")):
            text = open(os.path.join(ROOT, "include", name)).read()
            if name == header:
                assert needle in text
                text = text.replace(needle, needle + "_renamed")
            open(os.path.join(tmp, name), "w").write(text)
        r = _syntax_only(unit, tmp)
        assert r.returncode != 0 and needle in r.stderr, r.stderr[-2000:]
