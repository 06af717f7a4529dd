"""The C-ABI library loads without a GPU and exports every entry point include/*.h declares; the host-only
parts of kvz_api (configuration, pictures, chunks) behave like the reference expects.  No compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from kvazzup_amd import _native
    if not os.path.exists(_native.library_path()):
        _native.build_library()
    return _native.load_library()


def declared_functions():
    names = []
    for hdr in ("kvazaar.h", "openHevcWrapper.h", "kvazzup_amd.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"#define[^\n]*", "", text)
        for m in re.finditer(r"(?:KVZ_PUBLIC|OHEVC_PUBLIC)\s+[^;{]*?\b(\w+)\s*\(", text):
            names.append(m.group(1))
    return sorted(set(names))


def test_every_declared_entry_point_is_exported(lib):
    names = declared_functions()
    assert "kvz_api_get" in names and "libOpenHevcDecode" in names and "kvzx_encoder_encode_device" in names and len(names) >= 45
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_kvz_api_table_and_config_parsing(lib):
    from kvazzup_amd import _native as N
    assert not lib.kvz_api_get(10)                       # only 8-bit
    api = lib.kvz_api_get(8).contents
    for name, _ in N.KvzApi._fields_:
        assert getattr(api, name), name
    cfg = api.config_alloc()
    assert api.config_init(cfg) == 1
    ok = lambda k, v: api.config_parse(cfg, k.encode(), v.encode())
    # the exact sequence of KvazaarFilter::init (kvazaarfilter.cpp:172-284)
    for k, v in (("preset", "ultrafast"), ("input-res", "1920x1080"), ("input-fps", "30/1"), ("threads", "8"), ("owf", "2"), ("wpp", "1"),
                 ("qp", "32"), ("period", "64"), ("vps-period", "1"), ("intra-bits", ""), ("gop", "lp-g4d3t1"), ("scaling-list", "off"),
                 ("mv-constraint", "none"), ("mv-constraint", ""), ("vaq", "5"), ("rc-algorithm", "lambda"), ("slices", "wpp")):
        assert ok(k, v) == 1, (k, v)
    c = cfg.contents
    assert c.sao_type == 0                                   # ultrafast: SAO off
    assert (c.width, c.height, c.framerate_num, c.framerate_denom, c.qp, c.intra_period, c.vps_period, c.owf, c.wpp) == (1920, 1080, 30, 1, 32, 64, 1, 2, 1)
    assert ok("preset", "medium") == 1 and cfg.contents.sao_type == 3     # presets above ultrafast: SAO full ...
    assert cfg.contents.intra_in_p == 2 and cfg.contents.rdoq_enable == 1 and cfg.contents.signhide_enable == 0      # ... intra units in P pictures, rdoq from medium on
    assert ok("intra-in-p", "0") == 1 and cfg.contents.intra_in_p == 0 and ok("intra-in-p", "1") == 1 and ok("intra-in-p", "2") == 1 and ok("intra-in-p", "3") == 0
    assert cfg.contents.me_source == 0                                         # medium: the search on the reconstruction
    assert ok("preset", "veryfast") == 1 and cfg.contents.intra_in_p == 1      # the fast presets: 16x16 intra units only
    assert cfg.contents.me_source == 1 and ok("me-source", "0") == 1 and cfg.contents.me_source == 0 and ok("me-source", "1") == 1 and ok("me-source", "2") == 0      # ... and the search on the input picture ("uvgx search pipelining v1")
    assert ok("sao", "off") == 1 and cfg.contents.sao_type == 0           # ... unless a later option says otherwise
    assert ok("preset", "ultrafast") == 1 and cfg.contents.sao_type == 0 and cfg.contents.intra_in_p == 0 and cfg.contents.me_source == 0
    assert ok("scaling-list", "default") == 1 and cfg.contents.scaling_list == 2 and ok("scaling-list", "off") == 1 and cfg.contents.scaling_list == 0      # uvgComm's checkbox (kvazaarfilter.cpp:235-242)
    # rejected: unknown names and values outside the implemented tool set (kvazaarfilter.cpp:363-367 logs these)
    for k, v in (("no-such-option", "1"), ("qp", "99"), ("input-res", "axb"), ("tiles", "0x2"), ("scaling-list", "custom"), ("gop", "8"), ("preset", "warp9"), ("sao", "edge")):
        assert ok(k, v) == 0, (k, v)
    c.target_bitrate = 0
    c.mv_constraint = 4
    c.hash = 0
    assert api.config_destroy(cfg) == 1


def test_config_parse_takes_any_string(lib):
    """config_parse is what uvgComm's free-form "custom parameters" reach (kvazaarfilter.cpp:302-311, 363-367): every name and value -- empty, huge, negative,
    non-numeric, not UTF-8 -- is answered with 1 or 0, and what was accepted leaves the fields uvgComm reads inside their ranges"""
    import random
    api = lib.kvz_api_get(8).contents
    names = ["preset", "input-res", "input-fps", "threads", "owf", "wpp", "tiles", "slices", "qp", "period", "vps-period", "rc-algorithm", "intra-bits", "gop",
             "scaling-list", "mv-constraint", "vaq", "bitrate", "sao", "deblock", "subme", "me-range", "rdoq", "signhide", "lossless", "intra-in-p", "hash", "gpu",
             "me-early-termination", "intra-satd", "me-source", "input-hold", "recon-output", "null-input", "gpu-entropy", "roi", "cqmfile", "", "x" * 300]
    values = ["", "0", "1", "-1", "2147483647", "-2147483648", "99999999999999999999", "1e9", "0x10", "on", "off", "true", "lambda", "oba", "wpp", "tiles",
              "ultrafast", "placebo", "1920x1080", "0x0", "65536x65536", "-8x-8", "16x16", "7x7", "30/1", "1/0", "0/0", "-30/-1", "lp-g4d3t1", "8", "none", "frame",
              "frametilemargin", "full", "edge", "band", "default", "custom", "2x2", "20x22", "21x1", "1x23", "a" * 5000, "\xff\xfe"]
    rng = random.Random(7)
    cfg = api.config_alloc()
    assert api.config_init(cfg) == 1
    for _ in range(6000):
        k = rng.choice(names) if rng.random() < 0.9 else "".join(chr(rng.randrange(1, 256)) for _ in range(rng.randrange(1, 12)))
        v = rng.choice(values) if rng.random() < 0.8 else "".join(chr(rng.randrange(1, 256)) for _ in range(rng.randrange(0, 40)))
        rc = api.config_parse(cfg, k.encode("latin-1"), v.encode("latin-1"))
        assert rc in (0, 1), (k, v, rc)
        c = cfg.contents
        assert 0 <= c.qp <= 51 and c.intra_period >= 0 and 0 <= c.owf <= 64 and c.wpp in (0, 1), (k, v, c.qp, c.intra_period, c.owf, c.wpp)
        assert 0 <= c.width <= 16384 and 0 <= c.height <= 16384 and c.framerate_num >= 0 and c.framerate_denom >= 0, (k, v, c.width, c.height, c.framerate_num, c.framerate_denom)
    assert api.config_destroy(cfg) == 1


def test_pictures_and_chunks(lib):
    api = lib.kvz_api_get(8).contents
    pic = api.picture_alloc(64, 48)
    p = pic.contents
    assert (p.width, p.height, p.stride, p.refcount) == (64, 48, 64, 1)
    assert p.u - p.y == 64 * 48 and p.v - p.u == 64 * 48 // 4          # planar, stride == width (kvazaarfilter.cpp:410-418)
    C.memset(p.y, 7, 64 * 48 * 3 // 2)
    api.picture_free(pic)
    api.picture_free(None)
    api.chunk_free(None)
    assert not api.picture_alloc(63, 48) and not api.picture_alloc(0, 0)


def test_opening_codecs_without_a_gpu_fails_loudly(lib):
    if lib.kvzx_device_count() > 0:
        pytest.skip("a GPU is present: covered by the -m gpu tests")
    api = lib.kvz_api_get(8).contents
    cfg = api.config_alloc()
    api.config_init(cfg)
    api.config_parse(cfg, b"input-res", b"128x64")
    assert not api.encoder_open(cfg)                     # no CPU fallback
    api.config_destroy(cfg)
    h = lib.libOpenHevcInit(1, 2)
    assert lib.libOpenHevcStartDecoder(h) == -1
    assert lib.libOpenHevcDecode(h, b"\x00\x00\x00\x01\x40\x01", 6, 0) < 0
    lib.libOpenHevcClose(h)
    from kvazzup_amd.pipeline import Pipeline
    with pytest.raises(RuntimeError):
        Pipeline(128, 64)
