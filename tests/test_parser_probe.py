"""CPU: the PRODUCT's slice-data parser (kvazzup_amd/csrc/decoder.hip -- NAL units, parameter sets, slice headers, CABAC parse, merge / AMVP derivation)
through its parse-only hook (no device, no picture out): what it produces for the committed golden streams (twenty-one since round 6: two with 32- and 16-sample coding tree blocks) and for ten streams the checker's
encoder writes here equals, byte for byte (FNV-1a digest over every picture's 4x4 records, tables, transform blocks and level words), what the parser
produced when the whole GPU suite last compared the decoder with the checker's (tests/golden/parser_digests.json, make_parser_digests.py) -- with one
row thread and with four (the WPP hand-over between rows)."""
import json
import os

import pytest

import parser_probe as PP

WANT = json.load(open(os.path.join(PP.HERE, "golden", "parser_digests.json")))


@pytest.mark.parametrize("name", sorted(WANT))
def test_parser_output_is_pinned(name):
    nals = dict(PP.golden_cases())[name] if name.startswith("golden_") else PP.encoded_case(name)
    assert PP.probe(nals, 1) == WANT[name]
    assert PP.probe(nals, 4) == WANT[name]


def test_every_case_has_a_digest():
    assert sorted(WANT) == sorted(n for n, _ in PP.golden_cases()) + sorted(n for n, *_ in PP.ENCODED) or set(WANT) == {n for n, _ in PP.golden_cases()} | {n for n, *_ in PP.ENCODED}


def test_the_hook_never_outputs_a_picture_and_refuses_after_start():
    import ctypes as C
    lib = PP._lib()
    h = lib.libOpenHevcInit(1, 2)
    assert lib.kvzx_decoder_set_parse_only(h, 1) == 1
    assert lib.libOpenHevcStartDecoder(h) == 0
    assert lib.kvzx_decoder_set_parse_only(h, 1) == 0          # too late
    nals = PP.encoded_case("enc_flat_all_skip")
    assert all(lib.libOpenHevcDecode(h, n, len(n), 0) == 0 for n in nals)
    lib.libOpenHevcClose(h)


def test_short_lived_row_pools_close():
    """a decoder whose row pool lives for one all-skip picture: helpers that first run after the destructor's wake-up used to sleep on a generation word
    nobody changed again and libOpenHevcClose never returned (csrc/host_pool.h OrderedPool::worker)"""
    lib = PP._lib()
    nals = PP.encoded_case("enc_flat_all_skip")
    for _ in range(40):
        h = lib.libOpenHevcInit(1, 2)
        assert lib.kvzx_decoder_set_parse_only(h, 8) == 1
        assert lib.libOpenHevcStartDecoder(h) == 0
        for n in nals[:3]:
            assert lib.libOpenHevcDecode(h, n, len(n), 0) >= 0
        lib.libOpenHevcClose(h)


def test_exp_golomb_fields_beyond_31_bits_are_refused():
    """slice_pic_parameter_set_id written as a code word with 32 leading zeros: used to come out of BitReader::ue as a negative int that passed `pps_id > 63`
    and indexed the PPS table in front of its first entry (found by tools/fuzz_parser.py under AddressSanitizer)"""
    lib = PP._lib()
    nals = PP.encoded_case("enc_flat_all_skip")
    params = [n for n in nals if (n[4] >> 1) in (32, 33, 34)]
    assert len(params) >= 3
    for zeros in (31, 32, 33, 40, 64):
        bits = "1" + "0" * zeros + "1" + "1" * 40                      # first_slice_segment_in_pic_flag, then the code word
        bits += "0" * (-len(bits) % 8)
        payload = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
        # (emulation prevention: runs of zero bytes in the payload get their 0x03)
        out, z = bytearray(), 0
        for b in payload:
            if z >= 2 and b <= 3:
                out.append(3); z = 0
            out.append(b); z = z + 1 if b == 0 else 0
        slice_nal = b"\x00\x00\x00\x01" + bytes([1 << 1, 1]) + bytes(out)      # TRAIL_R, layer 0, temporal id 0
        h = lib.libOpenHevcInit(1, 2)
        assert lib.kvzx_decoder_set_parse_only(h, 1) == 1 and lib.libOpenHevcStartDecoder(h) == 0
        for n in params:
            assert lib.libOpenHevcDecode(h, n, len(n), 0) == 0
        assert lib.libOpenHevcDecode(h, slice_nal, len(slice_nal), 0) < 0, zeros
        lib.libOpenHevcClose(h)
