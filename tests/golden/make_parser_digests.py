#!/usr/bin/env python3
"""tests/golden/make_parser_digests.py -- digests of what the PRODUCT's slice-data parser (csrc/decoder.hip, host half of the decoder) produces for the
committed golden streams and for streams the checker's encoder writes here: per stream {pictures, transform blocks, level words, FNV-1a digest of every
picture's records / tables / blocks / levels}.  Written with a library whose parser had just passed the whole GPU suite against the checker's decoder
(round 5: the parser before its registers moved into locals); tests/test_parser_probe.py (CPU) then pins every later parser to the same output.

  KVAZZUP_AMD_LIBRARY=<that library> python tests/golden/make_parser_digests.py > tests/golden/parser_digests.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import parser_probe as PP                                                    # noqa: E402

if __name__ == "__main__":
    out = {}
    for name, nals in PP.all_cases():
        out[name] = PP.probe(nals, 1)
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()
