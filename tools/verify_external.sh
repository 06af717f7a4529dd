#!/bin/bash
# tools/verify_external.sh -- check the committed bitstreams (tests/golden/streams/*.hevc) with a decoder that is NOT part of this repository.
# Needs ffmpeg (any build with the native hevc decoder) and python3; nothing else.  Two independent checks per stream:
#   1. the decoder's own verification of the decoded picture hash SEI (MD5 per picture and colour component, H.265 D.2.19) that the HIP
#      encoder wrote: libavcodec compares it when asked to (-err_detect +crccheck+explode turns a mismatch into a failed run);
#   2. the MD5 of every decoded picture (cropped I420) against tests/golden/streams/index.json, which holds what this repository's CPU
#      checker (oracle/) and its HIP decoder produce for the same stream (tests/test_golden_streams.py).
# A stream that passes both decodes to exactly the pictures the encoder reconstructed -- which pins encoder AND checker to the standard
# as implemented by that external decoder.  (No such decoder exists in the build pool, which is why this is a script and not a test.)
set -e
cd "$(dirname "$0")/../tests/golden/streams"
command -v ffmpeg > /dev/null || { echo "ffmpeg not found"; exit 2; }
fail=0
for f in *.hevc; do
  name=${f%.hevc}
  if ! ffmpeg -v error -err_detect +crccheck+explode -i "$f" -f null - 2> /tmp/verify_external.err; then echo "FAIL $name: decoder reported errors"; cat /tmp/verify_external.err; fail=1; continue; fi
  if grep -qi "mismatch\|md5" /tmp/verify_external.err; then echo "FAIL $name: hash SEI mismatch"; cat /tmp/verify_external.err; fail=1; continue; fi
  ffmpeg -v error -i "$f" -pix_fmt yuv420p -f framemd5 - 2> /dev/null | grep -v '^#' | awk -F, '{gsub(/ /, "", $NF); print $NF}' > /tmp/verify_external.md5
  if python3 - "$name" <<'PY'
import json, sys
want = json.load(open("index.json"))[sys.argv[1]]["frame_md5"]
got = [l.strip() for l in open("/tmp/verify_external.md5") if l.strip()]
sys.exit(0 if got == want else 1)
PY
  then echo "ok   $name ($(wc -l < /tmp/verify_external.md5) pictures)"; else echo "FAIL $name: picture MD5s differ from index.json"; fail=1; fi
done
exit $fail
