#!/bin/bash
# ThreadSanitizer run of the multi-threaded host decoder (no GPU needed).  Exit code != 0 on a data race or a mismatch.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -I"$ROOT/include" "$ROOT/tools/tsan_host.cpp" "$ROOT/libjpeg_amd/csrc/host_decoder.cpp" "$ROOT/libjpeg_amd/csrc/encoder.cpp" -o /tmp/tsan_host
# big enough for the parallel paths: restart intervals, and a scan without restart markers for the speculative decoder
python3 - <<PY
import sys
sys.path.insert(0, "$ROOT")
from libjpeg_amd import synth
open("/tmp/tsan_dri.jpg", "wb").write(synth.synth_jpeg(1280, 720, 3, 85, "420", 4))
open("/tmp/tsan_nodri.jpg", "wb").write(synth.synth_jpeg(1280, 720, 4, 90, "420", 0))
open("/tmp/tsan_prog.jpg", "wb").write(synth.synth_jpeg(640, 480, 5, 85, "420", 8, progressive=True))
# no restart intervals: the scans of the frame run as a pipeline, a scan one MCU row behind the scan it refines
open("/tmp/tsan_prog_nodri.jpg", "wb").write(synth.synth_jpeg(640, 480, 6, 85, "420", 0, progressive=True))
# a tall JPEG XT frame with four hidden residual bits: rows of its residual planes are handed out a twelfth at a time
# (written by the reference encoder where it is built; tests/golden has small ones otherwise)
# beyond 2 MiB: the segment ends of a progressive file from one pass of the pool, the scans' marker searches side by side
open("/tmp/tsan_bigprog.jpg", "wb").write(synth.encode_jpeg(synth.synth_image(2048, 1400, 11), 96, "444", restart_mcus=8, progressive=True))
from oracle import oracle as O
if O.have_reference():
    # codestream boxes of more than a megabyte in 64 KiB segments: copied by the pool when the walk is through (materialize_boxes)
    open("/tmp/tsan_xt_big.jpg", "wb").write(O.reference_encode_hdr(synth.synth_hdr(1280, 960, 6), ["-r", "-q", "90", "-Q", "95", "-h", "-profile", "c", "-r12", "-s", "1x1,2x2,2x2", "-rR", "3", "-z", "4"]))
    open("/tmp/tsan_xt_rR4.jpg", "wb").write(O.reference_encode_hdr(synth.synth_hdr(192, 1536, 5), ["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-s", "1x1,2x2,2x2", "-rR", "4"]))
else:
    import shutil
    shutil.copy("$ROOT/tests/golden/xt_200x120_420_rR4.jpg", "/tmp/tsan_xt_rR4.jpg")
PY
TSAN_OPTIONS="halt_on_error=1 exitcode=66" /tmp/tsan_host /tmp/tsan_dri.jpg /tmp/tsan_nodri.jpg /tmp/tsan_prog.jpg "$ROOT"/tests/golden/pil_200x120_420_dri8.jpg "$ROOT"/tests/golden/ref_75x45_420_dri2.jpg "$ROOT"/tests/golden/xt_129x71_420.jpg "$ROOT"/tests/golden/refprog_64x64_444_dri5.jpg /tmp/tsan_prog_nodri.jpg "$ROOT"/tests/golden/xt_200x120_420_rR4.jpg "$ROOT"/tests/golden/xt_129x71_420_R2_rR3_dri3.jpg /tmp/tsan_xt_rR4.jpg /tmp/tsan_bigprog.jpg $( [ -f /tmp/tsan_xt_big.jpg ] && echo /tmp/tsan_xt_big.jpg )
