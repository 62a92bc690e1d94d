"""The C boundary lets no C++ exception through (reference convention: tools/environment.hpp:752-784,
interface/jpeg.cpp:205-220).  tests/cxx/alloc_fail.cpp -- a client of include/mijpeg.h and of the source-compatible class JPEG
-- replaces the global operator new with one that throws std::bad_alloc at the N-th allocation, for every N until a whole
host-side decode goes through (parse, boxes, the worker pool, pipelined refinement scans, the self-synchronising decoder): every
call comes back with a code (JPGERR_OUT_OF_MEMORY), the process is never terminated, no thread is left waiting."""
import os
import subprocess

import pytest

from conftest import GOLDEN_DIR, ROOT

SRC = os.path.join(ROOT, "tests", "cxx", "alloc_fail.cpp")
STREAMS = ["pil_200x120_420_dri8.jpg", "xt_129x71_420_R2_rR3_dri3.jpg", "refprog_120x88_420_qv.jpg", "ref_97x61_420_dnl.jpg",
           "pil_80x48_444.jpg", os.path.join("xt_alpha", "a8_residual_hidden.jpg"), os.path.join("xt_lossless", "hdr_ro.jpg"),
           os.path.join("xt_general", "a_r2_gamma.jpg"), os.path.join("damaged", "dri8_ff_run.jpg")]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("alloc_fail") / "alloc_fail")
    subprocess.run(["g++", "-O1", "-w", "-rdynamic", "-I", os.path.join(ROOT, "libjpeg_amd", "csrc"), SRC, "-o", exe, "-L", os.path.join(ROOT, "libjpeg_amd"),
                    "-lmijpeg", "-Wl,-rpath," + os.path.join(ROOT, "libjpeg_amd")], check=True)
    return exe


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("name", STREAMS)
def test_every_allocation_may_fail(harness, name, threads):
    r = subprocess.run([harness, os.path.join(GOLDEN_DIR, name), str(threads), "1"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, MIJPEG_DEVICE="-1"))
    assert r.returncode == 0 and r.stdout.startswith("OK "), (r.stdout[-300:], r.stderr[-1500:])
    # the harness did inject failures, and what came back for them was the reference's code for it -- nothing else
    assert " 0 x other code" in r.stdout and ", 0 x other;" in r.stdout, r.stdout
    assert int(r.stdout.split(";")[-1].split()[0]) > 10
