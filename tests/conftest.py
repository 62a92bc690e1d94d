import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_manifest():
    with open(os.path.join(GOLDEN_DIR, "manifest.json")) as f:
        return json.load(f)


MANIFEST = load_manifest()
SMALL_CASES = sorted(k for k, v in MANIFEST.items() if not v.get("big") and "kind" not in v)  # 8-bit integer output
XT_CASES = sorted(k for k, v in MANIFEST.items() if v.get("kind") == "xt_float32")
P12_CASES = sorted(k for k, v in MANIFEST.items() if v.get("kind") == "u16")
BIG_CASES = sorted(k for k, v in MANIFEST.items() if v.get("big"))


def golden_jpeg(name: str) -> bytes:
    with open(os.path.join(GOLDEN_DIR, name + ".jpg"), "rb") as f:
        return f.read()


def golden_pixels(name: str):
    import numpy as np

    ent = MANIFEST[name]
    fn = ent.get("pixels_file")
    if not fn:
        return None
    dtype = {"xt_float32": "<f4", "u16": "<u2"}.get(ent.get("kind"), np.uint8)
    with open(os.path.join(GOLDEN_DIR, fn), "rb") as f:
        return np.frombuffer(f.read(), dtype).reshape(ent["height"], ent["width"], ent["channels"])


def big_jpeg(name: str):
    """Regenerate a hash-only ('big') case; returns None if this box's encoder yields other bytes."""
    import hashlib

    from libjpeg_amd import synth

    ent = MANIFEST[name]
    data = synth.synth_jpeg(ent["width"], ent["height"], ent["seed"], ent["quality"], ent["sub"], ent["dri"])
    if hashlib.sha256(data).hexdigest() != ent["jpeg_sha256"]:
        return None
    return data


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build(ref=False)
    return O
