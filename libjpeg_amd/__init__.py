"""libjpeg_amd -- MI355X-native JPEG block-decode path behind the thorfdbg/libjpeg decode API.

    csrc/            HIP kernels (gfx950), host Huffman decoder, C ABI (include/mijpeg.h)
    api.py           ctypes binding used by tests and bench.py
    synth.py         deterministic synthetic images / streams
"""
__all__ = ["api", "synth"]
