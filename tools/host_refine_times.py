#!/usr/bin/env python3
"""Config 5 with hidden residual bits (`-rR 4`), the host side alone: where a read of the 4K stream spends its time
(MIJPEG_READ_TIMES / MIJPEG_SCAN_TIMES traces of the library), with the AC refinement chains on bit masks
(parse_block_refine_ac) and with the blocks themselves as the medium (MIJPEG_NO_DEFERRED_REFINE=1), alternating.
Needs oracle/_ref/jpeg (the reference encoder writes the stream); no device.

    python tools/host_refine_times.py [--size 4k] [--rounds 3]"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import os, sys, time
sys.path.insert(0, %r)
if os.environ.get("BIND_NODE"):  # this process and the library's worker pool on one socket
    spec = open("/sys/devices/system/node/node%%s/cpulist" %% os.environ["BIND_NODE"]).read().strip()
    cpus = set()
    for part in spec.split(","):
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    os.sched_setaffinity(0, cpus & os.sched_getaffinity(0))
from libjpeg_amd import api
data = open(sys.argv[1], "rb").read()
d = api.Decoder(None)
ts = []
for i in range(int(sys.argv[2])):
    t = time.perf_counter(); d.read(data, entropy="host"); ts.append((time.perf_counter() - t) * 1e3)
print("reads ms", " ".join("%%.1f" %% x for x in ts), "| entropy decoders %%.2f ms | threads %%d" %% (d.timing()["huffman"] * 1e3, api.default_threads()), flush=True)
""" % ROOT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="4k")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reads", type=int, default=8)
    a = ap.parse_args()
    import bench
    from libjpeg_amd import synth
    from oracle import oracle as O
    W, H = bench.SIZES[a.size] if a.size in bench.SIZES else tuple(int(v) for v in a.size.split("x"))
    hdr = synth.synth_hdr(W, H, 99)
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        files = {}
        for name, extra in (("r12_rR4", ["-rR", "4"]), ("r12", [])):
            blob = O.reference_encode_hdr(hdr, bench.XT_ARGS + extra)
            files[name] = os.path.join(d, name + ".jpg")
            open(files[name], "wb").write(blob)
            print(name, len(blob), "bytes", flush=True)
        script = os.path.join(d, "child.py")
        open(script, "w").write(CHILD)

        def run(name, env, reads):
            e = dict(os.environ, **env)
            r = subprocess.run([sys.executable, script, files[name], str(reads)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            return r.stdout.strip(), r.stderr

        print("--- one traced read of each flavour (after a first one that pays for the allocations)")
        for label, env in (("masks", {}), ("blocks", {"MIJPEG_NO_DEFERRED_REFINE": "1"})):
            out, err = run("r12_rR4", dict(env, MIJPEG_READ_TIMES="1", MIJPEG_SCAN_TIMES="1"), 2)
            lines = err.strip().splitlines()
            half = [i for i, l in enumerate(lines) if l.startswith("read:")]
            print(label + ":")
            print("\n".join("   " + l for l in (lines[half[0] + 1:] if len(half) > 1 else lines)))
        print("--- alternating, %d reads each" % a.reads)
        for _ in range(a.rounds):
            for label, env in (("masks ", {}), ("blocks", {"MIJPEG_NO_DEFERRED_REFINE": "1"}), ("blocks, first pass in order", {"MIJPEG_NO_DEFERRED_REFINE": "1", "MIJPEG_NO_SPEC_FIRST_PASS": "1"})):
                print(label, run("r12_rR4", env, a.reads)[0], flush=True)
        print("r12 (no hidden bits)", run("r12", {}, a.reads)[0])
        # the slow reads (+25-30 ms, in whatever step they hit): a process that is not bound to a socket?
        try:
            print("--- /proc/sys/kernel/numa_balancing =", open("/proc/sys/kernel/numa_balancing").read().strip(), "; masks, %d reads, bound to node 0 / not bound" % (2 * a.reads))
        except OSError:
            pass
        for _ in range(2):
            print("bound    ", run("r12_rR4", {"BIND_NODE": "0"}, 2 * a.reads)[0], flush=True)
            print("not bound", run("r12_rR4", {}, 2 * a.reads)[0], flush=True)


if __name__ == "__main__":
    main()
