#!/usr/bin/env python3
"""Does an AC refinement scan self-synchronise?  ("the self-synchronising walk you built for sequential scans
applies: one lane per subsequence, hand-over state = (bit position, block index, EOB-run), fixed point".)

A model, CPU only, no product code: N blocks with random non-zero histories (the density of config 5's hidden refinement scans:
about nineteen new coefficients and forty correction bits per block), coded as ONE successive-approximation AC refinement scan
the way T.81 G.1.2.3 / codestream/refinementscan.cpp:584-700 prescribe (run / size symbols with Huffman codes, a sign bit per new
coefficient, one correction bit per history coefficient passed on the way, EOB runs), then decoded
  (a) from the start: the truth -- the bit position at which every block begins;
  (b) by walkers that start at the TRUE bit position of block b but believe they are at block b + d (d = 1, 2, 8): the wrong
      histories.  Do they ever stand at the true start of a block again -- at any block index?  (For the sequential scans of
      DESIGN 4.1 the answer is "within a few blocks": there the bits a block takes do not depend on which block it is.)
  (c) by walkers that know their block index and start a few bits off;
  (d) the fixed-point iteration itself: S subsequences, every lane's start state taken from its predecessor's last round -- how many
      rounds until nothing changes?
Prints a table; profiles/r06/refine_sync_model.txt is its output."""
import random
import sys

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
SS, SE = 1, 63

# a plausible refinement alphabet: runs 0..15 with size 1, ZRL, EOB0..EOB3, and a COMPLETE prefix code for it (Huffman's
# construction over plausible frequencies: every bit string decodes to something, as with the optimised tables of `jpeg -h`)
import heapq

SYMS = [(r, 1) for r in range(16)] + [(15, 0)] + [(r, 0) for r in range(4)]
FREQ = {(r, 1): 0.5 * 0.55 ** r for r in range(16)}
FREQ[(15, 0)] = 0.002
for r in range(4):
    FREQ[(r, 0)] = 0.06 * 0.1 ** r
heap = [(f, i, [s]) for i, (s, f) in enumerate(sorted(FREQ.items()))]
heapq.heapify(heap)
LENS = {s_: 0 for s_ in SYMS}
tick = len(heap)
while len(heap) > 1:
    f1, _, a1 = heapq.heappop(heap)
    f2, _, a2 = heapq.heappop(heap)
    for s_ in a1 + a2:
        LENS[s_] += 1
    heapq.heappush(heap, (f1 + f2, tick, a1 + a2))
    tick += 1
order = sorted(SYMS, key=lambda s_: (LENS[s_], s_))
codes, code, prev = {}, 0, 0
for s_ in order:
    code <<= LENS[s_] - prev
    codes[s_] = (code, LENS[s_])
    prev = LENS[s_]
    code += 1
assert code == 1 << prev, "a Huffman code is complete"
DEC = {(c, l): s for s, (c, l) in codes.items()}
MAXLEN = max(LENS.values())


def make_block():
    """(history mask as a set of positions, list of new coefficient positions) -- history ~40 of 63, new ~19 of the rest"""
    hist = {k for k in range(SS, SE + 1) if random.random() < 0.63}
    free = [k for k in range(SS, SE + 1) if k not in hist]
    new = sorted(k for k in free if random.random() < 0.8)
    return hist, new


def encode(blocks):
    bits = []

    def put(v, n):
        for i in range(n - 1, -1, -1):
            bits.append((v >> i) & 1)

    starts = []
    for hist, new in blocks:
        starts.append(len(bits))
        k = SS
        pend = []  # correction bits of history positions passed since the last symbol

        def flush():
            bits.extend(pend)
            pend.clear()

        for t in new:
            run = 0
            zr = []
            for p in range(k, t):
                if p in hist:
                    zr.append(random.getrandbits(1))
                else:
                    run += 1
                    if run == 16:  # ZRL: sixteen zeros, with the corrections passed so far
                        put(*codes[(15, 0)])
                        bits.extend(zr)
                        zr = []
                        run = 0
            put(*codes[(run, 1)])
            put(random.getrandbits(1), 1)  # sign
            bits.extend(zr)
            k = t + 1
        if k <= SE:  # EOB0: the rest of the block only takes its corrections
            rest = [random.getrandbits(1) for p in range(k, SE + 1) if p in hist]
            if rest or True:
                put(*codes[(0, 0)])
                bits.extend(rest)
    starts.append(len(bits))
    return bits, starts


def decode_block(bits, pos, hist):
    """-> bit position behind the block, or None when the data runs out"""
    k = SS
    n = len(bits)
    while k <= SE:
        c, l, sym = 0, 0, None
        while l < MAXLEN and pos < n:
            c = (c << 1) | bits[pos]
            pos += 1
            l += 1
            if (c, l) in DEC:
                sym = DEC[(c, l)]
                break
        if sym is None:
            return None
        r, s = sym
        if s == 0 and r < 15:  # EOBn (the model never writes runs: n = 0)
            pos += r  # the run's low bits
            pos += sum(1 for p in range(k, SE + 1) if p in hist)
            return pos if pos <= n else None
        if s == 1:
            pos += 1  # sign
        zeros = r if s == 1 else 15
        while k <= SE:
            if k in hist:
                pos += 1
            elif zeros == 0:
                break
            else:
                zeros -= 1
            k += 1
        k += 1
        if pos > n:
            return None
    return pos


blocks = [make_block() for _ in range(N)]
bits, starts = encode(blocks)
# (a) the truth decodes
pos = 0
for b in range(N):
    assert pos == starts[b], (b, pos, starts[b])
    pos = decode_block(bits, pos, blocks[b][0])
assert pos == starts[N]
start_set = {p: b for b, p in enumerate(starts[:-1])}
print(f"model scan: {N} blocks, {len(bits)} bits ({len(bits) / N:.1f} per block), decodes from its start: yes")


def walk(first_block, pos, believed, limit):
    """a walker at bit `pos` that takes the histories of blocks believed, believed + 1, ...: -> (blocks walked until it stands at the
    true start of the block it believes in, or None; blocks walked until it stands at ANY true block start, or None; died)"""
    any_hit = None
    for j in range(limit):
        b = believed + j
        if b >= N:
            return None, any_hit, False
        pos = decode_block(bits, pos, blocks[b][0])
        if pos is None:
            return None, any_hit, True
        if pos in start_set:
            if any_hit is None:
                any_hit = j + 1
            if start_set[pos] == b + 1:
                return j + 1, any_hit, False
    return None, any_hit, False


LIMIT = 2000
print("\n(b) right bit position, wrong block index (200 walkers each, up to 2000 blocks):")
for d in (1, 2, 8):
    sync = anyhit = died = 0
    for t in range(200):
        b = random.randrange(0, N - LIMIT - 10)
        s, a, dead = walk(b, starts[b], b + d, LIMIT)
        sync += s is not None
        anyhit += a is not None
        died += dead
    print(f"  believes block b + {d}: back in step with the real decoder {sync} / 200, stood at some other block's true start {anyhit} / 200, ran out of data {died} / 200")
print("\n(c) right block index, wrong bit position (200 walkers each):")
for off in (1, 3, 17):
    sync = anyhit = died = 0
    dist = []
    for t in range(200):
        b = random.randrange(0, N - LIMIT - 10)
        s, a, dead = walk(b, starts[b] + off, b, LIMIT)
        sync += s is not None
        anyhit += a is not None
        died += dead
        if s is not None:
            dist.append(s)
    dist.sort()
    print(f"  {off:2d} bits late: back in step within 2000 blocks {sync} / 200 (median of those: {dist[len(dist) // 2] if dist else '-'} blocks, "
          f"90th percentile {dist[len(dist) * 9 // 10] if dist else '-'}), ran out of data {died} / 200")

# (d) the fixed point over S subsequences: lane i owns the bits [i * L, (i + 1) * L); its hand-over state is (bit position, block index)
print("\n(d) fixed-point iteration (a lane walks from its start state to the first block start at or behind its subsequence's end and hands that to its successor):")
for S in (16, 64, 256):
    L = (len(bits) + S - 1) // S
    state = [(0, 0)] + [(i * L, int(i * N / S)) for i in range(1, S)]  # guesses: the boundary bit, the proportional block
    truth = []
    for i in range(S):  # what the states would be if every predecessor were right
        tb = next(b for b in range(N + 1) if starts[b] >= i * L)
        truth.append((starts[tb], tb))
    rounds = 0
    while True:
        rounds += 1
        new = list(state)
        for i in range(S - 1):
            p, b = state[i]
            while p is not None and p < (i + 1) * L and b < N:
                p = decode_block(bits, p, blocks[b][0])
                b += 1
            if p is not None:
                new[i + 1] = (p, b)
        if new == state:
            break
        state = new
        if rounds > S + 2:
            break
    right = sum(1 for i in range(S) if state[i] == truth[i])
    print(f"  {S:4d} subsequences: settled after {rounds} rounds, {right} / {S} start states are the real decoder's   (a serial decode is {S} subsequences long)")
