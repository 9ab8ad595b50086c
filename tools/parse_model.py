#!/usr/bin/env python3
"""Executable model of the segment-parallel Rice parser used by k_decode_frames (sela_decode.hip).

A Golomb-Rice stream is a serial bit parse -- where codeword i+1 starts depends on codeword i -- but it
resynchronises: a parser dropped at an arbitrary bit lands on a true codeword boundary after a few
codewords (inside a unary run it is in step at the next terminator; inside a remainder field it has a
chance of ~E[quotient]/k per codeword).  The kernel gives every lane of a wave a ZONE of the stream:

  phase A  lane i parses from the first bit of its zone to the end of the zone, setting one bit per
           codeword start in a bitmap (only lane i writes its zone's words);
  phase B  lane i keeps going into the following zones until it stands on a start some later lane
           marked: from there on the two trajectories are the same one.  m_i = that position;
  resolve  the true trajectory begins at the stream's first bit in lane 0; it follows lane 0's path to
           m_0, which lies in the zone of some lane j (normally 1) and ON lane j's path, then lane j's
           path to m_j, ...  Lanes that are skipped were never in step and decode nothing.  Every lane
           on the chain counts its codewords from its true entry (bitmap popcount) -> exclusive scan
           -> index of its first value;
  pass 2   the lanes on the chain decode their codewords from their true entry, in parallel.

The coefficient stream (<= 100 codewords, other k) is parsed by the first C lanes in the same loops.
This file is the algorithm only (positions, marks, chain); tests/test_host_logic.py runs it against the
CPU oracle on encoder output and on hand-built streams.  Reference semantics: src/rice/rice_decoder.cpp:21-52.
"""
from __future__ import annotations

import numpy as np

WAVE = 64
MARGIN_WORDS = 4


def bits32(words, pos):
    """32 stream bits starting at bit `pos` (stream bit t = bit t % 32 of word t // 32); zero beyond the array."""
    w, sh = pos >> 5, pos & 31
    lo = int(words[w]) if w < len(words) else 0
    hi = int(words[w + 1]) if w + 1 < len(words) else 0
    return ((lo | (hi << 32)) >> sh) & 0xFFFFFFFF


def unzigzag(u):
    return -((u + 1) >> 1) if (u & 1) else (u >> 1)


class Lane:
    __slots__ = ("pos", "zone_end", "stream_end", "k", "in_run", "nA", "nB", "m", "done", "entry", "first_word", "head")


def step(words, ln):
    """Advance a lane over (part of) one codeword.  Returns True when a codeword was completed."""
    x = bits32(words, ln.pos)
    if x == 0xFFFFFFFF:  # 32 more ones: stay inside the codeword
        ln.pos += 32
        ln.in_run = True
        return False
    t = (~x & (x + 1)).bit_length() - 1  # trailing ones
    ln.pos += t + 1 + ln.k
    ln.in_run = False
    return True


def make_lanes(cw, rw, ck, rk, n_coef_lanes):
    """Zones in the unified bit space of the subframe's aligned words: coefficient stream = bits
    [24, 24 + 32 cw), residue stream = bits [32 (cw + 2), 32 (cw + 2 + rw))."""
    lanes = []
    C, R = n_coef_lanes, WAVE - n_coef_lanes
    ce, rs = 24 + 32 * cw, 32 * (cw + 2)
    zc = max(1, -(-(cw + 1) // C))
    zr = max(1, -(-rw // R))
    for i in range(WAVE):
        ln = Lane()
        if i < C:
            first = min(i * zc, cw + 1)
            ln.first_word = first
            ln.entry = 24 if i == 0 else 32 * first
            ln.zone_end = 32 * min((i + 1) * zc, cw + 1)
            ln.stream_end, ln.k, ln.head = ce, ck, i == 0
        else:
            r = i - C
            first = min(r * zr, rw)
            ln.first_word = cw + 2 + first
            ln.entry = rs + 32 * first
            ln.zone_end = rs + 32 * min((r + 1) * zr, rw)
            ln.stream_end, ln.k, ln.head = rs + 32 * rw, rk, r == 0
        ln.pos, ln.in_run, ln.nA, ln.nB, ln.m, ln.done = ln.entry, False, 0, 0, None, False
        lanes.append(ln)
    return lanes, zc, zr


def zone_of(pos, cw, zc, zr, C):
    w = pos >> 5
    if w < cw + 2:
        return min(w // zc, C - 1)
    return C + (w - (cw + 2)) // zr


def parse(words, cw, rw, ck, rk, order, n_values=2048, n_coef_lanes=4, stats=None):
    """-> (q[order], residues[n_values], overrun_coef, overrun_res).  `words` = the subframe's aligned words
    from the one holding the coefficient word count on (coefficient stream starts at bit 24)."""
    words = np.concatenate([np.asarray(words, np.uint32)[: cw + 2 + rw], np.zeros(MARGIN_WORDS, np.uint32)])
    C = n_coef_lanes
    lanes, zc, zr = make_lanes(cw, rw, ck, rk, C)
    marks = set()

    # ---- phase A: own zone, marking ----------------------------------------------------------------
    it_a = 0
    while True:
        act = [ln for ln in lanes if ln.pos < min(ln.zone_end, ln.stream_end)]
        if not act:
            break
        it_a += 1
        for ln in act:
            if not ln.in_run:
                marks.add(ln.pos)
                ln.nA += 1
            step(words, ln)

    # ---- phase B: continue until standing on a later lane's mark ---------------------------------------
    it_b = 0
    while True:
        act = [ln for ln in lanes if ln.m is None]
        if not act:
            break
        it_b += 1
        for ln in act:
            if not ln.in_run:
                if ln.pos >= ln.stream_end:
                    ln.m = -1  # END
                    continue
                if ln.pos in marks:
                    ln.m = ln.pos
                    continue
                ln.nB += 1
            step(words, ln)

    # ---- resolve: walk the chain(s) ------------------------------------------------------------------------
    out_q = np.zeros(order, np.int32)
    out_r = np.zeros(n_values, np.int32)
    overrun = [False, False]
    it_2 = 0
    hops = 0
    for head, need, out, which in ((0, order, out_q, 0), (C, n_values, out_r, 1)):
        cur, e, idx = head, lanes[head].entry, 0
        jobs = []
        while True:
            ln = lanes[cur]
            hops += 1
            own = sum(1 for p in marks if e <= p < ln.zone_end and p >= 32 * ln.first_word) if e < ln.zone_end else 0
            cnt = own + ln.nB
            jobs.append((cur, e, idx, cnt))
            idx += cnt
            if ln.m < 0:
                break
            nxt = zone_of(ln.m, cw, zc, zr, C)
            assert nxt > cur, (nxt, cur)
            cur, e = nxt, ln.m
        # ---- pass 2 ---------------------------------------------------------------------------------------
        for cur, e, idx0, cnt in jobs:
            ln = lanes[cur]
            pos, k = e, ln.k
            n_here = max(0, min(cnt, need - idx0))
            it_2 = max(it_2, n_here)
            for j in range(n_here):
                ones = 0
                while True:
                    x = bits32(words, pos)
                    if x != 0xFFFFFFFF:
                        break
                    ones += 32
                    pos += 32
                t = (~x & (x + 1)).bit_length() - 1
                ones += t
                pos += t + 1
                rem = 0
                field = bits32(words, pos) & ((1 << k) - 1) if k else 0
                for b in range(k):  # remainder is MSB first in the stream
                    rem = (rem << 1) | ((field >> b) & 1)
                pos += k
                u = ((ones << k) | rem) & 0xFFFFFFFF
                out[idx0 + j] = np.int32(unzigzag(u)) if unzigzag(u) < (1 << 31) else np.int32(unzigzag(u) - (1 << 32))
            if n_here == cnt and ln.m >= 0:
                assert pos == ln.m, (pos, ln.m)
            if n_here and pos > ln.stream_end:
                overrun[which] = True
        if idx < need:
            overrun[which] = True
    if stats is not None:
        stats.append((it_a, it_b, it_2, hops))
    return out_q, out_r, overrun[0], overrun[1]


# ---- any stream length: the same walks, segment by segment (sela_decode32.hip: parse_segment / parse_stream_segments) ----------
def parse_segment(words, w0, entry, n_seg_words, stream_end, k, need):
    """One segment: the words [w0, w0 + n_seg_words) of `words`, the true trajectory entering at bit `entry` (0..31) of word w0;
    positions are relative to bit 32 * w0.  A codeword that STARTS in front of the limit min(32 n_seg_words, stream_end) is this
    segment's however far it reaches.  -> (starts listed (<= need), the bit behind the last of them, overrun)."""
    W = n_seg_words
    zr = -(-W // WAVE)
    limit = min(32 * W, stream_end)
    base = 32 * w0

    def word_at(w):
        return int(words[w0 + w]) if w0 + w < len(words) else 0

    def bits_at(pos):
        w, sh = pos >> 5, pos & 31
        return ((word_at(w) | (word_at(w + 1) << 32)) >> sh) & 0xFFFFFFFF

    class L:
        pass

    lanes = []
    for i in range(WAVE):
        ln = L()
        first, end = min(i * zr, W), min((i + 1) * zr, W)
        ln.first_word, ln.end_word = first, end
        ln.pos = entry if i == 0 else 32 * first
        ln.zone_end = min(32 * end, limit)
        ln.in_run, ln.nB, ln.m = False, 0, None
        lanes.append(ln)

    def step(ln):
        x = bits_at(ln.pos)
        if x == 0xFFFFFFFF:
            ln.pos += 32
            ln.in_run = True
            return
        t = (~x & (x + 1)).bit_length() - 1
        ln.pos += t + 1 + k
        ln.in_run = False

    marks = set()
    for ln in lanes:  # phase A
        while ln.pos < ln.zone_end:
            if not ln.in_run:
                marks.add(ln.pos)
            step(ln)
    for ln in lanes:  # phase B (a lane's walk depends on the marks alone: lane by lane is the same as in lock step)
        while ln.m is None:
            if not ln.in_run:
                if ln.pos >= limit:
                    ln.m = -1
                    break
                if ln.pos in marks:
                    ln.m = ln.pos
                    break
                ln.nB += 1
            step(ln)
    # the chain from lane 0, every chain lane's codewords from its true entry
    starts, cur, e = [], 0, entry
    while True:
        ln = lanes[cur]
        own = sorted(p for p in marks if e <= p < 32 * ln.end_word and p >= 32 * ln.first_word) if e < 32 * ln.end_word else []
        pos = e
        walked = []
        for _ in range(len(own) + ln.nB):  # pass 2: the lane walks its codewords again and lists their starts
            walked.append(pos)
            while True:
                x = bits_at(pos)
                if x != 0xFFFFFFFF:
                    break
                pos += 32
            pos += (~x & (x + 1)).bit_length() - 1 + 1 + k
        assert walked[: len(own)] == own, "the marks behind a lane's true entry are its own trajectory"
        starts += [(p, cur) for p in walked]
        if ln.m < 0:
            end_of_last = pos
            break
        assert pos == ln.m and (ln.m >> 5) // zr > cur
        cur, e = (ln.m >> 5) // zr, ln.m
    listed = starts[:need]
    # the bit behind the last listed codeword
    p = listed[-1][0]
    while True:
        x = bits_at(p)
        if x != 0xFFFFFFFF:
            break
        p += 32
    nxt = p + (~x & (x + 1)).bit_length() - 1 + 1 + k
    overrun = nxt > stream_end  # (the listed codewords are in order: the last one decides whether any reaches beyond the stream's end)
    return [q for q, _ in listed], nxt, overrun, base


def parse_segments(words, first_bit, stream_end, k, count, seg_words=1072, seg_values=2048):
    """rice::RiceDecoder on `count` values of the stream in bits [first_bit, stream_end) of `words`, by segments of at most
    seg_words words and seg_values codewords sized by the stream's own words per value -> (values, overrun, segments)."""
    words = np.asarray(words, np.uint32)
    end_word = (stream_end + 31) >> 5
    n_stream_words = max(1, end_word - (first_bit >> 5))
    per_value_x256 = (256 * n_stream_words + count - 1) // max(count, 1) + 1
    out = np.zeros(count, np.int32)
    done, entry, boost, overrun, segments = 0, first_bit, 0, False, 0
    while done < count:
        if entry >= stream_end:
            overrun = True
            break
        need = min(count - done, seg_values)
        w0 = entry >> 5
        guess = (need * per_value_x256) >> 8
        W = max(1, min(min(seg_words, end_word - w0), (guess + (guess >> 4) + 2) << boost))
        starts, nxt, over, base = parse_segment(words, w0, entry & 31, W, stream_end - 32 * w0, k, need)
        assert starts, "a segment whose entry lies in front of its limit lists at least that codeword"
        for i, p in enumerate(starts):
            pos, ones = base + p, 0
            while True:
                x = _bits(words, pos)
                if x != 0xFFFFFFFF:
                    break
                ones += 32
                pos += 32
            t = (~x & (x + 1)).bit_length() - 1
            ones += t
            pos += t + 1
            field = _bits(words, pos) & ((1 << k) - 1) if k else 0
            rem = 0
            for b in range(k):
                rem = (rem << 1) | ((field >> b) & 1)
            u = ((ones << k) | rem) & 0xFFFFFFFF
            v = unzigzag(u)
            out[done + i] = np.int32(v) if v < (1 << 31) else np.int32(v - (1 << 32))
        done += len(starts)
        entry = 32 * w0 + nxt
        overrun |= over
        boost = boost + 1 if len(starts) < need and boost < 12 else boost
        segments += 1
    return out, overrun, segments


def _bits(words, pos):
    w, sh = pos >> 5, pos & 31
    lo = int(words[w]) if w < len(words) else 0
    hi = int(words[w + 1]) if w + 1 < len(words) else 0
    return ((lo | (hi << 32)) >> sh) & 0xFFFFFFFF


def subframe_words(coef_words, res_words):
    """The aligned words of one subframe from the word holding [coef word count u16 | order u8 | first
    coefficient byte]: coefficient words sit 3 bytes in (src/file/sela_file.cpp:121-129)."""
    cw, rw = len(coef_words), len(res_words)
    raw = bytes(3) + np.asarray(coef_words, "<u4").tobytes() + bytes(5) + np.asarray(res_words, "<u4").tobytes()
    return np.frombuffer(raw, "<u4").copy(), cw, rw
