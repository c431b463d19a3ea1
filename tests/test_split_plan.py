"""CPU: the plan that cuts heavy sort tiles into parts (emap_device.h: SplitView, emap_binned.hip: k_bin_scan's tail + tile_work), restated
in Python from the constants of the header and checked for the invariants the kernels rely on -- for ANY tile totals, any capacity the
host launched and any order in which the scan's threads reserve list entries:
  * a tile is either reduced whole by its own workgroup, or by exactly np workgroups (its own + np - 1 listed parts) whose record
    ranges tile [R0, R1) without gap or overlap -- np is what the last-arriving part compares its ticket with, and it travels in the
    high half of the tile's slot word: a tile takes as many parts as the launch has room for (round 5: a standing pool of extra
    workgroups splits the heavy tiles of a scene the host has not heard of yet), 2 <= np <= split_parts(n);
  * no more parts are listed than extra workgroups were launched; a tile that finds no room at all falls back to its own workgroup;
  * every list entry below the listed count is written in this frame (a part, or the empty marker), never stale.
The GPU tests (tests/test_hip_terrain.py) check the kernels' results bit for bit; this pins the arithmetic they share with the host."""
import os
import re

import numpy as np
from hypothesis import given, settings, strategies as st

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "elevation_mapping_cupy_amd", "csrc", "emap_device.h")


def _const(name):
    m = re.search(r"#define\s+%s\s+(\d+)u?\b" % name, open(HDR).read())
    assert m, name
    return int(m.group(1))


CAP, MAX_PARTS, MAX_SLOTS, MAX_EXTRA = (_const(n) for n in ("SPLIT_CAP", "SPLIT_MAX_PARTS", "SPLIT_MAX_SLOTS", "SPLIT_MAX_EXTRA"))
NONE = 0xFFFFFFFF


def split_parts(n):
    return max(1, min(MAX_PARTS, (n + CAP - 1) // CAP))


def scan_tail(totals, cap, sub, order):
    """k_bin_scan's tail: tile_start, and for the heavy tiles (visited in `order`: the threads reserve with LDS atomics in any order)
    a slot and list entries"""
    start = np.concatenate([[0], np.cumsum(totals)]).astype(np.int64)
    extra = np.full(max(cap, 1), 0xDEADBEEF, np.uint32)      # stale contents of the previous frame
    slot_of = {}
    n_slot = n_extra = 0
    for tt in order:
        n = int(totals[tt])
        if n <= CAP:
            continue
        want = split_parts(n) - 1
        s0, e0 = n_slot, n_extra
        n_slot += sub; n_extra += want
        take = min(want, max(cap - e0, 0))
        fits = s0 + sub <= MAX_SLOTS and take > 0
        for q in range(1, take + 1):
            extra[e0 + q - 1] = ((tt << 8) | q) if fits else NONE
        slot_of[tt] = (s0 | ((take + 1) << 16)) if fits else NONE
    return start, extra, slot_of, min(n_extra, cap), n_extra


def tile_work(block, cap, sub, n_tiles, start, extra, slot_of, n_listed):
    """emap_binned.hip: tile_work<true> -- what workgroup `block` of the grid (cap * sub extras in front, then one per tile) reduces"""
    eg = cap * sub
    part = 0
    if block >= eg:
        t, sb = block - eg, 0          # (sub == 1 mapping; stacked tiles only change which tile of the bin a workgroup filters for)
        if sub > 1:
            l = block - eg
            g, r = divmod(l, 8 * sub)
            sb, t = r >> 3, g * 8 + (r & 7)
        if t >= n_tiles:
            return None
    else:
        e, sb = divmod(block, sub)
        if e >= n_listed or extra[e] == NONE:
            return None
        assert extra[e] != 0xDEADBEEF, "a launched extra workgroup read a stale list entry"
        t, part = int(extra[e]) >> 8, int(extra[e]) & 255
    R0, R1 = int(start[t]), int(start[t + 1])
    n = R1 - R0
    nparts, r0, r1, slot = 1, R0, R1, NONE
    if n > CAP and slot_of.get(t, NONE) != NONE:
        nparts = slot_of[t] >> 16
        assert 2 <= nparts <= split_parts(n)
        slot = (slot_of[t] & 0xFFFF) + sb
        ln = (n + nparts - 1) // nparts
        r0 = min(R1, R0 + part * ln); r1 = min(R1, r0 + ln)
    return t, sb, part, nparts, r0, r1, slot


def test_constants_fit_their_encodings():
    assert MAX_PARTS < 256                                     # the part index is 8 bits of a list entry
    assert MAX_SLOTS + 4 < 1 << 16 and MAX_PARTS < 1 << 15    # slot | parts << 16 in one word, never the empty marker
    assert 16384 << 8 < NONE                                   # (bin << 8) | part never looks like the empty marker
    assert 16 * 1024 * 1024 // CAP <= MAX_EXTRA                # the parts a 16 M-point frame can need fit the list


@settings(max_examples=200, deadline=None)
@given(st.data())
def test_every_record_is_reduced_exactly_once(data):
    n_tiles = data.draw(st.integers(1, 40))
    sub = data.draw(st.sampled_from([1, 2, 4]))
    heavy = st.integers(CAP + 1, 40 * CAP)
    totals = np.array(data.draw(st.lists(st.one_of(st.integers(0, CAP), heavy, st.just(CAP), st.just(CAP + 1), st.just(MAX_PARTS * CAP + 7)),
                                         min_size=n_tiles, max_size=n_tiles)), np.int64)
    cap = 8 * data.draw(st.integers(0, 40))                    # what the host launched: a multiple of 8, possibly too few (or none)
    order = data.draw(st.permutations(range(n_tiles)))
    start, extra, slot_of, n_listed, need = scan_tail(totals, cap, sub, order)
    assert n_listed <= cap and need == sum(split_parts(int(n)) - 1 for n in totals if n > CAP)
    n_grid_tiles = n_tiles if sub == 1 else ((n_tiles + 7) // 8) * 8 * sub
    arrivals, covered = {}, {}
    for b in range(cap * sub + n_grid_tiles):
        w = tile_work(b, cap, sub, n_tiles, start, extra, slot_of, n_listed)
        if w is None:
            continue
        t, sb, part, nparts, r0, r1, slot = w
        assert 0 <= part < nparts <= MAX_PARTS and r0 <= r1
        key = (t, sb)
        arrivals.setdefault(key, []).append(nparts)
        covered.setdefault(key, []).append((r0, r1))
        if nparts > 1:
            assert slot != NONE and slot < MAX_SLOTS
    slots_seen = {}
    for t in range(n_tiles):
        for sb in range(sub):
            key = (t, sb)
            assert key in arrivals, "a tile without a workgroup"
            nps = set(arrivals[key])
            assert len(nps) == 1
            nparts = nps.pop()
            assert len(arrivals[key]) == nparts, "the ticket would wait for %d parts, %d workgroups come" % (nparts, len(arrivals[key]))
            spans = sorted(covered[key])
            assert spans[0][0] == start[t] and spans[-1][1] == start[t + 1]
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])), "gap or overlap between the parts"
            if nparts > 1:
                s = (slot_of[t] & 0xFFFF) + sb
                assert s not in slots_seen, "two tiles share a slot"
                slots_seen[s] = key
