#!/usr/bin/env python3
"""Lane-utilisation simulator for the SSR march schedule (k_sssr_trace): replays the per-ray step counts of a frame
(tools/trace_steps.py, from the oracle) through candidate schedules and reports wave-steps per original wave after the
pinned round (every ray runs 16 steps there).  Cost unit: one march step of one wave, whatever its lane count.

    python tools/trace_sim.py /tmp/sim/steps_4k.npy
"""
import heapq
import sys

import numpy as np


def blocks_of(steps, bw=32, bh=8):
    H, W = steps.shape
    for by in range(0, H - H % bh, bh):
        for bx in range(0, W - W % bw, bw):
            blk = steps[by:by + bh, bx:bx + bw]
            # wave w owns the 8x8 tile w of the block (row-major over tiles)
            tiles = [blk[ty:ty + 8, tx:tx + 8].ravel() for ty in range(0, bh, 8) for tx in range(0, bw, 8)]
            yield np.concatenate(tiles).astype(np.int32) - 16


def sim_rounds(rem, R=16, overhead=0.0):
    """current kernel: rounds of R steps, live rays compacted between rounds into full waves"""
    live = rem[rem > 0]
    cost = 0.0
    while live.size:
        n = live.size
        for w in range(0, n, 64):
            cost += min(R, int(live[w:w + 64].max())) + overhead
        live = live - R
        live = live[live > 0]
    return cost


def sim_refill(rem, nwaves=4, K=4, thr=1, c_refill=0.4, handoff=0, lifo=False, c_check=0.05):
    """barrier-free pool: the block's unfinished rays sit in one queue; `nwaves` waves march, every K steps a wave
    with >= thr idle lanes (or none active) refills them from the queue (cost c_refill per refill event);
    handoff > 0: a wave left with <= handoff live lanes while the queue is empty and another wave still marches
    pushes its rays back and stops."""
    queue = [int(r) for r in rem if r > 0]
    if lifo:
        queue.reverse()
    qi = 0
    cost = 0.0
    # event-ordered so that concurrent waves draw from the queue in time order
    heap = [(0.0, w) for w in range(nwaves)]
    lanes = [np.zeros(64, dtype=np.int32) for _ in range(nwaves)]
    marching = nwaves
    extra = []  # rays handed back
    while heap:
        t, w = heapq.heappop(heap)
        L = lanes[w]
        idle = int((L <= 0).sum())
        avail = (len(queue) - qi) + len(extra)
        if idle and avail and (idle >= thr or idle == 64):
            take = min(idle, avail)
            idx = np.flatnonzero(L <= 0)[:take]
            for j in idx:
                if extra:
                    L[j] = extra.pop()
                else:
                    L[j] = queue[qi]; qi += 1
            cost += c_refill; t += c_refill
        live = int((L > 0).sum())
        avail = (len(queue) - qi) + len(extra)
        if live == 0:
            if avail == 0:
                marching -= 1
                continue
            heapq.heappush(heap, (t, w)); continue
        if handoff and avail == 0 and live <= handoff and marching > 1:
            extra.extend(int(x) for x in L[L > 0]); L[:] = 0
            marching -= 1
            cost += c_refill
            continue
        k = min(K, int(L.max()))
        L -= k
        cost += k + c_check; t += k + c_check
        heapq.heappush(heap, (t, w))
    return cost


def main():
    steps = np.load(sys.argv[1])
    blocks = list(blocks_of(steps))
    nb = len(blocks)
    if len(sys.argv) > 2:
        blocks = blocks[::int(sys.argv[2])]
        nb = len(blocks)
    norm = nb * 4.0
    ideal = sum(float(b[b > 0].sum()) / 64 for b in blocks) / norm
    print(f"blocks {nb}  ideal {ideal:.2f} wave-steps per original wave")
    print(f"rounds16 (current)          {sum(sim_rounds(b) for b in blocks) / norm:.2f}   with 0.8/round {sum(sim_rounds(b, 16, 0.8) for b in blocks) / norm:.2f}")
    print(f"rounds8                     {sum(sim_rounds(b, 8, 0.8) for b in blocks) / norm:.2f}")
    for nw in (4, 2, 1):
        for K in (2, 4, 8):
            for thr in (1, 16, 32):
                for ho in (0, 24):
                    c = sum(sim_refill(b, nw, K, thr, handoff=ho) for b in blocks) / norm
                    print(f"refill waves={nw} K={K} thr={thr:2d} handoff={ho:2d}: {c:.2f}")




def sim_stream(steps, nwaves_total=3072, waves_per_block=4, K=4, thr_refill=8, thr_admit=16, handoff=16, c_refill=0.25,
               c_retire=0.15, c_epi=3.0, epi_batch=32, seed=0, c_check=0.05):
    """Streaming schedule: every wave runs a sequence of 8x8 tiles; survivors of a tile's pinned round wait in the wave's
    queue, idle march lanes are refilled from it, a new tile is admitted when the queue is empty and >= thr_admit lanes idle;
    finished rays collect in a done list whose epilogue runs in batches.  Returns wave-step equivalents per tile spent on
    (march after the pinned round + refill/retire overhead + epilogue), to compare with rounds: march + 3.0 (one full
    epilogue per tile).  End game: a wave out of tiles with <= handoff live rays gives them to the last wave of its block."""
    H, W = steps.shape
    tiles = [steps[y:y + 8, x:x + 8].ravel().astype(np.int32) - 16 for y in range(0, H - H % 8, 8) for x in range(0, W - W % 8, 8)]
    nt = len(tiles)
    nblocks = nwaves_total // waves_per_block
    total = 0.0
    tails = 0.0
    for b in range(nblocks):
        blist = list(range(b, nt, nblocks))  # strided static list of the block
        nxt = 0
        orphans = []
        arrived = 0
        wave_state = []
        for w in range(waves_per_block):
            wave_state.append(dict(L=np.zeros(64, dtype=np.int32), q=[], done=0, t=0.0, out=False))
        heap = [(0.0, w) for w in range(waves_per_block)]
        while heap:
            t, w = heapq.heappop(heap)
            S = wave_state[w]
            L = S['L']
            cost = 0.0
            # retire
            fin = int(((L <= 0) & (S.get('occ', np.zeros(64, bool)))).sum()) if 'occ' in S else 0
            if 'occ' not in S:
                S['occ'] = np.zeros(64, dtype=bool)
            occ = S['occ']
            fin_mask = occ & (L <= 0)
            nf = int(fin_mask.sum())
            if nf:
                S['done'] += nf; occ[fin_mask] = False; cost += c_retire
            idle = 64 - int(occ.sum())
            # admit
            if not S['q'] and idle >= thr_admit and nxt < len(blist):
                tile = tiles[blist[nxt]]; nxt += 1
                S['done'] += int((tile <= 0).sum())
                S['q'] = [int(r) for r in tile if r > 0]
            # epilogue batches
            while S['done'] >= epi_batch:
                n = min(64, S['done']); S['done'] -= n; cost += c_epi
            # refill
            if S['q'] and (idle >= thr_refill or idle == 64):
                take = min(idle, len(S['q']))
                idx = np.flatnonzero(~occ)[:take]
                for j in idx:
                    L[j] = S['q'].pop(); occ[j] = True
                cost += c_refill
            live = int(occ.sum())
            if live == 0 and not S['q']:
                if nxt < len(blist):
                    total += cost; heapq.heappush(heap, (t + cost + 0.01, w)); continue
                # out of work
                arrived += 1
                if arrived == waves_per_block and orphans:
                    S['q'] = orphans; orphans = []
                    total += cost; heapq.heappush(heap, (t + cost, w)); continue
                if S['done']:
                    cost += c_epi; S['done'] = 0
                total += cost
                continue
            if nxt >= len(blist) and not S['q'] and live <= handoff and arrived < waves_per_block - 1:
                arrived += 1
                orphans.extend(int(x) for x in L[occ]); occ[:] = False
                if S['done']:
                    cost += c_epi; S['done'] = 0
                total += cost + c_refill
                continue
            k = min(K, int(L[occ].max()))
            L[occ] -= k
            cost += k + c_check
            if nxt >= len(blist) and not S['q']:
                tails += k * (1 - live / 64.0)
            total += cost
            heapq.heappush(heap, (t + cost, w))
    return total / nt, tails / nt


def stream_report(path, sub=1):
    steps = np.load(path)
    if sub > 1:  # a sub-frame keeps tiles-per-wave when nwaves_total shrinks with it
        pass
    for nw, wpb in ((3072, 4), (4096, 4), (3072, 8), (2048, 4)):
        for K in (2, 4):
            for thr_refill, thr_admit in ((8, 16), (16, 16), (16, 32), (1, 8)):
                c, tl = sim_stream(steps, nw, wpb, K, thr_refill, thr_admit)
                print(f"stream waves={nw} per_block={wpb} K={K} refill>={thr_refill} admit>={thr_admit}: {c:.2f} per tile (tail waste {tl:.2f})")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "stream":
        stream_report(sys.argv[1])
    else:
        main()
