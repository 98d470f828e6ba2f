#!/usr/bin/env python3
"""Frame-wide pool simulator for the split SSR trace (k_trace_head / k_trace_march / k_trace_tail): replays the per-ray
step counts of a frame (tools/trace_steps.py, from the oracle) through a persistent march kernel whose waves refill idle
lanes from ONE frame-wide queue of the rays that survive the pinned round, and reports SIMD time in wave-steps per
original wave — the unit of tools/trace_sim.py (ideal 13.96 at 4K, the block-local rounds of round 3: 26.4).

Machine model: `simds` SIMDs, `wps` resident march waves each.  One tick = every wave with a live lane executes one march
step; a SIMD's time for the tick is the number of its waves that stepped, but at least `lat` while any steps (a lone wave
cannot hide the latency of its own dependent texel fetch).  Every K ticks a wave with >= thr idle lanes (or none live)
takes rays from the queue in wave order (cost c_refill wave-steps per refill event); the queue is in tile order, i.e. the
order in which the head kernel appends.

    python tools/trace_pool_sim.py /tmp/sim/steps_4k.npy
"""
import sys

import numpy as np


def queue_of(steps, chunk=(128, 64), tile=8):
    """survivor step counts (after the 16 of the head kernel) in the head kernel's append order: XCD chunks of
    128 x 64 px dealt round-robin are approximated by chunk-major order, 8x8 tiles row-major inside a chunk"""
    H, W = steps.shape
    out = []
    for cy in range(0, H, chunk[1]):
        for cx in range(0, W, chunk[0]):
            blk = steps[cy:cy + chunk[1], cx:cx + chunk[0]]
            for ty in range(0, blk.shape[0], tile):
                for tx in range(0, blk.shape[1], tile):
                    t = blk[ty:ty + tile, tx:tx + tile].ravel().astype(np.int32) - 16
                    out.append(t[t > 0])
    return np.concatenate(out)


def sim(q, simds=1024, wps=6, K=4, thr=8, c_refill=0.6, c_check=0.05, lat=2.5, endgame=None):
    nw = simds * wps
    rem = np.zeros((nw, 64), dtype=np.int32)
    qi = 0
    n = q.size
    simd_time = np.zeros(simds)
    tick = 0
    lanes_live = 0.0
    lanes_slots = 0.0
    while True:
        live = rem > 0
        if tick % K == 0:
            idle = 64 - live.sum(1)
            want = (idle >= thr) | (idle == 64)
            if qi < n and want.any():
                mask = (~live) & want[:, None]
                idx = np.flatnonzero(mask.ravel())  # idle lanes of the refilling waves, in wave order
                m = min(idx.size, n - qi)
                rem.ravel()[idx[:m]] = q[qi:qi + m]
                qi += m
                simd_time += np.bincount(np.unique(idx[:m] // 64) // wps, minlength=simds) * c_refill
                live = rem > 0
            elif qi >= n and endgame is not None:
                # end game: waves with <= endgame live lanes pool their rays (through the queue in HBM) into full waves
                cnt = live.sum(1)
                small = np.flatnonzero((cnt > 0) & (cnt <= endgame))
                if small.size > 1:
                    pool = rem[small][live[small]]
                    rem[small] = 0
                    nfull = (pool.size + 63) // 64
                    tgt = small[:nfull]
                    flat = np.zeros(nfull * 64, dtype=np.int32)
                    flat[:pool.size] = pool
                    rem[tgt] = flat.reshape(nfull, 64)
                    simd_time += np.bincount(small // wps, minlength=simds) * c_refill
                    live = rem > 0
        wave_live = live.any(1)
        if not wave_live.any():
            if qi >= n:
                break
        stepping = np.bincount((np.flatnonzero(wave_live) // wps), minlength=simds).astype(float)
        cost = np.where(stepping > 0, np.maximum(stepping, lat), 0.0)
        simd_time += cost + (c_check * stepping if tick % K == 0 else 0.0)
        lanes_live += live.sum()
        lanes_slots += wave_live.sum() * 64
        rem[live] -= 1
        tick += 1
    return simd_time, lanes_live / max(lanes_slots, 1), tick


def main():
    steps = np.load(sys.argv[1])
    q = queue_of(steps)
    nwaves_orig = steps.size / 64.0
    ideal = q.sum() / 64.0 / nwaves_orig
    print(f"rays {steps.size}  survivors {q.size} ({q.size / steps.size:.3f})  ideal {ideal:.2f} wave-steps per original wave")
    for wps in (4, 6, 8):
        for K in (2, 4, 8):
            for thr in (1, 8, 16):
                for eg in (None, 24):
                    st, util, ticks = sim(q, wps=wps, K=K, thr=thr, endgame=eg)
                    # the kernel ends when the slowest SIMD ends
                    print(f"wps={wps} K={K} thr={thr:2d} endgame={eg}: mean {st.mean() * 1024 / nwaves_orig:.2f}  max {st.max() * 1024 / nwaves_orig:.2f} "
                          f"wave-steps per original wave; lanes {util * 64:.1f}/64; ticks {ticks}", flush=True)


if __name__ == "__main__":
    main()


def sim_generations(q, bounds, c_reload=0.5, simds=1024, clk_step=440.0, clk_chain=1200.0, gap_us=3.0, ghz=2.4):
    """plain launches: generation g marches every ray still alive from step bounds[g] to bounds[g+1] in lock-step waves of
    64 consecutive queue entries (a wave ends when its last lane does), survivors are compacted frame-wide for the next."""
    rem = q.copy()
    total_ws = 0.0
    t_us = 0.0
    rows = []
    for g in range(len(bounds) - 1):
        R = bounds[g + 1] - bounds[g]
        n = rem.size
        if n == 0:
            break
        pad = (-n) % 64
        r = np.concatenate([rem, np.zeros(pad, dtype=rem.dtype)]).reshape(-1, 64)
        wmax = np.minimum(r.max(1), R)
        ws = float(wmax.sum()) + c_reload * r.shape[0]
        useful = float(np.minimum(r, R).sum()) / 64.0
        thr_clk = ws / simds * clk_step
        chain_clk = R * clk_chain
        t = max(thr_clk, chain_clk) / (ghz * 1e3) + gap_us
        rows.append((bounds[g], bounds[g + 1], n, ws, useful / max(ws, 1e-9), t))
        total_ws += ws
        t_us += t
        rem = rem - R
        rem = rem[rem > 0]
    return total_ws, t_us, rows


def gen_report(path):
    steps = np.load(path)
    q = queue_of(steps)
    nwaves_orig = steps.size / 64.0
    for b in ([16, 32, 48, 64, 80], [16, 24, 32, 48, 80], [16, 24, 32, 40, 48, 64, 80], [16, 20, 24, 28, 32, 40, 48, 64, 80], [16, 22, 30, 44, 80], [16, 28, 80], [16, 24, 40, 80]):
        ws, t, rows = sim_generations(q, b)
        print(f"bounds {b}: {ws / nwaves_orig:.2f} wave-steps per original wave, modelled {t:.1f} us")
        for r in rows:
            print(f"    steps {r[0]:2d}-{r[1]:2d}: {r[2]:8d} rays, {r[3] / nwaves_orig:6.2f} ws/orig wave, lane use {r[4]:.2f}, {r[5]:.1f} us")
