"""bench.py's timing rules: the clock-settling pre-run, windows of exactly K steps between barriers, medians."""
import time

import torch

AGREE = [None]      # set by main() when ranks > 1: all ranks leave the pre-run at the same chunk


def settle_clocks(fn, ms, cap_ms=None):
    """Run the step, untimed, until the device clocks sit on a plateau.  A GPU that was idle takes some 50 ms of
    sustained load before its clocks settle (measured: the same whole forward takes 230 us per batch in the first 30 ms
    after idle and 204 us from then on); the first process on a freshly leased box has been seen to need SECONDS (round 4:
    100-116 us per step for the first ~2 s, 87 us from then on, same kernel).  The K timed steps are a few milliseconds, so
    without this they would measure the ramp of a cold device instead of the steady state of a serving loop.

    Plateau (round-4 verdict, item 5): at least `ms` milliseconds AND the MEDIAN of the last 8 chunks of 64 steps within
    1 % of the median of the 8 chunks before them AND within 1 % of the fastest such 8-chunk median seen so far (chunks
    timed by HIP events on the current stream: the host clock around a 5 ms chunk is itself 1 % noisy).  Medians, because
    single chunks on this device scatter by several per cent from one to the next (BENCH_r04: 5.8 % between windows,
    which the round-4 rule — all of the last 8 chunks within 1 % of each other — could not reach in its 3 s cap); a clock
    that is still creeping moves the 8-chunk median by more than 1 % per 8 chunks or it is not worth waiting for.
    `ms` defaults to 2.5 s (round 5): every process on these boxes runs its first ~2 s of load on a FLAT slower level
    (105 us per step, then 87 — profiles/r05_bench_n1_with_150ms_settle.json: a plateau test alone leaves the pre-run on
    that level after 150 ms and the first window after it is 20 % slow; round 4 only got past it because its stricter
    rule always ran into its 3 s cap).  Gives up after `cap_ms` (default 10 x ms).
    Returns (chunks run, reached the plateau)."""
    if ms <= 0:
        return 0, True
    cap_ms = cap_ms if cap_ms is not None else 10 * ms
    t0 = time.perf_counter()
    chunks = []
    best8 = float("inf")
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(64):
            fn()
        e1.record()
        torch.cuda.synchronize()
        chunks.append(e0.elapsed_time(e1))
        elapsed = (time.perf_counter() - t0) * 1e3
        flat = False
        if len(chunks) >= 8:
            m_last = median(chunks[-8:])
            best8 = min(best8, m_last)
            if len(chunks) >= 16:
                m_prev = median(chunks[-16:-8])
                flat = abs(m_last - m_prev) <= 0.01 * m_last and m_last <= 1.01 * best8
        done = (elapsed >= ms and flat) or elapsed >= cap_ms
        if AGREE[0] is not None:
            done = AGREE[0](done)                 # a step may hold collectives: every rank runs the same number of chunks
        if done:
            SETTLE_NOISE[0] = (max(chunks[-8:]) - min(chunks[-8:])) / median(chunks[-8:]) if len(chunks) >= 8 else None
            return len(chunks), bool(elapsed >= ms and flat)


SETTLE_NOISE = [None]         # (max - min) / median of the last 8 chunks of the latest settle_clocks call


def median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def timed(fn, steps, sync_all):
    """Exactly `steps` calls bracketed by barrier + synchronize; wall ms and HIP-event ms."""
    sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    sync_all()
    return wall_ms, ev0.elapsed_time(ev1)
