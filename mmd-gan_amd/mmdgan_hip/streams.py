"""Side streams that do not share a hardware queue with the stream they are meant to overlap.

HIP multiplexes its streams onto a handful of hardware queues (4 per priority level by default) and a queue
executes in order: two streams that land on the same queue do not overlap, and a kernel issued early on one of
them (a weight-gradient launch, a collective) holds up everything issued later on the other.  Which queue a new
stream gets depends on how many streams the process created before - with the engine's four streams the same step
measured 2.43 ms (no sharing), 2.55 ms (power-iteration stream on the main stream's queue) and 2.89 ms
(weight-gradient stream on it), tools/queue_probe.py.  There is no API to ask for a queue, so candidates are
tested: a short kernel on one stream cannot finish while a spin kernel issued before it on the other is still
running iff the two share a queue.
"""
import os

import torch

_cycles_per_ms = {}


def _calibrate(device):
    key = torch.device(device).index
    if key not in _cycles_per_ms:
        torch.cuda._sleep(1000)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(2_000_000)
        e1.record()
        torch.cuda.synchronize(device)
        _cycles_per_ms[key] = 2_000_000 / max(e0.elapsed_time(e1), 1e-3)
    return _cycles_per_ms[key]


def shares_queue(a, b, ms=0.3):
    """True if streams a and b execute in order with respect to each other (same hardware queue)."""
    cyc = _calibrate(a.device)
    torch.cuda.synchronize(a.device)
    es, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(a):
        es.record(a)
        torch.cuda._sleep(int(ms * cyc))
        ea.record(a)
    with torch.cuda.stream(b):
        torch.cuda._sleep(10)
        eb.record(b)
    torch.cuda.synchronize(a.device)
    return es.elapsed_time(eb) > 0.5 * es.elapsed_time(ea)


def distinct_queue_streams(n, device, avoid=(), max_candidates=24):
    """n streams that share a hardware queue neither with the current stream, nor with `avoid`, nor with each other.
    Falls back to untested streams when the runtime has too few queues (GPU_MAX_HW_QUEUES < n + 1)."""
    if n <= 0:
        return []
    with torch.cuda.device(device):
        taken = [torch.cuda.current_stream(device)] + list(avoid)
        chosen, rejected = [], []
        for _ in range(max_candidates):
            cand = torch.cuda.Stream(device=device)
            with torch.cuda.stream(cand):
                torch.cuda._sleep(10)                    # first use creates the queue (milliseconds): not timed
            if any(shares_queue(t, cand) for t in taken + chosen):
                rejected.append(cand)
                continue
            chosen.append(cand)
            if len(chosen) == n:
                return chosen
        return chosen + rejected[:n - len(chosen)]
