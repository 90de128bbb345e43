"""HIP streams that really run side by side.

A HIP stream is bound to one of a handful of hardware queues when it is created (the runtime picks the least-referenced
queue, first one on a tie), and two streams on the same queue execute strictly one after the other.  Which queue a new stream
gets depends on every stream the process created (and destroyed) before, so "create the streams back to back" is not
enough in a long-lived process: the launches meant to overlap (bench.py / bench_secondary.py ``--overlap``, the batch
pipeline of ``generate_data``) then silently serialise.  ``concurrent_streams`` therefore *measures*: a candidate is kept
only if a one-thread spin kernel on it overlaps with the same kernel on every stream already chosen.
"""
from __future__ import annotations

import warnings
from typing import List

import torch

_SPIN = {}  # device -> spin cycles worth ~0.3 ms


def _spin_cycles(dev) -> int:
    key = str(dev)
    if key not in _SPIN:
        cycles = 100_000
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record()
            torch.cuda._sleep(cycles)
            e1.record()
            torch.cuda.synchronize(dev)
            if e0.elapsed_time(e1) >= 0.25:
                break
            cycles *= 4
        _SPIN[key] = cycles
    return _SPIN[key]


def overlap(sa: "torch.cuda.Stream", sb: "torch.cuda.Stream", dev, attempts: int = 3) -> bool:
    """True if work queued on `sa` and `sb` executes concurrently (they sit on different hardware queues).  Two streams on one
    queue can never look concurrent (the second kernel starts after the first has ended), while concurrent ones may look serial
    once in a while (the hardware queue of a stream is set up at its first launch; another process on the card): the answer is
    the best of a few attempts, after one untimed launch on each stream."""
    cycles = _spin_cycles(dev)
    for st in (sa, sb):
        with torch.cuda.stream(st):
            torch.cuda._sleep(1000)
    for _ in range(attempts):
        a0, a1, b1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(sa):
            a0.record(sa)
            torch.cuda._sleep(cycles)
            a1.record(sa)
        with torch.cuda.stream(sb):
            torch.cuda._sleep(cycles)
            b1.record(sb)
        torch.cuda.synchronize(dev)
        if a0.elapsed_time(b1) < 1.5 * a0.elapsed_time(a1):
            return True
    return False


def concurrent_streams(n: int, device, also: List["torch.cuda.Stream"] = ()) -> List["torch.cuda.Stream"]:
    """`n` new streams on `device`, pairwise concurrent and concurrent with the streams in `also`.  Falls back to plain new
    streams (with a warning) if the runtime offers fewer independent queues than asked for."""
    dev = torch.device(device)
    chosen: List[torch.cuda.Stream] = []
    rejected = []  # kept alive until the end: a destroyed stream would hand its queue to the next candidate again
    with torch.cuda.device(dev):
        for _ in range(4 * n + 4):
            if len(chosen) == n:
                break
            c = torch.cuda.Stream(device=dev)
            if all(overlap(o, c, dev) for o in list(also) + chosen):
                chosen.append(c)
            else:
                rejected.append(c)
        if len(chosen) < n:
            warnings.warn(f"only {len(chosen)} of {n} mutually concurrent HIP streams found; the remaining ones share a hardware queue")
            chosen += rejected[: n - len(chosen)]
            chosen += [torch.cuda.Stream(device=dev) for _ in range(n - len(chosen))]
    return chosen
