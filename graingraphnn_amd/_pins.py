"""Keeps alive what a captured hipGraph reads.

A captured step (training.GraphedTrainStep, rollout hipGraphs) bakes in the device addresses of every tensor
its launches touch.  Several of those tensors are owned only by small evictable caches (constant ones / zero
blocks of train_pack, the reverse CSR of training._topo_cache, the CSR + unit tables of engine._graph_cache);
an evicted entry's memory can be handed out again and the next replay would read garbage.  While a `collect`
context is active, every cache hands a reference of what it returns to the collector, and the capturing
object keeps that list for its own lifetime."""
import contextlib
from typing import List

_sinks: List[list] = []


def note(obj):
    """Called by the caches on every hit or miss; returns `obj`."""
    for s in _sinks:
        s.append(obj)
    return obj


@contextlib.contextmanager
def collect(sink: list):
    _sinks.append(sink)
    try:
        yield sink
    finally:
        for i in range(len(_sinks) - 1, -1, -1):   # by identity: two sinks that hold the same notes compare equal
            if _sinks[i] is sink:
                del _sinks[i]
                break
