"""Caches of device-side derived data (packed weights, folded FrozenBatchNorm constants, anchors) that stay correct when
the model is called from several host threads on several HIP streams (pipeline.StreamPipeline).

A cached tensor is produced by kernels enqueued on the stream of whoever fills the cache first.  A second caller on ANOTHER
stream that finds the entry must not read it before those kernels have run: every entry carries the event recorded behind
its fill, and a consumer on a different stream waits for that event (a no-op once the fill has completed).  The fill itself
runs under a lock, so two threads never produce the same entry twice or hand out each other's tensor."""
import threading
from typing import Any, Callable, Hashable, Optional

import torch


def _stream_id(device: Optional[torch.device]):
    if device is None or device.type != "cuda" or not torch.cuda.is_available():
        return None
    return (device.index, torch.cuda.current_stream(device).cuda_stream)


def _record_stream(val: Any, stream) -> None:
    """tell the caching allocator that `stream` reads the cached tensor(s): their blocks are not reused before the work queued
    on it so far has run (an entry may be invalidated / replaced while another stream still has readers in flight)"""
    if isinstance(val, torch.Tensor):
        if val.is_cuda:
            val.record_stream(stream)
    elif isinstance(val, (tuple, list)):
        for v in val:
            _record_stream(v, stream)
    elif isinstance(val, dict):
        for v in val.values():
            _record_stream(v, stream)


class StreamSafeEntry:
    """one value keyed on `key`; `make()` enqueues the work that produces it on the current stream of `device`"""

    def __init__(self):
        self._lock = threading.Lock()
        self.key: Hashable = None
        self.val: Any = None
        self._event = None
        self._filled_on = None

    def invalidate(self) -> None:
        """drop the value.  The cached tensors belong to the caching allocator's pool of the stream that FILLED the entry; `get`
        marks them as used by every other stream that receives them (``record_stream``), so a block is not handed out again
        before those streams' queued kernels have run.  Work already queued on the FILL stream is ordered by that stream itself."""
        with self._lock:
            self.key = self.val = self._event = self._filled_on = None

    # The caches hold DERIVED data (packed weights, folded constants, anchors) plus a lock and a HIP event, neither of which can be
    # copied or pickled.  A copied / unpickled module (copy.deepcopy for EMA / AveragedModel, torch.save(model), a model object sent
    # to a worker process) gets a fresh, EMPTY entry and refills it on its first call.
    def __deepcopy__(self, memo):
        new = type(self)()
        memo[id(self)] = new
        return new

    def __getstate__(self):
        return {"empty": True}          # (non-empty on purpose: pickle protocols 0 / 1 skip __setstate__ for a falsy state)

    def __setstate__(self, state):
        self.__init__()

    def get(self, key: Hashable, make: Callable[[], Any], device: Optional[torch.device] = None) -> Any:
        with self._lock:
            if self.val is None or key != self.key:
                val = make()
                ev, sid = None, _stream_id(device)
                if sid is not None:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(device))
                self.key, self.val, self._event, self._filled_on = key, val, ev, sid
            val, ev, filled_on = self.val, self._event, self._filled_on
        if ev is not None and _stream_id(device) != filled_on:
            cur = torch.cuda.current_stream(device)
            cur.wait_event(ev)
            _record_stream(val, cur)
        return val


class StreamSafeDict:
    """a few keyed entries (anchors per shape); cleared when it grows past `limit`"""

    def __init__(self, limit: int = 8):
        self._lock = threading.Lock()
        self._entries = {}
        self._limit = limit

    def __deepcopy__(self, memo):
        new = type(self)(self._limit)
        memo[id(self)] = new
        return new

    def __getstate__(self):
        return {"limit": self._limit}

    def __setstate__(self, state):
        self.__init__(state.get("limit", 8))

    def get(self, key: Hashable, make: Callable[[], Any], device: Optional[torch.device] = None) -> Any:
        with self._lock:
            e = self._entries.get(key)
            if e is None:
                if len(self._entries) > self._limit:
                    self._entries.clear()
                e = self._entries[key] = StreamSafeEntry()
        return e.get(key, make, device)
