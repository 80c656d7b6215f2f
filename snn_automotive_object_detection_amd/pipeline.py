"""Several batches in flight on one GPU: one host thread + one HIP stream per slot.

The detector's forward has host synchronisations (per-image proposal / detection counts come back to build the reference's
Python lists, rpn.py:493-499, roi_heads.py:1163-1172), so a single thread cannot keep the GPU fed across batch boundaries: while
it waits for one batch's counts nothing of the next batch is enqueued, and the small kernels of the post-processing stages
leave most of the chip idle.  With two slots the small kernels of one batch run beside the big contractions of the other
(end to end on 2 x 1024x2048 images: 139 -> 152 images/s; bench.py ``e2e.two_streams``).  Safe because every C-ABI entry point
enqueues on the stream it is given, the wrappers keep one workspace per (device, stream) (ops._Workspace), and the model's
lazily built device-side caches (packed weights, folded FrozenBatchNorm constants, anchors) are filled under a lock and carry
the event of their fill, which a consumer on another stream waits for (``_cache.StreamSafeEntry``) - a freshly constructed
model may be handed to ``map`` directly (tests/test_gpu_modules.py::test_stream_pipeline_on_a_cold_model)."""
import queue
import threading
from typing import Any, Callable, Iterable, List

import torch


class StreamPipeline:
    """``StreamPipeline(model, slots=2).map(batches)`` -> the model's outputs in the order of ``batches``.

    ``model`` is called as ``model(batch)`` under ``torch.no_grad()`` on the slot's stream; outputs are handed back after the
    slot's stream has been synchronised.  An exception in a slot is re-raised by ``map`` (remaining batches are dropped)."""

    def __init__(self, model: Callable[[Any], Any], slots: int = 2, device: torch.device = None):
        if slots < 1:
            raise ValueError("slots must be >= 1")
        self.model = model
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.streams = [torch.cuda.Stream(self.device) for _ in range(slots)]

    def map(self, batches: Iterable[Any]) -> List[Any]:
        work: "queue.Queue" = queue.Queue()
        n = 0
        for i, b in enumerate(batches):
            work.put((i, b))
            n += 1
        out: List[Any] = [None] * n
        errors: List[BaseException] = []
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))        # the batches were produced on the caller's stream

        def slot(stream):
            try:
                with torch.cuda.device(self.device), torch.no_grad(), torch.cuda.stream(stream):
                    stream.wait_event(ready)
                    while not errors:
                        try:
                            i, b = work.get_nowait()
                        except queue.Empty:
                            break
                        out[i] = self.model(b)
                    stream.synchronize()
            except BaseException as e:                              # re-raised in the caller's thread
                errors.append(e)

        threads = [threading.Thread(target=slot, args=(s,)) for s in self.streams]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return out
