"""AudioStreamer for the HIP path.  Same surface as the reference's (vibevoice/modular/streamer.py:
batch_size / stop_signal / timeout, put(audio_chunks, sample_indices), end(sample_indices=None), finished_flags,
audio_queues, iteration, get_stream) -- but put() never blocks the generation loop: the chunk batch is copied into a
pinned ring slot with ONE asynchronous D2H copy on the producing stream, an event marks it, and a background thread
hands finished chunks to the per-sample queues (the reference does `.detach().cpu()` per sample = one stream sync each)."""
import threading
from queue import Queue
from typing import Optional

import torch


class AudioStreamer:
    def __init__(self, batch_size: int, stop_signal=None, timeout: Optional[float] = None, ring_slots: int = 8):
        self.batch_size = batch_size
        self.stop_signal = stop_signal
        self.timeout = timeout
        self.audio_queues = [Queue() for _ in range(batch_size)]
        self.finished_flags = [False for _ in range(batch_size)]
        self._ring = [None] * ring_slots          # pinned host buffers, allocated on first use (shape of the first chunk)
        self._free = Queue()
        for i in range(ring_slots):
            self._free.put(i)
        self._work = Queue()
        self._thread = threading.Thread(target=self._drain, daemon=True)
        self._thread.start()

    # ---- producer side (generation loop) ----
    def put(self, audio_chunks: torch.Tensor, sample_indices: torch.Tensor):
        idxs = [int(i) for i in sample_indices.tolist()]
        live = [(row, idx) for row, idx in enumerate(idxs) if idx < self.batch_size and not self.finished_flags[idx]]
        if not live:
            return
        if not audio_chunks.is_cuda:
            for row, idx in live:
                self._work.put(("chunk", idx, audio_chunks[row].detach().clone(), None, None))
            return
        slot = self._free.get()                                    # back-pressure only if the consumer is > ring_slots behind
        need = audio_chunks.shape
        buf = self._ring[slot]
        if buf is None or buf.shape[1:] != need[1:] or buf.shape[0] < need[0] or buf.dtype != audio_chunks.dtype:
            buf = torch.empty((max(need[0], self.batch_size),) + tuple(need[1:]), dtype=audio_chunks.dtype).pin_memory()
            self._ring[slot] = buf
        buf[:need[0]].copy_(audio_chunks.detach(), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(audio_chunks.device))
        self._work.put(("slot", live, slot, ev, need[0]))

    def end(self, sample_indices=None):
        if sample_indices is None:
            idxs = list(range(self.batch_size))
        else:
            idxs = [int(i.item()) if torch.is_tensor(i) else int(i) for i in sample_indices]
        for idx in idxs:
            if idx < self.batch_size and not self.finished_flags[idx]:
                self.finished_flags[idx] = True
                self._work.put(("end", idx, None, None, None))       # ordered after that sample's pending chunks

    # ---- background thread: event -> per-sample queues, in production order ----
    def _drain(self):
        while True:
            kind, a, b, ev, n = self._work.get()
            if kind == "slot":
                ev.synchronize()
                buf = self._ring[b]
                for row, idx in a:
                    self.audio_queues[idx].put(buf[row].clone(), timeout=self.timeout)
                self._free.put(b)
            elif kind == "chunk":
                self.audio_queues[a].put(b, timeout=self.timeout)
            elif kind == "end":
                self.audio_queues[a].put(self.stop_signal, timeout=self.timeout)

    # ---- consumer side ----
    def __iter__(self):
        return AudioBatchIterator(self)

    def get_stream(self, sample_idx: int):
        if sample_idx >= self.batch_size:
            raise ValueError(f"Sample index {sample_idx} exceeds batch size {self.batch_size}")
        return AudioSampleIterator(self, sample_idx)


class AudioSampleIterator:
    def __init__(self, streamer: AudioStreamer, sample_idx: int):
        self.streamer = streamer
        self.sample_idx = sample_idx

    def __iter__(self):
        return self

    def __next__(self):
        value = self.streamer.audio_queues[self.sample_idx].get(timeout=self.streamer.timeout)
        if value is self.streamer.stop_signal or (not torch.is_tensor(value) and value == self.streamer.stop_signal):
            raise StopIteration()
        return value


class AudioBatchIterator:
    """Yields {sample index: chunk} for every sample that has a chunk ready, until all samples have ended."""

    def __init__(self, streamer: AudioStreamer):
        self.streamer = streamer
        self.active = set(range(streamer.batch_size))

    def __iter__(self):
        return self

    def __next__(self):
        import queue as _q
        import time
        while self.active:
            out = {}
            for idx in sorted(self.active):
                try:
                    v = self.streamer.audio_queues[idx].get(block=False)
                except _q.Empty:
                    continue
                if v is self.streamer.stop_signal or (not torch.is_tensor(v) and v == self.streamer.stop_signal):
                    self.active.discard(idx)
                else:
                    out[idx] = v
            if out:
                return out
            if self.active:
                time.sleep(0.001)
        raise StopIteration()
