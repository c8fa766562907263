"""AudioStreamer / AsyncAudioStreamer for the HIP path.  Same surface as the reference's
(vibevoice/modular/streamer.py: batch_size / stop_signal / timeout, put(audio_chunks, sample_indices),
end(sample_indices=None), finished_flags, audio_queues, iteration, get_stream; AsyncAudioStreamer :150-264) -- but put()
never blocks the generation loop: the chunk batch is copied into a pinned ring slot with ONE asynchronous D2H copy on a copy
stream ordered behind the producing stream, an event marks it, and a background thread hands finished chunks to the per-sample queues (the
reference does `.detach().cpu()` per sample = one stream sync each).

`pcm16=engine` converts each chunk to 16-bit PCM on the device before it leaves (vv_audio_to_pcm16: the arithmetic of the
gradio demo's convert_to_16_bit_wav, demo/gradio_demo.py:404-418,1058-1073), so the consumer receives int16 tensors and
the D2H copy is half the size of a bf16->fp32 round trip.

The drain thread exits when every sample has ended (or on close()); its pinned ring goes back to a process-wide pool.
"""
import asyncio
import threading
from queue import Queue
from typing import Optional

import torch

_CLOSE = object()

# Pinned ring buffers outlive their streamer: a serving process creates one AudioStreamer per request, and a fresh pinned
# allocation per ring slot (hipHostMalloc, ~1 ms apiece; torch's host allocator only starts re-using freed blocks a couple of
# requests later) sat in front of every request's FIRST chunk -- 14 ms instead of 10.8 ms to first audio on the 1.5B request.
# A finished streamer hands its buffers back (every copy into them has completed: the drain thread waited on their events);
# the next one takes buffers of the same shape / dtype.  Bounded: _POOL_CAP buffers (a 3200-sample chunk is 6-13 KB).
_POOL_LOCK = threading.Lock()
_POOL: dict = {}
_POOL_CAP = 64


# ... and so does the copy stream: a stream's first operations pay for its hardware queue (measured: 2 ms per wait_event on a
# streamer's fresh stream, in front of the first request's first chunk); one per device, created once, warmed by warmup()
_COPY_STREAMS: dict = {}


def _copy_stream_for(device):
    key = (device.type, device.index)
    with _POOL_LOCK:
        st = _COPY_STREAMS.get(key)
        if st is None:
            st = _COPY_STREAMS[key] = torch.cuda.Stream(device=device)
        return st


def _pinned_take(shape, dtype):
    with _POOL_LOCK:
        lst = _POOL.get((tuple(shape), dtype))
        if lst:
            return lst.pop()
    return torch.empty(tuple(shape), dtype=dtype).pin_memory()


def _pinned_give(bufs):
    with _POOL_LOCK:
        held = sum(len(v) for v in _POOL.values())
        for b in bufs:
            if b is not None and held < _POOL_CAP:
                _POOL.setdefault((tuple(b.shape), b.dtype), []).append(b)
                held += 1


class AudioStreamer:
    def __init__(self, batch_size: int, stop_signal=None, timeout: Optional[float] = None, ring_slots: int = 8, pcm16=None):
        self.batch_size = batch_size
        self.stop_signal = stop_signal
        self.timeout = timeout
        self.audio_queues = [self._make_queue() for _ in range(batch_size)]
        self.finished_flags = [False for _ in range(batch_size)]
        self._pcm_engine = pcm16                  # vibevoice_amd.Engine (or None: chunks keep the producer's dtype)
        self._ring = [None] * ring_slots          # pinned host buffers, allocated on first use (shape of the first chunk)
        self._pcm_dev = [None] * ring_slots       # device int16 staging per ring slot (pcm16 mode)
        self._copy_stream = None                  # D2H copies run here: the module's copy stream of the chunks' device
        self._free = Queue()
        for i in range(ring_slots):
            self._free.put(i)
        self._work = Queue()
        self._ended = 0                            # samples whose end marker has been queued
        self._lock = threading.Lock()
        self._closed = False
        self._thread = threading.Thread(target=self._drain, daemon=True)
        self._thread.start()

    def _make_queue(self):
        return Queue()

    def _deliver(self, idx, item):
        self.audio_queues[idx].put(item, timeout=self.timeout)

    # ---- producer side (generation loop) ----
    def put(self, audio_chunks: torch.Tensor, sample_indices: torch.Tensor):
        idxs = [int(i) for i in sample_indices.tolist()]
        live = [(row, idx) for row, idx in enumerate(idxs) if idx < self.batch_size and not self.finished_flags[idx]]
        if not live or self._closed:
            return
        if not audio_chunks.is_cuda:
            for row, idx in live:
                self._work.put(("chunk", idx, audio_chunks[row].detach().clone(), None, None))
            return
        slot = self._free.get()                                    # back-pressure only if the consumer is > ring_slots behind
        with self._lock:                                           # the drain thread hands the ring back to the pool under this lock
            if self._closed:
                self._free.put(slot)
                return
            self._put_slot(audio_chunks, live, slot)

    def _put_slot(self, audio_chunks, live, slot):
        src = audio_chunks.detach()
        if self._pcm_engine is not None:
            # int16 on device, on the producing stream, before the copy (demo/gradio_demo.py:1058-1073 per chunk)
            n = src.shape[0]
            flat = src.reshape(n, -1).to(torch.float32).contiguous()
            pd = self._pcm_dev[slot]
            if pd is None or pd.shape[0] < n or pd.shape[1] != flat.shape[1]:
                pd = torch.empty((max(n, self.batch_size), flat.shape[1]), dtype=torch.int16, device=src.device)
                self._pcm_dev[slot] = pd
            self._pcm_engine.audio_to_pcm16(flat, pd[:n], stream=torch.cuda.current_stream(src.device))
            src = pd[:n].view((n,) + tuple(audio_chunks.shape[1:]))
        need = src.shape
        buf = self._ring[slot]
        if buf is None or buf.shape[1:] != need[1:] or buf.shape[0] < need[0] or buf.dtype != src.dtype:
            if buf is not None:
                _pinned_give([buf])
            buf = _pinned_take((max(need[0], self.batch_size),) + tuple(need[1:]), src.dtype)
            self._ring[slot] = buf
        # The D2H copy runs on the streamer's OWN stream, ordered behind the producer by an event: the drain thread then waits on an
        # event of a stream that never enters hipGraph capture.  (Waiting on an event of the producing stream raced with the
        # engine's stream captures -- hipEventSynchronize refuses an event whose stream is capturing, hipErrorCapturedEvent --
        # whenever a new graph key was captured while a chunk was still in flight.)
        prod = torch.cuda.current_stream(audio_chunks.device)
        if self._copy_stream is None:
            self._copy_stream = _copy_stream_for(audio_chunks.device)
        ready = torch.cuda.Event()
        ready.record(prod)
        self._copy_stream.wait_event(ready)
        with torch.cuda.stream(self._copy_stream):
            buf[:need[0]].copy_(src, non_blocking=True)
        src.record_stream(self._copy_stream)
        ev = torch.cuda.Event()
        ev.record(self._copy_stream)
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

    def close(self):
        """Stop the drain thread and release the pinned ring (also happens by itself once every sample has ended)."""
        with self._lock:                     # under the lock put() queues its copies: every one of them precedes the marker below
            was_open = not self._closed
            self._closed = True
        if was_open:
            # a consumer blocked in get_stream() / __iter__ with timeout=None must wake up: every sample that has not ended
            # gets its stop signal (ordered after its pending chunks) before the thread stops
            for idx in range(self.batch_size):
                if not self.finished_flags[idx]:
                    self.finished_flags[idx] = True
                    self._work.put(("end", idx, None, None, None))
            self._work.put((_CLOSE, None, None, None, None))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- background thread: event -> per-sample queues, in production order ----
    def _drain(self):
        while True:
            kind, a, b, ev, n = self._work.get()
            if kind is _CLOSE:
                break
            if kind == "slot":
                ev.synchronize()
                buf = self._ring[b]
                for row, idx in a:
                    self._deliver(idx, buf[row].clone())
                self._free.put(b)
            elif kind == "chunk":
                self._deliver(a, b)
            elif kind == "end":
                self._deliver(a, self.stop_signal)
                self._ended += 1
                if self._ended >= self.batch_size:       # nothing can arrive any more (put() drops chunks of ended samples)
                    break
        with self._lock:                                 # no put() is between its closed check and its copy
            self._closed = True
        # a put() that raced with an end() / close() from another thread may have queued its copy behind the marker that ended the
        # loop: wait for those copies too before the buffers change hands
        import queue as _q
        while True:
            try:
                item = self._work.get_nowait()
            except _q.Empty:
                break
            if item[0] == "slot":
                item[3].synchronize()
                self._free.put(item[2])                  # a producer still blocked in _free.get() wakes up, sees _closed, returns
        with self._lock:
            _pinned_give(self._ring)                     # every copy into them has completed (their events have been waited on)
            self._ring = [None] * len(self._ring)
            self._pcm_dev = [None] * len(self._pcm_dev)  # release the device staging

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


class AsyncAudioStreamer(AudioStreamer):
    """Async variant (vibevoice/modular/streamer.py:150-264): per-sample asyncio queues owned by the event loop that was
    running at construction; the drain thread hands chunks over with loop.call_soon_threadsafe, so generate() -- running in
    a worker thread, as in the reference's gradio path -- never touches the loop and never syncs the stream.
        async for chunk in streamer.get_stream(0): ...      async for batch in streamer: ..."""

    def __init__(self, batch_size: int, stop_signal=None, timeout: Optional[float] = None, ring_slots: int = 8, pcm16=None):
        self.loop = asyncio.get_running_loop()
        super().__init__(batch_size, stop_signal, timeout, ring_slots=ring_slots, pcm16=pcm16)

    def _make_queue(self):
        return asyncio.Queue()

    def _deliver(self, idx, item):
        self.loop.call_soon_threadsafe(self.audio_queues[idx].put_nowait, item)

    def _is_stop(self, value):
        return value is self.stop_signal or (not torch.is_tensor(value) and value == self.stop_signal)

    async def get_stream(self, sample_idx: int):
        if sample_idx >= self.batch_size:
            raise ValueError(f"Sample index {sample_idx} exceeds batch size {self.batch_size}")
        while True:
            value = await self.audio_queues[sample_idx].get()
            if self._is_stop(value):
                break
            yield value

    def __iter__(self):
        raise TypeError("AsyncAudioStreamer is consumed with `async for`")

    def __aiter__(self):
        return AsyncAudioBatchIterator(self)


class AsyncAudioBatchIterator:
    """{sample index: chunk} for the samples whose next chunk is ready first; ends when every sample has ended."""

    def __init__(self, streamer: AsyncAudioStreamer):
        self.streamer = streamer
        self.active = set(range(streamer.batch_size))
        self._pending = {}                      # idx -> task waiting on that sample's queue (kept across calls: no chunk is lost)

    def __aiter__(self):
        return self

    async def __anext__(self):
        while self.active:
            for idx in self.active:
                if idx not in self._pending:
                    self._pending[idx] = asyncio.ensure_future(self.streamer.audio_queues[idx].get())
            done, _ = await asyncio.wait(self._pending.values(), return_when=asyncio.FIRST_COMPLETED, timeout=self.streamer.timeout)
            if not done:
                raise asyncio.TimeoutError()
            out = {}
            for idx in sorted(self.active):
                t = self._pending.get(idx)
                if t is not None and t in done:
                    del self._pending[idx]
                    v = t.result()
                    if self.streamer._is_stop(v):
                        self.active.discard(idx)
                    else:
                        out[idx] = v
            if out:
                return out
        raise StopAsyncIteration()
