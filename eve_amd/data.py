"""Input side of the hot path: decoded uint8 frames are normalised ON THE DEVICE, and host batches
are staged through pinned memory on a copy stream one batch ahead of the compute.

The reference normalises on the host (`EVESequencesBase.preprocess_frames` / `preprocess_screen_frames`,
/root/reference/src/datasources/eve_sequences.py:196-211) and moves float tensors with `.to(device, non_blocking=True)`
from pageable memory (src/core/training.py:257-261): 377 MB per 960-frame step for the eye patches alone.  Shipping the
uint8 frames is 4x less PCIe traffic, and the float values produced here are bit-identical to numpy's.

  preprocess_frames(u8)          [..., H, W, C] uint8 (device) -> [..., C, H, W] float32 in [-1, 1]
  preprocess_screen_frames(u8)   same, [0, 1]
  EyeNet.forward_sequence / RefineNet.forward_sequence / EVE accept the uint8 tensors directly (eye patches go straight
  into the stem kernel's packed bf16 layout, no float tensor is ever materialised).
  DevicePrefetcher(iterable)     pinned double-buffered H2D on a side stream
"""
import torch

from .kernels import default_kernels

EYE_SCALE, EYE_SHIFT = 2.0 / 255.0, -1.0        # eve_sequences.py:200-201
SCREEN_SCALE = 1.0 / 255.0                      # eve_sequences.py:209


def _fold(frames):
    if frames.dtype != torch.uint8 or frames.dim() < 4:
        raise TypeError('expected uint8 frames shaped [..., H, W, C], got %s %s' % (frames.dtype, tuple(frames.shape)))
    lead = tuple(frames.shape[:-3])
    return frames.reshape((-1,) + tuple(frames.shape[-3:])).contiguous(), lead


def preprocess_frames(frames):
    flat, lead = _fold(frames)
    out = default_kernels().frames_u8_to_nchw(flat, EYE_SCALE, EYE_SHIFT)
    return out.view(lead + tuple(out.shape[1:]))


def preprocess_screen_frames(frames):
    flat, lead = _fold(frames)
    out = default_kernels().frames_u8_to_nchw(flat, SCREEN_SCALE, None)
    return out.view(lead + tuple(out.shape[1:]))


class DevicePrefetcher(object):
    """Iterates `iterable` (dicts of CPU tensors, e.g. a DataLoader) and yields the same dicts on `device`.

    A worker thread pulls the next batch, copies pageable tensors into pinned staging buffers (allocated once per
    key / shape and reused; tensors that are already pinned -- DataLoader(pin_memory=True) -- are sent as they are) and
    enqueues the host-to-device copies on a dedicated copy stream, `depth` batches ahead of the consumer; the consumer's
    stream waits on the copy's event, never on the host.  Non-tensor entries pass through."""

    def __init__(self, iterable, device='cuda', depth=2):
        self.iterable = iterable
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.depth = max(1, int(depth))
        self.stream = torch.cuda.Stream(device=self.device)
        self._pinned = [dict() for _ in range(self.depth + 2)]     # staging slots, reused round-robin
        self._slot_events = [None] * len(self._pinned)             # the copy event of the batch last staged from each slot

    def _stage(self, batch, index):
        slot = self._pinned[index]
        # the host must not overwrite a pinned buffer whose host-to-device copy is still queued (the consumer only waits
        # stream-side, so the host can run ahead of the device): wait for this slot's previous copy
        if self._slot_events[index] is not None:
            self._slot_events[index].synchronize()
        out = {}
        with torch.cuda.stream(self.stream):
            for k, v in batch.items():
                if not isinstance(v, torch.Tensor):
                    out[k] = v
                    continue
                src = v
                if not v.is_pinned():
                    buf = slot.get(k)
                    if buf is None or buf.shape != v.shape or buf.dtype != v.dtype:
                        buf = slot[k] = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                    buf.copy_(v)                                   # (memcpy: releases the GIL)
                    src = buf
                out[k] = src.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._slot_events[index] = ev
        return out, ev

    def __iter__(self):
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        done = object()

        def worker():
            try:
                torch.cuda.set_device(self.device)
                for i, batch in enumerate(self.iterable):
                    q.put(self._stage(batch, i % len(self._pinned)))
                q.put(done)
            except BaseException as e:                              # surface loader errors in the consumer
                q.put(e)

        t = threading.Thread(target=worker, daemon=True)
        t.start()
        while True:
            item = q.get()
            if item is done:
                break
            if isinstance(item, BaseException):
                raise item
            out, ev = item
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for v in out.values():
                if isinstance(v, torch.Tensor):
                    v.record_stream(cur)
            yield out
        t.join()
