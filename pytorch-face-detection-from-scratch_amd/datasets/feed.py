"""Host -> device feed for uint8 frames.

The reference normalises on the host (`img / 255`, datasets/WIDERFace/dataset.py:146) and lets the
DataLoader collate float32 batches (datamodule.py:162-175): 4 bytes per pixel cross PCIe, 708 MB
for a (256,3,480,480) batch -- ~11 ms at 63 GB/s, twice the MI355X step time.  Here frames stay
uint8 until they are on the device: pinned staging buffers, one copy stream, `depth` slots (3: the slot
batch i+1 is copied into was last read by step i-2, so its copy never waits for the step that is running), and the /255 (or bilinear Resize + /255 for frames of another size) runs on the device
right behind the copy (fdet_u8_to_f32_norm / fdet_resize_bilinear_u8_norm).
"""
from typing import Optional, Tuple

import torch

from .. import hotpath as hp


class U8BatchFeeder:
    def __init__(self, batch_shape: Tuple[int, int, int, int], model_size: Tuple[int, int], device,
                 target_shape: Optional[Tuple[int, ...]] = None, depth: int = 3):
        B, C, H, W = batch_shape
        self.device = torch.device(device)
        self.model_size = (int(model_size[0]), int(model_size[1]))
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.depth = depth
        self._slots = []
        for _ in range(depth):
            slot = {"pin": torch.empty(batch_shape, dtype=torch.uint8).pin_memory(),
                    "u8": torch.empty(batch_shape, dtype=torch.uint8, device=self.device),
                    "x": torch.empty(B, C, *self.model_size, dtype=torch.float32, device=self.device),
                    "ready": torch.cuda.Event(), "free": torch.cuda.Event(), "y": None}
            if target_shape is not None:
                slot["ypin"] = torch.empty(target_shape, dtype=torch.float32).pin_memory()
                slot["y"] = torch.empty(target_shape, dtype=torch.float32, device=self.device)
            slot["free"].record(torch.cuda.current_stream(self.device))
            self._slots.append(slot)
        self._w = 0          # next slot to fill
        self._r = 0          # next slot to hand out
        self._inflight = 0

    def host_buffers(self):
        """(pinned uint8 frame buffer, pinned target buffer or None) of the NEXT slot: a loader that
        decodes straight into them saves the extra host copy; then call submit() without arguments."""
        if self._inflight == self.depth:
            raise RuntimeError("U8BatchFeeder: all slots are in flight; call get() first")
        s = self._slots[self._w]
        s["free"].synchronize()                      # the consumer of this slot's previous batch is done
        return s["pin"], s.get("ypin")

    def submit(self, frames_u8: Optional[torch.Tensor] = None, targets: Optional[torch.Tensor] = None) -> None:
        """Queue one host batch: a uint8 (B,C,H,W) CPU tensor [+ float targets], or nothing when the
        caller filled host_buffers() in place.  Returns as soon as the async copy and the device-side
        normalisation are enqueued on the copy stream."""
        if self._inflight == self.depth:
            raise RuntimeError("U8BatchFeeder: all slots are in flight; call get() first")
        s = self._slots[self._w]
        s["free"].synchronize()                      # the consumer of this slot's previous batch is done
        src = s["pin"]
        if frames_u8 is not None:
            if frames_u8.dtype != torch.uint8 or frames_u8.is_cuda:
                raise TypeError("U8BatchFeeder.submit expects a uint8 CPU tensor")
            if frames_u8.is_pinned() and frames_u8.is_contiguous() and tuple(frames_u8.shape) == tuple(s["pin"].shape):
                # a loader that collates into pinned memory (DataLoader(pin_memory=True)): the DMA engine reads the caller's
                # buffer directly -- no 177 MB host memcpy into the staging slot.  The slot keeps a reference until it is reused.
                src = s["src"] = frames_u8
            else:
                s["pin"].copy_(frames_u8)
        if targets is not None:
            s["ypin"].copy_(targets)
        with torch.cuda.stream(self.copy_stream):
            s["u8"].copy_(src, non_blocking=True)
            if s["y"] is not None:
                s["y"].copy_(s["ypin"], non_blocking=True)
            if tuple(s["u8"].shape[-2:]) == self.model_size:
                hp.u8_to_f32_norm(s["u8"], out=s["x"])
            else:
                hp.resize_bilinear_norm(s["u8"], self.model_size, out=s["x"])
            s["ready"].record(self.copy_stream)
        self._w = (self._w + 1) % self.depth
        self._inflight += 1

    def get(self):
        """(x float32 (B,C,Hm,Wm) in [0,1], targets or None) of the oldest queued batch; the current
        stream waits for its copy.  Call release() on the returned token when the step that
        consumes it has been enqueued."""
        if self._inflight == 0:
            raise RuntimeError("U8BatchFeeder: nothing queued")
        s = self._slots[self._r]
        torch.cuda.current_stream(self.device).wait_event(s["ready"])
        self._r = (self._r + 1) % self.depth
        self._inflight -= 1
        return s["x"], s["y"], s

    @staticmethod
    def release(token) -> None:
        token["free"].record(torch.cuda.current_stream(token["x"].device))
