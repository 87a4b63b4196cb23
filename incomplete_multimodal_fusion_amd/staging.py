"""Input staging on the GPU: raw sensor tiles -> normalised device tensors, with the host->device copy
of the next batch overlapped with the current step.

Replaces the per-sample CPU numpy/cv2 normalisation of the reference loaders (pretraining/utils/multimodal_dfc2023.py:
load_sar :127-139, load_rgb :114-124, load_dsm :99-111, constants :25-49): the DataLoader workers only read rasters;
dB conversion / clipping / area resize / z-scoring run in `mmae_stage_tiles` (csrc/staging.hip) after one H2D copy of the
RAW tiles (uint8 RGB travels at a quarter of the bytes of a normalised fp32 tile).

    stager = TileStager(device, image_size=256)
    stager.submit({'s1': sar_raw, 's2': rgb_raw, 'dem': dsm_raw})     # CPU tensors/arrays (B, C, H*f, W*f); returns at once
    x = stager.get()                                                   # {'s1','s2','dem'}: (B, C, 256, 256) fp32 on the device

`submit` copies into a pinned ring slot and enqueues H2D + the staging kernels on a side stream; `get` makes the
compute stream wait for that slot's event.  With `slots` >= 2 and the loop `x = get(); submit(next); step(x)` the copy
of batch i+1 runs under the step on batch i.  Tensors returned by get() stay valid until the `slots`-th following
submit(); submit() orders its overwrite after everything already enqueued on the compute stream.
"""
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr

SAR_DB, AFFINE, ZSCORE = 0, 1, 2

# per-domain recipe of the DFC2023 loaders (multimodal_dfc2023.py:25-49, :99-139)
DFC2023 = {
    's1': dict(kind=SAR_DB, mean=(-7.9447875,), std=(2.777256,)),
    's2': dict(kind=AFFINE, mean=(81.29692, 87.93711, 72.041306), std=(39.61512, 35.407978, 35.84708)),
    'dem': dict(kind=ZSCORE, mean=None, std=None),
}


def stage_tiles(raw: torch.Tensor, kind: int, image_size: int, mean=None, std=None, out: Optional[torch.Tensor] = None,
                stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
    """raw: device tensor (B, C, H*f, W*f), float32 or uint8 -> (B, C, image_size, image_size) float32."""
    assert raw.is_cuda and raw.dim() == 4 and raw.is_contiguous()
    assert raw.dtype in (torch.float32, torch.uint8), "raw tiles: float32 or uint8"
    B, C, Hr, Wr = raw.shape
    assert Hr == Wr and Hr % image_size == 0, "integer shrink factors only (cv2.INTER_AREA = block mean)"
    f = Hr // image_size
    if out is None:
        out = torch.empty(B, C, image_size, image_size, dtype=torch.float32, device=raw.device)
    import ctypes
    m = (ctypes.c_float * C)(*mean) if mean is not None else None
    s = (ctypes.c_float * C)(*std) if std is not None else None
    st = stream if stream is not None else torch.cuda.current_stream()
    call("mmae_stage_tiles", kind, _lib.F32 if raw.dtype == torch.float32 else 1, B, C, image_size, image_size, f, ptr(raw),
         ptr(out), m, s, ctypes.c_void_p(st.cuda_stream))
    return out


class TileStager:
    def __init__(self, device, image_size: int = 256, recipe: Dict[str, dict] = None, slots: int = 2):
        self.device = torch.device(device)
        self.image_size, self.recipe, self.slots = image_size, recipe or DFC2023, max(1, slots)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._ring = [dict(host={}, raw={}, out={}, ready=torch.cuda.Event()) for _ in range(self.slots)]
        self._head = self._tail = self._inflight = 0

    def _buffers(self, slot, d, arr):
        key = (tuple(arr.shape), arr.dtype)
        if slot['host'].get(d) is None or slot['host'][d][0] != key:
            host = torch.empty(arr.shape, dtype=arr.dtype).pin_memory()
            raw = torch.empty(arr.shape, dtype=arr.dtype, device=self.device)
            out = torch.empty(arr.shape[0], arr.shape[1], self.image_size, self.image_size, dtype=torch.float32, device=self.device)
            slot['host'][d], slot['raw'][d], slot['out'][d] = (key, host), raw, out
        return slot['host'][d][1], slot['raw'][d], slot['out'][d]

    def submit(self, batch: Dict[str, "np.ndarray | torch.Tensor"]):
        assert self._inflight < self.slots, "ring full: call get() before submitting more"
        slot = self._ring[self._head]
        self._head = (self._head + 1) % self.slots
        self._inflight += 1
        slot['ready'].synchronize()                      # this slot's previous DMA out of the pinned buffer has finished
        self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))   # consumers of its old tensors are enqueued
        with torch.cuda.stream(self.copy_stream):
            slot['doms'] = []
            for d, arr in batch.items():
                if d not in self.recipe:
                    continue
                t = torch.from_numpy(np.ascontiguousarray(arr)) if isinstance(arr, np.ndarray) else arr.contiguous()
                host, raw, out = self._buffers(slot, d, t)
                host.copy_(t)                              # pageable -> pinned (CPU), then one async DMA
                raw.copy_(host, non_blocking=True)
                r = self.recipe[d]
                stage_tiles(raw, r['kind'], self.image_size, r['mean'], r['std'], out=out, stream=self.copy_stream)
                slot['doms'].append(d)
            slot['ready'].record(self.copy_stream)

    def get(self) -> Dict[str, torch.Tensor]:
        """Staged tensors of the oldest submitted batch.  They stay valid until `slots` further submits; the consumer's
        stream is ordered after the staging kernels, no host synchronisation."""
        assert self._inflight > 0, "nothing submitted"
        slot = self._ring[self._tail]
        self._tail = (self._tail + 1) % self.slots
        self._inflight -= 1
        torch.cuda.current_stream(self.device).wait_event(slot['ready'])
        return {d: slot['out'][d] for d in slot['doms']}
