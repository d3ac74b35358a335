"""Data-parallel gradient exchange: one process per GPU, torch.distributed over RCCL/xGMI.

Replaces DeepSpeed ZeRO-2's bucketed reduce (script/zero2.json:16-22 of the reference: overlap_comm,
contiguous_gradients, reduce_bucket_size) - C1 in SURVEY.md section 2.3 - and the 7 per-step metric
all-gathers (C2).  Design for 8 x MI355X (xGMI, 7 links x ~153 GB/s per GPU):
  * parameters, fp32 master and Adam state are REPLICATED (108 GB << 288 GB per GPU): no ZeRO sharding,
    so the only collective on the data path is a SUM all-reduce of the bf16 gradient buffer;
  * the flat gradient buffer is laid out in backward-completion order (model.ParamStore), so buckets are
    contiguous slices that become final front to back; each slice is all-reduced asynchronously on
    RCCL's own stream as soon as backward has produced it and overlaps with the remaining layers;
  * small neighbouring slices are merged up to ``bucket_bytes`` (default 400 MB ~ one decoder layer) so a
    full fine-tune issues 18 large collectives per step (counted by bench.py's probe) instead of hundreds of small ones;
  * the 1/world averaging is folded into the gradient-clip factor (rv_grad_norm pre_scale), so no
    extra pass over the 13.5 GB buffer is needed.
Works unchanged on CPU tensors with the gloo backend (tests/test_dist_gloo.py, world_size 2).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from .trainer import GradReducer


def init_process_group_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher's environment (torch.distributed.run)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (the host driver has no legacy IPC)
        # RCCL next to 256-CU GEMMs: every RCCL channel is a persistent workgroup that takes a CU away
        # from the one-workgroup-per-CU GEMMs for as long as a collective runs, while the step only needs 2 x 7/8 x 13.5 GB
        # = 23.6 GB per GPU inside ~0.7 s of backward (34 GB/s).  RV_RCCL_CHANNELS=n caps RCCL at n channels; it is OPT-IN
        # (default 0 = RCCL's own choice) until an 8-GPU A/B exists - the single-GPU stand-in sweep of bench.py
        # (dp_standin_probe_1gpu: 4 / 8 / 16 / 32 persistent workgroups beside backward) prices only the CU-lending side.
        ch = int(os.environ.get("RV_RCCL_CHANNELS", "0"))
        if ch > 0:
            os.environ.setdefault("NCCL_MAX_NCHANNELS", str(ch))
            os.environ.setdefault("NCCL_MIN_NCHANNELS", str(min(ch, 4)))
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"      # "nccl" IS RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


class BucketedAllReduce(GradReducer):
    """Asynchronous bucketed SUM all-reduce over a flat gradient buffer."""

    def __init__(self, flat_grad: torch.Tensor, group=None, bucket_bytes: int = 400 << 20, force: bool = False):
        self.flat = flat_grad
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force          # issue the collectives even in a 1-rank group (exercises the RCCL path in tests)
        self.bucket_elems = max(1, bucket_bytes // flat_grad.element_size())
        self._pending: Optional[Tuple[int, int]] = None
        self._works: List = []
        self.launched: List[Tuple[int, int]] = []      # (start, end) of every collective of the current step

    def _launch(self, start: int, end: int):
        if (self.world_size == 1 and not self.force) or end <= start:
            return
        self.launched.append((start, end))
        self._works.append(dist.all_reduce(self.flat[start:end], op=dist.ReduceOp.SUM, group=self.group,
                                           async_op=True))

    def on_bucket_ready(self, name: str, start: int, end: int):
        """Called by backward when flat[start:end] holds final local gradients."""
        if self._pending is not None and self._pending[1] == start:
            start = self._pending[0]                    # merge with the adjacent unsent slice
        elif self._pending is not None:
            self._launch(*self._pending)
        self._pending = (start, end)
        if end - start >= self.bucket_elems:
            self._launch(start, end)
            self._pending = None

    def finish(self):
        """Flush the last partial bucket and make the current stream wait for every collective."""
        if self._pending is not None:
            self._launch(*self._pending)
            self._pending = None
        for w in self._works:
            w.wait()
        self._works = []
        done, self.launched = self.launched, []
        return done

    def reduce_metrics(self, t: torch.Tensor) -> torch.Tensor:
        """Cross-rank mean of a small metric vector in ONE collective."""
        if self.world_size == 1:
            return t
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world_size
