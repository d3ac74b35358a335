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
            # "nccl" IS RCCL on ROCm.  RV_DIST_BACKEND=gloo: rehearsal of the multi-rank code path on a box with fewer GPUs than
            # ranks (gloo stages device tensors through the host; several ranks may then share one device - tests only)
            backend = os.environ.get("RV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        elif torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


class BucketedAllReduce(GradReducer):
    """Asynchronous bucketed SUM all-reduce over a flat gradient buffer.

    mode (env RV_ALLREDUCE_MODE, default "overlap"; the first multi-GPU run can A/B all three in one command, bench.py dp_diag):
      overlap  every bucket is all-reduced asynchronously on RCCL's own stream the moment backward finalised it;
      serial   the same buckets, but the compute stream WAITS for each collective before backward goes on - the fallback
               should RCCL's persistent workgroups disturb the 256-workgroup GEMMs they run beside (two LDS-filling GEMMs side
               by side broke each other's XCD lockstep: 1.7-3.6 x, profiles/r03_wgrad_side_stream_negative_result.log);
      skip     no collective at all (WRONG gradients for world > 1: measurement only - step time without communication).
    ``timeline``: per-bucket (bytes, enqueue, complete) from device events, see ``last_timeline``.

    reduce_dtype (env RV_GRAD_REDUCE_DTYPE = "bf16" | "fp32", default the buffer's own dtype): the arithmetic of the cross-rank SUM.
      buffer dtype (bf16 on the device)  the collective sums the bf16 gradients as they lie: RCCL's ring adds in bf16, i.e. the
               sum of N ranks is rounded N - 1 times (DeepSpeed ZeRO-2 under --bf16 reduces bf16 gradients the same way);
      fp32     every bucket is widened to an fp32 staging buffer (a dtype-converting copy), summed in fp32 by the collective, and
               rounded ONCE when it is copied back: twice the bytes on the wire, and the fp32 staging buffer of a bucket (2 x the
               bucket) lives until its collective has completed - completed buckets are copied back and freed at the next
               launch (``_drain``), and at most ``max_staged`` (RV_GRAD_STAGE_MAX, default 4) buffers are ever in flight: the
               compute stream waits for the oldest collective before a fifth is staged (<= 4 x 800 MB at the default bucket
               size instead of 2 x the whole 13.5 GB gradient) - for a sum that equals the exact one rounded to bf16 up to fp32
               accumulation.  The
               two agree bit for bit at world size 2 (one addition, one rounding either way) and differ from 3 ranks on
               (tests/test_dist_gloo.py::test_fp32_gradient_sum_world3 measures both against the float64 sum)."""

    def __init__(self, flat_grad: torch.Tensor, group=None, bucket_bytes: int = 400 << 20, force: bool = False,
                 mode: Optional[str] = None, timeline: bool = False, reduce_dtype: Optional[str] = None):
        self.flat = flat_grad
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force          # issue the collectives even in a 1-rank group (exercises the RCCL path in tests)
        self.bucket_elems = max(1, bucket_bytes // flat_grad.element_size())
        self.mode = mode or os.environ.get("RV_ALLREDUCE_MODE", "overlap")
        if self.mode not in ("overlap", "serial", "skip"):
            raise ValueError(f"RV_ALLREDUCE_MODE must be overlap | serial | skip, got {self.mode!r}")
        self.timeline = bool(timeline) and flat_grad.is_cuda
        rd = (reduce_dtype or os.environ.get("RV_GRAD_REDUCE_DTYPE") or "").lower()
        if rd not in ("", "bf16", "fp32"):
            raise ValueError(f"RV_GRAD_REDUCE_DTYPE must be bf16 | fp32, got {rd!r}")
        self.widen = rd == "fp32" and flat_grad.dtype != torch.float32
        self._staged: List = []                        # (start, end, fp32 staging tensor) of the buckets in flight (widen only)
        self.max_staged = max(1, int(os.environ.get("RV_GRAD_STAGE_MAX", "4")))
        self._n_drained = 0                            # collectives of the current step already waited for (widen + overlap)
        self._pending: Optional[Tuple[int, int]] = None
        self._works: List = []
        self._events: List = []                        # (bytes, enqueue event, completion event) per collective
        self.launched: List[Tuple[int, int]] = []      # (start, end) of every collective of the current step
        self.last_timeline: Optional[dict] = None

    def _launch(self, start: int, end: int):
        if (self.world_size == 1 and not self.force) or end <= start:
            return
        self.launched.append((start, end))
        if self.mode == "skip":
            return
        ev = None
        if self.timeline:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()                              # compute stream: the bucket's gradients are final here
            self._events.append(((end - start) * self.flat.element_size(), ev[0], ev[1]))
        buf = self.flat[start:end]
        if self.widen:                                  # fp32 sum: widen -> all-reduce fp32 -> round once on the way back
            if self.mode != "serial":
                self._drain(block=len(self._staged) >= self.max_staged)
            buf = torch.empty(end - start, dtype=torch.float32, device=self.flat.device)
            buf.copy_(self.flat[start:end])
            self._staged.append((start, end, buf))
        w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.mode == "serial":
            w.wait()                                    # the compute stream waits: nothing of backward runs beside the collective
            if ev is not None:
                ev[1].record()
            if self.widen:
                a, b, t = self._staged.pop()
                self.flat[a:b].copy_(t)
        else:
            self._works.append(w)

    def _drain(self, block: bool = False):
        """widen + overlap: copy back (one rounding to bf16) and FREE the staging buffers of the collectives that have completed,
        oldest first (collectives of one communicator complete in order); ``block`` waits for the oldest one regardless."""
        while self._works and (block or self._works[0].is_completed()):
            w = self._works.pop(0)
            w.wait()                                    # stream-ordered: the copy below runs behind the collective
            if self.timeline and self._n_drained < len(self._events):
                self._events[self._n_drained][2].record()
            self._n_drained += 1
            a, b, t = self._staged.pop(0)
            self.flat[a:b].copy_(t)
            block = False

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
        t_end = None
        if self.timeline and self._events:
            t_end = torch.cuda.Event(enable_timing=True)
            t_end.record()                              # backward has been enqueued completely at this point
        for i, w in enumerate(self._works):
            w.wait()                                    # collectives of one communicator complete in order
            if self.timeline:
                self._events[self._n_drained + i][2].record()
        for a, b, t in self._staged:                    # (overlap mode) stream-ordered behind the waits above: one rounding to bf16
            self.flat[a:b].copy_(t)
        self._staged = []
        self._works = []
        self._n_drained = 0
        if self.timeline and self._events:
            self._timeline_pending = (self._events, t_end)
        self._events = []
        done, self.launched = self.launched, []
        return done

    def collect_timeline(self) -> Optional[dict]:
        """Host side of ``timeline`` (synchronises): per bucket MB, enqueue time and - in serial mode - completion time in ms after
        the first bucket became ready, and how long after the END of backward the last collective completed (the exposed
        communication).  In OVERLAP mode the per-bucket event is recorded on the compute stream inside finish(), i.e. after all
        of backward has been enqueued: it says when the COMPUTE STREAM passed that bucket's wait, never earlier than
        ``backward_end_ms``, and cannot show a bucket finishing under backward.  It is reported as
        ``wait_passed_ms`` there; only ``exposed_after_backward_ms`` is a statement about the collectives themselves."""
        pend = getattr(self, "_timeline_pending", None)
        if pend is None:
            return None
        events, t_end = pend
        self._timeline_pending = None
        torch.cuda.synchronize()
        t0 = events[0][1]
        done_key = "done_ms" if self.mode == "serial" else "wait_passed_ms"
        rows = [{"mb": round(nb / 2**20, 1), "enqueue_ms": round(t0.elapsed_time(a), 3), done_key: round(t0.elapsed_time(b), 3)}
                for nb, a, b in events]
        bw_end = t0.elapsed_time(t_end)
        self.last_timeline = dict(mode=self.mode, buckets=rows, backward_end_ms=round(bw_end, 3),
                                  exposed_after_backward_ms=round(max(rows[-1][done_key] - bw_end, 0.0), 3),
                                  total_mb=round(sum(r["mb"] for r in rows), 1))
        return self.last_timeline

    def reduce_metrics(self, t: torch.Tensor) -> torch.Tensor:
        """Cross-rank mean of a small metric vector in ONE collective."""
        if self.world_size == 1:
            return t
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world_size


# ----------------------------------------------------------------------------------------------------------------------
# Opt-in ZeRO-1: sharded optimizer.  The reference trains under DeepSpeed ZeRO-2 (script/zero2.json:16-22,
# script/train/llava15_train.sh:6): gradients reduce-scattered, optimizer state partitioned, updated parameters all-gathered.
# On 288 GB parts nothing FORCES the partition (replicated state: 108 GB), so the replicated BucketedAllReduce stays the default
# until a hardware A/B exists; what the partition buys is (N - 1) / N of the AdamW pass per step (32 ms of HBM streaming at
# N = 1: ~3 % of a step at N = 8) and (N - 1) / N of the 81 GB of fp32 state.  Bytes on the wire are those of the all-reduce it
# replaces: reduce-scatter (N - 1) / N x G  +  all-gather (N - 1) / N x P, with G = P = 13.5 GB of bf16.
# ----------------------------------------------------------------------------------------------------------------------
class ShardedGradReducer(BucketedAllReduce):
    """The bucket schedule of BucketedAllReduce with a REDUCE-SCATTER per launched range [a, b): c = floor((b - a) / (8 W)) * 8
    elements per rank, rank r receives the SUM of chunk [a + r c, a + (r + 1) c) into its compact ``g_shard`` (at a running
    offset), asynchronously on RCCL's stream like the all-reduce it replaces.  The < 8 W + 8 trailing elements of a range that do
    not divide (``remainders``) are summed by ONE small all-reduce in finish() and handled redundantly by every rank - no padding,
    no staging copy, no collective ever reads or writes outside [a, b)."""
    sharded = True

    def __init__(self, flat_grad: torch.Tensor, group=None, bucket_bytes: int = 400 << 20, force: bool = False,
                 mode: Optional[str] = None, timeline: bool = False):
        # ``timeline`` is accepted (make_reducer forwards one keyword set to either reducer) and ignored: the per-bucket device
        # events belong to the all-reduce path's diagnosis (bench.py dp_diag)
        super().__init__(flat_grad, group=group, bucket_bytes=bucket_bytes, force=force, mode=mode, reduce_dtype=None)
        if self.widen:
            raise ValueError("RV_GRAD_REDUCE_DTYPE=fp32 is implemented for the replicated all-reduce only")
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.collective = dist.is_initialized() and (self.world_size > 1 or force)     # else: one rank, plain copies
        W = self.world_size
        self.g_shard = torch.zeros(flat_grad.numel() // W + 8 * 1024, dtype=flat_grad.dtype, device=flat_grad.device)
        self.ranges: List[Tuple[int, int, int, int]] = []          # (a, b, c, offset into g_shard) of the LAST completed step
        self._step_ranges: List[Tuple[int, int, int, int]] = []
        self._off = 0
        self.rem_buf: Optional[torch.Tensor] = None               # packed reduced remainders of the last completed step
        self.rem_spans: List[Tuple[int, int]] = []

    def split(self, start: int, end: int) -> int:
        return ((end - start) // (8 * self.world_size)) * 8

    def plan(self, schedule) -> List[Tuple[int, int, int, int]]:
        """The ranges a step will launch for ``schedule`` = [(name, start, end), ...] (dry run of the merging logic)."""
        out, pending, off = [], None, 0

        def launch(a, b):
            nonlocal off
            if b <= a:                                  # _launch() returns early on an empty range: the plan must not list it either
                return
            c = self.split(a, b)
            out.append((a, b, c, off))
            off += c
        for _, a, b in schedule:
            if pending is not None and pending[1] == a:
                a = pending[0]
            elif pending is not None:
                launch(*pending)
            pending = (a, b)
            if b - a >= self.bucket_elems:
                launch(a, b)
                pending = None
        if pending is not None:
            launch(*pending)
        return out

    def _launch(self, start: int, end: int):
        if end <= start:
            return
        c = self.split(start, end)
        self._step_ranges.append((start, end, c, self._off))
        self.launched.append((start, end))
        off, self._off = self._off, self._off + c
        if self.mode == "skip" or c == 0:
            return
        W = self.world_size
        if not self.collective:
            self.g_shard[off:off + c].copy_(self.flat[start:start + c])
            return
        w = dist.reduce_scatter_tensor(self.g_shard[off:off + c], self.flat[start:start + W * c], op=dist.ReduceOp.SUM,
                                       group=self.group, async_op=True)
        if self.mode == "serial":
            w.wait()
        else:
            self._works.append(w)

    def finish(self):
        if self._pending is not None:
            self._launch(*self._pending)
            self._pending = None
        for w in self._works:
            w.wait()
        self._works = []
        W = self.world_size
        spans = [(a + W * c, b) for a, b, c, _ in self._step_ranges if b > a + W * c]
        if spans and self.mode != "skip" and self.collective:
            buf = torch.cat([self.flat[x:y] for x, y in spans])
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            self.rem_buf = buf
        else:
            self.rem_buf = torch.cat([self.flat[x:y] for x, y in spans]) if spans else self.flat[:0]
        self.rem_spans = spans
        if self._step_ranges:
            self.ranges = self._step_ranges
        self._step_ranges, self._off = [], 0
        done, self.launched = self.launched, []
        return done


def make_reducer(flat_grad: torch.Tensor, **kw) -> GradReducer:
    """The gradient exchange the environment selects: RV_ZERO1=1 -> ShardedGradReducer (opt-in ZeRO-1: reduce-scatter + sharded
    AdamW + parameter all-gather), else the replicated BucketedAllReduce (default)."""
    if os.environ.get("RV_ZERO1", "0") not in ("", "0"):
        return ShardedGradReducer(flat_grad, **kw)
    return BucketedAllReduce(flat_grad, **kw)


class _DeviceOptKernels:
    """The HIP kernels of the optimizer (rlaif-v_amd/ops.py); tests/test_dist_gloo.py substitutes CPU stand-ins with the same
    signatures to drive the partition / exchange logic under gloo."""

    @staticmethod
    def sumsq(g, out1, accumulate):
        from . import ops
        ops.grad_sumsq(g, out1, accumulate)

    @staticmethod
    def clip(sumsq, max_norm, out2, pre_scale):
        from . import ops
        ops.clip_from_sumsq(sumsq, max_norm, out2, pre_scale)

    @staticmethod
    def adamw(p, master, m, v, g, lr, b1, b2, eps, wd, step, clip):
        from . import ops
        ops.adamw_step(p, master, m, v, g, lr, b1, b2, eps, wd, step, clip=clip)

    @staticmethod
    def to_param(master, p):
        from . import ops
        ops.cast_f32_to_bf16(master, p)


class ShardedAdamW:
    """AdamW + clip_grad_norm_ on 1 / W of the parameters (the chunks ShardedGradReducer hands this rank) followed by an in-place
    all-gather of the updated bf16 parameters, issued in FORWARD order (embedding / projector, layer 0, ... lm_head = the reverse
    of backward's completion order) so that the parameters the next step needs first arrive first.
    Elementwise arithmetic identical to the replicated path (same kernels on the same values); the only difference is the
    summation order of the global gradient norm (per-rank partial sums), i.e. the clip factor agrees to fp32 rounding and is
    IDENTICAL whenever clipping is inactive (norm <= max_grad_norm)."""

    def __init__(self, train_p: torch.Tensor, n_decay: int, reducer: ShardedGradReducer, schedule, kernels=None,
                 full_state: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None):
        self.p, self.n_decay, self.red, self.k = train_p, n_decay, reducer, kernels or _DeviceOptKernels
        self.W, self.rank = reducer.world_size, reducer.rank
        self.ranges = reducer.plan(schedule)
        W = self.W
        if self.k is _DeviceOptKernels:
            # rv_grad_sumsq / rv_adamw_step process 8 elements per lane and require n % 8 == 0: every chunk (c is a multiple of 8 by
            # construction), every remainder span and both sides of the weight-decay split are multiples of 8 exactly when every
            # range boundary and n_decay are.  True for every layout the store produces today; a future one fails HERE, by name.
            bad = [(a, b) for a, b, _, _ in self.ranges if a % 8 or (b - a) % 8]
            if bad or n_decay % 8:
                raise ValueError(f"ZeRO-1 needs bucket boundaries and n_decay at multiples of 8 elements (kernel vector width): "
                                 f"ranges {bad[:4]}, n_decay {n_decay}")
        self.rem_spans = [(a + W * c, b) for a, b, c, _ in self.ranges if b > a + W * c]
        n_shard = sum(c for _, _, c, _ in self.ranges)
        n_rem = sum(y - x for x, y in self.rem_spans)
        dev = train_p.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.master, self.m, self.v = torch.zeros(n_shard, **f32), torch.zeros(n_shard, **f32), torch.zeros(n_shard, **f32)
        self.rem_master, self.rem_m, self.rem_v = torch.zeros(n_rem, **f32), torch.zeros(n_rem, **f32), torch.zeros(n_rem, **f32)
        self._sumsq = torch.zeros(1, **f32)
        if full_state is not None:
            self.load_full_state(*full_state)
        else:
            self.sync_master_from_params()

    # ---- this rank's chunk of a range, as (lo, hi) in trainable-relative indices
    def _mine(self, a, c):
        return a + self.rank * c, a + (self.rank + 1) * c

    def sync_master_from_params(self):
        for a, b, c, off in self.ranges:
            lo, hi = self._mine(a, c)
            self.master[off:off + c].copy_(self.p[lo:hi])
        o = 0
        for x, y in self.rem_spans:
            self.rem_master[o:o + y - x].copy_(self.p[x:y])
            o += y - x

    def _decay_pieces(self, lo, hi):
        """[lo, hi) split at the weight-decay boundary: (lo, hi, has_decay)"""
        nd = self.n_decay
        if hi <= nd:
            return [(lo, hi, True)]
        if lo >= nd:
            return [(lo, hi, False)]
        return [(lo, nd, True), (nd, hi, False)]

    def step(self, lr, b1, b2, eps, wd, step, max_norm, clip_out: torch.Tensor):
        red, k, W = self.red, self.k, self.W
        if [(a, b, c) for a, b, c, _ in red.ranges] != [(a, b, c) for a, b, c, _ in self.ranges]:
            raise RuntimeError("the step's gradient ranges differ from the optimizer's partition (bucket schedule changed?)")
        n_shard = self.master.numel()
        # ---- global gradient norm: local sum of squares (+ the replicated remainders, counted once: on rank 0) -> 1-float all-reduce
        k.sumsq(red.g_shard[:n_shard], self._sumsq, False)
        if self.rank == 0 and red.rem_buf.numel():
            k.sumsq(red.rem_buf, self._sumsq, True)
        if red.collective:
            dist.all_reduce(self._sumsq, op=dist.ReduceOp.SUM, group=red.group)
        k.clip(self._sumsq, max_norm, clip_out, 1.0 / W)
        # ---- AdamW on this rank's chunks and on the replicated remainders
        for a, b, c, off in self.ranges:
            lo, hi = self._mine(a, c)
            for x, y, dec in self._decay_pieces(lo, hi):
                s = slice(off + x - lo, off + y - lo)
                k.adamw(self.p[x:y], self.master[s], self.m[s], self.v[s], red.g_shard[s], lr, b1, b2, eps, wd if dec else 0.0, step, clip_out)
        o = 0
        for x0, y0 in self.rem_spans:
            for x, y, dec in self._decay_pieces(x0, y0):
                s = slice(o + x - x0, o + y - x0)
                k.adamw(self.p[x:y], self.rem_master[s], self.rem_m[s], self.rem_v[s], red.rem_buf[s], lr, b1, b2, eps, wd if dec else 0.0,
                        step, clip_out)
            o += y0 - x0
        # ---- updated bf16 parameters back to every rank: in-place all-gather per range, forward order
        works = []
        if red.collective:
            for a, b, c, off in reversed(self.ranges):
                if c == 0:
                    continue
                lo, hi = self._mine(a, c)
                works.append(dist.all_gather_into_tensor(self.p[a:a + W * c], self.p[lo:hi], group=red.group, async_op=True))
        for w in works:
            w.wait()

    # ---- checkpoints keep the REPLICATED format (full master / m / v in flat order): a sharded run resumes a replicated one and back
    def gather_full_state(self, all_ranks: bool = False):
        """(master, m, v) as full fp32 CPU tensors [n_train].  Collective: EVERY rank must call (the shards travel by all-gather);
        only rank 0 - the rank that writes the checkpoint - materialises the three host tensors (3 x 27 GB for the 7B full
        fine-tune; eight ranks of one node doing so would ask for 650 GB of host RAM and seven useless device-to-host copies),
        the other ranks return None.  ``all_ranks`` (tests) materialises them everywhere."""
        W, n = self.W, self.p.numel()
        keep = all_ranks or self.rank == 0
        out = []
        for shard, rem in ((self.master, self.rem_master), (self.m, self.rem_m), (self.v, self.rem_v)):
            full = torch.zeros(n, dtype=torch.float32) if keep else None
            for a, b, c, off in self.ranges:
                if c == 0:
                    continue
                tmp = torch.empty(W * c, dtype=torch.float32, device=shard.device)
                if self.red.collective:
                    dist.all_gather_into_tensor(tmp, shard[off:off + c].contiguous(), group=self.red.group)
                else:
                    tmp.copy_(shard[off:off + c])
                if keep:
                    full[a:a + W * c] = tmp.cpu()
            o = 0
            for x, y in self.rem_spans:
                if keep:
                    full[x:y] = rem[o:o + y - x].cpu()
                o += y - x
            out.append(full)
        return tuple(out) if keep else None

    def load_full_state(self, master, m, v):
        for shard, rem, full in ((self.master, self.rem_master, master), (self.m, self.rem_m, m), (self.v, self.rem_v, v)):
            for a, b, c, off in self.ranges:
                lo, hi = self._mine(a, c)
                shard[off:off + c].copy_(full[lo:hi])
            o = 0
            for x, y in self.rem_spans:
                rem[o:o + y - x].copy_(full[x:y])
                o += y - x
