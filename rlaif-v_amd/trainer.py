"""DPO trainer surface of the reference, re-hosted on the HIP kernels.

Mirrors /root/reference muffin/train/trainers.py:
  dpo_loss                  :91-126     -> rv_dpo_loss (loss, rewards and the closed-form gradient)
  get_beta_and_logps        :161-275    -> LlavaDPOModel.forward_logps (is_llava15 branch)
  collect_preference_metrics:140-158    -> same metric names, computed on device, ONE fused all-reduce
                                           instead of 7 x (_nested_gather + .item()) per step (C2)
  LLaVA15DPOTrainer.compute_loss :279-311
and the optimiser the entry script selects (muffin/train/train_llava15.py:75 adamw_torch; cosine schedule,
warm-up 5 %, weight decay 0.01 on matrices only, clip 1.0: script/train/llava15_train.sh:31-34).
The HF ``Trainer`` base class itself is third-party and is not reproduced; only the surface the entry
script uses (``train``, ``compute_loss``, ``log``, ``save_state``, ``_save``) is.
"""
from __future__ import annotations

import json
import math
import os
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import torch

from . import ops
from .model import LlavaDPOModel, StepOutput


@dataclass
class TrainingArguments:
    """Subset of the reference's TrainingArguments dataclass (train_llava15.py:73-100) that the DPO
    path reads; names unchanged so the shell scripts' flags map one to one."""
    output_dir: str = "./checkpoints"
    task: str = "DPO"
    dpo_use_average: bool = False
    dpo_token_weighted: bool = False
    dpo_token_weight: float = 1.0
    dpo_beta: float = 0.1
    learning_rate: float = 5e-7
    weight_decay: float = 0.01
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    warmup_ratio: float = 0.05
    lr_scheduler_type: str = "cosine"
    max_steps: int = 2672
    per_device_train_batch_size: int = 1
    gradient_accumulation_steps: int = 1
    logging_steps: int = 2
    save_steps: int = 167
    model_max_length: int = 2048
    past_index: int = -1
    bf16: bool = True
    seed: int = 42
    gradient_checkpointing: bool = False
    dataloader_num_workers: int = 0
    per_device_eval_batch_size: int = 8
    # LoRA flags of muffin/train/train_llava15_lora.py:111-116 (same names and defaults)
    fully_tune: bool = False
    lora_enable: bool = False
    lora_r: int = 64
    lora_alpha: int = 16
    lora_dropout: float = 0.05
    lora_weight_path: Optional[str] = None
    lora_bias: str = "none"

    def lora_config(self):
        """The LoraConfig init_model builds when --lora_enable (train_llava15_lora.py:304-312), else None."""
        if not self.lora_enable:
            return None
        from .model import LoraConfig
        return LoraConfig(r=self.lora_r, lora_alpha=self.lora_alpha, lora_dropout=self.lora_dropout, bias=self.lora_bias)


def forward_DPO(model, input_ids, labels, attention_mask, images, **kwargs):
    """trainers.py:66-88, the generic (non-LLaVA-1.5) branch of get_beta_and_logps: per-sequence log-probs - sum, or mean
    with ``dpo_use_average`` - under the model's label convention (``is_minicpm``: labels pre-shifted), or the per-token
    log-probs [S, L-1] when ``token_weighted`` (fed to compute_weighted_logp by the caller)."""
    token_weighted = kwargs.pop("token_weighted", False)
    dpo_use_average = kwargs.pop("dpo_use_average", False)
    is_minicpm = kwargs.pop("is_minicpm", False)
    if attention_mask is not None:
        raise NotImplementedError("attention_mask is None on the DPO path (trainers.py:199)")
    if token_weighted:
        if is_minicpm:
            raise NotImplementedError("per-token log-probs under the minicpm label convention")
        out = model.forward_logps(input_ids, labels, images, save_for_backward=False, all_rows=True)
        return out.per_token_logp.view(input_ids.shape[0], -1)
    out = model.forward_logps(input_ids, labels, images, save_for_backward=model.training, label_shift=0 if is_minicpm else 1)
    model.last_out = out
    return out.seq_logp / out.seq_cnt if dpo_use_average else out.seq_logp


def compute_weighted_logp(per_token_logp: torch.Tensor, labels: torch.Tensor, token_weight: torch.Tensor,
                          use_average: bool) -> torch.Tensor:
    """trainers.py:128-137 on the device: sum_t logp[s,t] * w[s,t] * (labels[s,t+1] != -100), optionally divided by the
    weighted token count.  (get_beta_and_logps raises for dpo_token_weighted with LLaVA-1.5, exactly like the reference,
    trainers.py:246-248 - its token weights are text-length, the log-probs spliced-length; this is the arithmetic the
    OmniLMM / MiniCPM branches use.)"""
    dev = per_token_logp.device
    S, Lm1 = per_token_logp.shape
    w = (token_weight.to(dev, torch.float32) * (labels[:, 1:].to(dev) != -100)).contiguous().view(-1)
    off = (torch.arange(S + 1, dtype=torch.int32) * Lm1).to(dev)
    s, c = ops.seq_sum(per_token_logp.to(torch.float32).contiguous().view(-1), off, S, weight=w)
    return s / c if use_average else s


def cosine_lr(step: int, total: int, base_lr: float, warmup_ratio: float) -> float:
    """transformers.get_cosine_schedule_with_warmup; ``step`` = optimizer steps already taken."""
    warm = math.ceil(total * warmup_ratio)
    if step < warm:
        return base_lr * step / max(1, warm)
    prog = (step - warm) / max(1, total - warm)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


def lr_at(kind: str, step: int, total: int, base_lr: float, warmup_ratio: float) -> float:
    """transformers.get_scheduler for the types the reference's scripts can select (`--lr_scheduler_type`, HF default
    "linear"; llava15_train.sh:34 passes "cosine").  ``step`` = optimizer steps already taken.  Anything else raises instead
    of silently running a different schedule."""
    if kind == "cosine":
        return cosine_lr(step, total, base_lr, warmup_ratio)
    if kind == "constant":
        return base_lr
    warm = math.ceil(total * warmup_ratio)
    if kind in ("linear", "constant_with_warmup"):
        if step < warm:
            return base_lr * step / max(1, warm)
        if kind == "constant_with_warmup":
            return base_lr
        return base_lr * max(0.0, (total - step) / max(1, total - warm))
    raise NotImplementedError(f"lr_scheduler_type {kind!r}: supported are cosine, linear, constant, constant_with_warmup")


def dpo_loss(policy_chosen_logps, policy_rejected_logps, reference_chosen_logps, reference_rejected_logps,
             beta: float, reference_free: bool = False):
    """Same signature / return as trainers.py:91-126 on device tensors: (losses, chosen_rewards,
    rejected_rewards)."""
    B = policy_chosen_logps.numel()
    dev = policy_chosen_logps.device
    s = torch.cat([policy_chosen_logps.float(), policy_rejected_logps.float()]).contiguous()
    ones = torch.ones(2 * B, dtype=torch.float32, device=dev)
    rw = reference_chosen_logps.to(dev, torch.float32).contiguous()
    rr = reference_rejected_logps.to(dev, torch.float32).contiguous()
    if reference_free:
        rw, rr = torch.zeros_like(rw), torch.zeros_like(rr)
    per_pair, _, _ = ops.dpo_loss(s, ones, rw, rr, beta, False, 0.0, 1.0)
    if reference_free:     # rewards still use the provided reference (trainers.py:121-124)
        return per_pair[0], beta * (policy_chosen_logps - reference_chosen_logps.to(dev)), \
            beta * (policy_rejected_logps - reference_rejected_logps.to(dev))
    return per_pair[0], per_pair[1], per_pair[2]


def get_beta_and_logps(data_dict: Dict, model: LlavaDPOModel, args, is_minicpm: bool = False,
                       is_llava15: bool = True, save_for_backward: bool = True):
    """trainers.py:161-275 for the LLaVA-1.5 branch.  Consumes the collator's batch dict (keys popped
    like the reference does) and returns (policy_win_logp, policy_rej_logp, ref_win_logp, ref_rej_logp,
    beta).  The StepOutput needed for backward is left on ``model.last_out``."""
    if not is_llava15 or is_minicpm:
        raise NotImplementedError("only the LLaVA-1.5 DPO branch is implemented (SURVEY.md section 8f.4)")
    if args.task != "DPO":
        raise NotImplementedError("task must be DPO (the KTO image branch is unused by the shipped scripts)")
    if args.dpo_token_weighted:
        raise NotImplementedError          # the reference raises here too for LLaVA-1.5 (trainers.py:246-248)
    for k in ("win_labels", "rej_labels", "win_attention_mask", "rej_attention_mask", "ref_win_per_token_logp",
              "ref_rej_per_token_logp", "concatenated_attention_mask", "win_token_weight", "rej_token_weight",
              "concatenated_token_weight"):
        data_dict.pop(k, None)
    win_input_ids = data_dict.pop("win_input_ids")
    rej_input_ids = data_dict.pop("rej_input_ids")
    ref_win_avg_logp = data_dict.pop("ref_win_avg_logp")
    ref_rej_avg_logp = data_dict.pop("ref_rej_avg_logp")
    ref_win_logp = data_dict.pop("ref_win_logp")
    ref_rej_logp = data_dict.pop("ref_rej_logp")
    if args.dpo_use_average:
        ref_win_logp, ref_rej_logp = ref_win_avg_logp, ref_rej_avg_logp
    beta = data_dict.pop("beta")
    images = data_dict.pop("images")
    ids = data_dict.pop("concatenated_input_ids")
    labels = data_dict.pop("concatenated_labels")
    assert win_input_ids.shape[0] == rej_input_ids.shape[0]
    out = model.forward_logps(ids, labels, images, save_for_backward=save_for_backward)
    model.last_out = out
    B = win_input_ids.shape[0]
    logp = out.seq_logp / out.seq_cnt if args.dpo_use_average else out.seq_logp
    dev = model.device
    return (logp[:B], logp[B:], ref_win_logp.to(dev, torch.float32).contiguous(),
            ref_rej_logp.to(dev, torch.float32).contiguous(), beta)


class GradReducer:
    """Interface of the data-parallel gradient exchange (see rlaif_v_amd.dist.BucketedAllReduce)."""
    world_size = 1

    def on_bucket_ready(self, name: str, start: int, end: int):   # pragma: no cover - trivial
        pass

    def finish(self):                                            # pragma: no cover - trivial
        pass

    def reduce_metrics(self, t: torch.Tensor) -> torch.Tensor:   # pragma: no cover - trivial
        return t


class LLaVA15DPOTrainer:
    """``LLaVA15DPOTrainer(model=, tokenizer=, args=, train_dataset=, eval_dataset=, data_collator=)``
    then ``.train()`` - the call surface of muffin/train/train_llava15.py:320-334."""

    def __init__(self, model: LlavaDPOModel, tokenizer=None, args: Optional[TrainingArguments] = None,
                 train_dataset=None, eval_dataset=None, data_collator: Optional[Callable] = None,
                 reducer: Optional[GradReducer] = None, **kwargs):
        self.model, self.tokenizer = model, tokenizer
        self.args = args or TrainingArguments()
        self.train_dataset, self.eval_dataset, self.data_collator = train_dataset, eval_dataset, data_collator
        self.reducer = reducer or GradReducer()
        a = self.args
        if a.gradient_accumulation_steps < 1:
            raise ValueError("gradient_accumulation_steps must be >= 1")
        lr_at(a.lr_scheduler_type, 0, max(1, a.max_steps), 1.0, a.warmup_ratio)      # unsupported schedule: fail now
        self._reduce_hook = self.reducer.on_bucket_ready \
            if (self.reducer.world_size > 1 or getattr(self.reducer, "force", False)) else None
        # --gradient_accumulation_steps (HF Trainer; script/train/llava15_train.sh:23 passes 1): micro-batch gradients are
        # summed in an fp32 side buffer; on the last micro-batch each finished slice of flat_g is replaced by the mean and
        # only then handed to the all-reduce (DDP no_sync semantics: one exchange per optimizer step)
        self._micro = 0                     # micro-batches already accumulated in the current window
        self._finalize_accum = False
        self._gacc: Optional[torch.Tensor] = None
        self._loss_window: Optional[torch.Tensor] = None
        self.model.grad_ready_hook = self._bucket_ready if (self._reduce_hook or a.gradient_accumulation_steps > 1) else None
        rank = int(os.environ.get("RANK", "0"))
        if getattr(self.model, "lora", None) is not None:
            self.model.dropout_rank = rank              # LoRA dropout masks must differ across data-parallel ranks
        if a.gradient_checkpointing:
            self.model.gradient_checkpointing = True
        self.state = dict(global_step=0, log_history=[], epoch=0, batches_in_epoch=0)
        self._clip = torch.zeros(2, dtype=torch.float32, device=model.device)
        self._pending_metrics: Optional[torch.Tensor] = None

    def _bucket_ready(self, name: str, start: int, end: int):
        """model.backward fires this when flat_g[start:end] is final for the current micro-batch."""
        if self._finalize_accum and end > start:
            ops.grad_accum(self._gacc[start:end], self.model.store.flat_g[start:end], 2,
                           1.0 / self.args.gradient_accumulation_steps)
        if self._reduce_hook is not None and (self._finalize_accum or self.args.gradient_accumulation_steps == 1):
            self._reduce_hook(name, start, end)

    # ---------------------------------------------------------------- loss
    def compute_loss(self, model: LlavaDPOModel, inputs: dict, return_outputs: bool = False,
                     num_items_in_batch=None):
        """trainers.py:281-311.  Returns the 0-d loss tensor (device).  Metrics are left on the device
        (``self._pending_metrics``) and reduced/logged without a per-metric host sync."""
        if self.args.past_index >= 0:
            raise NotImplementedError
        data_dict = inputs
        policy_win, policy_rej, ref_win, ref_rej, beta = get_beta_and_logps(
            data_dict, model, self.args, is_llava15=True, save_for_backward=model.training)
        out: StepOutput = model.last_out
        sft_w = float(os.environ.get("SFT_weight", 0.0))      # trainers.py:299-300 (read on every call)
        dpo_w = float(os.environ.get("DPO_weight", 1.0))
        per_pair, scalars, coef = ops.dpo_loss(out.seq_logp, out.seq_cnt, ref_win, ref_rej, beta,
                                               self.args.dpo_use_average, sft_w, dpo_w)
        out.loss, out.scalars, out.per_pair = scalars[0], scalars, per_pair
        model.last_coef = coef
        # [chosen, rejected, logp_rej, logp_win, ref_rej, ref_win, acc]  (collect_preference_metrics order)
        self._pending_metrics = torch.stack([scalars[1], scalars[2], scalars[6], scalars[5], ref_rej.mean(),
                                             ref_win.mean(), scalars[3]])
        self._pending_task = "train" if model.training else "test"
        return (out.loss, out) if return_outputs else out.loss

    def pop_metrics(self) -> Dict[str, float]:
        """One fused cross-rank mean + one host transfer (replaces gather_and_do_mean x7, trainers.py:285-286)."""
        if self._pending_metrics is None:
            return {}
        m = self.reducer.reduce_metrics(self._pending_metrics).tolist()
        t = self._pending_task
        self._pending_metrics = None
        return self._metrics_dict(m, t)

    @staticmethod
    def _metrics_dict(m: List[float], t: str) -> Dict[str, float]:
        """The reference's metric names (collect_preference_metrics, trainers.py:148-157) for task ``t`` (train | test)."""
        d = {f"rewards_{t}/chosen": m[0], f"rewards_{t}/rejected": m[1], f"logps_{t}/rejected": m[2],
             f"logps_{t}/chosen": m[3], f"logps_{t}/ref_rejected": m[4], f"logps_{t}/ref_chosen": m[5],
             f"rewards_{t}/accuracies": m[6]}
        d[f"rewards_{t}/margins"] = d[f"rewards_{t}/chosen"] - d[f"rewards_{t}/rejected"]
        return d

    # ---------------------------------------------------------------- optimisation
    def current_lr(self) -> float:
        a = self.args
        return lr_at(a.lr_scheduler_type, self.state["global_step"], a.max_steps, a.learning_rate, a.warmup_ratio)

    def optimizer_step(self, lr: Optional[float] = None):
        """clip_grad_norm_(max_grad_norm) + AdamW on the flat buffers, then refresh the W^T copies."""
        a, st = self.args, self.model.store
        self.reducer.finish()                                  # all gradient buckets reduced (SUM over ranks)
        step = self.state["global_step"] + 1
        lr = self.current_lr() if lr is None else lr
        if getattr(self.reducer, "sharded", False):            # opt-in ZeRO-1 (dist.ShardedGradReducer + ShardedAdamW)
            self._sharded_optimizer().step(lr, a.adam_beta1, a.adam_beta2, a.adam_epsilon, a.weight_decay, step, a.max_grad_norm,
                                           self._clip)
            st.refresh_transposes(trainable_only=True)
            self.state["global_step"] = step
            return
        ops.grad_norm(st.flat_g, a.max_grad_norm, self._clip, pre_scale=1.0 / self.reducer.world_size)
        nd, tp = st.n_decay, st.train_p          # the optimizer owns flat_p[t0:] (everything, or adapters + projector)
        ops.adamw_step(tp[:nd], st.flat_master[:nd], st.flat_m[:nd], st.flat_v[:nd], st.flat_g[:nd], lr,
                       a.adam_beta1, a.adam_beta2, a.adam_epsilon, a.weight_decay, step, clip=self._clip)
        ops.adamw_step(tp[nd:], st.flat_master[nd:], st.flat_m[nd:], st.flat_v[nd:], st.flat_g[nd:], lr,
                       a.adam_beta1, a.adam_beta2, a.adam_epsilon, 0.0, step, clip=self._clip)
        st.refresh_transposes(trainable_only=True)
        self.state["global_step"] = step

    def _sharded_optimizer(self):
        """The ZeRO-1 optimizer of this trainer, built on first use: takes this rank's shards of the fp32 state from the store's
        replicated buffers (fresh from the parameters, or just loaded from a checkpoint) and FREES those buffers."""
        if getattr(self, "_zero1", None) is None:
            from .dist import ShardedAdamW
            st = self.model.store
            full = (st.flat_master, st.flat_m, st.flat_v) if st.flat_master is not None else None
            self._zero1 = ShardedAdamW(st.train_p, st.n_decay, self.reducer, st.bucket_schedule(), full_state=full)
            st.flat_master = st.flat_m = st.flat_v = None      # (N - 1) / N of the 12 bytes per parameter are gone
            st.sharded_optimizer = self._zero1                 # load_state_dict / merge_lora re-sync the masters through the store
        return self._zero1

    def training_step(self, inputs: dict) -> torch.Tensor:
        """forward + backward (+ overlapped gradient all-reduce) of ONE micro-batch; the optimizer runs when the
        accumulation window (``gradient_accumulation_steps`` micro-batches) is complete.  Returns the device loss (mean
        over the window once it closes, HF's reporting convention)."""
        self.model.train()
        ga, st = self.args.gradient_accumulation_steps, self.model.store
        last = self._micro == ga - 1
        self._finalize_accum = ga > 1 and last
        loss = self.compute_loss(self.model, inputs)
        self.model.backward(self.model.last_out, self.model.last_coef)
        if ga == 1:
            self.optimizer_step()
            return loss
        self._loss_window = loss.detach().clone() if self._micro == 0 else self._loss_window + loss.detach()
        if not last:
            if self._gacc is None:
                self._gacc = torch.empty(st.n_train, dtype=torch.float32, device=self.model.device)
            ops.grad_accum(self._gacc, st.flat_g, 0 if self._micro == 0 else 1)
            self._micro += 1
            return loss
        self._micro, self._finalize_accum = 0, False
        self.optimizer_step()
        return self._loss_window / ga

    # ---------------------------------------------------------------- loop / logging / saving
    def log(self, logs: Dict[str, float]):
        logs = dict(logs, step=self.state["global_step"])
        self.state["log_history"].append(logs)
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps(logs), flush=True)

    def _rank_indices(self, dataset, epoch: int, shuffle: bool, batch_size: int) -> List[int]:
        """This rank's sample indices for one pass: a permutation seeded with seed + epoch (a new order every epoch, the same
        on every rank), truncated so that every rank gets the same number of full batches, rank-strided."""
        world, rank = self.reducer.world_size, int(os.environ.get("RANK", "0"))
        n = len(dataset)
        if shuffle:
            from torch.utils.data import RandomSampler
            g = torch.Generator().manual_seed(self.args.seed + epoch)
            perm = list(iter(RandomSampler(dataset, generator=g)))      # ZephyrTrainer._get_train_sampler (trainers.py:45-51)
        else:
            perm = list(range(n))
        n_batches = (n // world) // batch_size
        if n_batches == 0:
            raise ValueError(f"dataset of {n} samples gives rank {rank} of {world} no full batch of {batch_size}")
        return perm[rank:(n // world) * world:world][:n_batches * batch_size]

    def get_train_dataloader(self, epoch: Optional[int] = None, skip_batches: int = 0):
        from torch.utils.data import DataLoader
        bs = self.args.per_device_train_batch_size
        state = getattr(self, "state", None) or {}
        idx = self._rank_indices(self.train_dataset, state.get("epoch", 0) if epoch is None else epoch, True, bs)
        idx = idx[skip_batches * bs:]                 # resume: the batches this epoch already consumed
        nw = int(getattr(self.args, "dataloader_num_workers", 0) or 0)
        return DataLoader(self.train_dataset, batch_size=bs, sampler=idx, collate_fn=self.data_collator, drop_last=True,
                          num_workers=nw)

    def get_eval_dataloader(self, eval_dataset=None):
        """This rank's share of the WHOLE evaluation set (rank-strided, nothing truncated; the last batch may be partial),
        like the HF Trainer's evaluation loop.  INVARIANT: ranks may run DIFFERENT numbers of evaluation batches, which is only
        correct because the evaluation forward issues no collective (the one [sums, count] all-reduce of evaluate() happens
        once per rank, after the loop); a collective added to compute_loss / forward must pad the ranks to equal batch counts."""
        from torch.utils.data import DataLoader
        ds = eval_dataset if eval_dataset is not None else self.eval_dataset
        if ds is None:
            raise ValueError("evaluate() needs an eval_dataset")
        world, rank = self.reducer.world_size, int(os.environ.get("RANK", "0"))
        idx = list(range(len(ds)))[rank::world]
        bs = max(1, min(int(getattr(self.args, "per_device_eval_batch_size", 8)), len(idx)))
        return DataLoader(ds, batch_size=bs, sampler=idx, collate_fn=self.data_collator, drop_last=False,
                          num_workers=int(getattr(self.args, "dataloader_num_workers", 0) or 0))

    def evaluate(self, eval_dataset=None) -> Dict[str, float]:
        """The evaluation pass of the HF Trainer over ``compute_loss`` with ``model.training == False``: the reference
        then logs the same preference metrics under ``*_test/*`` (trainers.py:303).  Forward only (no activations kept);
        every sample of the set is evaluated once; returns the SAMPLE-weighted mean of every metric plus ``eval_loss``
        over all ranks (one fused all-reduce of [sums, count])."""
        was_training = self.model.training
        # evaluate() overwrites model.last_out / last_coef and the pending metrics: calling it between a training compute_loss
        # and its backward / pop_metrics would corrupt that step - saved here, restored below
        saved = (getattr(self.model, "last_out", None), getattr(self.model, "last_coef", None), self._pending_metrics,
                 getattr(self, "_pending_task", None))
        self.model.eval()
        dev = self.model.device
        tot, cnt = torch.zeros(8, dtype=torch.float32, device=dev), 0
        try:
            for batch in self.get_eval_dataloader(eval_dataset):
                nb = int(batch["win_input_ids"].shape[0])
                loss = self.compute_loss(self.model, batch)
                tot += nb * torch.cat([self._pending_metrics, loss.detach().reshape(1)])
                cnt += nb
        finally:
            # restored on EVERY exit (an exception in the loop or the empty-set error below must not leave an in-flight
            # training step's state clobbered: ADVICE r4)
            self.model.train(was_training)
            self.model.last_out, self.model.last_coef, self._pending_metrics, self._pending_task = saved
        # reduce_metrics returns the cross-rank MEAN: mean(sums) / mean(counts) = sum / count over all ranks
        red = self.reducer.reduce_metrics(torch.cat([tot, torch.tensor([float(cnt)], device=dev)]))
        if float(red[8]) <= 0:
            raise ValueError("evaluate(): empty evaluation set")
        mean = (red[:8] / red[8]).tolist()
        m = self._metrics_dict(mean[:7], "test")
        m["eval_loss"] = float(mean[7])
        self.log(m)
        return m

    def train(self, resume_from_checkpoint=None):
        a = self.args
        if resume_from_checkpoint:
            self.load_checkpoint(resume_from_checkpoint)
        t0 = time.time()
        while self.state["global_step"] < a.max_steps:
            done = False
            for batch in self.get_train_dataloader(self.state["epoch"], self.state["batches_in_epoch"]):
                loss = self.training_step(batch)
                self.state["batches_in_epoch"] += 1
                if self._micro != 0:
                    continue                               # inside an accumulation window: no optimizer step yet
                step = self.state["global_step"]
                if step % a.logging_steps == 0:
                    m = self.pop_metrics()
                    m.update(loss=float(loss), learning_rate=self.current_lr(), grad_norm=float(self._clip[0]))
                    if not math.isfinite(m["loss"]):       # reference: print + exit() on NaN (trainers.py:263-271)
                        raise FloatingPointError(f"non-finite loss at step {step}")
                    self.log(m)
                if a.save_steps and step % a.save_steps == 0:
                    self.save_checkpoint(os.path.join(a.output_dir, f"checkpoint-{step}"))
                if step >= a.max_steps:
                    done = True
                    break
            if not done:                                   # the pass over this rank's shard is complete
                self.state["epoch"] += 1
                self.state["batches_in_epoch"] = 0
        return dict(train_runtime=time.time() - t0, global_step=self.state["global_step"])

    def _save(self, output_dir: str, state_dict=None):
        """HF-layout weights (safe_save_model_for_hf_trainer, train_llava15.py:102-112)."""
        if int(os.environ.get("RANK", "0")) != 0:
            return
        from .checkpoint import save_lora_adapter, save_pretrained, save_state_dict_sharded
        if self.model.lora is not None and state_dict is None:
            save_lora_adapter(self.model, output_dir)        # adapter + non_lora_trainables.bin (train_llava15_lora.py:184-197)
        elif state_dict is None:
            save_pretrained(self.model, output_dir)          # sharded safetensors + index + config.json, HF names
        else:
            save_state_dict_sharded(state_dict, output_dir, self.model.cfg)

    def save_state(self):
        if int(os.environ.get("RANK", "0")) != 0:
            return
        os.makedirs(self.args.output_dir, exist_ok=True)
        with open(os.path.join(self.args.output_dir, "trainer_state.json"), "w") as f:
            json.dump(self.state, f)

    def _optimizer_layout(self) -> dict:
        """What the raw flat buffers of optimizer.pt mean: entry order / offsets (CRC), the gate|up row interleave of the
        fused MLP weight (RV_FUSE_SWIGLU) and the vocabulary padding.  A blob written under another layout has the same
        length but different row order - load_checkpoint refuses it instead of scrambling weights and Adam state."""
        import zlib
        st = self.model.store
        order = "|".join(f"{k}:{st.offsets[k][0]}:{'x'.join(map(str, st.offsets[k][1]))}" for k, _, _ in st.entries)
        return dict(version=1, interleave_gu=bool(st.interleave_gu), vocab_padded=int(self.model.cfg.vocab_padded),
                    n_train=int(st.n_train), t0=int(st.t0), entries_crc32=zlib.crc32(order.encode()))

    def save_checkpoint(self, path: str):
        st = self.model.store
        if getattr(self.reducer, "sharded", False):
            # ZeRO-1: the file keeps the REPLICATED layout (full fp32 master / m / v in flat order), so a sharded run resumes a
            # replicated one and back.  Collective: every rank takes part in the gather, rank 0 writes.
            full = self._sharded_optimizer().gather_full_state()        # host tensors on rank 0 only (None elsewhere)
            if int(os.environ.get("RANK", "0")) == 0:
                master, m_, v_ = full
                os.makedirs(path, exist_ok=True)
                torch.save(dict(master=master, m=m_, v=v_, state=self.state, dropout_step=int(self.model._dropout_step),
                                layout=self._optimizer_layout()), os.path.join(path, "optimizer.pt"))
                self._save(path)
            return
        if int(os.environ.get("RANK", "0")) != 0:
            return
        os.makedirs(path, exist_ok=True)
        # data position (epoch + batches consumed in it) rides in self.state; the LoRA dropout counter too, so that a
        # resumed run draws the masks the uninterrupted run would have drawn
        torch.save(dict(master=st.flat_master.cpu(), m=st.flat_m.cpu(), v=st.flat_v.cpu(), state=self.state,
                        dropout_step=int(self.model._dropout_step), layout=self._optimizer_layout()),
                   os.path.join(path, "optimizer.pt"))
        self._save(path)

    def load_checkpoint(self, path: str):
        st = self.model.store
        blob = torch.load(os.path.join(path, "optimizer.pt"), map_location="cpu")
        want = self._optimizer_layout()
        have = blob.get("layout")
        if have is None:
            # A blob from before the descriptor existed has an UNKNOWN layout: the same revision wrote interleaved gate|up rows
            # by default (RV_FUSE_SWIGLU=1) and block rows otherwise, with the same length - nothing in the blob tells them
            # apart.  Never guess: the caller names the layout the blob was written under.
            legacy = os.environ.get("RV_CKPT_LEGACY_LAYOUT", "")
            if legacy not in ("interleaved", "block"):
                raise ValueError(f"{path}/optimizer.pt carries no layout descriptor (written before checkpoints recorded the gate|up "
                                 "row order and vocabulary padding), so its row order cannot be verified.  Set "
                                 "RV_CKPT_LEGACY_LAYOUT=interleaved (written with the default RV_FUSE_SWIGLU=1) or =block "
                                 "(RV_FUSE_SWIGLU=0) to state it, or load the HF-layout weights (from_pretrained) and restart the "
                                 "optimizer")
            have = dict(version=0, interleave_gu=legacy == "interleaved", vocab_padded=want["vocab_padded"],
                        n_train=int(blob["master"].numel()), t0=want["t0"], entries_crc32=None)
        diff = [k for k in ("interleave_gu", "vocab_padded", "n_train", "t0") if have.get(k) != want[k]]
        if have.get("entries_crc32") not in (None, want["entries_crc32"]):
            diff.append("entries_crc32")
        if diff:
            raise ValueError(f"{path}/optimizer.pt was written under another parameter layout ({', '.join(diff)} differ: checkpoint "
                             f"{ {k: have.get(k) for k in diff} } vs this model { {k: want[k] for k in diff} }); resume with the "
                             "same RV_FUSE_SWIGLU / LoRA / vocabulary settings, or load the HF-layout weights "
                             "(from_pretrained) and restart the optimizer")
        if getattr(self.reducer, "sharded", False):
            z = self._sharded_optimizer()
            z.load_full_state(blob["master"], blob["m"], blob["v"])
            st.train_p.copy_(blob["master"])                     # fp32 -> bf16, round to nearest even like rv_cast_f32_to_bf16
        else:
            st.flat_master.copy_(blob["master"]), st.flat_m.copy_(blob["m"]), st.flat_v.copy_(blob["v"])
            ops.cast_f32_to_bf16(st.flat_master, st.train_p)
        st.refresh_transposes(trainable_only=True)
        self.state = dict(dict(epoch=0, batches_in_epoch=0), **blob["state"])
        self.model._dropout_step = int(blob.get("dropout_step", 0))
        self._micro = 0
