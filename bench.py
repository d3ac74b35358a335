#!/usr/bin/env python
"""Headline benchmark: preference-pairs/sec of one full DPO optimisation step (forward + backward +
gradient all-reduce + clip + AdamW) of LLaVA-1.5-7B (CLIP-ViT-L/14-336 + Vicuna-7B) in bf16, spliced
sequence length 2048, on N MI355X of one node (BASELINE.json configs[1] / configs[2]).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 5 --warmup 2

Synthetic (image, chosen, rejected) triples and HF-default random-init weights (no datasets/checkpoints
exist offline); inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: required by RCCL across processes on this driver

import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0      # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def flops_per_seq(n_tok: float, layers: int = 32, d: int = 4096, f: int = 11008, V: int = 32000,
                  lora_r: int = 0, n_tgt: float = None, kvd: int = None) -> float:
    """Algorithmic FLOPs of fwd + bwd over one sequence of n_tok tokens (causal attention at half, no recompute).
    Full fine-tune: forward + input gradients + weight gradients = 3 passes over the linear layers.  LoRA: the base
    weights are frozen -> 2 passes, plus the adapters (forward t = xA^T and tB^T; backward dt, dt A, dA, dB = 2x).
    ``n_tgt``: rows that reach the LM head (positions whose next label is a target); None = all n_tok rows, SURVEY's
    reference-layout accounting (the reference computes logits for every position)."""
    kvd = d if kvd is None else kvd                      # grouped-query attention: k / v projections are [kvd, d]
    per_tok_linear = layers * (4 * d * d + 4 * d * kvd + 6 * d * f)
    per_tok_attn = layers * 2 * d * n_tok
    per_tok_lora = layers * 2 * lora_r * (4 * 2 * d + 3 * (d + f))
    passes = 2 if lora_r else 3
    head = (n_tok if n_tgt is None else n_tgt) * passes * 2 * d * V
    return n_tok * (passes * per_tok_linear + 3 * per_tok_attn + 3 * per_tok_lora) + head


def flops_per_pair(L: int, layers: int = 32, d: int = 4096, f: int = 11008, V: int = 32000, lora_r: int = 0,
                   n_tgt: float = None, kvd: int = None, vision: float = 0.366e12 + 3 * 0.024e12) -> float:
    """algorithmic FLOPs of one pair (two sequences of L tokens), CLIP once per pair (forward
    only) + projector (fwd+bwd).  Full fine-tune at L = 2048: 169.4 TFLOP (n_tgt None = LM head on every position)."""
    return 2 * flops_per_seq(L, layers, d, f, V, lora_r, n_tgt, kvd) + vision


class GemmTimer:
    """HIP-event timing of every rv_gemm_nt_bf16 launch on the stream it is launched on."""

    def __init__(self):
        self.records = []

    def install(self):
        from rlaif_v_amd import ops, hip
        orig = ops.gemm_nt
        recs = self.records

        def timed(a, b, out=None, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(a, b, out=out, **kw)
            e.record()
            recs.append((s, e, 2.0 * a.shape[0] * b.shape[0] * a.shape[1],
                         2.0 * (a.shape[0] * a.shape[1] + b.shape[0] * b.shape[1] + a.shape[0] * b.shape[0])))
            return r

        ops.gemm_nt = timed
        orig_tn = ops.gemm_tn

        def timed_tn(p, q, out=None, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_tn(p, q, out=out, **kw)
            e.record()
            recs.append((s, e, 2.0 * p.shape[0] * p.shape[1] * q.shape[1],
                         2.0 * (p.shape[0] * p.shape[1] + q.shape[0] * q.shape[1] + p.shape[1] * q.shape[1])))
            return r

        ops.gemm_tn = timed_tn
        orig_lora = ops.gemm_nt_lora

        def timed_lora(a, b, a2, b2, **kw):       # fused LoRA GEMM: K + K2 contraction steps
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_lora(a, b, a2, b2, **kw)
            e.record()
            kk = a.shape[1] + b2.shape[1]
            recs.append((s, e, 2.0 * a.shape[0] * b.shape[0] * kk,
                         2.0 * (a.shape[0] * kk + b.shape[0] * kk + a.shape[0] * b.shape[0])))
            return r

        ops.gemm_nt_lora = timed_lora
        orig_nn = ops.gemm_nn

        def timed_nn(a, b, out=None, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_nn(a, b, out=out, **kw)
            e.record()
            recs.append((s, e, 2.0 * a.shape[0] * b.shape[1] * a.shape[1],
                         2.0 * (a.shape[0] * a.shape[1] + b.shape[0] * b.shape[1] + a.shape[0] * b.shape[1])))
            return r

        ops.gemm_nn = timed_nn
        self._restore = lambda: (setattr(ops, "gemm_nt", orig), setattr(ops, "gemm_tn", orig_tn),
                                 setattr(ops, "gemm_nt_lora", orig_lora), setattr(ops, "gemm_nn", orig_nn))

    def summary(self):
        tot_ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        tot_fl = sum(r[2] for r in self.records)
        alg_bytes = sum(r[3] for r in self.records)
        n = len(self.records)
        return dict(launches=n, total_ms=tot_ms, avg_ms=tot_ms / max(n, 1), tflops=tot_fl / max(tot_ms, 1e-9) / 1e9,
                    flops=tot_fl, alg_bytes=alg_bytes)


def cpu_baseline(seed: int = 0, pairs: int = 4, text_len: int = 512):
    """The oracle (a port of the reference's step) timed on THIS box's host cores at BASELINE config 1's shape
    (BASELINE.md section 2): 4 synthetic 336-px pairs, text length 512 -> spliced length 1087, fp32, one fwd + bwd + clip +
    AdamW step at full 7B widths; bounded sample = depths 1 and 2 of the language model (CLIP at full depth), the per-layer
    slope extrapolated linearly to 32 layers.  The reference's OWN functions, timed the same way in the build container
    (tools/cpu_reference_baseline.py -> profiles/r02_cpu_reference_baseline.json), ride along as ``reference_run``."""
    from oracle import dpo_oracle as O
    cores = os.cpu_count() or 1
    threads = min(cores, 128)
    torch.set_num_threads(threads)
    res = {}
    for depth in (1, 2):
        cfg = O.LlavaCfg(layers=depth, model_max_length=2048)
        W = O.make_weights(cfg, seed=seed, bf16_round=False)
        batch = O.make_synthetic_batch(cfg, pairs, text_len, 64, seed=seed, ragged=False)
        ph = {}
        O.dpo_train_step(batch, W, cfg, {}, lr=5e-7, step=1, sft_weight=0.0, dpo_weight=1.0, timings=ph)
        res[depth] = ph
        del W
    per_layer = {k: max(res[2][k] - res[1][k], 0.0) for k in ("fwd_s", "bwd_s", "opt_s")}
    fixed = {k: max(res[1][k] - per_layer[k], 0.0) for k in per_layer}
    full = {k: fixed[k] + 32 * per_layer[k] for k in per_layer}
    step = sum(full.values())
    out = dict(value=pairs / step, unit="pairs/s", cores=threads, kind="port",
               sample=f"oracle fp32 step, config 1 ({pairs} pairs, T={text_len}, L={text_len + 575}), depths 1,2 -> 32 layers: {step:.0f} s",
               host_cores=cores, step_s_extrapolated=step,
               phases_s={k[:-2]: round(v, 2) for k, v in full.items()},
               measured_s={str(d): {k[:-2]: round(v, 2) for k, v in res[d].items()} for d in res})
    try:
        with open(os.path.join(REPO, "profiles", "r02_cpu_reference_baseline.json")) as fh:
            ref = json.load(fh)
        out["reference_run"] = dict(kind="reference", where="build container", cores=ref["cores"], cpu=ref["cpu"],
                                    value=ref["pairs_per_s"], unit="pairs/s", step_s_extrapolated=ref["extrapolated_32_layers_s"]["step"])
    except Exception:
        pass
    return out


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without a launcher: re-run this script under torch.distributed.run with one rank per
    GPU (the command line the driver itself uses) and pass its output / exit code through.  Refuses when the node has
    fewer than N GPUs - a line with n_gpus < N is never printed."""
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node; refusing to measure fewer ranks")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs-per-gpu", type=int, default=8)
    ap.add_argument("--seq-len", type=int, default=2048, help="spliced length L (text length = L - 575)")
    ap.add_argument("--layers", type=int, default=32, help="debug only: anything but 32 is not the headline config")
    ap.add_argument("--lora", action="store_true",
                    help="LoRA-DPO workload (BASELINE.json configs[4]: rank 64 adapters on all decoder projections, "
                         "frozen base; use with --seq-len 4096).  Not the headline line.")
    ap.add_argument("--lora-r", type=int, default=64)
    ap.add_argument("--gradient-checkpointing", action="store_true", help="re-run each decoder layer in backward")
    ap.add_argument("--omnilmm", action="store_true",
                    help="BASELINE config 4 shape: OmniLMM-12B language side (Mistral-7B, 8 kv heads, f 14336) + Resampler, "
                         "64 image tokens; the frozen EVA02 tower is NOT run - synthetic precomputed tower tokens")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dp-probe", action="store_true",
                    help="skip the 1-rank RCCL probe (steps re-timed with the bucketed all-reduce forced on)")
    ap.add_argument("--no-gemm-timer", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))                 # no launcher: spawn one rank per GPU ourselves
    from rlaif_v_amd.dist import init_process_group_from_env, BucketedAllReduce
    rank, local, world = init_process_group_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from rlaif_v_amd.model import LlavaConfig, LlavaDPOModel, LoraConfig
    from rlaif_v_amd.trainer import LLaVA15DPOTrainer, TrainingArguments
    from rlaif_v_amd.data import SyntheticPreferenceDataset, DataCollatorForDPODataset
    import torch.distributed as dist

    L, B = args.seq_len, args.pairs_per_gpu
    lora = LoraConfig(r=args.lora_r) if args.lora else None        # peft defaults of train_llava15_lora.py:111-116
    if args.omnilmm:
        from rlaif_v_amd.omnilmm import OmniLMMConfig, OmniLMMDPOModel
        cfg = OmniLMMConfig(layers=args.layers, model_max_length=L)
        model = OmniLMMDPOModel(cfg, device=dev, lora=lora)
    else:
        cfg = LlavaConfig(layers=args.layers, model_max_length=L)
        model = LlavaDPOModel(cfg, device=dev, lora=lora)
    model.init_random(seed=0)            # identical weights on every rank
    reducer = BucketedAllReduce(model.store.flat_g) if world > 1 else None
    targs = TrainingArguments(max_steps=1000, per_device_train_batch_size=B, lora_enable=args.lora,
                              lora_r=args.lora_r, learning_rate=1e-5 if args.lora else 5e-7,
                              gradient_checkpointing=args.gradient_checkpointing)
    trainer = LLaVA15DPOTrainer(model=model, args=targs, reducer=reducer)

    class _Tok:
        pad_token_id = cfg.pad_token_id
    if args.omnilmm:      # the image span does not change the length: text length = L; 1024 tower tokens of width 1792 per image
        ds = SyntheticPreferenceDataset(n=B * world, vocab=32000, text_len=L, prompt_len=64 + cfg.num_query + 2, seed=1234,
                                        omnilmm=dict(tokens=(cfg.im_patch_token, cfg.im_start_token, cfg.im_end_token),
                                                     num_query=cfg.num_query, tower_tokens=(cfg.image_size // 14) ** 2,
                                                     width=cfg.vision_width))
    else:
        ds = SyntheticPreferenceDataset(n=B * world, vocab=cfg.vocab, text_len=L - (cfg.n_patches - 1), prompt_len=64,
                                        image_size=cfg.image_size, seed=1234)
    collate = DataCollatorForDPODataset(_Tok(), beta=0.1, mod_token_weight=1.0)
    batch = collate([ds[rank * B + i] for i in range(B)])
    batch["images"] = batch["images"].to(dev)       # inputs resident in HBM before the timed region

    def one_step():
        return trainer.training_step(dict(batch))

    for _ in range(args.warmup):
        one_step()
    timer = GemmTimer()
    if not args.no_gemm_timer:
        timer.install()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = None
    for _ in range(args.steps):
        loss = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    dp_probe = None
    if world == 1 and not args.no_dp_probe:
        # the data-parallel exchange on ONE rank: the same bucketed all-reduce schedule the N-GPU run issues (RCCL kernels
        # on RCCL's stream, overlapped with backward, optimizer waits on the handles), forced on in a 1-rank group.  It
        # prices the launch / stream / CU-sharing side of the overlap; the xGMI transfer itself needs >= 2 GPUs.
        try:
            os.environ.update(RANK="0", LOCAL_RANK=str(local), WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                              MASTER_PORT=str(_free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            red = BucketedAllReduce(model.store.flat_g, force=True)
            sent = []
            _launch = red._launch
            red._launch = lambda a, b: (sent.append(b - a), _launch(a, b))[1]
            trainer.reducer, trainer._reduce_hook = red, red.on_bucket_ready
            model.grad_ready_hook = trainer._bucket_ready
            if not args.no_gemm_timer:
                timer._restore()
            one_step()
            torch.cuda.synchronize()
            sent.clear()
            per_step = []
            for _ in range(3):        # median of three: RCCL's lazy channel setup can land in any one of the first steps
                t1 = time.perf_counter()
                one_step()
                torch.cuda.synchronize()
                per_step.append((time.perf_counter() - t1) * 1e3)
            ms_forced = sorted(per_step)[1]
            dp_probe = dict(ms_per_step=ms_forced, ms_each=[round(x, 1) for x in per_step], collectives_per_step=len(sent) // 3,
                            bytes_per_step=sum(sent) // 3 * model.store.flat_g.element_size())
            dist.destroy_process_group()
        except Exception as e:          # the probe must never cost the headline line
            dp_probe = dict(error=repr(e)[:200])

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        pairs_per_s = B * world * args.steps / dt
        lr_ = args.lora_r if args.lora else 0
        if args.omnilmm:     # resampler fwd + bwd per image: kv_proj, k / v / q / out / proj projections, 64 x N attention
            nt_, w_, d_, nq_ = (cfg.image_size // 14) ** 2, cfg.vision_width, cfg.hidden, cfg.num_query
            vis = 3.0 * (2 * nt_ * w_ * d_ + 2 * 2 * nt_ * d_ * d_ + 3 * 2 * nq_ * d_ * d_ + 4 * nq_ * nt_ * d_)
        else:
            vis = 0.366e12 + 3 * 0.024e12
        dims = dict(d=cfg.hidden, f=cfg.ffn, V=cfg.vocab, kvd=cfg.kv_dim, vision=vis)
        fp_nominal = flops_per_pair(L, layers=args.layers, lora_r=lr_, **dims)       # LM head on every position
        # FLOPs actually required (report the MFMA fraction on these): the LM head runs only on
        # the rows whose next label is a target, and the shared prefix of each pair is computed once
        plan = model.last_out.plan
        shared = plan.shared_len or [0] * B
        seq_dims = {k: v for k, v in dims.items() if k != "vision"}
        saved = sum(flops_per_seq(p, layers=args.layers, lora_r=lr_, n_tgt=0, **seq_dims) for p in shared) / max(len(shared), 1)
        fp = flops_per_pair(L, layers=args.layers, lora_r=lr_, n_tgt=plan.n_sel / (2.0 * B), **dims) - saved
        step_tflops_per_gpu = fp * (pairs_per_s / world) / 1e12
        line = {
            "metric": "preference-pairs/sec (DPO step) " + ("OmniLMM-12B (tower excluded)" if args.omnilmm else "LLaVA-1.5-7B") + " bf16", "value": pairs_per_s, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("OmniLMM-12B language side (Mistral-7B: 8 kv heads, f 14336) + Resampler (64 queries x 1024 "
                                    "tower tokens); frozen EVA02-E tower NOT run (precomputed synthetic tower tokens) "
                                    if args.omnilmm else "LLaVA-1.5-7B (CLIP-ViT-L/14-336 + Vicuna-7B) ")
                                   + (f"LoRA (r={args.lora_r}, all 7 decoder projections, dropout 0.05)" if args.lora else "full-FT")
                                   + f" DPO step, {cfg.image_size}px, seq_len={L}, {B} pairs/GPU, random-init weights",
                       "pairs_per_gpu": B, "global_batch_pairs": B * world, "seq_len": L, "llm_layers": args.layers,
                       "parallelism": f"dp{world}", "optimizer": "AdamW fp32 master + clip 1.0",
                       "trainable_params": int(model.store.n_train),
                       "gradient_checkpointing": bool(args.gradient_checkpointing), "shared_prefix_reuse": bool(model.share_prefix)},
            "loss": float(loss), "max_memory_allocated_gb": torch.cuda.max_memory_allocated() / 2**30,
            "max_memory_reserved_gb": torch.cuda.max_memory_reserved() / 2**30,
            "step_tflops_per_gpu": step_tflops_per_gpu, "step_mfma_frac": step_tflops_per_gpu / PEAK_BF16_TFLOPS,
            "flops_per_pair": fp, "flops_per_pair_reference_layout": fp_nominal,
            "shared_prefix_tokens_per_pair": sum(shared) / max(len(shared), 1),
            "tokens_per_step_per_gpu": plan.n_real_tokens,
        }
        if not args.no_gemm_timer:
            g = timer.summary()
            traffic, traffic_file = None, None
            for name in ("r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):   # newest committed PMC passes first
                try:   # HBM bytes per GEMM launch (profiles/, separate rocprofv3 --pmc runs of this same command)
                    with open(os.path.join(REPO, "profiles", name)) as fh:
                        traffic = json.load(fh)["gemm_all_launches_hbm_bytes_per_launch"]
                    traffic_file = name
                    break
                except Exception:
                    pass
            line["roofline"] = {"bound": "mfma", "kernel": "gemm_nn_a64_kernel / gemm_tn_256_kernel (+ gemm_nn_256 / gemm_nt_256 for the shapes they serve): 256x256 ping-pong tiles; all rv_gemm_nn_bf16 + rv_gemm_tn_bf16 + rv_gemm_nt_bf16 launches",
                                "achieved": g["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                "frac": g["tflops"] / PEAK_BF16_TFLOPS, "traffic": traffic,
                                "traffic_note": "HBM bytes per GEMM launch, rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + "
                                                f"WRITE_SIZE, separate passes (profiles/{traffic_file}); algorithmic "
                                                "operand+result bytes per launch: " + f"{g['alg_bytes'] / max(g['launches'], 1):.3e}",
                                "power_capped_mfma_ceiling": {"tflops": 1953.0, "frac": g["tflops"] / 1953.0,
                                                              "note": "pure register-operand v_mfma_f32_16x16x32_bf16 loop on all 256 CUs "
                                                                      "under the 1400 W package cap (32x32x16: 1750): "
                                                                      "profiles/r02_mfma_shape_power_probe.log"},
                                "launches": g["launches"], "avg_launch_ms": g["avg_ms"],
                                "gemm_ms_per_step": g["total_ms"] / args.steps}
        if dp_probe is not None:
            if "ms_per_step" in dp_probe:
                dp_probe["exposed_ms_per_step"] = dp_probe["ms_per_step"] - ms_per_step
            line["dp_overlap_probe_1rank"] = dp_probe
        if world == 1 and not args.no_cpu_baseline and not args.lora and not args.omnilmm:     # the CPU leg times the full-FT oracle step
            line["cpu_baseline"] = cpu_baseline()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio (flushed at exit, i.e. AFTER a Python print): drain it first so
        # that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
