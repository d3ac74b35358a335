"""RCCL path on the real device: a 1-rank NCCL(=RCCL) group with the collectives forced on, so the asynchronous
bucketed all-reduce of the bf16 gradient buffer, its overlap with backward and the optimizer's wait are exercised
on an MI355X (the 8-GPU run itself belongs to the driver).  Results must equal the run without a process group."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dpo_oracle as O  # noqa: E402


def test_training_step_with_rccl_bucket_reduce():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.distributed as dist
    from rlaif_v_amd.dist import BucketedAllReduce, init_process_group_from_env
    from rlaif_v_amd.model import LlavaConfig, LlavaDPOModel
    from rlaif_v_amd.trainer import LLaVA15DPOTrainer, TrainingArguments
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if not dist.is_initialized():
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg = O.tiny_cfg()
        W = O.make_weights(cfg, seed=9)
        batch = O.make_synthetic_batch(cfg, 2, 40, 12, seed=9)
        losses, masters = [], []
        for with_pg in (False, True):
            model = LlavaDPOModel(LlavaConfig(**O.asdict(cfg)))
            model.load_state_dict(W)
            red = BucketedAllReduce(model.store.flat_g, bucket_bytes=1 << 20, force=True) if with_pg else None
            tr = LLaVA15DPOTrainer(model=model, args=TrainingArguments(learning_rate=1e-3, warmup_ratio=0.0,
                                                                       lr_scheduler_type="constant"), reducer=red)
            for _ in range(2):
                loss = tr.training_step(dict(batch))
            if with_pg:
                m = tr.pop_metrics()
                assert "rewards_train/accuracies" in m and len(m) == 8
            torch.cuda.synchronize()
            losses.append(float(loss))
            masters.append(model.store.flat_master.clone())
        assert losses[0] == losses[1]
        assert torch.equal(masters[0], masters[1])        # SUM over one rank == identity, bit for bit
    finally:
        dist.destroy_process_group()


def test_zero1_sharded_optimizer_on_device_one_rank(tmp_path):
    """Opt-in ZeRO-1 on the REAL kernels (rv_grad_sumsq, rv_clip_from_sumsq, rv_adamw_step on chunk views, RCCL reduce-scatter /
    all-gather forced on in a 1-rank group): two training steps give bit-identical parameters and loss to the replicated path,
    the full-size fp32 buffers are gone, and a checkpoint written by the sharded run (replicated file format) resumes in a
    REPLICATED trainer to the same third step.  The partition logic for W > 1 is covered under gloo (tests/test_dist_gloo.py)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.distributed as dist
    from rlaif_v_amd.dist import ShardedGradReducer
    from rlaif_v_amd.model import LlavaConfig, LlavaDPOModel
    from rlaif_v_amd.trainer import LLaVA15DPOTrainer, TrainingArguments
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if not dist.is_initialized():
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg = O.tiny_cfg()
        W = O.make_weights(cfg, seed=9)
        batch = O.make_synthetic_batch(cfg, 2, 40, 12, seed=9)
        args = dict(learning_rate=1e-3, warmup_ratio=0.0, lr_scheduler_type="constant", output_dir=str(tmp_path))
        runs = {}
        for sharded in (False, True):
            model = LlavaDPOModel(LlavaConfig(**O.asdict(cfg)))
            model.load_state_dict(W)
            red = ShardedGradReducer(model.store.flat_g, bucket_bytes=1 << 20, force=True) if sharded else None
            tr = LLaVA15DPOTrainer(model=model, args=TrainingArguments(**args), reducer=red)
            for _ in range(2):
                loss = tr.training_step(dict(batch))
            torch.cuda.synchronize()
            runs[sharded] = (float(loss), model.store.flat_p.clone(), model, tr)
        assert runs[True][0] == runs[False][0] and torch.equal(runs[True][1], runs[False][1])
        m_sh, tr_sh = runs[True][2], runs[True][3]
        assert m_sh.store.flat_master is None and tr_sh._zero1.master.numel() == m_sh.store.n_train      # W = 1: the whole range
        full = tr_sh._zero1.gather_full_state()
        assert torch.equal(full[0], runs[False][2].store.flat_master.cpu()) and torch.equal(full[2], runs[False][2].store.flat_v.cpu())
        # checkpoint interoperability: sharded save -> replicated resume -> the same third step as the never-interrupted replicated run
        ck = os.path.join(str(tmp_path), "checkpoint-2")
        tr_sh.save_checkpoint(ck)
        model3 = LlavaDPOModel(LlavaConfig(**O.asdict(cfg)))
        model3.load_state_dict(W)
        tr3 = LLaVA15DPOTrainer(model=model3, args=TrainingArguments(**args))
        tr3.load_checkpoint(ck)
        l3 = tr3.training_step(dict(batch))
        l_ref = runs[False][3].training_step(dict(batch))
        torch.cuda.synchronize()
        assert float(l3) == float(l_ref) and torch.equal(model3.store.flat_p, runs[False][2].store.flat_p)
    finally:
        dist.destroy_process_group()


def _cat_batches(batches):
    """Union of collator batches: wins of all shards, then rejects of all shards (rows right-padded to a common width)."""
    def cat(key):
        ts = [b[key] for b in batches]
        if ts[0].dim() == 2:
            n = max(t.shape[1] for t in ts)
            fill = -100 if key.endswith("labels") else 0
            ts = [torch.nn.functional.pad(t, (0, n - t.shape[1]), value=fill) for t in ts]
        return torch.cat(ts, 0)
    out = {}
    for k, v in batches[0].items():
        if torch.is_tensor(v) and v.dim() > 0 and not k.startswith("concatenated"):
            out[k] = cat(k)
        elif not torch.is_tensor(v):
            out[k] = v
    from rlaif_v_amd.data import concate_pad
    out["concatenated_input_ids"] = concate_pad(out["win_input_ids"], out["rej_input_ids"], 0)
    out["concatenated_labels"] = concate_pad(out["win_labels"], out["rej_labels"], -100)
    out["concatenated_attention_mask"] = out["concatenated_input_ids"].ne(0)
    out["concatenated_token_weight"] = concate_pad(out["win_token_weight"], out["rej_token_weight"], 0)
    return out


def _two_rank_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    # RCCL's own account of what it built (ranks, channels, transport) goes to a per-rank file: the first multi-GPU run shows
    # "RCCL saw N ranks, C channels" next to the replicas-identical assertion
    log = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                       "gpurun_out", f"rccl_debug_rank{rank}.log")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    os.environ.setdefault("NCCL_DEBUG", "INFO")
    os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH")
    os.environ.setdefault("NCCL_DEBUG_FILE", log)
    import torch.distributed as dist
    from rlaif_v_amd.dist import BucketedAllReduce, init_process_group_from_env
    from rlaif_v_amd.model import LlavaConfig, LlavaDPOModel
    from rlaif_v_amd.trainer import LLaVA15DPOTrainer, TrainingArguments
    r, local, w = init_process_group_from_env()
    torch.cuda.set_device(local)
    cfg = O.tiny_cfg()
    model = LlavaDPOModel(LlavaConfig(**O.asdict(cfg)), device=f"cuda:{local}")
    model.load_state_dict(O.make_weights(cfg, seed=9))                      # identical replicas
    red = BucketedAllReduce(model.store.flat_g, bucket_bytes=1 << 20)
    tr = LLaVA15DPOTrainer(model=model, args=TrainingArguments(learning_rate=1e-3, warmup_ratio=0.0,
                                                               lr_scheduler_type="constant"), reducer=red)
    # (1) "2 ranks x B pairs == 1 rank x 2B pairs": the all-reduced gradient / world must equal the gradient of ONE process on the
    # union of the shards (loss = mean over pairs; the 1/world is folded into the clip factor, flat_g holds the SUM)
    shards = [O.make_synthetic_batch(cfg, 2, 40, 12, seed=50 + k) for k in range(w)]
    tr.compute_loss(model, dict(shards[rank]))
    model.backward(model.last_out, model.last_coef)
    red.finish()
    torch.cuda.synchronize()
    g_dp = model.store.flat_g.float() / w
    # the same exchange with the SUM carried in fp32 (RV_GRAD_REDUCE_DTYPE=fp32): at world size 2 one addition and
    # one rounding either way -> bit-identical to the bf16 sum; from 3 ranks on it is the closer one (tests/test_dist_gloo.py)
    red32 = BucketedAllReduce(model.store.flat_g, bucket_bytes=1 << 20, reduce_dtype="fp32")
    tr.reducer, tr._reduce_hook = red32, red32.on_bucket_ready
    tr.compute_loss(model, dict(shards[rank]))
    model.backward(model.last_out, model.last_coef)
    red32.finish()
    torch.cuda.synchronize()
    same_sum = bool(torch.equal(model.store.flat_g.float() / w, g_dp)) if w == 2 else True
    tr.reducer, tr._reduce_hook = red, red.on_bucket_ready
    union = _cat_batches(shards)
    model.grad_ready_hook = None                          # single-process reference: no exchange
    tr.compute_loss(model, dict(union))
    model.backward(model.last_out, model.last_coef)
    torch.cuda.synchronize()
    g_one = model.store.flat_g.float()
    cos = float((g_dp.double() @ g_one.double()) / (g_dp.double().norm() * g_one.double().norm()))
    nrel = abs(float(g_dp.norm()) - float(g_one.norm())) / float(g_one.norm())
    model.grad_ready_hook = tr._bucket_ready
    # (2) two optimizer steps on different shards: replicas stay bit-identical
    start = model.store.flat_master.clone()
    for step in range(2):
        batch = O.make_synthetic_batch(cfg, 2, 40, 12, seed=100 + 10 * step + rank)       # a different shard per rank
        loss = tr.training_step(dict(batch))
    m = tr.pop_metrics()
    torch.cuda.synchronize()
    mine = model.store.flat_master.clone()
    other = [torch.empty_like(mine) for _ in range(w)]
    dist.all_gather(other, mine)
    same = all(torch.equal(other[0], t) for t in other)
    moved = float((mine != start).float().mean())        # fraction of fp32 masters the two steps moved
    q.put((rank, same and same_sum, float(loss), len(m), moved, cos, nrel))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_rccl_replicas_stay_identical():
    """Two processes, one GPU each, RCCL over xGMI: different data shards, overlapped bucketed all-reduce, and after two
    optimizer steps the fp32 master weights of the two replicas are BIT-identical (the contract of replicated data
    parallelism).  Skips on a 1-GPU box; the driver's multi-GPU tier and bench.py --gpus N exercise the same path."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, loss, n_metrics, moved, cos, nrel in res:
        assert same and loss == loss and n_metrics == 8, (rank, same, loss, n_metrics)
        assert moved > 0.5, moved                                   # the two steps really moved the weights
        assert cos >= 0.999 and nrel <= 1e-2, (cos, nrel)           # 2 ranks x 2 pairs == 1 rank x 4 pairs
        print(f"rank {rank}: all-reduced gradient / world vs single-process gradient on the union: cosine {cos:.6f}, norm rel err {nrel:.2e}")
    import re
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for rank in range(2):
        path = os.path.join(root, "gpurun_out", f"rccl_debug_rank{rank}.log")
        if os.path.exists(path):
            txt = open(path, errors="replace").read()
            ch = re.findall(r"(\d+) coll channels|Channel (\d+)/(\d+)", txt)
            nr = re.findall(r"nranks (\d+)", txt)
            print(f"RCCL rank {rank}: nranks {sorted(set(nr))}, channel lines {len(ch)}: {ch[:4]}")
