"""world_size-2 data-parallel test on CPU (gloo): the bucketed gradient all-reduce driven by the model's
own bucket schedule, the fused metric reduce, and the rank-strided sampler."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, lora=False):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import dpo_oracle as O
    from rlaif_v_amd.dist import BucketedAllReduce, init_process_group_from_env
    from rlaif_v_amd.model import LlavaConfig, LoraConfig, ParamStore
    from rlaif_v_amd.trainer import LLaVA15DPOTrainer
    r, _, w = init_process_group_from_env("gloo")
    assert (r, w) == (rank, world)
    # full fine-tune: the whole model is reduced; LoRA: only adapters + projector (frozen base has no gradient slot)
    st = ParamStore(LlavaConfig(**O.asdict(O.tiny_cfg())), "cpu", lora=LoraConfig(r=16) if lora else None)
    assert (st.n_train < st.n_total) == lora
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(st.n_train, generator=g)
    st.flat_g = local.clone()          # fp32 on CPU (gloo); the GPU path reduces the bf16 buffer with RCCL
    red = BucketedAllReduce(st.flat_g, bucket_bytes=1 << 20)      # 1 MiB buckets -> several merges
    for name, a, b in st.bucket_schedule():
        red.on_bucket_ready(name, a, b)
    launched = red.finish()
    # every element reduced exactly once, collectives are contiguous and in schedule order
    assert launched[0][0] == 0 and launched[-1][1] == st.n_train
    assert all(launched[i][1] == launched[i + 1][0] for i in range(len(launched) - 1))
    others = [torch.randn(st.n_train, generator=torch.Generator().manual_seed(100 + k)) for k in range(world)]
    expect = sum(others)
    ok_sum = torch.allclose(st.flat_g, expect, rtol=1e-6, atol=1e-6)
    m = red.reduce_metrics(torch.tensor([float(rank), 1.0, -2.0 * rank]))
    ok_metric = torch.allclose(m, torch.tensor([0.5, 1.0, -1.0]))
    # second step reuses the reducer (state fully reset by finish)
    st.flat_g.copy_(local)
    for name, a, b in st.bucket_schedule():
        red.on_bucket_ready(name, a, b)
    red.finish()
    ok_again = torch.allclose(st.flat_g, expect, rtol=1e-6, atol=1e-6)
    # RV_ALLREDUCE_MODE: "serial" gives the same sums through the same buckets, "skip" issues nothing (measurement only)
    st.flat_g.copy_(local)
    red_s = BucketedAllReduce(st.flat_g, bucket_bytes=1 << 20, mode="serial", timeline=True)    # timeline is a no-op on CPU tensors
    for name, a, b in st.bucket_schedule():
        red_s.on_bucket_ready(name, a, b)
    ok_again = ok_again and red_s.finish() == launched and torch.allclose(st.flat_g, expect, rtol=1e-6, atol=1e-6) \
        and red_s.collect_timeline() is None
    st.flat_g.copy_(local)
    red_k = BucketedAllReduce(st.flat_g, bucket_bytes=1 << 20, mode="skip")
    for name, a, b in st.bucket_schedule():
        red_k.on_bucket_ready(name, a, b)
    ok_again = ok_again and red_k.finish() == launched and torch.equal(st.flat_g, local)
    # rank-strided shards of one permutation are disjoint and cover the dataset

    class T:
        pass
    tr = LLaVA15DPOTrainer.__new__(LLaVA15DPOTrainer)
    from rlaif_v_amd.trainer import TrainingArguments
    tr.args, tr.reducer, tr.train_dataset, tr.data_collator = TrainingArguments(per_device_train_batch_size=1), red, \
        list(range(10)), (lambda x: x)
    tr.state = dict(global_step=0, epoch=0, batches_in_epoch=0)
    idx = [b[0] for b in tr.get_train_dataloader()]
    gathered = [None] * world
    dist.all_gather_object(gathered, idx)
    ok_shard = sorted(sum(gathered, [])) == list(range(10))
    q.put((rank, ok_sum, ok_metric, ok_again, ok_shard, len(launched)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("lora", [False, True])
def test_bucketed_allreduce_gloo_world2(lora):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, lora)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_sum, ok_metric, ok_again, ok_shard, n in res:
        assert ok_sum and ok_metric and ok_again and ok_shard, (rank, ok_sum, ok_metric, ok_again, ok_shard)
        assert n >= 2


def _worker_sum(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from rlaif_v_amd.dist import BucketedAllReduce, init_process_group_from_env
    init_process_group_from_env("gloo")
    n = 1 << 16
    locals_ = [(torch.randn(n, generator=torch.Generator().manual_seed(7 + k)) * (1.0 + k)).to(torch.bfloat16) for k in range(world)]
    exact = sum(t.double() for t in locals_)
    out = {}
    for rd in ("bf16", "fp32"):
        for mode in ("overlap", "serial"):
            flat = locals_[rank].clone()
            red = BucketedAllReduce(flat, bucket_bytes=1 << 14, mode=mode, reduce_dtype=rd)
            for a in range(0, n, 5000):
                red.on_bucket_ready("x", a, min(a + 5000, n))
            red.finish()
            assert flat.dtype == torch.bfloat16 and not red._staged
            out[(rd, mode)] = flat
    ok_modes = torch.equal(out[("bf16", "overlap")], out[("bf16", "serial")]) and torch.equal(out[("fp32", "overlap")], out[("fp32", "serial")])
    once = exact.to(torch.bfloat16)                     # the exact sum rounded ONCE
    err = {rd: float((out[(rd, "overlap")].double() - exact).abs().mean()) for rd in ("bf16", "fp32")}
    q.put((rank, ok_modes, bool(torch.equal(out[("fp32", "overlap")], once)), err["bf16"], err["fp32"],
           int((out[("bf16", "overlap")] != once).sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_fp32_gradient_sum_world3(world):
    """RV_GRAD_REDUCE_DTYPE=fp32: bf16 gradient buckets widened to fp32 for the cross-rank SUM and rounded once.
    World 2: one addition, one rounding either way - both sums are the exact sum rounded once, bit for bit.  World 3: the fp32 sum
    still is; the bf16 ring sum (two roundings) is measurably further from the float64 sum."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sum, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_modes, fp32_is_exact_once, e_bf16, e_fp32, n_diff in res:
        assert ok_modes and fp32_is_exact_once, (rank, ok_modes, fp32_is_exact_once)
        if world == 2:
            assert n_diff == 0
        else:
            assert n_diff > 0 and e_fp32 < e_bf16, (n_diff, e_bf16, e_fp32)
