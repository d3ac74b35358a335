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
