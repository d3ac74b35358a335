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
    ok_metric = torch.allclose(m, torch.tensor([(world - 1) / 2, 1.0, -float(world - 1)]))
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
        list(range(5 * world)), (lambda x: x)
    tr.state = dict(global_step=0, epoch=0, batches_in_epoch=0)
    idx = [b[0] for b in tr.get_train_dataloader()]
    gathered = [None] * world
    dist.all_gather_object(gathered, idx)
    ok_shard = sorted(sum(gathered, [])) == list(range(5 * world)) and all(len(g_) == 5 for g_ in gathered)   # every sample once, equal shares
    # a set that does not divide: the same number of batches on every rank, no sample twice
    tr.train_dataset = list(range(5 * world + world - 1))
    idx = [b[0] for b in tr.get_train_dataloader()]
    dist.all_gather_object(gathered, idx)
    flat_idx = sum(gathered, [])
    ok_shard = ok_shard and len(set(flat_idx)) == len(flat_idx) == 5 * world and all(len(g_) == 5 for g_ in gathered)
    q.put((rank, ok_sum, ok_metric, ok_again, ok_shard, len(launched)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("lora,world", [(False, 2), (True, 2), (False, 8), (True, 8)])
def test_bucketed_allreduce_gloo_world2(lora, world):
    """world 2, and world 8 = the target node (full fine-tune and LoRA bucket schedules, the fused metric
    reduce and the rank-strided sampler at the world size the machine has)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, lora)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(world)]
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


# ---------------------------------------------------------------------------------------------- opt-in ZeRO-1 (sharded optimizer)
class _CpuOptKernels:
    """torch stand-ins for the optimizer's HIP kernels (same signatures as dist._DeviceOptKernels): the partition / exchange logic
    is what runs under gloo; both the replicated and the sharded path use THESE, so equality of the two is a statement about it."""

    @staticmethod
    def sumsq(g, out1, accumulate):
        s = g.double().pow(2).sum().float()
        out1[0] = out1[0] + s if accumulate else s

    @staticmethod
    def clip(sumsq, max_norm, out2, pre_scale):
        nrm = sumsq[0].sqrt() * pre_scale
        out2[0] = nrm
        out2[1] = pre_scale * (torch.clamp(max_norm / (nrm + 1e-6), max=1.0) if max_norm > 0 else 1.0)

    @staticmethod
    def adamw(p, master, m, v, g, lr, b1, b2, eps, wd, step, clip):
        gg = g.float() * clip[1]
        if wd:
            master.mul_(1.0 - lr * wd)
        m.mul_(b1).add_(gg, alpha=1 - b1)
        v.mul_(b2).addcmul_(gg, gg, value=1 - b2)
        master.addcdiv_(m / (1 - b1 ** step), (v / (1 - b2 ** step)).sqrt() + eps, value=-lr)
        p.copy_(master)


def _worker_zero1(rank, world, port, q, lora, grad_scale):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import dpo_oracle as O
    from rlaif_v_amd.dist import BucketedAllReduce, ShardedAdamW, ShardedGradReducer, init_process_group_from_env
    from rlaif_v_amd.model import LlavaConfig, LoraConfig, ParamStore
    init_process_group_from_env("gloo")
    BF = torch.bfloat16
    K = _CpuOptKernels

    def fresh():
        st = ParamStore(LlavaConfig(**O.asdict(O.tiny_cfg())), "cpu", lora=LoraConfig(r=16) if lora else None)
        st.flat_p.copy_((torch.randn(st.n_total, generator=torch.Generator().manual_seed(5)) * 0.05).to(BF))     # identical replicas
        st.sync_master_from_params()
        return st

    def local_grad(st, step):
        g = torch.Generator().manual_seed(1000 * step + rank)
        return (torch.randn(st.n_train, generator=g) * grad_scale).to(BF)

    hp = dict(lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=0.01)
    # ---- replicated reference (the default path's arithmetic with the CPU stand-ins)
    a = fresh()
    red = BucketedAllReduce(a.flat_g, bucket_bytes=1 << 18)
    clip_a = torch.zeros(2)
    for step in (1, 2):
        a.flat_g.copy_(local_grad(a, step))
        for name, s0, s1 in a.bucket_schedule():
            red.on_bucket_ready(name, s0, s1)
        red.finish()
        ss = torch.zeros(1)
        K.sumsq(a.flat_g, ss, False)
        K.clip(ss, 1.0, clip_a, 1.0 / world)
        nd = a.n_decay
        K.adamw(a.train_p[:nd], a.flat_master[:nd], a.flat_m[:nd], a.flat_v[:nd], a.flat_g[:nd], hp["lr"], hp["b1"], hp["b2"], hp["eps"],
                hp["wd"], step, clip_a)
        K.adamw(a.train_p[nd:], a.flat_master[nd:], a.flat_m[nd:], a.flat_v[nd:], a.flat_g[nd:], hp["lr"], hp["b1"], hp["b2"], hp["eps"],
                0.0, step, clip_a)
    # ---- ZeRO-1
    b = fresh()
    sred = ShardedGradReducer(b.flat_g, bucket_bytes=1 << 18)
    opt = ShardedAdamW(b.train_p, b.n_decay, sred, b.bucket_schedule(), kernels=K, full_state=(b.flat_master, b.flat_m, b.flat_v))
    clip_b = torch.zeros(2)
    for step in (1, 2):
        b.flat_g.copy_(local_grad(b, step))
        for name, s0, s1 in b.bucket_schedule():
            sred.on_bucket_ready(name, s0, s1)
        launched = sred.finish()
        opt.step(hp["lr"], hp["b1"], hp["b2"], hp["eps"], hp["wd"], step, 1.0, clip_b)
    n_shard, n_rem = opt.master.numel(), opt.rem_master.numel()
    covers = launched[0][0] == 0 and launched[-1][1] == b.n_train and world * n_shard + n_rem == b.n_train
    master, m_, v_ = opt.gather_full_state(all_ranks=True)
    same_p = bool(torch.equal(a.train_p, b.train_p))
    same_state = bool(torch.equal(master, a.flat_master) and torch.equal(m_, a.flat_m) and torch.equal(v_, a.flat_v))
    close_p = float((a.train_p.float() - b.train_p.float()).abs().max())
    # a checkpoint in the replicated format restores the shards (round trip through load_full_state)
    opt2 = ShardedAdamW(b.train_p, b.n_decay, sred, b.bucket_schedule(), kernels=K, full_state=(master, m_, v_))
    rt = bool(torch.equal(opt2.master, opt.master) and torch.equal(opt2.rem_v, opt.rem_v))
    q.put((rank, covers, same_p, same_state, close_p, float(clip_a[1]), float(clip_b[1]), rt, n_shard, n_rem, len(launched)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("lora,world,grad_scale", [(False, 2, 1e-3), (True, 2, 1e-3), (False, 2, 1.0), (False, 3, 1e-3),
                                                   (False, 8, 1e-3), (True, 8, 1e-3)])
def test_zero1_sharded_optimizer_equals_replicated(lora, world, grad_scale):
    """Opt-in ZeRO-1 (dist.ShardedGradReducer + ShardedAdamW; VERDICT r4 next 6b, script/zero2.json:16-22): reduce-scatter of the
    gradient ranges, clip + AdamW on 1 / W of the parameters, in-place all-gather of the updated bf16 parameters - against the
    replicated all-reduce path on the same per-rank gradients, two optimizer steps.  World 2 with the clip factor inactive
    (gradient norm < max_grad_norm): parameters AND the gathered fp32 master / m / v are BIT-IDENTICAL to the replicated run's
    (elementwise arithmetic on identical values; the bf16 SUM of two ranks is one rounding either way).  With clipping active the
    global norm is summed in another order (per-rank partial sums): the clip factors agree to fp32 rounding and the parameters to
    a bf16 ulp.  World 3: the bf16 ring sum and the reduce-scatter may round differently (3 addends) - asserted close."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_zero1, args=(r, world, port, q, lora, grad_scale)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, covers, same_p, same_state, close_p, ca, cb, rt, n_shard, n_rem, n_launch in res:
        assert covers and rt and n_launch >= 2 and n_shard > 0, (rank, covers, rt, n_launch, n_shard, n_rem)
        assert abs(ca - cb) <= 2e-6 * abs(ca), (ca, cb)
        if world == 2 and grad_scale < 1.0:
            assert abs(ca - 1.0 / world) < 1e-9                      # clipping inactive: the factor is exactly 1 / world
            assert same_p and same_state, (rank, same_p, same_state, close_p)
        else:
            assert close_p <= 2e-3, close_p


def _worker_chunks(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from rlaif_v_amd.dist import ShardedAdamW, ShardedGradReducer, init_process_group_from_env
    init_process_group_from_env("gloo")
    K = _CpuOptKernels
    # ranges launched one by one (1-element buckets): widths below 8 W (no chunk at all: c = 0, everything is remainder), exactly
    # 8 W (one 8-element chunk per rank, no remainder), 8 W + 8 W - 8 (the largest remainder) and a long one; an EMPTY range in
    # the schedule (plan() must skip it like _launch() does); the weight-decay boundary inside a chunk and inside a remainder
    W8 = 8 * world
    widths = [40, W8, 2 * W8 - 8, 0, 24, 5 * W8 + 16, 8]
    sched, a = [], 0
    for i, w in enumerate(widths):
        sched.append((f"r{i}", a, a + w))
        a += w
    n = a
    n_decay = 40 + W8 + W8 + 8                     # inside the third range: in rank 1's chunk for W = 8 ... and below, inside a remainder
    out = []
    for nd in (n_decay, 40 + W8 + 2 * W8 - 4, n):
        p0 = (torch.randn(n, generator=torch.Generator().manual_seed(3)) * 0.05).to(torch.bfloat16)
        # small integers x 2^-12: every partial sum of W <= 8 of them is exact in bf16, so the ring all-reduce and the reduce-scatter
        # agree bit for bit whatever their summation orders (random bf16 values differ from 3 ranks on, see the test above)
        grads = [(torch.randint(-8, 9, (n,), generator=torch.Generator().manual_seed(50 + r)).float() * 2.0 ** -12).to(torch.bfloat16)
                 for r in range(world)]
        # replicated reference: the sum every rank would hold after an all-reduce, one AdamW step over everything
        g_sum = grads[0].clone()
        dist.all_reduce(g_sum := grads[rank].clone(), op=dist.ReduceOp.SUM)
        pa = p0.clone()
        ma, m_, v_ = pa.float(), torch.zeros(n), torch.zeros(n)
        ss, clip = torch.zeros(1), torch.zeros(2)
        K.sumsq(g_sum, ss, False)
        K.clip(ss, 1.0, clip, 1.0 / world)
        K.adamw(pa[:nd], ma[:nd], m_[:nd], v_[:nd], g_sum[:nd], 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, clip)
        K.adamw(pa[nd:], ma[nd:], m_[nd:], v_[nd:], g_sum[nd:], 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, clip)
        # sharded
        pb = p0.clone()
        flat = grads[rank].clone()
        red = ShardedGradReducer(flat, bucket_bytes=1)
        opt = ShardedAdamW(pb, nd, red, sched, kernels=K)
        for name, x, y in sched:
            red.on_bucket_ready(name, x, y)
        launched = red.finish()
        clip_b = torch.zeros(2)
        opt.step(1e-3, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, clip_b)
        cs = [c for _, _, c, _ in opt.ranges]
        covered = sum(world * c for c in cs) + sum(y - x for x, y in opt.rem_spans) == n and all(b > a_ for a_, b, _, _ in opt.ranges)
        full = opt.gather_full_state(all_ranks=True)
        out.append((bool(torch.equal(pa, pb)), bool(torch.equal(full[0], ma) and torch.equal(full[2], v_)), covered, cs, len(launched),
                    opt.gather_full_state() is None))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 8])
def test_zero1_chunking_rule_small_ranges(world):
    """The chunking rule c = floor((b - a) / 8 W) * 8 at the world size of the target node (exercised at W = 2, 3
    only before), on ranges NARROWER than 8 W, exactly 8 W wide and with the largest possible remainder, with an empty range in
    the schedule and the weight-decay boundary inside a chunk, inside a remainder and at the end: parameters and gathered state
    bit-identical to the replicated step (clipping inactive); gather_full_state() hands host tensors to rank 0 only."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_chunks, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        for same_p, same_state, covered, cs, n_launch, none_elsewhere in out:
            assert same_p and same_state and covered, (rank, same_p, same_state, covered, cs)
            W8 = 8 * world
            widths = [40, W8, 2 * W8 - 8, 24, 5 * W8 + 16, 8]                            # the schedule's non-empty ranges
            assert cs == [(w // W8) * 8 for w in widths] and n_launch == 6, (cs, n_launch)    # the empty range is neither planned nor launched
            if world == 8:
                assert cs == [0, 8, 8, 0, 40, 0]
            assert none_elsewhere == (rank != 0)
