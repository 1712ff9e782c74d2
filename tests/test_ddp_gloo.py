"""World-size-2 data-parallel plumbing on CPU (gloo): the gradient exchange of HotPathTrainer is ONE all-reduce of a
contiguous arena range per backward, averaged over ranks, and it always walks the current arena (also after the
parameters were re-packed).  No kernels are launched here (the render path itself needs a GPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import contrastive_lift_amd as cl
        from contrastive_lift_amd.trainer import HotPathTrainer, default_config
        torch.manual_seed(0)
        m = cl.TensorVMSplit([6, 7, 8], num_semantic_classes=3, dim_feature_instance=6, use_semantic_mlp=True,
                             use_instance_mlp=True, slow_fast_mode=True, device="cpu")
        r = cl.TensoRFRenderer(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), [6, 7, 8], semantic_weight_mode="softmax")
        tr = HotPathTrainer(m, r, default_config())
        assert tr.world == world
        # identical initial weights on every rank (same seed), rank-dependent gradients
        m.grad_flat.copy_(torch.arange(m.arena.total, dtype=torch.float32) * (rank + 1))
        a, b = tr.main_range
        i0, i1 = tr.inst_range
        before_inst = m.grad_flat[i0:i1].clone()
        tr._allreduce(tr.main_range)
        want = torch.arange(m.arena.total, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        ok = torch.allclose(m.grad_flat[a:b], want[a:b]) and torch.equal(m.grad_flat[i0:i1], before_inst)
        tr._allreduce(tr.inst_range)
        ok = ok and torch.allclose(m.grad_flat[i0:i1], want[i0:i1])
        # main range = grids + nets, contiguous and disjoint from the instance range; slow net is in neither (DINO style)
        s0, s1 = m.arena.range_of("inst_slow")
        ok = ok and (a == 0 and b == i0 and i1 == s0)
        # parameters are views of the arena: the grad views alias grad_flat
        g = m.get_parameter("render_instance_mlp.mlp.0.bias").grad
        sl = m.arena.by_name["render_instance_mlp.mlp.0.bias"]
        ok = ok and g.data_ptr() == m.grad_flat[sl.offset:].data_ptr() and torch.allclose(g, want[sl.offset:sl.offset + sl.numel])
        # re-pack (what upsample/shrink do) and check the exchange follows the new buffers
        m.pack()
        tr.setup_optimizers()
        m.grad_flat.fill_(float(rank + 1))
        tr._allreduce(tr.main_range)
        ok = ok and torch.allclose(m.grad_flat[tr.main_range[0]:tr.main_range[1]], torch.tensor((world + 1) / 2.0))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gradient_exchange_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}


def test_state_dict_roundtrip_and_layout():
    """Arena-backed parameters keep the reference shapes; channels-last tables and pitched matrices round-trip through
    state_dict / export_state_dict."""
    import contrastive_lift_amd as cl
    torch.manual_seed(1)
    m = cl.TensorVMSplit([5, 6, 7], num_semantic_classes=4, dim_feature_instance=6, use_semantic_mlp=True, use_instance_mlp=True,
                         slow_fast_mode=True, device="cpu")
    sd = m.export_state_dict()
    assert sd["density_plane.0"].shape == (1, 16, 6, 5) and sd["density_plane.0"].is_contiguous()
    assert sd["appearance_line.2"].shape == (1, 48, 5, 1)
    assert sd["render_appearance_mlp.mlp.0.weight"].shape == (128, 150) and sd["render_appearance_mlp.mlp.0.weight"].is_contiguous()
    assert sd["render_semantic_mlp.mlp.8.weight"].shape == (4, 256)
    assert sd["render_instance_mlp.slow_mlp.6.weight"].shape == (3, 256)
    assert float(sd["render_appearance_mlp.mlp.4.bias"].abs().max()) == 0.0          # tensoRF.py:398
    m2 = cl.TensorVMSplit([5, 6, 7], num_semantic_classes=4, dim_feature_instance=6, use_semantic_mlp=True, use_instance_mlp=True,
                          slow_fast_mode=True, device="cpu")
    m2.load_state_dict(sd)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k]), k
    p = m2.get_parameter("density_plane.1")
    assert p.stride()[1] == 1 and p.data_ptr() >= m2.param_flat.data_ptr()           # channels-last view into the arena
    w = m2.get_parameter("render_appearance_mlp.mlp.0.weight")
    assert w.stride() == (160, 1)          # 150 inputs padded to the 160-float pitch of the persistent 128-wide layer kernel
    groups = m2.get_optimizable_parameters(1e-2, 5e-4, 1e-8)
    assert len(groups) == 7 and groups[0]["lr"] == 1e-2 and groups[-1]["lr"] == 5e-4
    assert len(m2.get_optimizable_instance_parameters(1e-2, 5e-4, using_DINO=True)) == 1
    assert len(m2.get_optimizable_instance_parameters(1e-2, 5e-4, using_DINO=False)) == 2


def _fake_render(model, renderer, rays, chunk, white_bg):
    """Deterministic per-ray stand-in for the GPU renderer with the real output signature (rgb, semantics, instances, dist)."""
    o, d = rays[:, 0:3], rays[:, 3:6]
    return (o * 2 + d, torch.cat([o, d, o * d], 1)[:, :5], torch.cat([d, o], 1), rays[:, 7] * 3 - rays[:, 6])


def _shard_worker(rank, world, port, q, P):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from contrastive_lift_amd import inference as inf
        g = torch.Generator().manual_seed(9)
        rays = torch.randn((P, 8), generator=g)
        got = inf.render_rays_sharded(None, None, rays, 0, False, render_fn=_fake_render)
        want = _fake_render(None, None, rays, 0, False)
        ok = all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(got, want))
        b = inf.tile_bounds(P, world)
        ok = ok and b[0] == 0 and b[-1] == P and max(b[i + 1] - b[i] for i in range(world)) - min(b[i + 1] - b[i] for i in range(world)) <= 1
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_frame_render_row_tiles_world2():
    """Inference sharding (BASELINE configs[4]): contiguous row-tiles per rank + one all-gather == the unsharded render, for
    even, odd and smaller-than-world ray counts."""
    for P in (64, 1001, 1):
        port = _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=_shard_worker, args=(r, 2, port, q, P)) for r in range(2)]
        for p in ps:
            p.start()
        for p in ps:
            p.join(180)
            assert p.exitcode == 0, P
        res = [q.get(timeout=10) for _ in ps]
        assert all(ok for _, ok in res), (P, res)
