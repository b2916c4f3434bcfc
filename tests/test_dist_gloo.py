"""The N > 1 path on CPU: two gloo ranks shard the frame by strips, each renders its strips
(with the CPU oracle standing in for the GPU), and the gathered frame must equal the
single-process frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import raytracing_amd as ra
from raytracing_amd import tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, height, width, tile_rows, out_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, cam = ra.teaching_scene(), ra.teaching_camera(width, height)
    prm = ra.Params(spp=1, max_depth=3, tile_index=rank, tile_count=world, tile_rows=tile_rows)
    rgba, hit, rays = oracle.render(scene, cam, prm.c)
    rows = tiles.owned_rows(height, rank, world, tile_rows)
    fg = tiles.FrameGather(height, width, 4, world, rank, tile_rows, torch.device("cpu"))
    assert fg.n_local == len(rows)
    fg.local[: len(rows)].copy_(torch.from_numpy(rgba[rows]))
    total = torch.tensor([float(rays[1:].sum())], dtype=torch.float64)
    dist.all_reduce(total)
    final = fg.gather()
    # the pipelined form must give the same frame
    h = fg.gather_async()
    final2 = fg.finish(h)
    if rank == 0:
        assert torch.equal(final, final2)
    if rank == 0:
        np.save(out_path, final.numpy())
        np.save(out_path + ".rays.npy", total.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("height,width,tile_rows,world", [(48, 40, 8, 2), (50, 33, 8, 2), (20, 16, 4, 3)])
def test_strip_sharding_and_gather(tmp_path, oracle_mod, height, width, tile_rows, world):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), height, width, tile_rows, out), nprocs=world, join=True)
    got = np.load(out)
    scene, cam = ra.teaching_scene(), ra.teaching_camera(width, height)
    ref, _, rays = oracle_mod.render(scene, cam, ra.Params(spp=1, max_depth=3).c)
    assert np.array_equal(got, ref, equal_nan=True)
    assert float(np.load(out + ".rays.npy")[0]) == float(rays[1:].sum())


def test_owned_rows_partition():
    for h, n, tr in [(1080, 1, 8), (1520, 2, 8), (2176, 4, 8), (3072, 8, 8), (50, 3, 8), (7, 4, 2)]:
        seen = sorted(r for k in range(n) for r in tiles.owned_rows(h, k, n, tr))
        assert seen == list(range(h))
        assert tiles.max_owned_rows(h, n, tr) >= (h + n - 1) // n


def _bcast_worker(rank, world, port, which, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    make = {"teaching": ra.teaching_scene, "mesh": lambda: ra.tessellated_mesh_scene(quads=6),
            "textured": lambda: ra.textured_quad_scene(1, 3.0)}[which]
    scene = make() if rank == 0 else ra.Scene()
    scene = tiles.broadcast_scene(scene, torch.device("cpu"))
    d = scene.desc()  # must be a valid description on every rank
    assert d.n_materials == len(scene.materials)
    if rank == 1:
        import pickle
        with open(out_path, "wb") as f:
            pickle.dump({k: getattr(scene, k) for k in tiles._SCENE_ARRAYS}
                        | {"shading": bytes(scene.shading), "mat_texture": list(scene.mat_texture),
                           "textures": scene.textures}, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("which", ["teaching", "mesh", "textured"])
def test_scene_broadcast_is_byte_identical(tmp_path, which):
    """The scene every other rank uploads is the one rank 0 marshalled, array by array."""
    import pickle
    out = str(tmp_path / "scene.pkl")
    mp.spawn(_bcast_worker, args=(2, _free_port(), which, out), nprocs=2, join=True)
    got = pickle.load(open(out, "rb"))
    ref = {"teaching": ra.teaching_scene, "mesh": lambda: ra.tessellated_mesh_scene(quads=6),
           "textured": lambda: ra.textured_quad_scene(1, 3.0)}[which]()
    for k in tiles._SCENE_ARRAYS:
        a, b = getattr(ref, k), got[k]
        assert (a is None) == (b is None), k
        if a is not None:
            assert np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes(), k
            assert np.asarray(a).shape == np.asarray(b).shape, k
    assert bytes(ref.shading) == got["shading"]
    want_mt = list(ref.mat_texture) + [-1] * (len(ref.materials) - len(ref.mat_texture))
    assert got["mat_texture"] == want_mt or (not ref.textures and set(got["mat_texture"]) <= {-1})
    assert len(got["textures"]) == len(ref.textures)
    for (ia, wa), (ib, wb) in zip(ref.textures, got["textures"]):
        assert wa == wb and np.array_equal(ia, ib)
