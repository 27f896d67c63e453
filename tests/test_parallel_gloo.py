"""N > 1 path on CPU: world_size-2 gloo processes run the shard + single all-gather logic of
clip_cpp_amd.parallel.  The per-shard compute is played by the CPU oracle (test infrastructure standing in
for the HIP encoder, which needs a GPU); what is under test is sharding, padding, ordering and the collective."""
import os
import socket

import numpy as np
import pytest

from oracle import fixtures, ref

torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, B, out_dir):
    import torch
    import torch.distributed as dist
    from clip_cpp_amd import parallel
    from oracle import fixtures, ref
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = ref.OracleModel(path)
    imgs = torch.from_numpy(fixtures.synthetic_images(B, 32, seed=77))

    def encode_shard(x):
        if x.shape[0] == 0:
            return torch.empty((0, 32), dtype=torch.float32)
        return torch.from_numpy(orc.image_batch_encode(x.numpy(), normalize=True, mode=ref.MODE_FAITHFUL, n_threads=1))

    full = parallel.encode_images_data_parallel(encode_shard, imgs)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), full.numpy())
    dist.destroy_process_group()


def test_shard_bounds_cover_batch():
    from clip_cpp_amd.parallel import shard_bounds
    for total in (0, 1, 5, 8, 13, 256, 1024, 1023):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            per = -(-total // world)
            assert all(b - a == per for a, b in spans if b < total) and all(0 <= b - a <= per for a, b in spans)


def test_shard_bounds_is_the_c_abi_rule(clip_lib):
    """one partitioning for both multi-GPU forms: parallel.shard_bounds == clip_amd_shard_bounds"""
    from clip_cpp_amd.parallel import shard_bounds
    for total in (0, 1, 5, 8, 13, 100, 256, 1023, 1024):
        for world in (1, 2, 3, 4, 7, 8):
            for r in range(world):
                assert shard_bounds(total, r, world) == clip_lib.shard_bounds(total, world, r)[:2], (total, world, r)


@pytest.mark.parametrize("B", [5, 8, 1])
def test_data_parallel_encode_world2_gloo(B, tmp_path, fixture_cache):
    import torch.multiprocessing as mp
    path = fixtures.cached_model(fixture_cache, "tiny", "q4_0", text=False, vision=True)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, path, B, str(tmp_path)), nprocs=2, join=True)
    want = ref.OracleModel(path).image_batch_encode(fixtures.synthetic_images(B, 32, seed=77), normalize=True, n_threads=1)
    for r in range(2):
        got = np.load(str(tmp_path / ("r%d.npy" % r)))
        assert got.shape == want.shape
        assert np.array_equal(got, want), r
