"""Data-parallel batch encoding across the GPUs of one node: one process per GPU, contiguous batch
shards, ONE all-gather of the final embeddings (RCCL over xGMI when the backend is "nccl").

The reference has no multi-device path at all; images of a batch never interact
(attention is per image, reference clip.cpp:1366,1382), so the hot path shards with no data-path
collective other than the final gather of [B, proj] f32 rows — a few hundred KiB per rank, latency
bound, one hop on the fully connected xGMI mesh.

The functions are backend-agnostic (`torch.distributed`): "nccl" on GPUs, "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous shard [lo, hi) of `total` items for `rank`: shards of ceil(total / world), trailing shards short or empty — ONE rule for
    both multi-GPU forms: this is clip_amd_shard_bounds (csrc/host_pipeline.cpp::multi_shard, SURVEY 8e `B_g = ceil(B / G)`) restated in
    Python so that the per-process form needs no library call; tests/test_parallel_gloo.py holds the two against each other."""
    per = -(-total // world) if world > 0 else 0
    return min(total, rank * per), min(total, (rank + 1) * per)


def all_gather_rows(local, total_rows, group=None):
    """All-gather row shards of unequal size: pads every shard to the largest one, issues ONE
    all_gather_into_tensor, and returns the [total_rows, D] tensor in global order on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(total_rows, rank, world)
    assert local.shape[0] == hi - lo, "shard has %d rows, expected %d" % (local.shape[0], hi - lo)
    max_rows = -(-total_rows // world)
    if local.shape[0] == max_rows:
        padded = local.contiguous()
    else:
        padded = torch.zeros((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    out = torch.empty((world * max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    if total_rows == world * max_rows:
        return out
    parts = []
    for r in range(world):
        a, b = shard_bounds(total_rows, r, world)
        parts.append(out[r * max_rows: r * max_rows + (b - a)])
    return torch.cat(parts, 0)


def encode_images_data_parallel(encode_shard, images, group=None):
    """images: the FULL batch [B, S, S, 3] (every rank holds it, or at least its own shard's rows are valid).
    encode_shard(images[lo:hi]) -> [hi-lo, proj] tensor on this rank's device.  Returns [B, proj] on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    B = images.shape[0]
    lo, hi = shard_bounds(B, rank, world)
    local = encode_shard(images[lo:hi])
    return all_gather_rows(local, B, group=group)


def clip_shard_encoder(clip, normalize=True):
    """encode_shard callable over a clip_cpp_amd.Clip bound to this rank's GPU: device tensor in, device tensor out."""
    proj = clip.vision_config["projection_dim"]

    def run(shard):
        shard = shard.contiguous()
        out = torch.empty((shard.shape[0], proj), dtype=torch.float32, device=shard.device)
        if shard.shape[0]:
            clip.set_stream(torch.cuda.current_stream().cuda_stream)
            clip.encode_images_device(shard.data_ptr(), shard.shape[0], out.data_ptr(), normalize)
        return out

    return run
