"""Frame sharding across the GPUs of one node + clip assembly.

Frames of a clip are independent given the frame-invariant state: rank r renders
frames {t : t % world == r}; every rank runs the (cheap) per-clip work redundantly; the only
communication is ONE all-gather of the finished frame blocks (RCCL over xGMI; gloo in the CPU
tests).  The reference has no counterpart (it runs one process per scene on one GPU,
test_animating/CLAW/test_all_CLAW_scenes.py:86-96).
"""
import torch
import torch.distributed as dist


def shard_frames(N, rank, world):
    """Frame indices rendered by ``rank`` (round-robin: equal Euler depth mix per rank)."""
    return list(range(rank, N, world))


def frames_per_rank(N, world):
    return (N + world - 1) // world


def gather_clip(local_frames, N, rank, world, group=None):
    """local_frames [len(shard_frames(N,rank,world)),3,H,W] -> [N,3,H,W] on every rank.
    Shards are padded to ceil(N/world) frames so a single all_gather_into_tensor suffices."""
    if world == 1:
        return local_frames
    per = frames_per_rank(N, world)
    C, H, W = local_frames.shape[1:]
    send = local_frames.new_zeros(per, C, H, W)
    send[:local_frames.shape[0]] = local_frames
    recv = local_frames.new_empty(world * per, C, H, W)
    dist.all_gather_into_tensor(recv, send, group=group)
    # recv[r*per + i] is frame r + i*world  ->  clip order
    clip = recv.view(world, per, C, H, W).transpose(0, 1).reshape(per * world, C, H, W)
    return clip[:N]
