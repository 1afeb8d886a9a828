"""Shared by tools/ovl_debug*.py: the side-stream feature generator that exposes the open concurrency issue."""
import torch
_side = None


def features_ahead_overlap(clip, frames):
    """The removed option of slr_sfs_amd.pipeline (round 1): frame i+1's features on a side HIP stream while the
    caller's stream runs frame i's decoder; tensors that cross streams are registered with the caching allocator."""
    global _side
    frames = list(frames)
    main = torch.cuda.current_stream()
    if _side is None:
        _side = torch.cuda.Stream()
    side = _side
    side.wait_stream(main)

    def launch(t):
        with torch.cuda.stream(side):
            out = clip.features(t)
            ev = torch.cuda.Event()
            ev.record(side)
        return out, ev

    nxt = launch(frames[0])
    for i in range(len(frames)):
        out, ev = nxt
        if i + 1 < len(frames):
            nxt = launch(frames[i + 1])
        main.wait_event(ev)
        for x in (out if isinstance(out, tuple) else (out,)):
            x.record_stream(main)
        yield out
    side.wait_stream(main)
