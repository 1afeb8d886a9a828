"""Drop-in for the reference's ``models/projection/euler_integration_manipulator.py``.

``euler_integration`` / ``EulerIntegration`` keep the reference's signatures, assertions and
return values (bit-exact), but one HIP launch integrates all steps (the reference runs a Python
loop of ~15 torch kernels per step, euler_integration_manipulator.py:36-55).
``euler_integration_all`` is the O(N) all-frames pass the frame-synthesis pipeline uses.
"""
import torch
import torch.nn as nn

from ._lib import check, lib, ptr, require_device, stream_of


def _steps(destination_frame):
    # the models pass a python int or a 1-element index tensor (animating_softmax_splating.py:847)
    if torch.is_tensor(destination_frame):
        return int(destination_frame.reshape(-1)[0].item())
    return int(destination_frame)


def euler_integration(motion, destination_frame, return_all_frames=False):
    """euler_integration_manipulator.py:7-56.

    :param motion: Eulerian motion field [1,2,H,W] (ch0 = x, ch1 = y velocity).
    :param destination_frame: number of integration steps.
    :return: (displacements [1,2,H,W], visible_pixels [1,1,H,W]); invalid pixels carry
             max(H,W)+1 in both channels.
    ``return_all_frames=True`` crashes in the reference (:31,50); here it returns the
    displacement maps of frames 0..destination_frame ([n+1,2,H,W], [n+1,1,H,W]).
    """
    assert (motion.dim() == 4)
    b, c, height, width = motion.shape
    assert (b == 1), 'Function only implemented for batch = 1'
    assert (c == 2), f'Input motion field should be Bx2xHxW. Given tensor is: {motion.shape}'
    n = _steps(destination_frame)
    if return_all_frames:
        return euler_integration_all(motion, n)
    motion = motion.contiguous()
    require_device(motion)
    if torch.is_grad_enabled() and motion.requires_grad:
        # joint training feeds the motion regressor's output through here (animating_softmax_splating.py:515-580):
        # the reference's loop is differentiable w.r.t. `motion`, so is this
        return _EulerIntegrate.apply(motion, n)
    return _integrate(motion, n)


def _integrate(motion, n):
    _, _, height, width = motion.shape
    disp = torch.empty_like(motion)
    vis = motion.new_empty(1, 1, height, width)
    with torch.cuda.device(motion.device):
        check(lib().slr_euler_integrate(ptr(motion), height, width, n, 1.0, ptr(disp), ptr(vis),
                                        stream_of(motion)), "slr_euler_integrate")
    return disp, vis


class _EulerIntegrate(torch.autograd.Function):
    """euler_integration with the gradient torch autograd derives from the reference's loop (:36-55): every step of
    a pixel that stays inside the image passes the pixel's displacement gradient to the motion cell it gathered
    from; pixels that left the image (constant displacement, :55) and `visible_pixels` carry none."""

    @staticmethod
    def forward(ctx, motion, n):
        ctx.save_for_backward(motion)
        ctx.n = n
        disp, vis = _integrate(motion, n)
        ctx.mark_non_differentiable(vis)
        return disp, vis

    @staticmethod
    def backward(ctx, grad_disp, _grad_vis):
        motion, = ctx.saved_tensors
        grad_disp = grad_disp.contiguous()
        require_device(grad_disp)
        _, _, height, width = motion.shape
        grad_motion = torch.empty_like(motion)
        with torch.cuda.device(motion.device):
            check(lib().slr_euler_backward(ptr(motion), height, width, ctx.n, 1.0, ptr(grad_disp), ptr(grad_motion),
                                           stream_of(motion)), "slr_euler_backward")
        return grad_motion, None


def euler_integration_all(motion, nmax, sign=1.0, want_visible=True):
    """Displacement maps to every frame t = 0..nmax in one pass:
    out[t] == euler_integration(sign*motion, t).  -> ([nmax+1,2,H,W], [nmax+1,1,H,W] or None)"""
    assert motion.dim() == 4 and motion.shape[0] == 1 and motion.shape[1] == 2
    if torch.is_grad_enabled() and motion.requires_grad:
        raise RuntimeError("euler_integration_all has no backward (inference pipeline); use euler_integration per "
                           "frame when gradients w.r.t. the motion field are needed")
    motion = motion.contiguous()
    require_device(motion)
    _, _, H, W = motion.shape
    disp = motion.new_empty(nmax + 1, 2, H, W)
    vis = motion.new_empty(nmax + 1, 1, H, W) if want_visible else None
    with torch.cuda.device(motion.device):
        check(lib().slr_euler_integrate_all(ptr(motion), H, W, int(nmax), float(sign), ptr(disp), ptr(vis),
                                            stream_of(motion)), "slr_euler_integrate_all")
    return disp, vis


def _steps_on_device(destination_frame, batch, device):
    """Per-sample step counts as an int64 tensor on `device` without reading anything back: the training step hands over
    `middle_index.long() - start_index.long()` (animating_softmax_splating.py:579-580), already on the device."""
    if torch.is_tensor(destination_frame):
        steps = destination_frame.reshape(-1)
    else:
        steps = torch.as_tensor([int(v) for v in destination_frame], dtype=torch.int64)
    assert steps.numel() >= batch, f'{steps.numel()} step counts for a batch of {batch}'
    return steps[:batch].to(device=device, dtype=torch.int64, non_blocking=True).contiguous()


def _integrate_batch(motion, steps):
    b, _, height, width = motion.shape
    disp = torch.empty_like(motion)
    vis = motion.new_empty(b, 1, height, width)
    with torch.cuda.device(motion.device):
        check(lib().slr_euler_integrate_batch(ptr(motion), ptr(steps), b, height, width, 1.0, ptr(disp), ptr(vis),
                                              stream_of(motion)), "slr_euler_integrate_batch")
    return disp, vis


class _EulerIntegrateBatch(torch.autograd.Function):
    """The batch of _EulerIntegrate: one launch forward, one backward, step counts never leave the device."""

    @staticmethod
    def forward(ctx, motion, steps):
        ctx.save_for_backward(motion, steps)
        disp, vis = _integrate_batch(motion, steps)
        ctx.mark_non_differentiable(vis)
        return disp, vis

    @staticmethod
    def backward(ctx, grad_disp, _grad_vis):
        motion, steps = ctx.saved_tensors
        grad_disp = grad_disp.contiguous()
        require_device(grad_disp)
        b, _, height, width = motion.shape
        grad_motion = torch.empty_like(motion)
        with torch.cuda.device(motion.device):
            check(lib().slr_euler_backward_batch(ptr(motion), ptr(steps), b, height, width, 1.0, ptr(grad_disp),
                                                 ptr(grad_motion), stream_of(motion)), "slr_euler_backward_batch")
        return grad_motion, None


class EulerIntegration(nn.Module):
    """euler_integration_manipulator.py:58-71 (batch wrapper, per-sample step counts).  The reference loops over the
    samples in Python and reads every step count on the host; here the whole batch is ONE launch (`slr_euler_integrate_batch`)
    and `destination_frame` stays where it is -- no `.item()`, no per-sample slice copies."""

    def __init__(self, opt=None):
        super().__init__()
        self.opt = opt

    def forward(self, motion, destination_frame, return_all_frames=False, show_visible_pixels=False):
        assert (motion.dim() == 4)
        assert (motion.shape[1] == 2), f'Input motion field should be Bx2xHxW. Given tensor is: {motion.shape}'
        if motion.shape[0] == 0:
            displacements, visible_pixels = torch.empty_like(motion), motion.new_empty(0, 1, motion.shape[2], motion.shape[3])
        else:
            motion = motion.contiguous()
            require_device(motion)
            steps = _steps_on_device(destination_frame, motion.shape[0], motion.device)
            if torch.is_grad_enabled() and motion.requires_grad:
                displacements, visible_pixels = _EulerIntegrateBatch.apply(motion, steps)
            else:
                displacements, visible_pixels = _integrate_batch(motion, steps)
        if show_visible_pixels:
            return displacements, visible_pixels
        else:
            return displacements
