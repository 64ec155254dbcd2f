"""Training-loop pieces of the reference (pytorch/bts_main.py) around the HIP training graph (bts_amd/train.py).

Only what the model-side protocol needs: which encoder layers the reference freezes (``set_misc``,
bts_main.py:160-190), its optimiser (AdamW, two parameter groups, bts_main.py:354-356), its polynomial learning-rate
decay (bts_main.py:420, 602-604), the supervised loss on valid ground-truth pixels (bts_main.py:412, 551-565) and the
one-process-per-GPU wrapper (DistributedDataParallel over RCCL, bts_main.py:295-317).  The data loader, the C3D /
photometric losses (external ``c3d`` package), TensorBoard and checkpoint rotation are out of scope.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


def fixing_layers(encoder_name: str, fix_first_conv_blocks: bool = False, fix_first_conv_block: bool = False):
    """Name fragments of the encoder parameters the reference keeps frozen (bts_main.py:165-182): always the stem
    convolution and every norm layer's affine parameters, optionally the first one or two blocks."""
    resnet = 'resne' in encoder_name
    stem, norm = ('base_model.conv1', '.bn') if resnet else ('conv0', 'norm')
    first = ['base_model.layer1.0', 'base_model.layer1.1'] if resnet else ['denseblock1.denselayer1', 'denseblock1.denselayer2']
    if fix_first_conv_blocks:
        return [stem] + first + [norm]
    if fix_first_conv_block:
        return [stem] + first[:1] + [norm]
    return [stem, norm]


def set_misc(model, encoder_name: str, fix_first_conv_blocks: bool = False, fix_first_conv_block: bool = False,
             bn_no_track_stats: bool = False):
    """bts_main.py:160-190 on a BtsModel (or a DataParallel/DDP wrapper of one): returns the frozen parameter names."""
    from .bts import bn_init_as_tf
    if bn_no_track_stats:
        model.apply(bn_init_as_tf)
    frags = fixing_layers(encoder_name, fix_first_conv_blocks, fix_first_conv_block)
    core = model.module if hasattr(model, "module") else model
    frozen = []
    for name, p in core.encoder.named_parameters():
        if any(f in name for f in frags):
            p.requires_grad = False
            frozen.append(name)
    return frozen


def make_optimizer(model, learning_rate: float = 1e-4, weight_decay: float = 1e-2, adam_eps: float = 1e-3,
                   fused: Optional[bool] = None):
    """AdamW with weight decay on the encoder only (bts_main.py:354-356; defaults of arguments_train_eigen.txt).
    ``fused`` (default: on when every parameter lives on a GPU): torch's single-kernel-per-chunk implementation of the
    same update instead of the multi-tensor one (~8 passes over the 47 M parameters)."""
    core = model.module if hasattr(model, "module") else model
    if fused is None:
        fused = all(p.is_cuda for p in core.parameters())
    opt = torch.optim.AdamW([{'params': core.encoder.parameters(), 'weight_decay': weight_decay},
                             {'params': core.decoder.parameters(), 'weight_decay': 0}],
                            lr=learning_rate, eps=adam_eps, fused=fused)
    # torch's fused optimisers rewrite the parameters WITHOUT bumping Tensor._version, which is what the packed-weight
    # caches, recorded plans and captured graphs fingerprint: age them after every step, whatever mode the model is in
    # (a model kept in eval() and stepped by hand would otherwise go on computing with the previous step's packs).
    # Writers that bypass the optimiser and go through `.data` call bts_amd.workspace.invalidate_packs() themselves.
    from . import workspace as _workspace
    opt.register_step_post_hook(lambda optimizer, args, kwargs: _workspace.invalidate_packs())
    return opt


def poly_lr(global_step: int, num_total_steps: int, learning_rate: float, end_learning_rate: float = -1.0) -> float:
    """bts_main.py:420, 603: (lr - end) * (1 - step/total)^0.9 + end, end = 0.1*lr unless given."""
    end = end_learning_rate if end_learning_rate != -1 else 0.1 * learning_rate
    return (learning_rate - end) * (1 - global_step / num_total_steps) ** 0.9 + end


def gt_mask(depth_gt: torch.Tensor, dataset: str) -> torch.Tensor:
    """Valid ground-truth pixels: depth > 1.0 for KITTI, > 0.1 for NYU (bts_main.py:551-553)."""
    return depth_gt > (1.0 if dataset == 'kitti' else 0.1)


def train_step(model, optimizer, criterion, image, focal, depth_gt, mask: Optional[torch.Tensor] = None,
               lr: Optional[float] = None, dataset: str = 'kitti'):
    """One iteration of bts_main.py:465-606 without the external losses: zero_grad, forward, silog on valid pixels,
    backward, learning-rate update, optimiser step.  Returns (loss, the model's 6 outputs)."""
    optimizer.zero_grad(set_to_none=True)
    outs = model(image, focal)
    if mask is None:
        mask = gt_mask(depth_gt, dataset)
    loss = criterion(outs[4], depth_gt, mask.to(torch.bool))
    loss.backward()
    if lr is not None:
        for group in optimizer.param_groups:
            group['lr'] = lr
    optimizer.step()
    return loss, outs


def wrap_ddp(model, device: torch.device):
    """One process per GPU over RCCL (backend 'nccl' on ROCm), as bts_main.py:295-317 does with
    find_unused_parameters=True (ResNet encoders carry an unused ``fc``)."""
    if not dist.is_initialized():
        raise RuntimeError("wrap_ddp: torch.distributed is not initialised (launch with torch.distributed.run)")
    return torch.nn.parallel.DistributedDataParallel(model.to(device), device_ids=[device.index],
                                                     find_unused_parameters=True)
