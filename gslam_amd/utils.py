"""Hot-path helpers of gslam/utils.py: create_batch (:17-23), edge_aware_tv (:136-161), StopOnPlateau (:164-186)."""
from __future__ import annotations

from typing import Callable, List, Optional

import torch


def create_batch(things: List, getter: Optional[Callable] = None) -> torch.Tensor:
    if getter is not None:
        things = [getter(thing) for thing in things]
    if len(things) == 1:
        return things[0].unsqueeze(0)           # same values as torch.stack, without the copy kernel
    return torch.stack(things, dim=0)


def edge_aware_tv(depth: torch.Tensor, rgb: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """depth [B,H,W], rgb [B,H,W,3], mask [B,H,W] bool.  SUM (not mean) over masked pixels, as the reference."""
    gdx = (depth[..., :, :-1] - depth[..., :, 1:]).abs()
    gdy = (depth[..., :-1, :] - depth[..., 1:, :]).abs()
    gix = (rgb[..., :, :-1, :] - rgb[..., :, 1:, :]).abs().mean(-1)
    giy = (rgb[..., :-1, :, :] - rgb[..., 1:, :, :]).abs().mean(-1)
    gdx = gdx * torch.exp(-gix)
    gdy = gdy * torch.exp(-giy)
    if mask is None:
        return gdx.sum() + gdy.sum()
    return (gdx * mask[..., :, :-1]).sum() + (gdy * mask[..., :-1, :]).sum()


class StopOnPlateau:
    """Stop optimisation if the loss doesn't decrease appreciably for a bit (gslam/utils.py:164-186, verbatim
    state machine including its quirks: the counter counts DEcreases below min_loss)."""

    def __init__(self, patience, min_loss):
        self.patience = patience
        self.counter = 0
        self.min_loss = min_loss
        self.last_loss = None

    def stop(self, loss):
        if self.last_loss is None:
            self.last_loss = loss
            return False
        if loss > self.min_loss:
            return False
        elif self.last_loss > loss:
            self.counter += 1
            if self.counter >= self.patience:
                return True
        else:
            self.counter = 0
        self.last_loss = loss
        return False
