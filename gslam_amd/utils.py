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
    """Early-stop rule of the mapping loop (same interface and same decisions as gslam/utils.py:164-186, pinned by the
    reference-generated trace in tests/golden/utils.npz).  Restated: the first loss only seeds the comparison value and
    never stops.  A loss above ``min_loss`` is ignored altogether - it neither stops, nor counts, nor becomes the
    comparison value.  A loss at or below ``min_loss`` that improves on the comparison value adds one to a streak and
    stops once the streak reaches ``patience``; one that does not improve resets the streak.  Either way it becomes the
    new comparison value (unless it just stopped)."""

    def __init__(self, patience, min_loss):
        self.patience = patience
        self.min_loss = min_loss
        self.counter = 0            # current streak of improving losses below min_loss
        self.last_loss = None       # comparison value

    def stop(self, loss):
        seeded = self.last_loss is not None
        if not seeded:
            self.last_loss = loss
        if not seeded or loss > self.min_loss:
            return False
        improved = loss < self.last_loss
        self.counter = self.counter + 1 if improved else 0
        if improved and self.counter >= self.patience:
            return True
        self.last_loss = loss
        return False
