"""Device-side hand-off of the map in the backend -> frontend SYNC message.

The reference ships ``splats.no_grad_clone()`` (seven tensor clones, gslam/backend.py:508-519) and the frontend
``deepcopy``-s it again (gslam/frontend.py:253-273): 14 copy kernels and 14 allocations per sync, and the frontend's map
tensors change address every time, which would throw away its captured tracking graph.  Here the message tuple keeps
its shape, but the map inside it is a *view*:

* ``MapMailbox.publish(splats)`` (producer side) copies all per-Gaussian arrays into the free one of two capacity-sized
  device slots with ONE launch (``gsx_concat_rows`` with an empty second part) on the producer's stream, records an
  event and returns a ``GaussianSplattingData`` whose parameters are views of the slot (no grad, event attached);
* ``receive(dst, payload)`` (consumer side) makes the consumer's stream wait for that event (no host sync) and copies,
  again in one launch, into the consumer's own persistent map when the number of Gaussians is unchanged - the addresses
  the tracker's HIP graph has captured stay valid - and only re-allocates when the map was densified or pruned.

A slot is reused two publishes later.  Two guards make that safe when the consumer lags: ``publish`` makes the producer's
stream wait for the event the consumer recorded after its last copy out of that slot (no overwrite under a running copy),
and every payload carries the generation of its slot - ``receive`` drops a payload whose slot has been re-published in the
meantime (``stale``: the newer map is already in the consumer's queue) instead of reading a torn or re-sized map."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from ._lib import check, lib, stream_ptr
from .map import GaussianSplattingData

_PARAMS = GaussianSplattingData._per_splat_params


def _row_words(t: torch.Tensor) -> int:
    per_row = t[0].numel() if t.dim() > 1 else 1
    return per_row * t.element_size() // 4


def copy_rows(dsts: Sequence[torch.Tensor], srcs: Sequence[torch.Tensor], n_rows: int):
    """dst_k[:n_rows] = src_k[:n_rows] for every k in ONE launch (4-byte-granular rows, contiguous tensors)."""
    if n_rows == 0:
        return
    for d, s in zip(dsts, srcs):
        if not (d.is_contiguous() and s.is_contiguous() and d.dtype == s.dtype and d.shape[1:] == s.shape[1:]):
            raise RuntimeError("copy_rows needs contiguous tensors of matching dtype and row shape")
        if d.shape[0] < n_rows or s.shape[0] < n_rows:
            raise RuntimeError("copy_rows: fewer rows than requested")
    m = len(dsts)
    check(lib.gsx_concat_rows(m, (C.c_void_p * m)(*[s.data_ptr() for s in srcs]), n_rows,
                              (C.c_void_p * m)(*([None] * m)), 0, (C.c_void_p * m)(*[d.data_ptr() for d in dsts]),
                              (C.c_int * m)(*[_row_words(d) for d in dsts]), stream_ptr(dsts[0].device)),
          "gsx_concat_rows")


class MapMailbox:
    GROW = 1.25

    def __init__(self):
        self._slots: List[Optional[List[torch.Tensor]]] = [None, None]
        self._generation = [0, 0]
        self._consumed: List[Optional[torch.cuda.Event]] = [None, None]    # recorded by receive() after its copy
        self._next = 0
        self.publishes = 0

    def _ensure(self, i: int, splats: GaussianSplattingData):
        n = int(splats.means.shape[0])
        slot = self._slots[i]
        ok = slot is not None and slot[0].shape[0] >= n and slot[0].device == splats.means.device and all(
            s.shape[1:] == getattr(splats, p).shape[1:] for s, p in zip(slot, _PARAMS))
        if not ok:
            cap = max(int(n * self.GROW), 1)
            self._slots[i] = [torch.empty((cap,) + tuple(getattr(splats, p).shape[1:]), dtype=getattr(splats, p).dtype,
                                          device=splats.means.device) for p in _PARAMS]
        return self._slots[i]

    @torch.no_grad()
    def publish(self, splats: GaussianSplattingData) -> GaussianSplattingData:
        n = int(splats.means.shape[0])
        i = self._next
        self._next ^= 1
        done = self._consumed[i]
        if done is not None:                                  # back-pressure: a consumer copy out of this slot may be in flight
            torch.cuda.current_stream().wait_event(done)
            self._consumed[i] = None
        slot = self._ensure(i, splats)
        self._generation[i] += 1
        copy_rows(slot, [getattr(splats, p).detach().contiguous() for p in _PARAMS], n)
        view = GaussianSplattingData(*[s[:n] for s in slot])
        for p in _PARAMS:
            getattr(view, p).requires_grad_(False)
        ev = torch.cuda.Event()
        ev.record()                                           # on the producer's current stream
        view._mail_event = ev
        view._mail_slot = (self, i, self._generation[i])
        self.publishes += 1
        return view


@torch.no_grad()
def receive(dst: Optional[GaussianSplattingData], payload: GaussianSplattingData):
    """-> (map to use, replaced).  ``replaced`` is False when ``dst`` was updated in place (same N: captured graphs
    over its tensors stay valid); otherwise a new no-grad map was allocated."""
    slot = getattr(payload, "_mail_slot", None)
    if slot is not None and slot[0]._generation[slot[1]] != slot[2]:
        return dst, False                                     # stale: its slot was re-published; the newer payload follows
    ev = getattr(payload, "_mail_event", None)
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)
    n = int(payload.means.shape[0])
    same = dst is not None and int(dst.means.shape[0]) == n and dst.means.device == payload.means.device and all(
        getattr(dst, p).shape == getattr(payload, p).shape and getattr(dst, p).is_contiguous() for p in _PARAMS)
    def consumed():
        if slot is not None:
            done = torch.cuda.Event()
            done.record()
            slot[0]._consumed[slot[1]] = done

    if same:
        copy_rows([getattr(dst, p).data for p in _PARAMS], [getattr(payload, p).data for p in _PARAMS], n)
        consumed()
        return dst, False
    out = GaussianSplattingData(*[torch.empty_like(getattr(payload, p).data) for p in _PARAMS])
    for p in _PARAMS:
        getattr(out, p).requires_grad_(False)
    copy_rows([getattr(out, p).data for p in _PARAMS], [getattr(payload, p).data for p in _PARAMS], n)
    consumed()
    return out, True
