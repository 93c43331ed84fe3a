"""Island model over torch.distributed (RCCL on MI355X, gloo in the CPU tests).

The population shards across ranks with no data-path collective; the only exchange is one
all-gather of each island's best `num_elites` rows per generation.  A row is
[fitness, v0..v(D-1), s0..s(D-1)] (include/sots_hip.h, island section).  Immigrants
overwrite the tail of the receiving island's PARENT rows, so they take part in the next
recombination (ocl_program.cl:99-112 only ever reads parent blocks).

Two schedules:
  overlap=False  pack -> all-gather -> inject inside the same generation (the collective's
                 latency, ~tens of microseconds, sits on the critical path);
  overlap=True   the all-gather started after generation g is injected after generation g+1
                 has been sorted, so it runs underneath g+1's synthesis/FFT.  Every generation
                 still sends and receives elites; they arrive one generation later.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class IslandExchange:
    def __init__(self, rank: int, world: int, num_elites: int, num_dims: int, device, overlap: bool = False):
        self.rank, self.world, self.E = rank, world, num_elites
        self.width = 2 * num_dims + 1
        self.device = torch.device(device)
        self.overlap = overlap
        nbuf = 2 if overlap else 1
        self.mine = [torch.empty(num_elites, self.width, dtype=torch.float32, device=self.device) for _ in range(nbuf)]
        self.all = [torch.empty(world * num_elites, self.width, dtype=torch.float32, device=self.device) for _ in range(nbuf)]
        self.cur = 0
        self.pending = None  # (work handle, buffer index) of the collective in flight

    @property
    def num_immigrants(self) -> int:
        return (self.world - 1) * self.E

    def _all_gather(self, idx: int, async_op: bool):
        try:
            return dist.all_gather_into_tensor(self.all[idx], self.mine[idx], async_op=async_op)
        except (RuntimeError, NotImplementedError):
            parts = list(self.all[idx].chunk(self.world))
            return dist.all_gather(parts, self.mine[idx], async_op=async_op)

    def _others(self, idx: int) -> torch.Tensor:
        lo, hi = self.rank * self.E, (self.rank + 1) * self.E
        return torch.cat([self.all[idx][:lo], self.all[idx][hi:]])

    def _exchange(self, pack, inject) -> None:
        """pack(buffer index) fills self.mine[idx]; inject(buffer index) consumes self.all[idx]."""
        if self.world == 1:
            return
        if not self.overlap:
            pack(0)
            self._all_gather(0, async_op=False)
            inject(0)
            return
        if self.pending is not None:
            work, idx = self.pending
            work.wait()  # GPU backends: the current stream waits; gloo: the host does
            inject(idx)
            self.pending = None
        idx = self.cur
        pack(idx)
        self.pending = (self._all_gather(idx, async_op=True), idx)
        self.cur ^= 1

    def finish(self) -> None:
        """Drains the collective still in flight (overlap schedule); its rows are dropped."""
        if self.pending is not None:
            self.pending[0].wait()
            self.pending = None

    # ---- device path: rows never leave HBM -------------------------------------------------
    def migrate_device(self, es) -> None:
        """es: HipES whose stream is the current torch stream."""
        self._exchange(lambda i: es.pack_elites_device(self.mine[i].data_ptr(), self.E),
                       lambda i: es.inject_gathered_device(self.all[i].data_ptr(), self.world, self.rank, self.E))

    # ---- host path (gloo tests, oracle islands) ---------------------------------------------
    def migrate_host(self, pack, inject) -> None:
        """pack(n) -> ndarray [n, width]; inject(ndarray [(world-1)*E, width])."""
        self._exchange(lambda i: self.mine[i].copy_(torch.from_numpy(pack(self.E))),
                       lambda i: inject(self._others(i).cpu().numpy()))
