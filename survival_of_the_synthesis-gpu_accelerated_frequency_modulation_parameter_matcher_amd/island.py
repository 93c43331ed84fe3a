"""Island model over torch.distributed (RCCL on MI355X, gloo in the CPU tests).

The population shards across ranks with no data-path collective; the only exchange is one
all-gather of each island's best `num_elites` rows per migration.  A row is
[fitness, v0..v(D-1), s0..s(D-1)] (include/sots_hip.h, island section).  Immigrants
overwrite the tail of the receiving island's PARENT rows, so they take part in the next
recombination (ocl_program.cl:99-112 only ever reads parent blocks).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class IslandExchange:
    def __init__(self, rank: int, world: int, num_elites: int, num_dims: int, device):
        self.rank, self.world, self.E = rank, world, num_elites
        self.width = 2 * num_dims + 1
        self.device = torch.device(device)
        self.mine = torch.empty(num_elites, self.width, dtype=torch.float32, device=self.device)
        self.all = torch.empty(world * num_elites, self.width, dtype=torch.float32, device=self.device)
        self.immigrants = torch.empty((world - 1) * num_elites, self.width, dtype=torch.float32,
                                      device=self.device)

    @property
    def num_immigrants(self) -> int:
        return (self.world - 1) * self.E

    def all_gather(self) -> None:
        """all-gather self.mine into self.all (rank order)."""
        try:
            dist.all_gather_into_tensor(self.all, self.mine)
        except (RuntimeError, NotImplementedError):
            parts = list(self.all.chunk(self.world))
            dist.all_gather(parts, self.mine)

    def gather(self) -> torch.Tensor:
        """all-gather self.mine; returns the other islands' rows in rank order."""
        if self.world == 1:
            return self.immigrants
        self.all_gather()
        lo, hi = self.rank * self.E, (self.rank + 1) * self.E
        self.immigrants[:lo].copy_(self.all[:lo])
        self.immigrants[lo:].copy_(self.all[hi:])
        return self.immigrants

    # ---- device path: rows never leave HBM -------------------------------------------------
    def migrate_device(self, es) -> None:
        """es: HipES whose stream is the current torch stream."""
        if self.world == 1:
            return
        es.pack_elites_device(self.mine.data_ptr(), self.E)
        self.all_gather()
        es.inject_gathered_device(self.all.data_ptr(), self.world, self.rank, self.E)

    # ---- host path (gloo tests, oracle islands) ---------------------------------------------
    def migrate_host(self, pack, inject) -> None:
        """pack(n) -> ndarray [n, width]; inject(ndarray [(world-1)*E, width])."""
        if self.world == 1:
            return
        self.mine.copy_(torch.from_numpy(pack(self.E)))
        imm = self.gather()
        inject(imm.cpu().numpy())
