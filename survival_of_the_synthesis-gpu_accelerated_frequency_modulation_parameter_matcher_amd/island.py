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
    """Elite exchange of one island (rank) over torch.distributed.  The object carries the exchange in flight (overlapped
    schedule) and an exchange count; `restart()` drops both and belongs wherever the population is re-initialised.
    `generation()` and `migrate_device()` also restart by themselves when they see a context whose generation count has
    gone back to zero (es.init_population) while rows of the old population are still in flight."""
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
        # generation(): the overlapped all-gather runs on a side stream; the HOST waits for it (arrived event) right before it
        # enqueues the sort that takes its rows, so the island's stream never waits for an event of another stream (on this
        # runtime that costs the waiting stream ~18 us per generation, tools/ubench/cross_stream.hip)
        self.exchanges = 0
        self.side = None
        if self.device.type == "cuda" and world > 1:
            self.side = torch.cuda.Stream(device=self.device)
            self.packed = [torch.cuda.Event(), torch.cuda.Event()]
            self.arrived = [torch.cuda.Event(), torch.cuda.Event()]

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
        if self.side is not None:
            self.side.synchronize()

    def restart(self) -> None:
        """A new population (init_population): no rows in flight, exchange counting starts again."""
        self.finish()
        self.exchanges = 0
        self.cur = 0

    # ---- device path: rows never leave HBM -------------------------------------------------
    def generation(self, es) -> None:
        """One generation of `es` (a HipES whose stream is the current torch stream) and its exchange, with pack and
        inject folded into the generation's sort kernel (sots_fuse_exchange_next_sort) instead of a launch each:
        the same rows at the same places as execute_generations(1) + migrate_device(es)."""
        if self.world == 1:
            es.execute_generations(1)
            return
        if es.generation == 0 and (self.exchanges > 0 or self.pending is not None):
            # the population was re-initialised under us (init_population resets the generation count): rows gathered
            # from the OLD population must not reach the new one.  Every rank initialises at the same point of the
            # program, so every rank restarts here together.
            self.restart()
        if not es.sort_places(self.E):  # elites beyond the rows the sort places: the separate launches
            es.execute_generations(1)
            self.migrate_device(es)
            return
        if not self.overlap:
            es.fuse_exchange_next_sort(self.mine[0].data_ptr(), self.E, None, self.world, self.rank, self.E)
            es.execute_generations(1)
            self._all_gather(0, async_op=False)
            es.inject_gathered_device(self.all[0].data_ptr(), self.world, self.rank, self.E)
            return
        # overlapped: exchange x (buffers x & 1) takes the rows gathered at exchange x - 1.  The host waits for that
        # collective inside execute_generations, right before the sort is enqueued (by then this generation's variation,
        # synthesis and spectral kernels are on the stream); having seen exchange x - 1 complete, which ran behind exchange
        # x - 2 on the side stream, also frees mine[x & 1] (filled at exchange x - 2) for this sort to overwrite
        x, idx = self.exchanges, self.exchanges & 1
        if x > 0:
            es.fuse_exchange_next_sort(self.mine[idx].data_ptr(), self.E, self.all[idx ^ 1].data_ptr(), self.world, self.rank, self.E,
                                       self.arrived[idx ^ 1].cuda_event)
        else:
            es.fuse_exchange_next_sort(self.mine[idx].data_ptr(), self.E, None, self.world, self.rank, self.E)
        es.execute_generations(1)
        self.packed[idx].record()  # on the island's stream: mine[idx] is complete
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.packed[idx])
            self._all_gather(idx, async_op=False)  # the SIDE stream waits for the collective; the host does not (gloo: it does)
            self.arrived[idx].record()
        self.exchanges += 1

    def migrate_device(self, es) -> None:
        """es: HipES whose stream is the current torch stream."""
        if es.generation <= 1 and self.pending is not None:
            self.restart()  # called after the first generation of a re-initialised population: drop the old one's rows
        self._exchange(lambda i: es.pack_elites_device(self.mine[i].data_ptr(), self.E),
                       lambda i: es.inject_gathered_device(self.all[i].data_ptr(), self.world, self.rank, self.E))

    # ---- host path (gloo tests, oracle islands) ---------------------------------------------
    def migrate_host(self, pack, inject) -> None:
        """pack(n) -> ndarray [n, width]; inject(ndarray [(world-1)*E, width])."""
        self._exchange(lambda i: self.mine[i].copy_(torch.from_numpy(pack(self.E))),
                       lambda i: inject(self._others(i).cpu().numpy()))
