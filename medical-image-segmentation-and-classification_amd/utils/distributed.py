"""Data-parallel plumbing of the drop-in ``train()`` (utils/helpers.py): rank discovery and loader sharding.

The reference trains on one device (utils/helpers.py:317-342: one loader, one model); BASELINE.json's north star shards the
batch across the GPUs of a node.  The rule used here keeps every number of the reference's epoch log well defined:

* the BATCHES of an epoch are the loader's own (its batch size, its order, its last short batch): batch ``b`` of the epoch goes
  to rank ``b % world``, so optimiser step ``k`` consumes batches ``k*world .. k*world + world - 1`` — a global batch of
  ``world x batch_size`` samples whose gradient is the mean over ranks of the per-rank means (mi355.dp.DataParallel);
* the epoch's order is drawn ONCE, on rank 0, and broadcast (a ``shuffle=True`` loader draws from a per-process RNG);
* training pads the last optimiser step with batches from the start of the epoch when ``len(loader) % world != 0`` (what
  ``DistributedSampler`` does with samples): every rank takes part in every gradient exchange.  Validation is not padded —
  it needs no collective per batch — so its sums run over exactly the loader's batches, each evaluated once.
"""
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader


def dist_info():
    """(rank, world) of the default process group; (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class _ListBatches:
    """A batch sampler over a fixed list of index batches."""

    def __init__(self, batches):
        self.batches = batches

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def shard_batches(batches, rank, world, pad):
    """Rank ``rank``'s batches of an epoch: every ``world``-th one; ``pad``: wrap around so that all ranks get the same count."""
    batches = list(batches)
    if pad and batches and len(batches) % world:
        batches = batches + batches[:world - len(batches) % world]
    return batches[rank::world]


class RankShard:
    """Rank-local view of a loader for helpers.train(): iterating yields this rank's batches of the epoch (module docstring).
    Works on ``torch.utils.data.DataLoader`` (any sampler / collate_fn / workers: the epoch's index batches are taken from its
    ``batch_sampler``) and on ``utils.dataset.GpuBatchLoader``.  ``dataset`` / ``global_batches`` keep the WHOLE loader's
    sizes: the epoch log divides all-reduced sums by them."""

    def __init__(self, loader, rank, world, pad, group=None):
        self.loader, self.rank, self.world, self.pad, self.group = loader, int(rank), int(world), bool(pad), group
        self.dataset = loader.dataset
        self.global_batches = len(loader)
        if not (hasattr(loader, "epoch_batches") or isinstance(loader, DataLoader)):
            raise TypeError(f"cannot shard a {type(loader).__name__}: expected a DataLoader or a GpuBatchLoader")
        if isinstance(loader, DataLoader) and loader.batch_sampler is None:
            raise TypeError("cannot shard a DataLoader built with batch_size=None")

    def __len__(self):
        n = self.global_batches
        if self.pad and n % self.world:
            n += self.world - n % self.world
        return len(range(self.rank, n, self.world))

    def _epoch_batches(self):
        payload = [None]
        if self.rank == 0:
            ld = self.loader
            payload[0] = ld.epoch_batches() if hasattr(ld, "epoch_batches") else [list(b) for b in ld.batch_sampler]
        if self.world > 1:
            dist.broadcast_object_list(payload, src=0, group=self.group)
        return shard_batches(payload[0], self.rank, self.world, self.pad)

    def __iter__(self):
        mine = self._epoch_batches()
        ld = self.loader
        if hasattr(ld, "epoch_batches"):
            for idxs in mine:
                yield ld.load(idxs)
            return
        yield from DataLoader(ld.dataset, batch_sampler=_ListBatches(mine), num_workers=ld.num_workers, collate_fn=ld.collate_fn,
                              pin_memory=ld.pin_memory, worker_init_fn=ld.worker_init_fn)


def all_reduce_sums(*tensors, group=None):
    """SUM over ranks of a few 0-dim device tensors in ONE collective; returns them as a tuple of 0-dim tensors."""
    flat = torch.stack([t.to(torch.float64) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return tuple(flat[i] for i in range(len(tensors)))
