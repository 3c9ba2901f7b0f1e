"""Row counts as DEVICE data: what lets ONE captured hipGraph of the segmentation network serve packed batches whose cloud boundaries
-- and therefore every level's row count -- change each step (segmentation/util/data_util.py:15-23: the reference's collate function
concatenates clouds of whatever sizes the loader drew; consumed at segmentation/tool/train.py:280-290).

A captured graph freezes grid sizes and scalar kernel arguments.  Under a `Capacity`, tensors are allocated for the LARGEST batch the
step will see (`capacity` rows at level 0, capacity // 4 at level 1, ...), every launch is sized for those capacities, and each kernel
that reduces over rows -- the row GEMMs' BatchNorm sums, the weight gradients, the finalizes, the scatter / gather backward passes --
reads the batch's actual count from a small int32 table in device memory (`rows_dev` of include/repsurf_hip.h) that the host refills
before each replay.  Rows beyond the count are never read by a reduction; elementwise kernels may touch them (their indices are
padded with row 0, their values are whatever the buffers hold) and nothing valid depends on them.

`dev(rows)` is how the launch helpers of repsurf_amd.mlp_hip / ops find the count that belongs to a capacity: the capacities of one
step are pairwise distinct by construction (level rows N, N/4, N/16, ..., grouped rows 32 x those, fan rows 9 x N; `capacity` must be
a multiple of 256), so the number of rows a launch is sized for identifies its level.  Outside a `Capacity` block `dev()` returns
None and every kernel takes its row count from its scalar argument, as before."""
import torch

_active = None


class Capacity:
    """levels: row capacities of the network's levels (level 0 first); nsample / fan: rows per group of the grouped stacks / of the
    constructor.  `table` (int32, device) holds, for every distinct capacity, the matching count of the current batch; `fill(counts)`
    writes the counts of a new batch (host list of level row counts) through pinned memory on the current stream."""

    def __init__(self, levels, nsample, fan, device):
        self.levels, self.nsample, self.fan = list(levels), nsample, fan
        self.slots = {}
        self.formulas = []                       # slot -> (level, multiplier)
        for li, cap in enumerate(self.levels):
            for mul in ((1, fan) if li == 0 else (1, nsample)):
                rows = cap * mul
                if rows in self.slots:
                    raise ValueError(f"ragged.Capacity: capacity {rows} appears twice (levels {self.levels}): use a level-0 capacity that is a multiple of 256")
                self.slots[rows] = len(self.formulas)
                self.formulas.append((li, mul))
        self.table = torch.zeros((len(self.formulas),), dtype=torch.int32, device=device)
        self.host = torch.zeros((len(self.formulas),), dtype=torch.int32)
        if torch.device(device).type == "cuda":
            self.host = self.host.pin_memory()

    def fill(self, counts):
        """counts[l] = rows of level l in the batch about to run (each <= its capacity)."""
        for li, (n, cap) in enumerate(zip(counts, self.levels)):
            if n > cap or n < 0:
                raise ValueError(f"ragged.Capacity: level {li} holds {n} rows, captured for at most {cap}")
        vals = [counts[li] * mul for li, mul in self.formulas]
        self.host.copy_(torch.tensor(vals, dtype=torch.int32))
        self.table.copy_(self.host, non_blocking=True)

    def ptr(self, rows):
        slot = self.slots.get(int(rows))
        return None if slot is None else self.table.data_ptr() + 4 * slot

    def __enter__(self):
        global _active
        if _active is not None:
            raise RuntimeError("ragged.Capacity blocks do not nest")
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = None


def dev(rows):
    """Device address of the row count that belongs to a launch sized for `rows` rows, or None (no Capacity active / not a capacity)."""
    return None if _active is None else _active.ptr(rows)


def active():
    return _active
