"""Host random draws of the hot path, and their static-buffer form for hipGraph replay.

The reference's CPU path draws two things per forward from the CPU default generator:
the per-cloud normal flip (classification/modules/recons_utils.py:50) and one FPS start index per
sampling stage (classification/modules/pointnet2_utils.py:66).  Eagerly we make the same calls and
copy the few bytes to the device.  A captured step cannot contain host work, so under
`StaticDraws` every draw is served from a device buffer that lives as long as the graph; `refill()`
repeats the same CPU-generator calls in the same order before each replay and overwrites the buffers.
"""
import torch

_active = None


def _cpu_draw(kind, b, n):
    if kind == "flip":      # same call as recons_utils.py:50
        return (torch.randint(0, 2, (b, 1, 1)).float() * 2. - 1.).view(b)
    if kind == "npflip":    # segmentation: numpy's global generator, segmentation/modules/recons_utils.py:29-35
        import numpy as np
        return torch.from_numpy(np.where(np.random.rand(b) < 0.5, 1.0, -1.0).astype(np.float32))
    return torch.randint(0, n, (b,), dtype=torch.long).to(torch.int32)   # pointnet2_utils.py:66


class StaticDraws:
    """Context manager: record the draws made while capturing, replay them with fresh numbers later.
    All draws live in ONE device arena (int32 words; flips are stored as float bits) fed from one pinned host
    arena, so a refill is a single host-to-device copy however many draws the model makes."""
    ARENA = 4096          # words; a forward of the shipped models draws B * (1 + number of sampling stages) values

    def __init__(self, device):
        self.device = device
        self.slots = []        # (kind, b, n, device view, (lo, words) in the arenas)
        self.cursor = None     # not None while re-running the same code path (warm-up iterations)
        self.dev = torch.zeros((self.ARENA,), dtype=torch.int32, device=device)
        # TWO pinned host arenas, used alternately: the host runs ahead of the GPU under graph replay, so the H2D copy of
        # refill s may still be pending when refill s+1 writes its numbers -- into the other arena; an arena is only
        # rewritten after the copy that read it has finished (`copied` event).
        cuda = device.type == "cuda"
        self.hosts = [torch.zeros((self.ARENA,), dtype=torch.int32).pin_memory() if cuda
                      else torch.zeros((self.ARENA,), dtype=torch.int32) for _ in range(2)]
        self.copied = [torch.cuda.Event() if cuda else None for _ in range(2)]
        self.pending = [False, False]
        self.turn = 0
        self.used = 0

    def __enter__(self):
        global _active
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = None

    def begin_pass(self):
        self.cursor = 0

    @staticmethod
    def _typed(words, kind):
        return words.view(torch.float32) if kind in ("flip", "npflip") else words

    def draw(self, kind, b, n):
        if self.cursor is not None and self.cursor < len(self.slots):
            k, bb, nn, buf, _ = self.slots[self.cursor]
            assert (k, bb, nn) == (kind, b, n), "draw order changed between passes"
            self.cursor += 1
            return buf
        lo = self.used
        self.used = lo + (b + 3) // 4 * 4                   # 16-byte aligned views
        if self.used > self.ARENA:
            raise RuntimeError("StaticDraws arena exhausted")
        buf = self._typed(self.dev[lo:lo + b], kind)
        value = _cpu_draw(kind, b, n)
        buf.copy_(value)                              # first use: a real draw (never run a kernel on garbage)
        self.slots.append((kind, b, n, buf, (lo, b)))
        if self.cursor is not None:
            self.cursor += 1
        return buf

    def refill(self):
        """Fresh CPU-generator draws, in forward order, into the static buffers: one async H2D copy on the current
        stream."""
        t = self.turn
        self.turn = 1 - t
        if self.pending[t]:
            self.copied[t].synchronize()              # the copy issued two refills ago has read this arena
            self.pending[t] = False
        arena = self.hosts[t]
        for kind, b, n, _, (lo, words) in self.slots:
            self._typed(arena[lo:lo + words], kind).copy_(_cpu_draw(kind, b, n))
        if self.used:
            self.dev[:self.used].copy_(arena[:self.used], non_blocking=True)
            if self.copied[t] is not None:
                self.copied[t].record()
                self.pending[t] = True


def draw(kind, b, n, device):
    """kind: "flip" / "npflip" -> (b,) float +-1 (torch / numpy generator);  "fps" -> (b,) int32 in [0, n)."""
    if _active is not None:
        return _active.draw(kind, b, n)
    return _cpu_draw(kind, b, n).to(device, non_blocking=True)
