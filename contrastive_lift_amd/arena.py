"""Flat fp32 arenas for parameters, gradients and Adam moments.

All trainable tensors of the field live in ONE flat device buffer, in optimizer-group order, so that
  * the Adam step is one kernel launch per (lr, weight_decay) range (clift_adam),
  * the data-parallel gradient exchange is ONE RCCL all-reduce of a contiguous range per backward,
  * VM tables can be laid out channels-last and weight matrices with a 16-byte-aligned row pitch while the
    ``nn.Parameter`` objects keep the reference's shapes (state_dict compatible).
"""
import torch


def _round_up(x, m):
    return (x + m - 1) // m * m


class Slot:
    __slots__ = ("name", "shape", "kind", "offset", "numel", "pitch", "group")

    def __init__(self, name, shape, kind, group, pitch_align=4):
        self.name, self.shape, self.kind, self.group = name, tuple(shape), kind, group
        if kind == "grid":            # (1,C,H,W) stored channels-last: [H][W][C]
            _, c, h, w = self.shape
            self.pitch = c
            self.numel = h * w * c
        elif kind == "matrix":        # (out,in) stored row-major with in padded to a multiple of pitch_align (>= 4) floats
            o, i = self.shape
            self.pitch = _round_up(i, max(4, pitch_align))
            self.numel = o * self.pitch
        else:                         # "vector": 1-D
            self.pitch = 1
            self.numel = self.shape[0]
        self.offset = 0

    def view(self, flat):
        seg = flat[self.offset:self.offset + self.numel]
        if self.kind == "grid":
            _, c, h, w = self.shape
            return seg.view(h, w, c).permute(2, 0, 1).unsqueeze(0)
        if self.kind == "matrix":
            o, i = self.shape
            return seg.view(o, self.pitch)[:, :i]
        return seg


class Arena:
    """Layout = ordered list of Slots; ``groups`` maps group name -> (start, end) float offsets (contiguous)."""

    def __init__(self, slots, device):
        self.slots = list(slots)
        self.groups = {}
        off = 0
        cur = None
        for s in self.slots:
            off = _round_up(off, 4)            # every slot 16-byte aligned
            if s.group != cur:
                if s.group in self.groups:
                    raise ValueError(f"arena group {s.group} is not contiguous")
                if cur is not None:
                    self.groups[cur] = (self.groups[cur][0], off)
                self.groups[s.group] = (off, off)
                cur = s.group
            s.offset = off
            off += s.numel
        off = _round_up(off, 4)
        if cur is not None:
            self.groups[cur] = (self.groups[cur][0], off)
        self.total = off
        self.device = device
        self.by_name = {s.name: s for s in self.slots}

    def new_buffer(self):
        return torch.zeros(self.total, dtype=torch.float32, device=self.device)

    def views(self, flat):
        return {s.name: s.view(flat) for s in self.slots}

    def range_of(self, *groups):
        """Contiguous [start, end) covering the given groups (must be adjacent in layout order)."""
        a = min(self.groups[g][0] for g in groups)
        b = max(self.groups[g][1] for g in groups)
        covered = sum(self.groups[g][1] - self.groups[g][0] for g in groups)
        if covered != b - a:
            raise ValueError(f"groups {groups} are not adjacent in the arena")
        return a, b
