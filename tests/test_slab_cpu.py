"""z-slab driver on CPU: partitioning, and the exchange choreography over gloo with two ranks.

The stand-in engine below is a 1-D Yee line along z (one value per layer and component) with the
same half_e/half_h(part) contract and ghost layers as the 3-D engine; two ranks exchanging ghosts
through fdtd_amd.slab must reproduce the single-rank run bit for bit.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fdtd_amd import slab


def test_partition_is_contiguous_and_balanced():
    for Nz, world in [(1024, 8), (1000, 7), (8, 8), (13, 4), (5, 1)]:
        parts = slab.partition(Nz, world)
        assert parts[0][0] == 0 and parts[-1][1] == Nz
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        sizes = [hi - lo for lo, hi in parts]
        assert min(sizes) >= 1 and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        slab.partition(3, 4)


class Line:
    """E(k), k in [0,Nz]; H(k), k in [0,Nz): E(k) += c (H(k) - H(k-1)) for 1 <= k < Nz, source at kJ;
    H(k) += c (E(k+1) - E(k)).  Local storage: planes kl = 0..nzl+1, global k = z_lo - 1 + kl."""

    def __init__(self, Nz, z_lo, z_hi, kJ):
        self.Nz, self.z_lo, self.z_hi, self.kJ = Nz, z_lo, z_hi, kJ
        self.nzl = z_hi - z_lo
        self.last = z_hi == Nz
        self.E = torch.zeros(self.nzl + 2, dtype=torch.float64)
        self.H = torch.zeros(self.nzl + 2, dtype=torch.float64)
        self.log = []

    def _planes(self, part, electric):
        lo, hi = 1, (self.nzl + 2 if self.last else self.nzl + 1)
        if part == slab.PART_ALL:
            return range(lo, hi)
        if electric:
            return range(lo, lo + 1) if part == slab.PART_EDGE else range(lo + 1, hi)
        top = self.nzl
        return range(top, top + 1) if part == slab.PART_EDGE else [p for p in range(lo, hi) if p != top]

    def half_e(self, n, part, stream=None):
        for kl in self._planes(part, True):
            k = self.z_lo - 1 + kl
            if 1 <= k < self.Nz:
                self.E[kl] = self.E[kl] + 0.5 * (self.H[kl] - self.H[kl - 1])
            if k == self.kJ:
                self.E[kl] = self.E[kl] + np.sin(0.3 * (n + 1))

    def half_h(self, part, stream=None):
        for kl in self._planes(part, False):
            k = self.z_lo - 1 + kl
            if k < self.Nz:
                self.H[kl] = self.H[kl] + 0.5 * (self.E[kl + 1] - self.E[kl])


class FusedLine(Line):
    """The same line with the fused engine's contract: two buffer sets, fused(n, part) that computes
    E_new(k) and, one plane behind, H_new(k-1) chunk by chunk (a chunk recomputes E_new of the plane
    that closes it), the top layer's H left to half_h(PART_EDGE) on a slab below another one."""
    chunk = 4

    def __init__(self, Nz, z_lo, z_hi, kJ):
        super().__init__(Nz, z_lo, z_hi, kJ)
        self.bufs = [(self.E, self.H), (torch.zeros_like(self.E), torch.zeros_like(self.H))]
        self.cur, self.phase = 0, 0

    def parity(self):
        return self.cur

    def fused_top(self, stream=None):
        assert self.phase != 1
        if not self.last:
            self.half_h(slab.PART_EDGE)

    def _e_new(self, n, kl, Eo, Ho):
        k = self.z_lo - 1 + kl
        v = Eo[kl].clone()
        if 1 <= k < self.Nz:
            v = v + 0.5 * (Ho[kl] - Ho[kl - 1])
        if k == self.kJ:
            v = v + np.sin(0.3 * (n + 1))
        return v

    def fused(self, n, part, stream=None):
        lo, hi = 1, (self.nzl + 2 if self.last else self.nzl + 1)
        few = min(self.chunk, 3)                              # the engine also uses short EDGE / TOP parts
        cut = min(lo + few, hi)
        cut2 = max(cut, hi - few)
        first = part in (slab.PART_ALL, slab.PART_EDGE)
        assert (self.phase == 0) if first else (self.phase == 1 if part == slab.PART_TOP else self.phase != 0)
        kl0, kl1 = {slab.PART_ALL: (lo, hi), slab.PART_EDGE: (lo, cut), slab.PART_TOP: (cut2, hi),
                    slab.PART_INTERIOR: (cut, cut2 if self.phase == 2 else hi)}[part]
        old, new = (self.cur, 1 - self.cur) if first else (1 - self.cur, self.cur)
        (Eo, Ho), (En, Hn) = self.bufs[old], self.bufs[new]
        for cb in range(kl0, kl1, self.chunk):
            ce = min(cb + self.chunk, kl1)
            end = ce + 1 if ce < hi else ce
            e_prev = None
            for kl in range(cb, end):
                e = self._e_new(n, kl, Eo, Ho)
                if kl < ce:
                    En[kl] = e
                if kl > cb and self.z_lo - 1 + (kl - 1) < self.Nz:
                    Hn[kl - 1] = Ho[kl - 1] + 0.5 * (e - e_prev)
                e_prev = e
                if not self.last and kl == hi - 1 and kl < ce:     # only the chunk that owns the top layer carries it over
                    Hn[kl] = Ho[kl]
        if first:
            self.cur = 1 - self.cur
            self.E, self.H = self.bufs[self.cur]
        self.phase = 1 if part == slab.PART_EDGE else (2 if part == slab.PART_TOP else 0)


def _fused_worker(rank, world, port, Nz, steps, out, chunk=4):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z_lo, z_hi = slab.partition(Nz, world)[rank]
    eng = FusedLine(Nz, z_lo, z_hi, kJ=3)
    eng.chunk = chunk
    xs = [slab.HaloExchange(rank, world, [H[eng.nzl:eng.nzl + 1]], [H[0:1]], [E[1:2]], [E[eng.nzl + 1:eng.nzl + 2]]) for E, H in eng.bufs]
    st = slab.FusedSlabStepper(eng, xs)
    st.prime()
    st.run(0, steps)
    st.finish()
    owned_e = eng.E[1:eng.nzl + 1 + (1 if eng.last else 0)].clone()
    owned_h = eng.H[1:eng.nzl + 1].clone()
    gathered = [None] * world
    dist.all_gather_object(gathered, (owned_e.numpy(), owned_h.numpy()))
    if rank == 0:
        np.save(out + "_e.npy", np.concatenate([g[0] for g in gathered]))
        np.save(out + "_h.npy", np.concatenate([g[1] for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, Nz, steps, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z_lo, z_hi = slab.partition(Nz, world)[rank]
    eng = Line(Nz, z_lo, z_hi, kJ=3)
    x = slab.HaloExchange(rank, world, [eng.H[eng.nzl:eng.nzl + 1]], [eng.H[0:1]], [eng.E[1:2]], [eng.E[eng.nzl + 1:eng.nzl + 2]])
    st = slab.SlabStepper(eng, x)
    st.prime()
    st.run(0, steps)
    st.finish()
    owned_e = eng.E[1:eng.nzl + 1 + (1 if eng.last else 0)].clone()
    owned_h = eng.H[1:eng.nzl + 1].clone()
    gathered = [None] * world
    dist.all_gather_object(gathered, (owned_e.numpy(), owned_h.numpy()))
    if rank == 0:
        np.save(out + "_e.npy", np.concatenate([g[0] for g in gathered]))
        np.save(out + "_h.npy", np.concatenate([g[1] for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_exchange_reproduces_single_rank(tmp_path, world):
    Nz, steps = 23, 40
    ref = Line(Nz, 0, Nz, kJ=3)
    for n in range(steps):
        ref.half_e(n, slab.PART_ALL)
        ref.half_h(slab.PART_ALL)
    out = str(tmp_path / "slab")
    mp.spawn(_worker, args=(world, _free_port(), Nz, steps, out), nprocs=world, join=True)
    e, h = np.load(out + "_e.npy"), np.load(out + "_h.npy")
    assert np.array_equal(e, ref.E[1:Nz + 2].numpy())
    assert np.array_equal(h, ref.H[1:Nz + 1].numpy())
    assert np.abs(e).max() > 0.1


def test_fused_line_equals_half_steps():
    """The stand-in's fused step is the same arithmetic as its two half steps (single rank)."""
    Nz, steps = 23, 41
    ref, fl = Line(Nz, 0, Nz, kJ=3), FusedLine(Nz, 0, Nz, kJ=3)
    for n in range(steps):
        ref.half_e(n, slab.PART_ALL)
        ref.half_h(slab.PART_ALL)
        fl.fused(n, slab.PART_ALL)
    assert torch.equal(fl.E, ref.E) and torch.equal(fl.H, ref.H)


@pytest.mark.parametrize("world,chunk", [(2, 4), (3, 4), (2, 1)])
def test_fused_exchange_reproduces_single_rank(tmp_path, world, chunk):
    Nz, steps = 23, 41
    ref = Line(Nz, 0, Nz, kJ=3)
    for n in range(steps):
        ref.half_e(n, slab.PART_ALL)
        ref.half_h(slab.PART_ALL)
    out = str(tmp_path / "fslab")
    mp.spawn(_fused_worker, args=(world, _free_port(), Nz, steps, out, chunk), nprocs=world, join=True)
    e, h = np.load(out + "_e.npy"), np.load(out + "_h.npy")
    assert np.array_equal(e, ref.E[1:Nz + 2].numpy())
    assert np.array_equal(h, ref.H[1:Nz + 1].numpy())
    assert np.abs(e).max() > 0.1
