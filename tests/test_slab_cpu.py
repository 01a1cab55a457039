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


class PairLine:
    """Stand-in for an engine that advances two steps per launch on a slab (fdtd3d_step_pairs): the same 1-D line,
    layers [s_lo, s_hi) stored plus one ghost plane either side, two buffer sets.  Like the kernel it recomputes what
    it needs from what is stored -- values that would need a plane beyond the ghosts come out wrong, which is why
    PairSlabStepper lets slabs overlap and refreshes the two planes beyond every cut.  `aux` follows E with a
    recurrence of its own and, like the engine's B-side CPML state, changes buffers on two-step launches only."""

    def __init__(self, Nz, s_lo, s_hi, kJ):
        self.Nz, self.z_lo, self.z_hi, self.kJ = Nz, s_lo, s_hi, kJ
        self.nzl = s_hi - s_lo
        self.last = s_hi == Nz
        n = self.nzl + 2
        self.fields = [[torch.zeros(n, dtype=torch.float64) for _ in range(2)] for _ in range(2)]   # [set][E, H]
        self.aux = [torch.zeros(n, dtype=torch.float64) for _ in range(2)]
        self.cur, self.cur_aux = 0, 0

    def parity(self):
        return self.cur

    def planes(self, id_, k, count):
        kl = k - (self.z_lo - 1)
        assert 0 <= kl and kl + count <= self.nzl + 2, (k, count, self.z_lo, self.z_hi)
        arr = self.aux[self.cur_aux] if id_ == 2 else self.fields[self.cur][id_]
        return arr[kl:kl + count]

    def _one_step(self, E, H, A, n):
        """One whole step of every stored plane from (E, H, A); neighbours outside the storage count as zero."""
        m = E.numel()
        E1, H1 = E.clone(), H.clone()
        for kl in range(m):
            k = self.z_lo - 1 + kl
            if 1 <= k < self.Nz:
                E1[kl] = E[kl] + 0.5 * (H[kl] - (H[kl - 1] if kl >= 1 else 0.0))
            if k == self.kJ:
                E1[kl] = E1[kl] + np.sin(0.3 * (n + 1))
        for kl in range(m):
            k = self.z_lo - 1 + kl
            if 0 <= k < self.Nz:
                H1[kl] = H[kl] + 0.5 * ((E1[kl + 1] if kl + 1 < m else 0.0) - E1[kl])
        return E1, H1, 0.5 * A + E1

    def _owned(self):
        return range(1, self.nzl + 2 if self.last else self.nzl + 1)

    def step_pairs(self, n, pairs, stream=None):
        assert pairs == 1
        E, H = self.fields[self.cur]
        A = self.aux[self.cur_aux]
        E1, H1, A1 = self._one_step(E, H, A, n)
        E2, H2, A2 = self._one_step(E1, H1, A1, n + 1)
        nE, nH = self.fields[self.cur ^ 1]
        nA = self.aux[self.cur_aux ^ 1]
        nE.fill_(float("nan")); nH.fill_(float("nan")); nA.fill_(float("nan"))       # ghosts of the new set: stale
        for kl in self._owned():
            nE[kl], nH[kl], nA[kl] = E2[kl], H2[kl], A2[kl]
        self.cur ^= 1
        self.cur_aux ^= 1

    def fused(self, n, part, stream=None):
        assert part == slab.PART_ALL
        E, H = self.fields[self.cur]
        A = self.aux[self.cur_aux]
        E1, H1, A1 = self._one_step(E, H, A, n)
        nE, nH = self.fields[self.cur ^ 1]
        nE.fill_(float("nan")); nH.fill_(float("nan"))
        for kl in self._owned():
            nE[kl] = E1[kl]
            nH[kl] = H[kl] if (kl == self.nzl and not self.last) else H1[kl]     # top layer of a slab below another: left as it was
            A[kl] = A1[kl]                                                       # in place
        self.cur ^= 1


class SplitPairLine(PairLine):
    """The same with the two-step launch in two parts (fdtd3d_pair_part): the three planes at either end first -- the
    buffer sets change roles there -- then the rest, so that the stepper can post the exchange in between."""

    def pair_part(self, n, part, stream=None):
        owned = list(self._owned())
        edge = [kl for kl in owned if kl < owned[0] + 3 or kl > owned[-1] - 3]
        if part == slab.PART_EDGE:
            E, H = self.fields[self.cur]
            A = self.aux[self.cur_aux]
            E1, H1, A1 = self._one_step(E, H, A, n)
            self._res = self._one_step(E1, H1, A1, n + 1)
            for arr in self.fields[self.cur ^ 1] + [self.aux[self.cur_aux ^ 1]]:
                arr.fill_(float("nan"))
            self.cur ^= 1
            self.cur_aux ^= 1
            todo = edge
        else:
            assert part == slab.PART_INTERIOR
            todo = [kl for kl in owned if kl not in edge]
        nE, nH = self.fields[self.cur]
        nA = self.aux[self.cur_aux]
        for kl in todo:
            nE[kl], nH[kl], nA[kl] = self._res[0][kl], self._res[1][kl], self._res[2][kl]


def _pair_worker(rank, world, port, Nz, runs, out, split=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    parts = slab.partition(Nz, world)
    own = parts[rank]
    s_lo, s_hi = slab.pair_slab_range(own[0], own[1], Nz)
    eng = (SplitPairLine if split else PairLine)(Nz, s_lo, s_hi, kJ=3)
    st = slab.PairSlabStepper(eng, rank, world, own, [0, 1, 2], lambda e, id_, k, count: e.planes(id_, k, count))
    assert st.split == split
    st.prime()
    n = 0
    for count in runs:
        st.run(n, count)
        n += count
    st.finish()
    top = own[1] + (1 if eng.last else 0)
    got = [eng.planes(id_, own[0], (top if id_ == 0 else own[1]) - own[0]).clone().numpy() for id_ in (0, 1, 2)]
    # after the last exchange the planes beyond the cuts hold the neighbours' values, not leftovers
    stored_ok = all(bool(torch.isfinite(eng.planes(id_, s_lo - (1 if rank else 0), 1)).all()) for id_ in (0, 1, 2))
    gathered = [None] * world
    dist.all_gather_object(gathered, (got, stored_ok))
    if rank == 0:
        assert all(g[1] for g in gathered)
        for q, name in enumerate("eha"):
            np.save(out + f"_{name}.npy", np.concatenate([g[0][q] for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("world,runs", [(2, [8, 5, 4]), (3, [3, 6, 7]), (2, [1, 1, 2])])
def test_pair_exchange_reproduces_single_rank(tmp_path, world, runs, split):
    """Two steps per launch on overlapping slabs: pairs, odd steps in between (the buffer sets then change roles
    unevenly), several ranks -- equal to one rank stepping the whole line one step at a time."""
    Nz = 29
    ref = Line(Nz, 0, Nz, kJ=3)
    aux = torch.zeros(Nz + 2, dtype=torch.float64)
    for n in range(sum(runs)):
        ref.half_e(n, slab.PART_ALL)
        ref.half_h(slab.PART_ALL)
        aux = 0.5 * aux + ref.E
    out = str(tmp_path / "pslab")
    mp.spawn(_pair_worker, args=(world, _free_port(), Nz, runs, out, split), nprocs=world, join=True)
    e, h, a = (np.load(out + f"_{c}.npy") for c in "eha")
    assert np.array_equal(e, ref.E[1:Nz + 2].numpy())
    assert np.array_equal(h, ref.H[1:Nz + 1].numpy())
    assert np.array_equal(a[:Nz], aux[1:Nz + 1].numpy())
    assert np.abs(e).max() > 0.1


def test_pair_slabs_possible_and_ranges():
    assert slab.pair_slab_range(0, 10, 40) == (0, 11) and slab.pair_slab_range(10, 20, 40) == (9, 21) and slab.pair_slab_range(30, 40, 40) == (29, 40)
    assert slab.pair_slabs_possible([(0, 128), (128, 256), (256, 384)], 384, (20, 20))
    assert not slab.pair_slabs_possible([(0, 20), (20, 384)], 384, (20, 20))        # the cut lies in the low z layers
    assert not slab.pair_slabs_possible([(0, 1), (1, 40)], 40, (0, 0))              # a slab of one layer
