"""z-slab decomposition of the 3-D grid over the GPUs of one node, one process per GPU.

Rank r owns the cell layers [z_lo, z_hi) of every component (the last rank also the closing wall
plane k = Nz).  Two nearest-neighbour exchanges per step, each one x-y plane of two tangential
components:

  after the H-half: Hx, Hy of the top owned layer go up   (the D update at the next slab's first
                    layer needs H(k) - H(k-1), reference sweeps FDTD3D.cpp:545, :627)
  after the E-half: Ex, Ey of the lowest owned layer go down (the B update at this slab's top layer
                    needs E(k+1) - E(k), FDTD3D.cpp:799, :912)

With x unit-stride and z slowest each message is one contiguous device range, sent straight out of
and received straight into the engine's arrays with torch.distributed (RCCL over xGMI on the GPU
box, gloo in the CPU tests).  Each exchange is started right after the single-plane "edge" kernel
that produces the data and overlaps the next half step's interior kernel, which does not touch the
ghost plane; the wait sits in front of the one edge kernel that needs it.
"""
from __future__ import annotations

import time
from typing import List, Sequence, Tuple

PART_ALL, PART_EDGE, PART_INTERIOR, PART_TOP = 0, 1, 2, 3

# Host time this process has spent inside HaloExchange.wait() and the number of waits that had something to
# wait for.  With RCCL a wait only makes the launch stream wait (the device-side idle time shows up in the
# engine's launch timeline instead); with a host-staged backend it is the real blocking time.
WAIT_CLOCK = {"seconds": 0.0, "calls": 0}


def partition(Nz: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced layer ranges; every rank gets at least one layer."""
    if world < 1 or Nz < world:
        raise ValueError(f"cannot split {Nz} layers over {world} ranks")
    base, extra = divmod(Nz, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


class HaloExchange:
    """Posts the up / down plane exchanges of one rank and hands back the requests."""

    def __init__(self, rank: int, world: int, send_up: Sequence, recv_lo: Sequence, send_down: Sequence, recv_hi: Sequence, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.send_up, self.recv_lo, self.send_down, self.recv_hi = list(send_up), list(recv_lo), list(send_down), list(recv_hi)

    def _staged(self) -> bool:
        """Device planes over a backend without device point-to-point (gloo): stage through the host.
        Only for rehearsing the multi-rank path on a box without RCCL peers."""
        import torch.distributed as dist
        t = (self.send_up + self.recv_lo + self.send_down + self.recv_hi)[0]
        return bool(getattr(t, "is_cuda", False)) and dist.get_backend(self.group) == "gloo"

    def _post(self, sends, send_to, recvs, recv_from):
        import torch.distributed as dist
        if not self._staged():
            ops = [dist.P2POp(dist.isend, t, send_to, self.group) for t in sends]
            ops += [dist.P2POp(dist.irecv, t, recv_from, self.group) for t in recvs]
            return dist.batch_isend_irecv(ops) if ops else []
        reqs = [_StagedSend(dist.isend(t.cpu(), send_to, self.group)) for t in sends]
        reqs += [_StagedRecv(t, recv_from, self.group) for t in recvs]
        return reqs

    def start_up(self):
        """Top owned H planes -> rank+1's low ghost; own low ghost <- rank-1."""
        up, down = self.rank + 1 < self.world, self.rank > 0
        return self._post(self.send_up if up else [], self.rank + 1, self.recv_lo if down else [], self.rank - 1)

    def start_down(self):
        """Lowest owned E planes -> rank-1's high ghost; own high ghost <- rank+1."""
        up, down = self.rank + 1 < self.world, self.rank > 0
        return self._post(self.send_down if down else [], self.rank - 1, self.recv_hi if up else [], self.rank + 1)

    @staticmethod
    def wait(reqs) -> None:
        if not reqs:
            return
        t0 = time.perf_counter()
        for r in reqs:
            r.wait()
        WAIT_CLOCK["seconds"] += time.perf_counter() - t0
        WAIT_CLOCK["calls"] += 1


class _StagedSend:
    def __init__(self, req):
        self.req = req

    def wait(self):
        self.req.wait()


class _StagedRecv:
    def __init__(self, target, src, group):
        import torch
        import torch.distributed as dist
        self.target = target
        self.buf = torch.empty(target.shape, dtype=target.dtype, device="cpu")
        self.req = dist.irecv(self.buf, src, group)

    def wait(self):
        self.req.wait()
        self.target.copy_(self.buf)


class FusedSlabStepper:
    """One rank's time loop with the fused whole-step kernel.

    `engine` needs fused(n, part), fused_top() and parity(); `exchanges` holds one HaloExchange per
    buffer parity (the planes alias whichever D/B buffers are current after the step's first launch).
    Per step: the lowest planes first, so that D of layer z_lo starts travelling down; then the highest
    planes; as soon as the upper neighbour's D plane is there, the single-plane H kernel finishes the top
    layer's B, which starts travelling up; the bulk of the slab is computed while both exchanges are in
    flight.  The next step's lowest planes wait for the B plane from below.
    """

    def __init__(self, engine, exchanges: Sequence[HaloExchange], stream=None):
        self.engine, self.xs, self.stream = engine, list(exchanges), stream
        self._pending_b: list = []

    def step(self, n: int) -> None:
        e, s = self.engine, self.stream
        HaloExchange.wait(self._pending_b)           # B of layer z_lo-1, for the lowest chunk
        e.fused(n, PART_EDGE, s)
        x = self.xs[e.parity() % len(self.xs)]
        pending_d = x.start_down()
        e.fused(n, PART_TOP, s)
        HaloExchange.wait(pending_d)                 # new D of layer z_hi
        e.fused_top(s)
        self._pending_b = x.start_up()
        e.fused(n, PART_INTERIOR, s)
        if getattr(e, "probe_count", 0):
            e.probe_record(s)

    def prime(self) -> None:
        """One exchange in each direction before the first step: sets up the point-to-point channels
        (seconds with RCCL) outside any timed region.  The planes carry the initial state, which is
        what the ghosts hold anyway."""
        x = self.xs[self.engine.parity() % len(self.xs)]
        HaloExchange.wait(x.start_down())
        HaloExchange.wait(x.start_up())

    def run(self, n0: int, count: int) -> None:
        for n in range(n0, n0 + count):
            self.step(n)

    def finish(self) -> None:
        HaloExchange.wait(self._pending_b)
        self._pending_b = []


class SlabStepper:
    """One rank's time loop.  `engine` needs half_e(n, part), half_h(part); `exchange` a HaloExchange."""

    def __init__(self, engine, exchange: HaloExchange, stream=None):
        self.engine, self.x, self.stream = engine, exchange, stream
        self._pending_h: list = []       # H planes travelling up since the previous step

    def step(self, n: int) -> None:
        e, s = self.engine, self.stream
        # E-half: layers above the lowest one never look at the low ghost
        e.half_e(n, PART_INTERIOR, s)
        HaloExchange.wait(self._pending_h)
        e.half_e(n, PART_EDGE, s)
        pending_e = self.x.start_down()
        # H-half: layers below the top one never look at the high ghost
        e.half_h(PART_INTERIOR, s)
        HaloExchange.wait(pending_e)
        e.half_h(PART_EDGE, s)
        self._pending_h = self.x.start_up()
        if getattr(e, "probe_count", 0):
            e.probe_record(s)

    def prime(self) -> None:
        """One exchange in each direction before the first step (see FusedSlabStepper.prime)."""
        HaloExchange.wait(self.x.start_down())
        HaloExchange.wait(self.x.start_up())

    def run(self, n0: int, count: int) -> None:
        for n in range(n0, n0 + count):
            self.step(n)

    def finish(self) -> None:
        """Complete the exchange left in flight by the last step."""
        HaloExchange.wait(self._pending_h)
        self._pending_h = []


def pair_slab_range(z_lo: int, z_hi: int, Nz: int) -> Tuple[int, int]:
    """Layers a rank has to STORE to advance its layers [z_lo, z_hi) two steps per launch: one more either side of
    every cut (see PairSlabStepper)."""
    return max(0, z_lo - 1), min(Nz, z_hi + 1)


def pair_slabs_possible(parts: Sequence[Tuple[int, int]], Nz: int, npml_z: Tuple[int, int]) -> bool:
    """Every slab at least two layers thick, every cut clear of the z layers by the planes that cross it."""
    cuts = [lo for lo, _ in parts[1:]]
    return all(hi - lo >= 2 for lo, hi in parts) and all(c - 2 > npml_z[0] + 1 and c + 2 < Nz - npml_z[1] - 1 for c in cuts)


class PairSlabStepper:
    """One rank's time loop with two time steps per kernel launch (fdtd3d_step_pairs on a slab).

    The kernel recomputes its halos from what the slab stores, and next to a cut the slab's single ghost plane
    cannot feed a second step: after a launch everything is exact except the outermost stored layer either side
    and the ghost planes.  So neighbouring slabs overlap: this rank stores [z_lo-1, z_hi+1) (`pair_slab_range`),
    answers for [z_lo, z_hi), and after every launch the two planes beyond each cut -- D, B and the x- and y-face
    CPML state of both families -- are overwritten with the neighbour's exact ones: planes z_hi-2, z_hi-1 go up,
    planes z_lo, z_lo+1 go down (each a contiguous device range per array).  One exchange per two steps instead of
    two per step; the launch is cut in two (pair_part: the three planes at either end first, then the bulk) so that
    the planes travel while the bulk is computed -- the neighbour's bulk launch neither reads the buffers they land in
    nor writes those planes.  An odd step at the end is one launch of the one-step kernel followed by the same exchange.

    `engine` needs step_pairs(n, 1), fused(n, part), parity() and planes via `tensor_of(engine, id, k, count)`.
    """

    def __init__(self, engine, rank: int, world: int, own: Tuple[int, int], ids: Sequence[int], tensor_of, group=None, stream=None):
        self.engine, self.rank, self.world, self.own, self.ids = engine, rank, world, own, list(ids)
        self.tensor_of, self.group, self.stream = tensor_of, group, stream
        self.split = hasattr(engine, "pair_part")     # launch in parts so that the exchange overlaps the bulk
        self._x = {}                      # HaloExchange per set of current buffers
        self._singles = 0                 # one-step launches so far: they swap D and B but not the B-side CPML buffers

    def _post(self):
        """Start the exchange of the planes beyond the cuts out of / into the buffers that are current now."""
        e = self.engine
        key = (e.parity(), self._singles & 1)
        if key not in self._x:
            lo, hi = self.own
            up, down = self.rank + 1 < self.world, self.rank > 0
            t = lambda k: [self.tensor_of(e, id_, k, 2) for id_ in self.ids]
            self._x[key] = HaloExchange(self.rank, self.world, t(hi - 2) if up else [], t(lo - 2) if down else [],
                                        t(lo) if down else [], t(hi) if up else [], self.group)
        x = self._x[key]
        return x.start_up() + x.start_down()

    def _exchange(self) -> None:
        HaloExchange.wait(self._post())

    def pair(self, n: int) -> None:
        e = self.engine
        if not self.split:
            e.step_pairs(n, 1, self.stream)
            self._exchange()
            return
        # the planes next to the cuts first (the buffers change roles with this launch), on their way while the bulk runs
        e.pair_part(n, PART_EDGE, self.stream)
        reqs = self._post()
        e.pair_part(n, PART_INTERIOR, self.stream)
        HaloExchange.wait(reqs)

    def single(self, n: int) -> None:
        self.engine.fused(n, PART_ALL, self.stream)
        self._singles += 1
        self._exchange()

    def prime(self) -> None:
        """One exchange before the first step: sets the point-to-point channels up outside any timed region (the
        planes carry the initial state, which is what both sides hold anyway)."""
        self._exchange()

    def run(self, n0: int, count: int) -> None:
        n = n0
        for _ in range(count // 2):
            self.pair(n)
            n += 2
        if count % 2:
            self.single(n)

    def finish(self) -> None:
        pass


def make_gpu_pair_stepper(engine, rank: int, world: int, own: Tuple[int, int], group=None, stream=None):
    """PairSlabStepper over an Engine3D created with z_range = pair_slab_range(*own, Nz): the exchanged arrays are
    D, B and every x- and y-face CPML array the engine holds."""
    from .engine import BX, DX, PHI_E, PHI_M, planes_tensor
    ids = [DX + c for c in range(3)] + [BX + c for c in range(3)]
    for base in (PHI_E, PHI_M):
        for comp in range(3):
            for axis in (0, 1):
                for side in (0, 1):
                    id_ = base + (comp * 3 + axis) * 2 + side
                    if axis != comp and engine.size(id_):
                        ids.append(id_)
    return PairSlabStepper(engine, rank, world, own, ids, planes_tensor, group, stream)


def make_gpu_stepper(engine, rank: int, world: int, group=None, stream=None):
    """Wire an Engine3D's own ghost planes into a stepper (tensors alias engine memory): a
    FusedSlabStepper for MODE_FUSED engines, a SlabStepper otherwise."""
    from .engine import HALO_RECV_HI, HALO_RECV_LO, HALO_SEND_DOWN, HALO_SEND_UP, halo_tensor

    def exchange(parity=None):
        planes = {w: [halo_tensor(engine, w, c, parity) for c in (0, 1)] for w in (HALO_SEND_UP, HALO_RECV_LO, HALO_SEND_DOWN, HALO_RECV_HI)}
        return HaloExchange(rank, world, planes[HALO_SEND_UP], planes[HALO_RECV_LO], planes[HALO_SEND_DOWN], planes[HALO_RECV_HI], group)

    if not getattr(engine, "is_fused", False):
        return SlabStepper(engine, exchange(), stream)
    return FusedSlabStepper(engine, [exchange(0), exchange(1)], stream)
