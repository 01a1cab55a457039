"""Run under torch.distributed.run (gloo, all ranks on cuda:0): every rank advances its slab two steps per launch
with PairSlabStepper and compares the layers it answers for -- D, B and all CPML state -- with a whole-grid engine
it steps itself, bit for bit.  Prints OK per rank."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fdtd_amd  # noqa: E402
from fdtd_amd import engine as E, slab  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    prec = int(os.environ.get("PAIR_PREC", "32"))
    has_mu = os.environ.get("PAIR_MU", "1") == "1"
    N = tuple(int(v) for v in os.environ.get("PAIR_N", "140,37,60").split(","))
    npml = tuple(int(v) for v in os.environ.get("PAIR_NPML", "6,7,5,4,6,5").split(","))
    steps = int(os.environ.get("PAIR_STEPS", "9"))
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    kJ = int(os.environ.get("PAIR_KJ", str(N[2] // 2)))
    rng = np.random.default_rng(3)
    wf, mag = po.default_source3d(N[0], N[1], steps, d, dt)
    mag = rng.random(mag.shape)
    parts = slab.partition(N[2], world)
    assert slab.pair_slabs_possible(parts, N[2], (npml[4], npml[5])), (parts, npml)
    own = parts[rank]
    ext = slab.pair_slab_range(own[0], own[1], N[2])

    def make(z_range):
        eng = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=E.MODE_FUSED, has_mu=has_mu, z_range=z_range)
        for a in range(3):
            for s in range(2):
                if npml[a * 2 + s]:
                    tabs = po.cpml_face(npml[a * 2 + s], s, d[a], d[a], dt)
                    for k in range(6):
                        eng.set(po.tab_id(a, s, k), tabs[k])
        eng.set(po.JX_WF, wf)
        eng.set(po.JX_MAG, mag)
        eng.fill_materials(99, K["eps0"], 4 * K["eps0"], K["mu0"], (2 if has_mu else 1) * K["mu0"])
        return eng

    whole, mine = make((0, N[2])), make(ext)
    mine.pair_shape(int(os.environ.get("PAIR_CHUNK", "5")), int(os.environ.get("PAIR_ROWS", "9")))
    st = slab.make_gpu_pair_stepper(mine, rank, world, own)
    st.prime()
    first = 4 if steps > 5 else 2
    st.run(0, first + 1)                      # pairs, then an odd step: the buffers change roles unevenly
    st.run(first + 1, steps - first - 1)
    st.finish()
    for n in range(steps):
        whole.step(n, 1)
    torch.cuda.synchronize()
    lo, hi = own
    bad = []
    for id_ in [po.DX, po.DY, po.DZ, po.BX, po.BY, po.BZ] + po.ALL_PHI3D:
        if not whole.size(id_):
            continue
        shape = po.shape3d(id_, N, npml)
        w = whole.get(id_).reshape(shape)
        m = mine.get(id_).reshape(shape)
        axis_z_face = id_ >= 100 and ((id_ - (200 if id_ >= 200 else 100)) // 2) % 3 == 2
        if axis_z_face:
            continue                          # z-face state belongs to the end slabs; compared through the fields it feeds
        top = hi + 1 if (rank == world - 1 and shape[2] == N[2] + 1) else hi
        if not np.array_equal(w[:, :, lo:top], m[:, :, lo:top]):
            bad.append(id_)
        # the reductions over the layers this rank answers for
        if id_ < 18:
            ss_w, cs_w = whole.reduce_range(id_, lo, top)
            ss_m, cs_m = mine.reduce_range(id_, lo, top)
            if cs_w != cs_m:
                bad.append(("checksum", id_))
    print(f"rank {rank} layers {own} stored {ext}: {'OK' if not bad else 'MISMATCH ' + str(bad)}", flush=True)
    dist.barrier()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
