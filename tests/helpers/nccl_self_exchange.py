"""One rank, RCCL backend: a ghost-plane exchange of an engine with itself through
torch.distributed.batch_isend_irecv (send and receive both addressed to rank 0), between two kernels
on the compute stream.  Checks what a one-GPU box can check of the RCCL path: the backend accepts the
tensors that alias engine memory, and the send sees the planes the preceding kernel wrote, the
following kernel the planes the receive delivered.  Prints OK on success."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import fdtd_amd  # noqa: E402
from fdtd_amd import engine as E  # noqa: E402
from fdtd_amd import setup as fs  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

N = (64, 48, 40)
K = fs.constants()
d = (1053e-9 / 20,) * 3
dt = d[0] / (2 * K["c0"])
eng = fdtd_amd.Engine3D(N, (4,) * 6, d, dt, K["mu0"], 12, 64, precision=32, has_mu=True, mode=E.MODE_FUSED, z_range=(8, 24))
fs.load_cpml3d(eng, d, dt)
wf, mag = fs.default_source3d(N[0], N[1], 64, d, dt)
eng.set(E.JX_WF, wf)
eng.set(E.JX_MAG, mag)
eng.fill_materials(3, K["eps0"], 4 * K["eps0"], K["mu0"], 2 * K["mu0"])
planes = [{w: [E.halo_tensor(eng, w, c, par) for c in (0, 1)] for w in range(4)} for par in (0, 1)]
for n in range(12):
    eng.fused(n, E.PART_EDGE)                       # writes the new D of the lowest layer
    cur = planes[eng.parity()]
    ops = [dist.P2POp(dist.isend, t, 0) for t in cur[E.HALO_SEND_DOWN]] + [dist.P2POp(dist.irecv, t, 0) for t in cur[E.HALO_RECV_HI]]
    reqs = dist.batch_isend_irecv(ops)
    eng.fused(n, E.PART_TOP)
    for r in reqs:
        r.wait()
    for c in (0, 1):                                # the high ghost now holds what the lowest layer held
        assert torch.equal(cur[E.HALO_RECV_HI][c], cur[E.HALO_SEND_DOWN][c]), f"step {n}: plane {c} did not arrive"
    eng.fused_top()
    eng.fused(n, E.PART_INTERIOR)
torch.cuda.synchronize()
assert float(cur[E.HALO_SEND_DOWN][0].abs().max()) > 0, "nothing propagated: the exchange was not exercised"
assert np.isfinite(eng.sumsq(E.DX))

# The two-steps-per-launch stepper's messages: two consecutive planes of D, B and of the x- and y-face CPML arrays, taken
# with fdtd3d_planes from whichever buffers are current after a two-step launch.  Sent to itself: planes z_lo, z_lo+1
# land in the planes beyond the upper cut (z_hi, z_hi+1), as the upper neighbour's would.
from fdtd_amd import slab  # noqa: E402
own = (9, 23)
st = slab.make_gpu_pair_stepper(eng, 0, 1, own)
assert len(st.ids) > 6, "the CPML arrays are part of the exchange"
for n in range(12, 20, 2):
    eng.step_pairs(n, 1)
    send = [E.planes_tensor(eng, id_, own[0], 2) for id_ in st.ids]
    recv = [E.planes_tensor(eng, id_, own[1], 2) for id_ in st.ids]
    reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, t, 0) for t in send] + [dist.P2POp(dist.irecv, t, 0) for t in recv])
    for r in reqs:
        r.wait()
    for q, (a, b) in enumerate(zip(send, recv)):
        assert torch.equal(a, b), f"two-step launch at {n}: planes of array {st.ids[q]} did not arrive"
torch.cuda.synchronize()
assert float(send[0].abs().max()) > 0
dist.destroy_process_group()
print("OK")
