"""Two steps per launch (fdtd3d_step_pairs) against the reference vectors, the twin and the one-step kernel (needs a GPU)."""
import os

import numpy as np
import pytest

import fdtd_amd
from fdtd_amd.engine import MODE_FUSED
from oracle import pyoracle as po
from tests import util
from tests.test_gpu_parity import FIX3D, engine3d_from_meta

pytestmark = pytest.mark.gpu


class Paired:
    """Back end that advances an engine two steps per launch wherever it can."""

    def __init__(self, eng):
        self.eng = eng

    def set(self, id_, a):
        self.eng.set(id_, a)

    def get(self, id_):
        return self.eng.get(id_)

    def step(self, n0, count):
        self.eng.step_pairs(n0, count // 2)
        if count % 2:
            self.eng.step(n0 + count - 1, 1)


@pytest.mark.parametrize("prec", [32, 64])
@pytest.mark.parametrize("rows,chunk", [(16, 48), (6, 2), (9, 5)])
def test_pairs_equal_reference_cavity(prec, rows, chunk):
    """BASELINE's vacuum cavity (no CPML): fp64 against the reference's own vectors, fp32 against the fp32 twin, bit for bit."""
    meta, ins, outs, _ = util.load_fixture("g2_3d_cavity")
    assert not any(meta["npml"])
    eng = engine3d_from_meta(meta, prec, mode=MODE_FUSED)
    eng.pair_shape(chunk, rows)
    util.feed(eng, ins)
    got = util.run_snapshots(Paired(eng), meta, outs)
    if prec == 64:
        ref = outs
    else:
        tw = util.twin3d_from_meta(meta, 32)
        util.feed(tw, ins)
        ref = util.run_snapshots(tw, meta, outs)
    for key in outs:
        util.assert_equal_arrays(got[key], ref[key], f"cavity pairs fp{prec} step {key[0]} array {key[1]}")


@pytest.mark.parametrize("name", FIX3D)
@pytest.mark.parametrize("prec", [32, 64])
@pytest.mark.parametrize("rows,chunk", [(16, 48), (6, 2), (9, 5)])
def test_pairs_equal_reference_3d(name, prec, rows, chunk):
    """Every 3-D fixture (CPML on all, some or no faces): fp64 against the reference's vectors, fp32 against the
    fp32 twin, bit for bit, fields and CPML state."""
    meta, ins, outs, _ = util.load_fixture(name)
    eng = engine3d_from_meta(meta, prec, mode=MODE_FUSED)
    eng.pair_shape(chunk, rows)
    util.feed(eng, ins)
    got = util.run_snapshots(Paired(eng), meta, outs)
    if prec == 64:
        ref = outs
    else:
        tw = util.twin3d_from_meta(meta, 32)
        util.feed(tw, ins)
        ref = util.run_snapshots(tw, meta, outs)
    for key in outs:
        util.assert_equal_arrays(got[key], ref[key], f"{name} pairs fp{prec} step {key[0]} array {key[1]}")


@pytest.mark.parametrize("prec,N,npml", [
    (32, (300, 9, 11), (5, 6, 2, 2, 3, 3)),       # three x tiles
    (32, (260, 20, 6), (3, 20, 1, 0, 0, 2)),      # the boundary between x tiles 1 and 2 (column 250) inside the high CPML strip
    (32, (130, 40, 9), (20, 20, 20, 16, 4, 4)),   # more rows than a workgroup owns, CPML rows in two row groups
    (64, (140, 6, 19), (62, 20, 0, 3, 4, 4)),     # fp64 tiles own 60 columns: boundary inside the low strip
    (64, (12, 33, 9), (4, 4, 4, 4, 4, 4)),
    (32, (64, 12, 40), (2, 2, 2, 2, 18, 19)),     # nearly every plane in a z layer
    (32, (256, 7, 9), (20, 20, 2, 2, 2, 2)),      # Nx = row pitch: the right end of a row ends at the wall, not in padding
    (64, (140, 6, 19), (10, 8, 0, 3, 4, 4)),      # fp64, both strips in x tile 0, two tiles after it
])
@pytest.mark.parametrize("has_mu", [False, True])
@pytest.mark.parametrize("rows,chunk", [(16, 48), (6, 2), (11, 3)])
@pytest.mark.parametrize("wrap", ["1", "0"])      # x tile 0 holding both ends of a row (where both strips fit it), or tiles in a row
def test_pairs_with_cpml_equal_twin_ragged(prec, N, npml, has_mu, rows, chunk, wrap, monkeypatch):
    monkeypatch.setenv("FDTD_PAIR_WRAP", wrap)
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    steps = 24
    kJ = N[2] // 2
    rng = np.random.default_rng(13)
    wf, mag = po.default_source3d(N[0], N[1], steps, d, dt)
    mag = rng.random(mag.shape)
    eps = [K["eps0"] * (1 + 3 * rng.random(po.shape3d(po.EPS_X + c, N, npml))) for c in range(3)]
    mu = [K["mu0"] * (1 + rng.random(po.shape3d(po.MU_X + c, N, npml))) for c in range(3)]
    tw = po.Twin3D(prec, N, npml, d, dt, K["mu0"], kJ, steps, has_mu)
    pair = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=MODE_FUSED, has_mu=has_mu)
    pair.pair_shape(chunk, rows)
    for be in (tw, pair):
        for a in range(3):
            for s in range(2):
                tabs = po.cpml_face(npml[a * 2 + s], s, d[a], d[a], dt)
                for k in range(6):
                    if npml[a * 2 + s]:
                        be.set(po.tab_id(a, s, k), tabs[k])
        be.set(po.JX_WF, wf)
        be.set(po.JX_MAG, mag)
        for c in range(3):
            be.set(po.EPS_X + c, eps[c])
            if has_mu:
                be.set(po.MU_X + c, mu[c])
    # pairs, one single step (the buffers change roles an odd number of times), pairs again, then single steps
    tw.step(0, 6); pair.step_pairs(0, 3)
    tw.step(6, 1); pair.step(6, 1)
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        if pair.size(id_):
            util.assert_equal_arrays(pair.get(id_), tw.get(id_), f"N={N} step 7 array {id_}")
    tw.step(7, 14); pair.step_pairs(7, 7)
    tw.step(21, 3); pair.step(21, 3)
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        if pair.size(id_):
            util.assert_equal_arrays(pair.get(id_), tw.get(id_), f"N={N} step 24 array {id_}")


@pytest.mark.parametrize("prec,N", [
    (32, (300, 9, 11)),       # three x tiles of 124 owned columns
    (32, (126, 40, 9)),       # Nx = the first tile's owned columns exactly; more rows than one workgroup
    (32, (256, 5, 7)),        # Nx = row pitch: the wall column is not stored
    (32, (7, 6, 5)),
    (64, (140, 6, 19)),       # fp64 tiles own 60 columns
    (64, (12, 33, 9)),
    (64, (64, 17, 4)),
])
@pytest.mark.parametrize("has_mu", [False, True])
@pytest.mark.parametrize("rows,chunk", [(16, 48), (6, 2), (11, 3)])
def test_pairs_equal_single_steps_ragged(prec, N, has_mu, rows, chunk):
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    steps = 24
    kJ = N[2] // 2
    npml = (0,) * 6
    rng = np.random.default_rng(11)
    wf, mag = po.default_source3d(N[0], N[1], steps, d, dt)
    mag = rng.random(mag.shape)
    eps = [K["eps0"] * (1 + 3 * rng.random(po.shape3d(po.EPS_X + c, N, npml))) for c in range(3)]
    mu = [K["mu0"] * (1 + rng.random(po.shape3d(po.MU_X + c, N, npml))) for c in range(3)]
    tw = po.Twin3D(prec, N, npml, d, dt, K["mu0"], kJ, steps, has_mu)
    pair = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=MODE_FUSED, has_mu=has_mu)
    pair.pair_shape(chunk, rows)
    for be in (tw, pair):
        be.set(po.JX_WF, wf)
        be.set(po.JX_MAG, mag)
        for c in range(3):
            be.set(po.EPS_X + c, eps[c])
            if has_mu:
                be.set(po.MU_X + c, mu[c])
    # 3 pairs, one single step (the buffers change roles an odd number of times), then pairs that run past the
    # end of the source waveform
    tw.step(0, 6); pair.step_pairs(0, 3)
    tw.step(6, 1); pair.step(6, 1)
    for id_ in po.FIELDS3D:
        if pair.size(id_):
            util.assert_equal_arrays(pair.get(id_), tw.get(id_), f"N={N} step 7 array {id_}")
    tw.step(7, 20); pair.step_pairs(7, 10)
    for id_ in po.FIELDS3D:
        if pair.size(id_):
            util.assert_equal_arrays(pair.get(id_), tw.get(id_), f"N={N} step 27 array {id_}")


def test_pairs_refuse_what_they_do_not_carry():
    K = po.consts()
    # a slab takes them (its driver lets slabs overlap and refreshes the planes next to the cuts, see PairSlabStepper)
    slab = fdtd_amd.Engine3D((16, 16, 16), (0,) * 6, (1e-7,) * 3, 1e-16, K["mu0"], 4, 3, precision=32, mode=MODE_FUSED, z_range=(0, 8))
    slab.step_pairs(0, 1)
    assert not slab.pair_info()["used_by_step"]
    lean = fdtd_amd.Engine3D((16, 16, 16), (0,) * 6, (1e-7,) * 3, 1e-16, K["mu0"], 4, 3, precision=32, mode=fdtd_amd.engine.MODE_LEAN)
    with pytest.raises(fdtd_amd.EngineError, match="FUSED"):
        lean.step_pairs(0, 1)
    probed = fdtd_amd.Engine3D((16, 16, 16), (0,) * 6, (1e-7,) * 3, 1e-16, K["mu0"], 4, 3, precision=32, mode=MODE_FUSED, z_range=(0, 8))
    probed.probe_add(po.DX, 3, 3, 3)
    with pytest.raises(fdtd_amd.EngineError, match="probes"):
        probed.step_pairs(0, 1)


@pytest.mark.parametrize("prec,N,npml,has_mu", [
    (32, (40, 21, 17), (5, 6, 4, 3, 3, 4), True),
    (64, (30, 18, 22), (4, 4, 0, 5, 6, 0), False),
    (32, (24, 20, 28), (0,) * 6, False),
])
def test_probe_series_of_pairs_equal_single_steps(prec, N, npml, has_mu):
    """Probes on two-step launches: the sample after the first step of every launch comes from one-plane launches of
    the one-step kernel before it (which must leave no trace), the sample after the second from the arrays.  The series
    of D, E, B and H probes -- inside CPML layers, next to walls, several per plane -- and the final state equal those
    of an engine that takes one step per launch; step() pairs by itself (FDTD_PAIR=1) and step_pairs() explicitly."""
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    steps = 15
    kJ = N[2] // 2
    rng = np.random.default_rng(5)
    wf, mag = po.default_source3d(N[0], N[1], steps, d, dt)
    mag = rng.random(mag.shape)
    cells = [(po.DX, 1, 2, kJ), (po.EX, 3, 4, kJ), (po.EY, 2, 1, 1), (po.EZ, N[0] - 2, N[1] - 2, N[2] - 2), (po.BX, 1, 1, 1),
             (po.HY, 2, 3, kJ), (po.HZ, N[0] - 2, 2, 2), (po.BZ, 4, N[1] - 2, N[2] - 1), (po.DZ, 5, 5, 0), (po.HX, 3, 3, N[2] - 1)]
    runs = []
    for mode in ("single", "auto", "explicit"):
        os.environ["FDTD_PAIR"] = "0" if mode == "single" else "1"
        try:
            eng = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=MODE_FUSED, has_mu=has_mu)
            for a in range(3):
                for s_ in range(2):
                    if npml[a * 2 + s_]:
                        tabs = po.cpml_face(npml[a * 2 + s_], s_, d[a], d[a], dt)
                        for k in range(6):
                            eng.set(po.tab_id(a, s_, k), tabs[k])
            eng.set(po.JX_WF, wf)
            eng.set(po.JX_MAG, mag)
            eng.fill_materials(99, K["eps0"], 4 * K["eps0"], K["mu0"], (2 if has_mu else 1) * K["mu0"])
            for c in cells:
                eng.probe_add(*c)
            if mode == "explicit":
                eng.step_pairs(0, 3); eng.step(6, 1); eng.step_pairs(7, 4)
            else:
                eng.step(0, 7); eng.step(7, 8)
            series = eng.probe_read()
            state = {id_: eng.get(id_) for id_ in [po.DX, po.DY, po.DZ, po.BX, po.BY, po.BZ] + po.ALL_PHI3D if eng.size(id_)}
            runs.append((series, state))
        finally:
            os.environ.pop("FDTD_PAIR", None)
    assert runs[0][0].shape == (steps, len(cells))
    assert np.any(runs[0][0] != 0)
    for mode, (series, state) in zip(("auto", "explicit"), runs[1:]):
        util.assert_equal_arrays(series, runs[0][0], f"probe series, {mode} pairs, N={N}")
        for id_, v in state.items():
            util.assert_equal_arrays(v, runs[0][1][id_], f"state after probed {mode} pairs, N={N}, array {id_}")


@pytest.mark.parametrize("seed", range(24))
def test_pairs_equal_single_steps_random_configurations(seed):
    """Random grids, CPML layer counts per face, mu grid or not, precision, workgroup shape and step counts: the
    engine stepping in pairs against a second engine stepping one step per launch -- checksums and sums of squares
    of all six stored arrays and of every CPML auxiliary array, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    prec = int(rng.choice([32, 64]))
    tile = 124 if prec == 32 else 60
    N = (int(rng.choice([rng.integers(5, 40), tile - 2, tile - 1, tile, tile + 1, rng.integers(tile, 3 * tile)])),
         int(rng.integers(4, 46)), int(rng.integers(4, 40)))
    def layers(n):
        lo, hi = int(rng.integers(0, 1 + min(12, (n - 2) // 2))), int(rng.integers(0, 1 + min(12, (n - 2) // 2)))
        return (lo if rng.random() < 0.8 else 0, hi if rng.random() < 0.8 else 0)
    npml = layers(N[0]) + layers(N[1]) + layers(N[2])
    if npml[0] and npml[1] and (N[0] - npml[1]) // 4 * 4 < -(-(npml[0] + 1) // 4) * 4:
        npml = (0,) + npml[1:]
    has_mu = bool(rng.integers(0, 2))
    rows, chunk = int(rng.integers(6, 17)), int(rng.choice([1, 2, 3, 5, 8, 48]))
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    steps = 2 * int(rng.integers(3, 9)) + 1
    kJ = int(rng.integers(0, N[2] + 1))
    wf, mag = po.default_source3d(N[0], N[1], steps, d, dt)
    mag = rng.random(mag.shape)
    engs = [fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=MODE_FUSED, has_mu=has_mu) for _ in range(2)]
    engs[0].pair_shape(chunk, rows)
    for eng in engs:
        for a in range(3):
            for s_ in range(2):
                if npml[a * 2 + s_]:
                    tabs = po.cpml_face(npml[a * 2 + s_], s_, d[a], d[a], dt)
                    for k in range(6):
                        eng.set(po.tab_id(a, s_, k), tabs[k])
        eng.set(po.JX_WF, wf)
        eng.set(po.JX_MAG, mag)
        eng.fill_materials(77 + seed, K["eps0"], 4 * K["eps0"], K["mu0"], (2 if has_mu else 1) * K["mu0"])
    first = 2 * int(rng.integers(1, steps // 2))
    engs[0].step_pairs(0, first // 2)
    engs[0].step(first, 1)
    engs[0].step_pairs(first + 1, (steps - first - 1) // 2)
    for n in range(steps):
        engs[1].step(n, 1)
    what = f"N={N} npml={npml} mu={has_mu} fp{prec} rows={rows} chunk={chunk} steps={steps}"
    for id_ in [po.DX, po.DY, po.DZ, po.BX, po.BY, po.BZ] + po.ALL_PHI3D:
        if engs[0].size(id_):
            util.assert_equal_arrays(engs[0].get(id_), engs[1].get(id_), f"{what} array {id_}")


@pytest.mark.parametrize("ranks,env", [
    (2, {"PAIR_PREC": "32", "PAIR_MU": "1"}),
    (3, {"PAIR_PREC": "64", "PAIR_MU": "0", "PAIR_CHUNK": "48", "PAIR_ROWS": "16"}),
    (3, {"PAIR_PREC": "32", "PAIR_MU": "1", "PAIR_N": "300,20,47", "PAIR_NPML": "5,6,0,3,4,4", "PAIR_STEPS": "12", "PAIR_KJ": "21"}),
])
def test_pair_slab_stepper_equals_whole_grid(ranks, env):
    """Slabs that overlap by one layer either side of a cut, two steps per launch, the two planes beyond each cut
    exchanged after every launch (gloo, all ranks on cuda:0): every rank's own layers -- D, B, x- and y-face CPML
    state, range checksums -- equal a whole-grid engine's bit for bit (tests/helpers/pair_slab_check.py)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "helpers", "pair_slab_check.py")]
    r = subprocess.run(cmd, cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert r.stdout.count(": OK") == ranks, r.stdout[-1500:]
