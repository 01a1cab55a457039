"""The C++ host drivers (CMake targets fdtd3d / fdtd2d) end to end on the GPU: with fp64 arithmetic
their output files must equal, byte for byte, the files the reference's own run() wrote for the
same small problems (tests/golden/run3d, run2d: device observables + writers + output cadence)."""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

import fdtd_amd
from oracle import pyoracle as po
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(binary, env, cwd):
    exe = os.path.join(ROOT, binary, "build", binary)
    if not os.path.exists(exe):
        pytest.fail(f"{exe} missing: run __graft_entry__.build()")
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([exe], cwd=cwd, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout[-2000:]
    return r.stdout


def _compare_dir(got_dir, fixture_dir):
    names = sorted(f[:-3] for f in os.listdir(fixture_dir) if f.endswith(".gz"))
    assert sorted(os.listdir(got_dir)) == names
    for n in names:
        want = gzip.open(os.path.join(fixture_dir, n + ".gz"), "rb").read()
        got = open(os.path.join(got_dir, n), "rb").read()
        assert got == want, f"{n}: differs from the reference's file"
    return names


def test_fdtd3d_driver_writes_the_reference_files(tmp_path):
    fx = os.path.join(util.GOLDEN, "run3d")
    meta = json.load(open(os.path.join(fx, "meta.json")))
    out = _run("fdtd3d", dict(FDTD3D_NX=meta["N"][0], FDTD3D_NY=meta["N"][1], FDTD3D_NZ=meta["N"][2], FDTD3D_NPML=meta["npml"][0],
                              FDTD3D_STEPS=meta["steps"], FDTD3D_PRECISION=64), str(tmp_path))
    assert "[grid] Nx=16, Ny=14, Nz=24" in out and "[n=40] frame dumped." in out and "Done. VTI in ./output_3d" in out
    names = _compare_dir(str(tmp_path / "output_3d"), fx)
    assert "E_mag_rms_t00040.vti" in names and "Ex_t00020.vti" in names


def test_fdtd3d_driver_with_the_window_inside_two_step_launches(tmp_path):
    """FDTD_PAIR=1 makes the engine take two steps per launch on this small grid too: the driver's runs of steps then go
    through the two-step kernel with the RMS window advanced inside it, completed windows kept on the way (the window
    of 20 steps completes on the first step of a launch) -- and the files are still the reference's, byte for byte."""
    fx = os.path.join(util.GOLDEN, "run3d")
    meta = json.load(open(os.path.join(fx, "meta.json")))
    env = dict(FDTD3D_NX=meta["N"][0], FDTD3D_NY=meta["N"][1], FDTD3D_NZ=meta["N"][2], FDTD3D_NPML=meta["npml"][0],
               FDTD3D_STEPS=meta["steps"], FDTD3D_PRECISION=64, FDTD_PAIR=1)
    out = _run("fdtd3d", env, str(tmp_path))
    assert "[n=40] frame dumped." in out
    _compare_dir(str(tmp_path / "output_3d"), fx)
    _run("fdtd3d", dict(env, FDTD3D_FRAME_EVERY=7), str(tmp_path))          # dumps at steps that are first steps of a launch as well
    names = sorted(os.listdir(str(tmp_path / "output_3d")))
    for n in names:
        if n.startswith("E_mag_rms"):
            want = gzip.open(os.path.join(fx, n + ".gz"), "rb").read()
            assert open(str(tmp_path / "output_3d" / n), "rb").read() == want, n
    assert "Ex_t00007.vti" in names and "Ex_t00035.vti" in names


def test_fdtd3d_driver_on_overlapping_slabs_with_the_window_inside_the_launches(tmp_path):
    """FDTD3D_SLABS with FDTD_PAIR=1: slabs that overlap by one layer either side of a cut, two steps per launch, the RMS
    window advanced inside the launches (the top layer of a slab finished after the planes from above have arrived) --
    the reference's files, byte for byte."""
    fx = os.path.join(util.GOLDEN, "run3d")
    meta = json.load(open(os.path.join(fx, "meta.json")))
    env = dict(FDTD3D_NX=meta["N"][0], FDTD3D_NY=meta["N"][1], FDTD3D_NZ=meta["N"][2], FDTD3D_NPML=meta["npml"][0],
               FDTD3D_STEPS=meta["steps"], FDTD3D_PRECISION=64, FDTD_PAIR=1)
    for slabs in (2, 3):
        out = _run("fdtd3d", dict(env, FDTD3D_SLABS=slabs), str(tmp_path))
        assert f"{slabs} z-slabs" in out
        _compare_dir(str(tmp_path / "output_3d"), fx)
    _run("fdtd3d", dict(env, FDTD3D_SLABS=2, FDTD_GROUP_TRANSPORT="rccl"), str(tmp_path))      # the planes by ncclSend / ncclRecv
    _compare_dir(str(tmp_path / "output_3d"), fx)


def test_fdtd3d_driver_full_storage_and_fp32(tmp_path):
    fx = os.path.join(util.GOLDEN, "run3d")
    meta = json.load(open(os.path.join(fx, "meta.json")))
    env = dict(FDTD3D_NX=meta["N"][0], FDTD3D_NY=meta["N"][1], FDTD3D_NZ=meta["N"][2], FDTD3D_NPML=meta["npml"][0], FDTD3D_STEPS=meta["steps"])
    for lean in (0, 1):           # the two-kernel paths: D,E,B,H resident / D,B resident (the default is the fused kernel)
        _run("fdtd3d", dict(env, FDTD3D_PRECISION=64, FDTD3D_FUSED=0, FDTD3D_LEAN=lean), str(tmp_path))
        _compare_dir(str(tmp_path / "output_3d"), fx)
    _run("fdtd3d", dict(env, FDTD3D_PRECISION=32), str(tmp_path))
    # fp32: same files, values within the stated drift (relative L2 <= 1e-6 of the field's norm)
    def values(path, opener):
        txt = opener(path).read().decode()
        body = txt.split('NumberOfComponents="1">\n')[1].split("        </DataArray>")[0]
        return np.array(body.split(), dtype=np.float64)
    for n in ("Ex_t00040.vti", "E_mag_rms_t00040.vti"):
        got = values(str(tmp_path / "output_3d" / n), lambda p: open(p, "rb"))
        want = values(os.path.join(fx, n + ".gz"), lambda p: gzip.open(p, "rb"))
        assert util.rel_l2(got, want) < 1e-6


def test_fdtd3d_driver_on_slabs_writes_the_reference_files(tmp_path):
    """FDTD3D_SLABS: the grid cut into z-slabs driven from the driver's one host thread (one per GPU on a
    multi-GPU node; all on the one device here) -- same files, byte for byte, in every storage mode."""
    fx = os.path.join(util.GOLDEN, "run3d")
    meta = json.load(open(os.path.join(fx, "meta.json")))
    env = dict(FDTD3D_NX=meta["N"][0], FDTD3D_NY=meta["N"][1], FDTD3D_NZ=meta["N"][2], FDTD3D_NPML=meta["npml"][0],
               FDTD3D_STEPS=meta["steps"], FDTD3D_PRECISION=64)
    out = _run("fdtd3d", dict(env, FDTD3D_SLABS=3), str(tmp_path))
    assert "3 z-slabs" in out
    _compare_dir(str(tmp_path / "output_3d"), fx)
    _run("fdtd3d", dict(env, FDTD3D_SLABS=5, FDTD3D_FUSED=0, FDTD3D_LEAN=1), str(tmp_path))
    _compare_dir(str(tmp_path / "output_3d"), fx)
    _run("fdtd3d", dict(env, FDTD3D_SLABS=2, FDTD3D_FUSED=0, FDTD3D_LEAN=0), str(tmp_path))
    _compare_dir(str(tmp_path / "output_3d"), fx)
    _run("fdtd3d", dict(env, FDTD3D_SLABS=0), str(tmp_path))             # one slab per visible GPU
    _compare_dir(str(tmp_path / "output_3d"), fx)


def test_fdtd3d_driver_binary_vti(tmp_path):
    """Opt-in raw appended VTI: the Float32 payload equals the reference's ASCII values rounded to fp32."""
    fx = os.path.join(util.GOLDEN, "run3d")
    meta = json.load(open(os.path.join(fx, "meta.json")))
    _run("fdtd3d", dict(FDTD3D_NX=meta["N"][0], FDTD3D_NY=meta["N"][1], FDTD3D_NZ=meta["N"][2], FDTD3D_NPML=meta["npml"][0],
                        FDTD3D_STEPS=meta["steps"], FDTD3D_PRECISION=64, FDTD3D_BINARY_VTI=1), str(tmp_path))
    cells = meta["N"][0] * meta["N"][1] * meta["N"][2]
    for n in ("Ex_t00040.vti", "E_mag_t00020.vti", "E_mag_rms_t00040.vti"):
        raw = open(str(tmp_path / "output_3d" / n), "rb").read()
        head, rest = raw.split(b'<AppendedData encoding="raw">\n   _', 1)
        assert b'header_type="UInt64"' in head and b'type="Float32"' in head and b'format="appended" offset="0"' in head
        nbytes = int(np.frombuffer(rest[:8], dtype="<u8")[0])
        assert nbytes == 4 * cells
        got = np.frombuffer(rest[8:8 + nbytes], dtype="<f4")
        assert rest[8 + nbytes:] == b"\n  </AppendedData>\n</VTKFile>\n"
        txt = gzip.open(os.path.join(fx, n + ".gz"), "rb").read().decode()
        want = np.array(txt.split('NumberOfComponents="1">\n')[1].split("        </DataArray>")[0].split(), dtype=np.float64)
        assert np.array_equal(got, want.astype(np.float32))


def test_fdtd2d_driver_writes_the_reference_files(tmp_path):
    fx = os.path.join(util.GOLDEN, "run2d")
    meta = json.load(open(os.path.join(fx, "meta.json")))
    out = _run("fdtd2d", dict(FDTD2D_NX=meta["Nx"], FDTD2D_NZ=meta["Nz"], FDTD2D_NPML=meta["npml"][0], FDTD2D_STEPS=meta["steps"],
                              FDTD2D_IJY=meta["i_Jy"], FDTD2D_KJY=meta["k_Jy"], FDTD2D_PRECISION=64), str(tmp_path))
    assert "Done. CSV in ./output_2d" in out
    names = _compare_dir(str(tmp_path / "output_2d"), fx)
    # the two-kernel path writes the same files
    _run("fdtd2d", dict(FDTD2D_NX=meta["Nx"], FDTD2D_NZ=meta["Nz"], FDTD2D_NPML=meta["npml"][0], FDTD2D_STEPS=meta["steps"],
                        FDTD2D_IJY=meta["i_Jy"], FDTD2D_KJY=meta["k_Jy"], FDTD2D_PRECISION=64, FDTD2D_FUSED=0), str(tmp_path))
    _compare_dir(str(tmp_path / "output_2d"), fx)
    # two steps per launch with the RMS window advanced inside the launches; one engine call per step
    for extra in (dict(FDTD2D_PAIR=1), dict(FDTD2D_PAIR=1, FDTD_Z_CHUNK2D_PAIR=3), dict(FDTD2D_STEP_BY_STEP=1)):
        _run("fdtd2d", dict(FDTD2D_NX=meta["Nx"], FDTD2D_NZ=meta["Nz"], FDTD2D_NPML=meta["npml"][0], FDTD2D_STEPS=meta["steps"],
                            FDTD2D_IJY=meta["i_Jy"], FDTD2D_KJY=meta["k_Jy"], FDTD2D_PRECISION=64, **extra), str(tmp_path))
        _compare_dir(str(tmp_path / "output_2d"), fx)
    assert "dims.txt" in names and "eymag_t00080.csv" in names and "ey_t00081.csv" in names


@pytest.mark.parametrize("prec,mode", [(32, 0), (32, 1), (64, 1)])
def test_device_observables_equal_twin(prec, mode):
    meta, ins, _, _ = util.load_fixture("g3_3d_hetero")
    eng = fdtd_amd.Engine3D(meta["N"], meta["npml"], (meta["dx"], meta["dy"], meta["dz"]), meta["dt"], meta["mu0"], meta["k_Jx"],
                            meta["steps"], precision=prec, mode=mode)
    tw = util.twin3d_from_meta(meta, prec)
    util.feed(eng, ins)
    util.feed(tw, ins)
    for n in range(12):
        eng.step(n, 1)
        tw.step(n, 1)
        eng.observe_accumulate(0.1, n % 5 == 0)
        tw.observe_accumulate(0.1, n % 5 == 0)
    util.assert_equal_arrays(eng.observe_rms(), tw.observe_rms(), "RMS window")
    a, b = eng.observe_frame()
    ta, tb = tw.observe_frame()
    util.assert_equal_arrays(a, ta, "centred Ex")
    util.assert_equal_arrays(b, tb, "|E|")
    assert np.abs(a).max() > 0


def test_device_observables_2d_equal_twin():
    meta, ins, _, _ = util.load_fixture("g4_2d_slab")
    eng = fdtd_amd.Engine2D(meta["Nx"], meta["Nz"], meta["npml"], meta["dx"], meta["dz"], meta["dt"], meta["mu0"], meta["eps0"],
                            meta["i_Jy"], meta["k_Jy"], meta["steps"], precision=32)
    tw = util.twin2d_from_meta(meta, 32)
    util.feed(eng, ins)
    util.feed(tw, ins)
    for n in range(30):
        eng.step(n, 1)
        tw.step(n, 1)
        eng.observe_accumulate(0.05, n % 10 == 0)
        tw.observe_accumulate(0.05, n % 10 == 0)
    util.assert_equal_arrays(eng.observe_rms(), tw.observe_rms(), "2-D RMS window")
