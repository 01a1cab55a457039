"""Readers of the drivers' output formats against files the reference itself wrote
(tests/golden/run3d, run2d) and against hand-made files in both VTI encodings."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from fdtd_amd import fieldio
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_vti_files_read_back():
    fx = os.path.join(util.GOLDEN, "run3d")
    meta = json.load(open(os.path.join(fx, "meta.json")))
    N = tuple(meta["N"])
    ex, dims, origin, spacing = fieldio.read_vti(os.path.join(fx, "Ex_t00040.vti.gz"))
    mag, _, _, _ = fieldio.read_vti(os.path.join(fx, "E_mag_t00040.vti.gz"))
    assert dims == N and ex.shape == N and origin == (0.0, 0.0, 0.0)
    assert spacing[0] == pytest.approx(1053e-9 / 20, rel=1e-5)
    # file order is x fastest: value number q sits at (q % Nx, (q // Nx) % Ny, q // (Nx Ny))
    txt = gzip.open(os.path.join(fx, "Ex_t00040.vti.gz"), "rb").read().decode()
    flat = np.array(txt.split('NumberOfComponents="1">\n')[1].split("        </DataArray>")[0].split(), dtype=np.float64)
    for q in (0, 5, N[0], N[0] * N[1] + 3, flat.size - 1):
        assert ex[q % N[0], (q // N[0]) % N[1], q // (N[0] * N[1])] == flat[q]
    assert np.all(mag >= 0) and np.all(mag + 1e-300 >= np.abs(ex))            # |E| bounds its x component
    assert [fieldio.frame_step(f) for f in fieldio.list_frames(fx, "E_mag")] == [0, 20, 40]
    assert [fieldio.frame_step(f) for f in fieldio.list_frames(fx, "E_mag_rms")] == [20, 40]


def test_appended_and_ascii_encodings_agree(tmp_path):
    rng = np.random.default_rng(3)
    N = (5, 4, 3)
    vals = rng.standard_normal(N[0] * N[1] * N[2]).astype(np.float32)
    head = ('<?xml version="1.0"?>\n<VTKFile type="ImageData" version="1.0" byte_order="LittleEndian"%s>\n'
            '  <ImageData WholeExtent="0 4 0 3 0 2" Origin="0 0 0" Spacing="1 2 3">\n    <Piece Extent="0 4 0 3 0 2">\n'
            '      <PointData Scalars="f">\n')
    a = tmp_path / "f_t00001.vti"
    a.write_text(head % "" + '        <DataArray type="Float64" Name="f" format="ascii" NumberOfComponents="1">\n'
                 + " ".join(repr(float(v)) for v in vals) + " \n        </DataArray>\n      </PointData>\n    </Piece>\n  </ImageData>\n</VTKFile>\n")
    b = tmp_path / "f_t00002.vti"
    b.write_bytes((head % ' header_type="UInt64"').encode()
                  + b'        <DataArray type="Float32" Name="f" format="appended" offset="0" NumberOfComponents="1"/>\n'
                  b'      </PointData>\n    </Piece>\n  </ImageData>\n  <AppendedData encoding="raw">\n   _'
                  + np.array([vals.nbytes], dtype="<u8").tobytes() + vals.tobytes() + b"\n  </AppendedData>\n</VTKFile>\n")
    da, dims, _, spacing = fieldio.read_vti(str(a))
    db, dims_b, _, _ = fieldio.read_vti(str(b))
    assert dims == dims_b == N and spacing == (1.0, 2.0, 3.0)
    assert np.array_equal(da, db) and np.array_equal(da[:, 0, 0], vals[:5].astype(np.float64))
    bad = tmp_path / "g_t00003.vti"
    bad.write_bytes(b.read_bytes().replace(b'WholeExtent="0 4', b'WholeExtent="0 5'))
    with pytest.raises(ValueError):
        fieldio.read_vti(str(bad))


def test_reference_csv_files_read_back():
    fx = os.path.join(util.GOLDEN, "run2d")
    d = fieldio.read_dims2d(os.path.join(fx, "dims.txt.gz"))
    meta = json.load(open(os.path.join(fx, "meta.json")))
    assert (d["Nx"], d["Nz"]) == (meta["Nx"], meta["Nz"]) and d["dx"] > 0
    ey, x, z = fieldio.read_csv2d(os.path.join(fx, "ey_t00080.csv.gz"), d["Nx"], d["Nz"])
    assert ey.shape == (d["Nx"], d["Nz"]) and np.abs(ey).max() > 0
    assert x[1] - x[0] == pytest.approx(d["dx"]) and z[1] - z[0] == pytest.approx(d["dz"])
    assert np.unravel_index(np.abs(fieldio.read_csv2d(os.path.join(fx, "ey_t00020.csv.gz"), d["Nx"], d["Nz"])[0]).argmax(), ey.shape)[0] in range(meta["i_Jy"] - 6, meta["i_Jy"] + 7)
    with pytest.raises(ValueError):
        fieldio.read_csv2d(os.path.join(fx, "ey_t00080.csv.gz"), d["Nx"] + 1, d["Nz"])


def test_energy_report_and_series_verdict():
    vol = np.zeros((40, 40, 40))
    vol[18:22, 18:22, 18:22] = 2.0
    r = fieldio.energy_report(vol, rim=5)
    assert r["finite"] and r["total"] == 4 * 64 and r["centre"] == r["total"] and r["rim_share"] == 0
    vol[0, :, :] = 1.0
    r = fieldio.energy_report(vol, rim=5)
    # the face x = 0 is counted once as the x rim and its edges again in the y and z rims, as the reference counts
    assert r["rim"] == 1600 + 4 * (5 * 40) and r["total"] == 256 + 1600
    vol[3, 3, 3] = np.inf
    assert not fieldio.energy_report(vol, rim=5)["finite"]
    assert fieldio.check_series([1.0, 5.0, 50.0]) == (True, 50.0)
    assert fieldio.check_series([1.0, 500.0])[0] is False and fieldio.check_series([1.0, 0.01])[0] is False
    assert fieldio.check_series([2.0]) == (True, 0.0)


def test_verify_tool_on_reference_files(tmp_path):
    for sub, extra in (("run3d", ["--rim", "3"]), ("run2d", ["--csv", "--rim", "5"])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_output.py"), os.path.join(util.GOLDEN, sub)] + extra,
                           capture_output=True, text=True, timeout=120)
        assert "energy ratio" in r.stdout, r.stdout + r.stderr
        assert r.returncode in (0, 1)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_output.py"), str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "no E_mag" in r.stdout
