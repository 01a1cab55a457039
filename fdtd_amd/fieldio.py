"""Readers for the files the host drivers write, and the reference's after-the-fact checks on them.

The 3-D driver writes one VTI image per observable and frame -- ASCII Float64 exactly as the
reference's OutputManager does (fdtd3d/src/OutputManager.cpp:23-85), or, opt-in, a raw appended
Float32 block.  The 2-D driver writes "x,z,val" CSV files plus dims.txt (fdtd2d/src/OutputManager.cpp:20-44).
The reference reads its VTI files back with fdtd3d/scripts/verify_vti.py:48-82 (ASCII only) and
judges a run by the finiteness and the frame-to-frame energy ratio (:140-148, :255-262) and by where
the energy sits (fdtd3d/scripts/diagnose_energy.py:19-33); the same checks are offered here for
both encodings.  Files may be gzip-compressed (name ending in .gz).
"""
from __future__ import annotations

import gzip
import os
import re

import numpy as np

_STEP = re.compile(r"_t(\d+)\.(?:vti|csv)(?:\.gz)?$")
_APPENDED_MARK = b'<AppendedData encoding="raw">'


def _slurp(path: str) -> bytes:
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        return f.read()


def _attr(header: str, element: str, name: str) -> str:
    m = re.search(r"<%s\b[^>]*?\b%s=\"([^\"]*)\"" % (element, name), header)
    if not m:
        raise ValueError(f"VTI header has no {element}/{name}")
    return m.group(1)


def read_vti(path: str):
    """-> (data[x, y, z] float64, dims (Nx, Ny, Nz), origin, spacing) of a single-array image file."""
    raw = _slurp(path)
    cut = raw.find(_APPENDED_MARK)
    header = (raw if cut < 0 else raw[:cut]).decode("ascii", errors="replace")
    ext = [int(v) for v in _attr(header, "ImageData", "WholeExtent").split()]
    dims = (ext[1] - ext[0] + 1, ext[3] - ext[2] + 1, ext[5] - ext[4] + 1)
    origin = tuple(float(v) for v in _attr(header, "ImageData", "Origin").split())
    spacing = tuple(float(v) for v in _attr(header, "ImageData", "Spacing").split())
    count = dims[0] * dims[1] * dims[2]
    fmt = _attr(header, "DataArray", "format")
    if fmt == "ascii":
        start = header.index(">", header.index("<DataArray")) + 1
        stop = header.index("</DataArray>", start)
        flat = np.array(header[start:stop].split(), dtype=np.float64)
    elif fmt == "appended":
        if cut < 0:
            raise ValueError(f"{path}: appended format without an AppendedData block")
        vtype = _attr(header, "DataArray", "type")
        dtype = {"Float32": "<f4", "Float64": "<f8"}.get(vtype)
        if dtype is None:
            raise ValueError(f"{path}: unsupported appended type {vtype}")
        hdr_type = re.search(r'header_type="(\w+)"', header)
        hsize = 8 if (hdr_type and hdr_type.group(1) == "UInt64") else 4
        offset = int(_attr(header, "DataArray", "offset"))
        block = raw[cut + len(_APPENDED_MARK):]
        under = block.index(b"_") + 1 + offset
        nbytes = int(np.frombuffer(block[under:under + hsize], dtype="<u8" if hsize == 8 else "<u4")[0])
        if nbytes != count * np.dtype(dtype).itemsize:
            raise ValueError(f"{path}: block of {nbytes} bytes for {count} points of {vtype}")
        flat = np.frombuffer(block[under + hsize:under + hsize + nbytes], dtype=dtype).astype(np.float64)
    else:
        raise ValueError(f"{path}: unsupported DataArray format {fmt}")
    if flat.size != count:
        raise ValueError(f"{path}: {flat.size} values for {count} points")
    # the files hold x fastest, then y, then z
    return flat.reshape(dims[2], dims[1], dims[0]).transpose(2, 1, 0), dims, origin, spacing


def read_dims2d(path: str) -> dict:
    """dims.txt of the 2-D driver -> {"Nx": int, "Nz": int, "dx": float, "dz": float}."""
    out = {}
    for line in _slurp(path).decode().splitlines():
        key, val = line.split()
        out[key] = int(val) if key in ("Nx", "Nz") else float(val)
    return out


def read_csv2d(path: str, Nx: int, Nz: int):
    """-> (values[i, k], x[i], z[k]) of one "x,z,val" file (rows ordered i-major, k fastest)."""
    rows = np.loadtxt(_slurp(path).decode().splitlines(), delimiter=",", comments="#", ndmin=2)
    if rows.shape != (Nx * Nz, 3):
        raise ValueError(f"{path}: {rows.shape[0]} rows for a {Nx} x {Nz} grid")
    return rows[:, 2].reshape(Nx, Nz), rows[::Nz, 0].copy(), rows[:Nz, 1].copy()


def frame_step(path: str):
    """Time-step number in a frame's file name, or None."""
    m = _STEP.search(os.path.basename(path))
    return int(m.group(1)) if m else None


def list_frames(directory: str, prefix: str):
    """Frame files `<prefix>_tNNNNN.*` of a directory, ordered by step."""
    pat = re.compile(r"^%s_t\d+\.(?:vti|csv)(?:\.gz)?$" % re.escape(prefix))
    files = [os.path.join(directory, f) for f in os.listdir(directory) if pat.match(f)]
    return sorted(files, key=frame_step)


def energy_report(data: np.ndarray, rim: int = 20, core=None) -> dict:
    """Where the energy of one |E| frame sits: total of data**2, the share in the `rim` outermost cells
    of every face (each face counted, so corners count more than once, as the reference does) and the
    energy of the central block (default: the middle tenth of every axis)."""
    finite = bool(np.all(np.isfinite(data)))
    dens = np.square(data, dtype=np.float64)
    total = float(dens.sum())
    rim_sum = 0.0
    for axis in range(data.ndim):
        lo = [slice(None)] * data.ndim
        hi = [slice(None)] * data.ndim
        lo[axis], hi[axis] = slice(0, rim), slice(data.shape[axis] - rim, data.shape[axis])
        rim_sum += float(dens[tuple(lo)].sum() + dens[tuple(hi)].sum())
    if core is None:
        core = tuple(slice(n // 2 - max(n // 20, 1), n // 2 + max(n // 20, 1)) for n in data.shape)
    return {"finite": finite, "nonfinite": int(data.size - np.count_nonzero(np.isfinite(data))), "max": float(np.max(data)) if data.size else 0.0,
            "total": total, "centre": float(dens[core].sum()), "rim": rim_sum, "rim_share": rim_sum / total if total > 0 else 0.0}


def check_series(energies, low: float = 0.1, high: float = 100.0):
    """The reference's series verdict: last / first total energy must lie within [low, high].
    -> (ok, ratio); a series with fewer than two frames or a zero first frame is reported as ratio 0."""
    if len(energies) < 2 or not energies[0] > 0:
        return len(energies) < 2, 0.0
    ratio = float(energies[-1] / energies[0])
    return low <= ratio <= high, ratio
