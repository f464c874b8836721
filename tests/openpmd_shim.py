"""The subset of openPMD-viewer's OpenPMDTimeSeries that the reference's checksum backend uses
(/root/reference/tests/checksum/backend/openpmd_backend.py:17-62), reading the HDF5 files of hipace_amd/openpmd_writer.py
(through hipace_amd/h5lite.py: the HDF5 C library by ctypes; the image has neither h5py nor openPMD-viewer) or its npz
container, and that backend's two reductions restated on top of it."""
import glob
import json
import os
import re

import numpy as np


class _Arrays(dict):
    """{path: array} with numpy's NpzFile interface (`files`)"""
    @property
    def files(self):
        return list(self)


class OpenPMDTimeSeries:
    def __init__(self, path, container=None):
        """container: "h5" | "npz" | None = the HDF5 files if the writer has made them (the reference's container), else npz"""
        h5 = sorted(glob.glob(os.path.join(path, "openpmd_*.h5")))
        self.container = container or ("h5" if h5 else "npz")
        files = h5 if self.container == "h5" else sorted(glob.glob(os.path.join(path, "openpmd_*.npz")))
        assert files, f"no openPMD iterations under {path}"
        self._files = {int(re.search(r"openpmd_(\d+)\.(npz|h5)$", f).group(1)): f for f in files}
        self.iterations = np.array(sorted(self._files))
        z, attrs = self._open(self.iterations[-1])
        base = f"/data/{self.iterations[-1]}"
        self.avail_fields = sorted({k[len(base) + 8:] for k in z.files if k.startswith(base + "/fields/")})
        species = sorted({k[len(base) + 11:].split("/")[0] for k in list(z.files) + list(attrs) if k.startswith(base + "/particles/")})
        self.avail_species = species or None
        self.avail_record_components = {}
        for s in species:
            comps = set()
            p = f"{base}/particles/{s}/"
            for k in list(z.files) + [a for a in attrs if "value" in attrs[a]]:
                if not k.startswith(p):
                    continue
                rec = k[len(p):].split("/")
                if rec[0] == "position":
                    comps.add(rec[1])
                elif rec[0] == "momentum":
                    comps.add("u" + rec[1])
                elif rec[0] == "weighting":
                    comps.add("w")
                elif rec[0] in ("id", "charge", "mass"):
                    comps.add(rec[0])
            self.avail_record_components[s] = sorted(comps)

    def _open(self, iteration):
        if self.container == "h5":
            from hipace_amd import openpmd_writer as W
            arrays, attrs = W.read_hdf5(self._files[int(iteration)])
            return _Arrays(arrays), attrs
        z = np.load(self._files[int(iteration)])
        return z, json.loads(bytes(z["__attrs__"]).decode())

    def get_field(self, field, iteration):
        z, attrs = self._open(iteration)
        path = f"/data/{int(iteration)}/fields/{field}"
        return z[path], attrs[path]

    def get_particle(self, var_list, species, iteration):
        z, attrs = self._open(iteration)
        p = f"/data/{int(iteration)}/particles/{species}"
        out = []
        for v in var_list:
            if v in ("x", "y", "z"):
                out.append(z[f"{p}/position/{v}"])
            elif v in ("ux", "uy", "uz"):
                out.append(z[f"{p}/momentum/{v[1]}"])
            elif v == "w":
                out.append(z[f"{p}/weighting"])
            elif v == "id":
                out.append(z[f"{p}/id"])
            else:       # constant record component
                a = attrs[f"{p}/{v}"]
                out.append(np.full(a["shape"][0], a["value"]))
        return out


def checksums(path):
    """openpmd_backend.py:40-62 -- sum |Q| of every field and of every record component of every species of the last
    iteration, in the layout of the reference's benchmark JSON files."""
    ts = OpenPMDTimeSeries(path)
    out = {"lev=0": {}}
    for f in ts.avail_fields:
        Q = ts.get_field(field=f, iteration=ts.iterations[-1])[0]
        out["lev=0"][f] = float(np.sum(np.abs(Q)))
    for s in ts.avail_species or []:
        out[s] = {}
        for a in ts.avail_record_components[s]:
            Q = ts.get_particle(var_list=[a], species=s, iteration=ts.iterations[-1])
            c = np.sum(np.abs(Q))
            out[s][a] = int(c) if isinstance(c, (np.int64, np.uint64)) else float(c)
    return out


H5PY_PYTHON = os.environ.get("HPS_H5PY_PYTHON", "/opt/conda/bin/python3.9")      # an interpreter that has h5py (the system one has not)


def h5py_checksums(path):
    """The same reductions by h5py itself: tests/h5py_reader.py (openPMD-viewer's sequence of h5py accesses) run with the
    image's conda interpreter.  None where there is no such interpreter."""
    import subprocess
    if not os.path.exists(H5PY_PYTHON):
        return None
    probe = subprocess.run([H5PY_PYTHON, "-c", "import h5py"], capture_output=True, cwd="/tmp")
    if probe.returncode != 0:
        return None
    r = subprocess.run([H5PY_PYTHON, os.path.join(os.path.dirname(os.path.abspath(__file__)), "h5py_reader.py"), path],
                       capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout)
