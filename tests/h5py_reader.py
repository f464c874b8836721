"""Read the openPMD HDF5 files of hipace_amd/openpmd_writer.py with h5py the way openPMD-viewer's h5py backend does, and print
the reference's checksums (tests/checksum/backend/openpmd_backend.py:40-62) as JSON.

Stand-alone on purpose: the image's system python has no h5py, its /opt/conda/bin/python3.9 has (h5py 3.3 on HDF5 1.10) --
tests/test_abi_and_oracle_ops.py runs this file with that interpreter when it is there.  What is restated from
openPMD-viewer (openpmd_timeseries/data_reader/h5py_reader: utilities.py, params_reader.py, field_reader.py,
particle_reader.py) is its sequence of h5py accesses: the root attributes `openPMD`, `basePath`, `meshesPath`,
`particlesPath`, `iterationEncoding` decoded from fixed-length strings; the iterations as the keys of `/data`; a mesh's
`geometry`, `axisLabels`, `gridSpacing`, `gridGlobalOffset`, `gridUnitSI`, `position`; a record component as a dataset, or --
constant -- a group whose attributes `value` and `shape` stand for `value * ones(shape)`; `unitSI` applied to what is read;
a momentum divided by mass * c (the viewer shows u = p / (m c)).

usage: python3.9 tests/h5py_reader.py <directory with openpmd_*.h5>"""
import glob
import json
import os
import re
import sys

import h5py
import numpy as np

C_SI = 299792458.0


def dec(v):
    if isinstance(v, bytes):
        return v.decode()
    if isinstance(v, np.ndarray) and v.dtype.kind in "SO":
        return [dec(x) for x in v]
    return v


def is_scalar_record(rec):
    return isinstance(rec, h5py.Dataset) or "value" in rec.attrs


def get_data(dset):
    if isinstance(dset, h5py.Group):             # constant record component
        data = dset.attrs["value"] * np.ones(tuple(int(x) for x in dset.attrs["shape"]))
    else:
        data = dset[...]
    u = dset.attrs["unitSI"]
    return data if u == 1.0 else data * u


def main(path):
    files = sorted(glob.glob(os.path.join(path, "openpmd_*.h5")))
    assert files, "no openpmd_*.h5 under " + path
    its = {}
    for fn in files:
        with h5py.File(fn, "r") as f:
            assert dec(f.attrs["openPMD"]).startswith("1.") and dec(f.attrs["iterationEncoding"]) == "fileBased"
            for k in f["/data"].keys():
                its[int(k)] = fn
    last = max(its)
    out = {"iterations": sorted(its), "lev=0": {}, "meta": {}}
    with h5py.File(its[last], "r") as f:
        bpath = f[dec(f.attrs["basePath"]).replace("%T", str(last))]
        out["meta"]["time"] = float(bpath.attrs["time"] * bpath.attrs["timeUnitSI"])
        meshes = bpath[dec(f.attrs["meshesPath"])]
        for name in meshes.keys():
            field = meshes[name]
            assert is_scalar_record(field) and dec(field.attrs["geometry"]) == "cartesian"
            out["meta"][name] = dict(axisLabels=dec(field.attrs["axisLabels"]), shape=list(field.shape), dataOrder=dec(field.attrs["dataOrder"]),
                                     gridSpacing=[float(x) for x in field.attrs["gridSpacing"]],
                                     gridGlobalOffset=[float(x) * float(field.attrs["gridUnitSI"]) for x in field.attrs["gridGlobalOffset"]],
                                     position=[float(x) for x in field.attrs["position"]])
            out["lev=0"][name] = float(np.sum(np.abs(get_data(field))))
        pp = dec(f.attrs["particlesPath"]).strip("/")
        if pp in bpath:
            for sname in bpath[pp].keys():
                species = bpath[pp][sname]
                comps = []
                for rname in species.keys():
                    if rname == "particlePatches":
                        continue
                    rec = species[rname]
                    comps += [rname] if is_scalar_record(rec) else [rname + "/" + c for c in rec.keys()]
                cs = {}
                for comp in comps:
                    if comp.startswith("positionOffset"):
                        continue
                    short = {"position/x": "x", "position/y": "y", "position/z": "z", "momentum/x": "ux", "momentum/y": "uy", "momentum/z": "uz",
                             "weighting": "w"}.get(comp, comp)
                    data = get_data(species[comp])
                    if short in ("x", "y", "z"):
                        data = data + get_data(species["positionOffset/" + short])
                    if short in ("ux", "uy", "uz"):
                        data = data / (get_data(species["mass"]) * C_SI)
                    c = np.sum(np.abs(data))
                    cs[short] = int(c) if isinstance(c, (np.int64, np.uint64)) else float(c)
                out[sname] = cs
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1])
