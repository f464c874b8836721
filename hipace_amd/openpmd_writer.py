"""openPMD output of the slice engine's diagnostics (the reference: diagnostics/OpenPMDWriter.cpp:55-450).

One file per output iteration, `<prefix>/openpmd_%06d.npz`, holding the openPMD 1.1 hierarchy the reference writes
through openPMD-api -- `/data/<iteration>/fields/<name>` (meshes: data order C, axes z y x, gridSpacing,
gridGlobalOffset, position; OpenPMDWriter.cpp:85-180) and `/data/<iteration>/particles/<beam>/{position, momentum,
weighting, id, charge, mass, positionOffset}` (:290-450) -- with the same record and attribute names.  The CONTAINER is
numpy's npz (a zip of .npy arrays + one JSON document of attributes) instead of HDF5 / ADIOS2, because this image has no
HDF5 library and no openPMD-api; the hierarchy paths are the array names.  With `json_too=True` the same iteration is also
written as `openpmd_%06d.json` in the layout of openPMD-api's JSON backend (plain text: groups as nested objects, every
group's attributes under "attributes" as {"datatype", "value"}, datasets as {"attributes", "datatype", "data": nested
lists}, constant record components as groups with the attributes "value" and "shape", "platform_byte_widths" at the root)
-- the one openPMD container that needs no library to write.  That layout is restated from openPMD-api 0.14/0.15's
JSONIOHandlerImpl; it could not be opened with openPMD-api here (not installed), `tests/test_abi_and_oracle_ops.py` only
checks it against itself.  `tests/openpmd_shim.py` offers the
subset of openPMD-viewer's `OpenPMDTimeSeries` that the reference's checksum backend uses
(tests/checksum/backend/openpmd_backend.py:17-62), so that backend's reductions run on these files unchanged.
"""
import json
import os

import numpy as np

OPENPMD_VERSION = "1.1.0"


def write_iteration(prefix, iteration, time, dt, geometry, fields=None, beams=None, normalized=True, constants=None, json_too=False):
    """Write one openPMD iteration.

    geometry: dict(lo=(x, y, z), hi=(x, y, z)) of the (possibly coarsened) diagnostic grid.
    fields: {name: array [nz, ny, nx]} (diag_type xyz, as hps_engine_field_diagnostic returns them).
    beams: {name: dict(x, y, z, ux, uy, uz, w, [id], charge, mass)} -- u = proper velocity / c as the engine keeps it
    (normalised units) or in m/s times gamma (SI), written as the reference does (OpenPMDWriter.cpp:376-385: momentum =
    u * mass * c with unitSI attributes).
    Returns the file name.
    """
    os.makedirs(prefix, exist_ok=True)
    base = f"/data/{iteration}"
    arrays, attrs = {}, {}
    attrs["/"] = dict(openPMD=OPENPMD_VERSION, openPMDextension=0, basePath="/data/%T/", meshesPath="fields/",
                      particlesPath="particles/", iterationEncoding="fileBased", iterationFormat="openpmd_%06T",
                      software="hpslice (hipace_amd)", softwareVersion="0.2")
    attrs[base] = dict(time=float(time), dt=float(dt), timeUnitSI=1.0)
    for name, arr in (fields or {}).items():
        a = np.ascontiguousarray(arr)
        nz, ny, nx = a.shape
        lo, hi = geometry["lo"], geometry["hi"]
        path = f"{base}/fields/{name}"
        arrays[path] = a
        attrs[path] = dict(geometry="cartesian", dataOrder="C", axisLabels=["z", "y", "x"],
                           gridSpacing=[(hi[2] - lo[2]) / nz, (hi[1] - lo[1]) / ny, (hi[0] - lo[0]) / nx],
                           gridGlobalOffset=[lo[2], lo[1], lo[0]], gridUnitSI=1.0, position=[0.5, 0.5, 0.5],
                           unitSI=1.0, timeOffset=0.0)
    for bname, b in (beams or {}).items():
        p = f"{base}/particles/{bname}"
        n = len(b["x"])
        for comp in "xyz":
            arrays[f"{p}/position/{comp}"] = np.asarray(b[comp], dtype=np.float64)
            arrays[f"{p}/momentum/{comp}"] = np.asarray(b["u" + comp], dtype=np.float64)
            attrs[f"{p}/positionOffset/{comp}"] = dict(value=0.0, shape=[n], unitSI=1.0)
        arrays[f"{p}/weighting"] = np.asarray(b["w"], dtype=np.float64)
        arrays[f"{p}/id"] = np.asarray(b.get("id", np.arange(1, n + 1)), dtype=np.uint64)
        # constant record components (one value for all particles, OpenPMDWriter.cpp:340-347)
        attrs[f"{p}/charge"] = dict(value=float(b["charge"]), shape=[n], unitSI=1.0, macroWeighted=0, weightingPower=1.0)
        attrs[f"{p}/mass"] = dict(value=float(b["mass"]), shape=[n], unitSI=1.0, macroWeighted=0, weightingPower=1.0)
        attrs[f"{p}/weighting"] = dict(macroWeighted=1, weightingPower=1.0, unitSI=1.0)
        attrs[f"{p}/position"] = dict(macroWeighted=0, weightingPower=0.0, unitSI=1.0)
        attrs[f"{p}/momentum"] = dict(macroWeighted=0, weightingPower=1.0, unitSI=1.0)
        attrs[p] = {"HiPACE++_use_reference_unitSI": True, "normalized_units": bool(normalized)}
    fn = os.path.join(prefix, "openpmd_%06d.npz" % iteration)
    np.savez(fn, __attrs__=np.frombuffer(json.dumps(attrs).encode(), dtype=np.uint8), **{k: v for k, v in arrays.items()})
    if json_too:
        with open(os.path.join(prefix, "openpmd_%06d.json" % iteration), "w") as f:
            json.dump(to_openpmd_json(attrs, arrays), f)
    return fn


# ---- openPMD-api JSON backend layout -------------------------------------------------------------------------------------
_BYTE_WIDTHS = dict(BOOL=1, CHAR=1, DOUBLE=8, FLOAT=4, INT=4, LONG=8, LONGLONG=8, LONG_DOUBLE=16, SHORT=2, UCHAR=1, UINT=4,
                    ULONG=8, ULONGLONG=8, USHORT=2)
_ATTR_TYPES = dict(openPMDextension="UINT", timeOffset="FLOAT", unitDimension="ARR_DBL_7", shape="VEC_ULONG", macroWeighted="UINT")


def _json_attr(name, v):
    if name in _ATTR_TYPES:
        return {"datatype": _ATTR_TYPES[name], "value": v}
    if isinstance(v, bool):
        return {"datatype": "BOOL", "value": v}
    if isinstance(v, str):
        return {"datatype": "STRING", "value": v}
    if isinstance(v, int):
        return {"datatype": "INT", "value": v}
    if isinstance(v, float):
        return {"datatype": "DOUBLE", "value": v}
    if isinstance(v, (list, tuple)):
        if v and isinstance(v[0], str):
            return {"datatype": "VEC_STRING", "value": list(v)}
        return {"datatype": "VEC_DOUBLE", "value": [float(x) for x in v]}
    raise TypeError(f"openPMD attribute {name}: {type(v)}")


def to_openpmd_json(attrs, arrays):
    """{hierarchy path: attribute dict}, {hierarchy path: array} -> the document openPMD-api's JSON backend keeps"""
    root = {}

    def node(path):
        n = root
        for part in [q for q in path.split("/") if q]:
            n = n.setdefault(part, {})
        return n

    for path, a in attrs.items():
        n = node(path)
        at = n.setdefault("attributes", {})
        for k, v in a.items():
            at[k] = _json_attr(k, v)
        comp = path.rstrip("/").split("/")
        if len(comp) >= 2 and comp[-2] in ("fields",) or (comp and comp[-1] in ("position", "momentum", "weighting", "charge", "mass", "positionOffset", "id")):
            at.setdefault("unitDimension", _json_attr("unitDimension", [0.0] * 7))      # (record level; the engine's units are the deck's)
            at.setdefault("timeOffset", _json_attr("timeOffset", 0.0))
    for path, arr in arrays.items():
        n = node(path)
        n["datatype"] = "ULONG" if arr.dtype == np.uint64 else "DOUBLE"
        n["data"] = arr.tolist()
        n.setdefault("attributes", {}).setdefault("unitSI", _json_attr("unitSI", 1.0))
    root["platform_byte_widths"] = dict(_BYTE_WIDTHS)
    return root


def read_openpmd_json(fn):
    """The inverse (tests): {path: array}, {path: {attribute: value}} of a document written by `to_openpmd_json`."""
    doc = json.load(open(fn))
    arrays, attrs = {}, {}

    def walk(n, path):
        if "attributes" in n:
            attrs[path or "/"] = {k: v["value"] for k, v in n["attributes"].items()}
        if "data" in n and "datatype" in n:
            arrays[path] = np.array(n["data"], dtype=np.uint64 if n["datatype"] == "ULONG" else np.float64)
            return
        for k, c in n.items():
            if k not in ("attributes", "platform_byte_widths") and isinstance(c, dict):
                walk(c, f"{path}/{k}")

    walk(doc, "")
    return arrays, attrs


def write_engine_output(engine, prefix, iteration, time=0.0, beam_name="beam", beam=None, json_too=False):
    """diagnostic.output of one step of a SliceEngine: the fields of its field diagnostic (set_field_diagnostic before
    the step) and, if given, the beam as (7, n) rows x y z ux uy uz w."""
    d = engine.deck
    fields = engine.field_diagnostic()
    beams = None
    if beam is not None:
        beams = {beam_name: dict(x=beam[0], y=beam[1], z=beam[2], ux=beam[3], uy=beam[4], uz=beam[5], w=beam[6],
                                 charge=d["beam_charge"], mass=d.get("beam_mass", 1.0) or 1.0)}
    return write_iteration(prefix, iteration, time, d.get("dt", 0.0), dict(lo=d["lo"], hi=d["hi"]), fields, beams,
                           normalized=not d.get("si_units", 0), json_too=json_too)
