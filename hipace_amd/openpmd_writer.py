"""openPMD output of the slice engine's diagnostics (the reference: diagnostics/OpenPMDWriter.cpp:55-450).

One file per output iteration holding the openPMD 1.1 hierarchy the reference writes through openPMD-api --
`/data/<iteration>/fields/<name>` (meshes: data order C, axes z y x, gridSpacing, gridGlobalOffset, position;
OpenPMDWriter.cpp:85-180) and `/data/<iteration>/particles/<beam>/{position, momentum, weighting, id, charge, mass,
positionOffset}` (:290-450) -- with the same record and attribute names, in up to three containers:

* `<prefix>/openpmd_%06d.h5`: HDF5, the reference's container and file name (`hipace.file_prefix/openpmd_%06T.h5`), written
  through the HDF5 C library by ctypes (`hipace_amd/h5lite.py`; the image has libhdf5 1.10 but neither h5py nor openPMD-api)
  in the layout of openPMD-api's HDF5 backend: groups = hierarchy, contiguous datasets, constant record components as groups
  with the attributes `value` and `shape`, attribute datatypes as openPMD-api writes them.  Checked here by reading it back
  (`read_hdf5`, `tests/openpmd_shim.py`) and with the library's own `h5dump`; openPMD-viewer / openPMD-api themselves are
  not installed, so "the reference's `checksumAPI.py` opens it unmodified" is what the layout is built for, not something
  this image can run.  Written when the library can be loaded (`hdf5=None`) or on request;
* `openpmd_%06d.npz`: numpy's npz (a zip of .npy arrays + one JSON document of attributes; the hierarchy paths are the array
  names) -- always written, needs nothing;
* with `json_too=True` `openpmd_%06d.json` in the layout of openPMD-api's JSON backend (plain text: groups as nested objects,
  every group's attributes under "attributes" as {"datatype", "value"}, datasets as {"attributes", "datatype", "data": nested
  lists}, constant record components as groups with the attributes "value" and "shape", "platform_byte_widths" at the
  root), restated from openPMD-api 0.14/0.15's JSONIOHandlerImpl.

`tests/openpmd_shim.py` offers the subset of openPMD-viewer's `OpenPMDTimeSeries` that the reference's checksum backend uses
(tests/checksum/backend/openpmd_backend.py:17-62) on the HDF5 files (or the npz container), so that backend's reductions
run on these files unchanged.
"""
import json
import os

import numpy as np

from . import h5lite

OPENPMD_VERSION = "1.1.0"


def write_iteration(prefix, iteration, time, dt, geometry, fields=None, beams=None, normalized=True, constants=None, json_too=False,
                    hdf5=None):
    """Write one openPMD iteration.

    geometry: dict(lo=(x, y, z), hi=(x, y, z)[, cells=(nx, ny, nz) of the simulation grid][, cell_volume=dx dy dz of it]) of the (possibly coarsened)
    diagnostic grid.
    fields: {name: array [nz, ny, nx]} (diag_type xyz, as hps_engine_field_diagnostic returns them).
    beams: {name: dict(x, y, z, ux, uy, uz, w, [id], charge, mass)} -- u = proper velocity / c as the engine keeps it
    (normalised units) or in m/s times gamma (SI), written as the reference does (OpenPMDWriter.cpp:376-385: momentum =
    u * mass * c with unitSI attributes).
    hdf5: also write `openpmd_%06d.h5` (None: if the HDF5 C library can be loaded, `h5lite.available()`).
    Returns the name of the npz file.
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
        # multipliers from the engine's units to SI as the reference writes them (OpenPMDWriter.cpp:344-384): the momentum
        # datasets hold u as the beam keeps it; their unitSI is mass (SI run) or mass * c (normalised run: openPMD-viewer
        # divides a momentum by mass * c to show u, the reference's "openpmd_viewer_u_workaround", on by default)
        m_e, q_e, c_SI, ep0 = 9.1093837015e-31, 1.602176634e-19, 299792458.0, 8.8541878128e-12
        mass = float(b["mass"])
        to_si = dict(pos=1.0, w=1.0, mom=mass, charge=1.0, mass=1.0)
        mom_unit = mass
        if normalized:
            kp_inv = c_SI / (q_e * np.sqrt(1.0 / (ep0 * m_e)))          # n_0 = 1 (OpenPMDWriter.cpp:353-357)
            lo, hi = geometry["lo"], geometry["hi"]
            # dx dy dz of the SIMULATION grid: given outright (`cell_volume`: the diagnostic grid may be a slice or a patch of the
            # box, whose lo / hi are not the simulation's), else from lo, hi and `cells`
            cell = geometry.get("cell_volume")
            if cell is None:
                cell = 1.0
                for ax, nn in zip(range(3), geometry.get("cells", (0, 0, 0))):
                    cell *= (hi[ax] - lo[ax]) / nn if nn else 1.0
            to_si = dict(pos=kp_inv, w=cell * kp_inv**3, mom=mass * m_e * c_SI, charge=q_e, mass=m_e)
            mom_unit = mass * c_SI
        ref = "HiPACE++_reference_unitSI"
        for comp in "xyz":
            arrays[f"{p}/position/{comp}"] = np.asarray(b[comp], dtype=np.float64)
            arrays[f"{p}/momentum/{comp}"] = np.asarray(b["u" + comp], dtype=np.float64)
            attrs[f"{p}/position/{comp}"] = {"unitSI": 1.0, ref: to_si["pos"]}
            attrs[f"{p}/momentum/{comp}"] = {"unitSI": mom_unit, ref: to_si["mom"]}
            attrs[f"{p}/positionOffset/{comp}"] = {"value": 0.0, "shape": [n], "unitSI": 1.0, ref: to_si["pos"]}
        arrays[f"{p}/weighting"] = np.asarray(b["w"], dtype=np.float64)
        arrays[f"{p}/id"] = np.asarray(b.get("id", np.arange(1, n + 1)), dtype=np.uint64)
        # constant record components (one value for all particles, OpenPMDWriter.cpp:340-347)
        # unitDimension = powers of (L, M, T, I, theta, N, J) (utils/IOUtil.cpp:105-140)
        attrs[f"{p}/charge"] = {"value": float(b["charge"]), "shape": [n], "unitSI": 1.0, "macroWeighted": 0, "weightingPower": 1.0,
                                "unitDimension": [0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 0.0], ref: to_si["charge"]}
        attrs[f"{p}/mass"] = {"value": mass, "shape": [n], "unitSI": 1.0, "macroWeighted": 0, "weightingPower": 1.0,
                              "unitDimension": [0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0], ref: to_si["mass"]}
        attrs[f"{p}/weighting"] = {"macroWeighted": 1, "weightingPower": 1.0, "unitSI": 1.0, ref: to_si["w"]}
        attrs[f"{p}/position"] = dict(macroWeighted=0, weightingPower=0.0, unitDimension=[1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
        attrs[f"{p}/positionOffset"] = dict(unitDimension=[1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
        attrs[f"{p}/momentum"] = dict(macroWeighted=0, weightingPower=1.0, unitDimension=[1.0, 1.0, -1.0, 0.0, 0.0, 0.0, 0.0])
        attrs[p] = {"HiPACE++_use_reference_unitSI": True, "normalized_units": bool(normalized)}
        if normalized:
            attrs[p]["HiPACE++_Plasma_Density"] = 1.0
    fn = os.path.join(prefix, "openpmd_%06d.npz" % iteration)
    np.savez(fn, __attrs__=np.frombuffer(json.dumps(attrs).encode(), dtype=np.uint8), **{k: v for k, v in arrays.items()})
    if json_too:
        with open(os.path.join(prefix, "openpmd_%06d.json" % iteration), "w") as f:
            json.dump(to_openpmd_json(attrs, arrays), f)
    if hdf5 is None:
        hdf5 = h5lite.available()
    if hdf5:
        write_hdf5(os.path.join(prefix, "openpmd_%06d.h5" % iteration), attrs, arrays)
    return fn


# ---- HDF5 container (the reference's: hipace.file_prefix/openpmd_%06T.h5 through openPMD-api) -----------------------------
_RECORDS = ("position", "momentum", "weighting", "charge", "mass", "positionOffset", "id")
_H5_KINDS = dict(UINT="UINT", FLOAT="FLOAT", ARR_DBL_7="DOUBLE", VEC_ULONG="ULONG")


def _complete(attrs, arrays):
    """the attributes the openPMD standard requires at record / record-component level added where the hierarchy above leaves
    them out (unitDimension, timeOffset; unitSI of every component)"""
    full = {p: dict(a) for p, a in attrs.items()}
    for path in list(full) + list(arrays):
        comp = path.rstrip("/").split("/")
        if (len(comp) >= 2 and comp[-2] == "fields") or (comp and comp[-1] in _RECORDS):
            a = full.setdefault(path, {})
            a.setdefault("unitDimension", [0.0] * 7)      # (the engine's units are the deck's; the reference writes its own table)
            a.setdefault("timeOffset", 0.0)
    for path in arrays:
        full.setdefault(path, {}).setdefault("unitSI", 1.0)
    return full


def write_hdf5(fn, attrs, arrays):
    """{hierarchy path: attribute dict}, {hierarchy path: array} -> one HDF5 file laid out as openPMD-api's HDF5 backend does
    (groups = hierarchy, contiguous datasets, constant record components as groups with `value` and `shape`, attribute
    datatypes as in `hipace_amd/h5lite.py`)."""
    full = _complete(attrs, arrays)
    with h5lite.File(fn, "w") as f:
        for path, arr in arrays.items():
            f.dataset(path, arr)
        for path in full:
            if path not in arrays and path != "/":
                f.group(path)
        for path, a in full.items():
            for k, v in a.items():
                f.attr(path, k, v, _H5_KINDS.get(_ATTR_TYPES.get(k)))
    return fn


def read_hdf5(fn):
    """The inverse (tests, tests/openpmd_shim.py): {path: array}, {path: {attribute: value}}."""
    arrays, attrs = {}, {}
    with h5lite.File(fn, "r") as f:
        attrs["/"] = f.attrs("/")
        for path, is_ds in f.walk("/"):
            attrs[path] = f.attrs(path)
            if is_ds:
                arrays[path] = f.read(path)
    return arrays, attrs


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


def write_engine_output(engine, prefix, iteration, time=0.0, beam_name="beam", beam=None, json_too=False, hdf5=None):
    """diagnostic.output of one step of a SliceEngine: the fields of its field diagnostic (set_field_diagnostic before
    the step) and, if given, the beam as (7, n) rows x y z ux uy uz w."""
    d = engine.deck
    fields = engine.field_diagnostic()
    beams = None
    if beam is not None:
        beams = {beam_name: dict(x=beam[0], y=beam[1], z=beam[2], ux=beam[3], uy=beam[4], uz=beam[5], w=beam[6],
                                 charge=d["beam_charge"], mass=d.get("beam_mass", 1.0) or 1.0)}
    lo, hi = d["lo"], d["hi"]
    if hasattr(engine, "field_diagnostic_geometry"):       # (a slice or a patch: the diagnostic grid's own box)
        _, lo, hi = engine.field_diagnostic_geometry()
    cell_volume = 1.0
    for ax, nn in enumerate((d["nx"], d["ny"], d["nz"])):          # the simulation's cells, whatever the diagnostic grid is
        cell_volume *= (d["hi"][ax] - d["lo"][ax]) / nn
    return write_iteration(prefix, iteration, time, d.get("dt", 0.0), dict(lo=lo, hi=hi, cells=(d["nx"], d["ny"], d["nz"]), cell_volume=cell_volume), fields, beams,
                           normalized=not d.get("si_units", 0), json_too=json_too, hdf5=hdf5)


# ---- beam.injection_type = from_file ------------------------------------------------------------------------------------
def read_beam(container, deck, iteration=0, species=None, plasma_density=0.0):
    """<beam>.injection_type = from_file (InitBeamFromFile, particles/beam/BeamParticleContainerInit.cpp:768-1090): the beam of
    one openPMD iteration as the (7, n) array x y z ux uy uz w that SliceEngine.set_beam_particles takes, in `deck`'s units.

    container: an HDF5 file of this writer or of any openPMD writer (`read_hdf5`), or the (arrays, attrs) pair itself.
    As the reference: the records are found by their unitDimension -- position L, proper velocity L/T or momentum M L/T, and
    for the weights `weighting`, else a charge (I T), else a mass (M) record (:797-885, :906-927); record components may be
    datasets or constant (`value` + `shape`); every value is scaled by its unitSI -- or, for a file that carries
    HiPACE++_use_reference_unitSI (one of the reference's, or this writer's), by HiPACE++_reference_unitSI (:1011-1040) --
    over the simulation's unit in SI: 1 m, c (or m c for a momentum), one particle (charge: q, mass: m) in an SI run;
    1/kp, and n0 dx dy dz / kp^3 particles, in a normalised run, n0 = plasma_density or the file's HiPACE++_Plasma_Density
    (:990-1009).  species None: the first one in the file."""
    arrays, attrs = read_hdf5(container) if isinstance(container, str) else container
    base = f"/data/{iteration}/particles"
    names = sorted({p[len(base) + 1:].split("/")[0] for p in list(arrays) + list(attrs) if p.startswith(base + "/")})
    if not names:
        raise ValueError(f"no particle species in iteration {iteration}")
    if species is None:
        species = names[0]
    if species not in names:
        raise ValueError(f"species {species!r} not in the file (it holds {names})")
    root = f"{base}/{species}"
    records = sorted({p[len(root) + 1:].split("/")[0] for p in list(arrays) + list(attrs) if p.startswith(root + "/")})

    def dims(rec):
        return tuple(float(v) for v in attrs.get(f"{root}/{rec}", {}).get("unitDimension", [0.0] * 7))

    def comps(rec):
        cs = sorted({p[len(root) + len(rec) + 2:].split("/")[0] for p in list(arrays) + list(attrs) if p.startswith(f"{root}/{rec}/")})
        return cs or [None]                  # None: the record is its own (scalar) component

    L, V, P = (1.0, 0, 0, 0, 0, 0, 0), (1.0, 0, -1.0, 0, 0, 0, 0), (1.0, 1.0, -1.0, 0, 0, 0, 0)
    M, Q, none = (0, 1.0, 0, 0, 0, 0, 0), (0, 0, 1.0, 1.0, 0, 0, 0), (0,) * 7
    name_r = name_u = name_w = None
    u_is_momentum = False
    kind = None
    for rec in records:
        dm = dims(rec)
        if dm == L and ("position" not in records or rec == "position"):
            name_r = rec
        elif dm == V:
            name_u, u_is_momentum = rec, False
        elif dm == P:
            name_u, u_is_momentum = rec, True
    for want, dm, k in (("weighting", none, "weighting"), (None, Q, "charge"), (None, M, "mass")):
        for rec in records:
            if name_w is None and dims(rec) == dm and (want is None or rec == want):
                name_w, kind = rec, k
    if name_r is None or name_u is None or name_w is None:
        raise ValueError("the file needs a position (L), a velocity or momentum (L/T, M L/T) and a weighting, charge or mass record")

    def component(rec, axis):
        for c in comps(rec):
            if c is None or c.lower() == axis or axis is None:
                path = f"{root}/{rec}" + (f"/{c}" if c is not None else "")
                a = attrs.get(path, {})
                if path in arrays:
                    return np.asarray(arrays[path], dtype=np.float64), a
                if "value" in a:
                    return np.full(int(np.atleast_1d(a["shape"])[0]), float(a["value"])), a
        raise ValueError(f"record {rec!r} has no {axis!r} component")

    si = deck.get("si_units", 0)
    c_SI, q_SI, m_SI, ep0 = 299792458.0, 1.602176634e-19, 9.1093837015e-31, 8.8541878128e-12
    mass, charge = deck.get("beam_mass", 1.0) or 1.0, deck["beam_charge"]
    m_e_sim, q_e_sim, c_sim = (m_SI, q_SI, c_SI) if si else (1.0, 1.0, 1.0)
    to_pos = 1.0
    to_mom = mass * (m_SI / m_e_sim) * c_SI if u_is_momentum else c_SI
    to_w = dict(weighting=1.0, charge=charge * (q_SI / q_e_sim), mass=mass * (m_SI / m_e_sim))[kind]
    sp_attrs = attrs.get(root, {})
    if not si:
        n0 = plasma_density or float(sp_attrs.get("HiPACE++_Plasma_Density", 0.0))
        if not n0:
            raise ValueError("a normalised run needs the plasma density of the external beam (plasma_density, or HiPACE++_Plasma_Density in the file)")
        kp_inv = c_SI / (q_SI * np.sqrt(n0 / (ep0 * m_SI)))
        cell = np.prod([(deck["hi"][k] - deck["lo"][k]) / deck[("nx", "ny", "nz")[k]] for k in range(3)])
        to_pos = kp_inv
        to_w *= n0 * cell * kp_inv ** 3
    restart = bool(sp_attrs.get("HiPACE++_use_reference_unitSI", False))

    def scaled(rec, axis, unit):
        v, a = component(rec, axis)
        return v * (float(a["HiPACE++_reference_unitSI"] if restart else a.get("unitSI", 1.0)) / unit)

    out = np.array([scaled(name_r, "x", to_pos), scaled(name_r, "y", to_pos), scaled(name_r, "z", to_pos),
                    scaled(name_u, "x", to_mom) * c_sim, scaled(name_u, "y", to_mom) * c_sim, scaled(name_u, "z", to_mom) * c_sim,
                    np.abs(scaled(name_w, None, to_w))])
    return out
