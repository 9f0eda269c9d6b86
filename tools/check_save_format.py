"""Build-container check of the -save writer (rmhd_case_save) against the mesh files the reference
itself reads: /root/reference/data/{periodic-cube,cube01_hex}.mesh (MFEM mesh v1.0).

For each lattice the reference file and the file this repository writes for the same mesh at -rs 0 are parsed into
numbering-independent canonical forms --
  geometry : every element as the sorted tuple of its 8 corner coordinates;
  topology : every pair of elements that share vertices, as (element, element, number of shared vertices), elements
             named by their geometry (captures the periodic identification);
  grammar  : the sequence of section keywords, the element / boundary line shapes and the counts --
and compared.  The canonical forms' hashes (DATA, not file text) go to tests/golden/save_format.json, where
tests/test_case_host.py::test_save_matches_reference_mesh_files re-derives them from the writer without the reference.

    python tools/check_save_format.py            (needs /root/reference; rewrites the fixture)
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KEYWORDS = ("dimension", "elements", "boundary", "vertices", "nodes", "FiniteElementSpace", "FiniteElementCollection:", "VDim:", "Ordering:")


def parse_mesh(path):
    """MFEM mesh v1.0 text -> dict(elements [ne][8] vertex ids, nb, nv, fec, vdim, ordering, nodes flat array, keywords)"""
    toks = []
    for line in open(path):
        line = line.split("#")[0]
        toks += line.split()
    assert toks[:3] == ["MFEM", "mesh", "v1.0"], toks[:3]
    kw = [t for t in toks if t in KEYWORDS]
    i = toks.index("dimension")
    dim = int(toks[i + 1])
    i = toks.index("elements")
    ne = int(toks[i + 1])
    el = np.array(toks[i + 2:i + 2 + 10 * ne], dtype=np.int64).reshape(ne, 10)
    assert (el[:, 1] == 5).all()  # Geometry::CUBE
    i = toks.index("boundary")
    nb = int(toks[i + 1])
    if nb:
        bd = np.array(toks[i + 2:i + 2 + 6 * nb], dtype=np.int64).reshape(nb, 6)
        assert (bd[:, 1] == 3).all()  # Geometry::SQUARE
    i = toks.index("vertices")
    nv = int(toks[i + 1])
    i = toks.index("nodes")
    assert toks[i + 1] == "FiniteElementSpace"
    fec = toks[toks.index("FiniteElementCollection:") + 1]
    vdim = int(toks[toks.index("VDim:") + 1])
    io = toks.index("Ordering:")
    ordering = int(toks[io + 1])
    nodes = np.array(toks[io + 2:], dtype=np.float64)
    return dict(dim=dim, elements=el[:, 2:], nb=nb, nv=nv, fec=fec, vdim=vdim, ordering=ordering, nodes=nodes, keywords=kw)


def corners(m):
    """[ne][8][3] corner coordinates of every element"""
    ne = len(m["elements"])
    if m["fec"] == "Linear" or m["fec"].startswith("H1_3D_P1"):
        xyz = m["nodes"].reshape(3, m["nv"]).T if m["ordering"] == 0 else m["nodes"].reshape(m["nv"], 3)
        return xyz[m["elements"]]
    if m["fec"].startswith("L2_T1_3D_P"):
        order = int(m["fec"].split("P")[-1])
        n1 = order + 1
        nd = n1**3
        if m["ordering"] == 1:
            x = m["nodes"].reshape(ne, nd, 3)
        else:
            x = m["nodes"].reshape(3, ne, nd).transpose(1, 2, 0)
        idx = [ax + n1 * (ay + n1 * az) for az in (0, order) for ay in (0, order) for ax in (0, order)]
        return x[:, idx, :]
    raise ValueError(m["fec"])


def canonical(m):
    c = np.round(corners(m), 6)
    names = [tuple(sorted(map(tuple, ce.tolist()))) for ce in c]
    geometry = sorted(names)
    assert len(set(names)) == len(names)
    rank = {n: i for i, n in enumerate(geometry)}
    el = m["elements"]
    topo = set()
    for a in range(len(el)):
        for b in range(len(el)):
            if a != b:
                shared = len(set(el[a].tolist()) & set(el[b].tolist()))
                if shared:
                    topo.add((rank[names[a]], rank[names[b]], shared))
    return geometry, sorted(topo)


def digest(obj):
    return hashlib.sha256(json.dumps(obj, sort_keys=True).encode()).hexdigest()


def our_file(mesh):
    from remhos_amd.case import Case, load_host_library, make_config

    lib = load_host_library()
    c = Case(lib, make_config(mesh, 0, 2, 10 if mesh == "cube01_hex" else 0, -1.0, 0.5))
    d = tempfile.mkdtemp()
    path = os.path.join(d, "mesh.mesh")
    c.save(0.0, None, path)
    return path


def summary(mesh, m):
    g, t = canonical(m)
    return {"elements": len(m["elements"]), "vertices": m["nv"], "keywords": m["keywords"], "geometry_sha256": digest(g),
            "topology_sha256": digest(t), "shared_vertex_pairs": len(t)}


def main():
    ref_dir = "/root/reference/data"
    if not os.path.isdir(ref_dir):
        raise SystemExit("needs the reference checkout (build container only)")
    out = {"_what": "Numbering-independent canonical forms (hashes) of the meshes the reference reads, derived from "
                    "/root/reference/data/*.mesh by tools/check_save_format.py; the -save writer must reproduce them at -rs 0.  "
                    "Known, intended differences of the written files: nodes are order-2 L2 fields (what remhos writes after "
                    "SetCurvature(2), remhos.cpp:509-527), and the periodic mesh is written without the 54 interior 'boundary' "
                    "quads that data/periodic-cube.mesh inherits from the mesh it was made periodic from."}
    for mesh in ("periodic-cube", "cube01_hex"):
        ref = parse_mesh(os.path.join(ref_dir, mesh + ".mesh"))
        ours = parse_mesh(our_file(mesh))
        sr, so = summary(mesh, ref), summary(mesh, ours)
        print(mesh, "reference:", sr)
        print(mesh, "written  :", so)
        assert sr["geometry_sha256"] == so["geometry_sha256"], "corner coordinates differ"
        assert sr["topology_sha256"] == so["topology_sha256"], "vertex sharing (topology / periodic identification) differs"
        assert sr["elements"] == so["elements"] and sr["vertices"] == so["vertices"]
        assert sr["keywords"] == so["keywords"], "section grammar differs"
        if mesh == "cube01_hex":
            assert ref["nb"] == ours["nb"] == 24
        out[mesh] = {**sr, "reference_file": f"data/{mesh}.mesh", "reference_boundary_elements": ref["nb"], "written_boundary_elements": ours["nb"],
                     "reference_nodes": ref["fec"], "written_nodes": ours["fec"]}
    path = os.path.join(ROOT, "tests", "golden", "save_format.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
