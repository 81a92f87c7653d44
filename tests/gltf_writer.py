"""Minimal GLB (binary glTF 2.0) writer for the loader tests: builds small files that exercise what the reference's
LoadGLTF (src/loader.cpp:13-401) reads -- float POSITION / NORMAL / TEXCOORD_0, uint16 indices, pbrMetallicRoughness,
KHR_materials_emissive_strength, node TRS / matrix / camera nodes, interleaved buffer views, several primitives per
mesh.  Own code; the files it writes are test data."""
import json
import struct

import numpy as np

FLOAT, USHORT = 5126, 5123


class Glb:
    def __init__(self):
        self.bin = bytearray()
        self.j = {"asset": {"version": "2.0"}, "buffers": [{}], "bufferViews": [], "accessors": [], "meshes": [], "nodes": [],
                  "materials": [], "scenes": [{"nodes": []}], "scene": 0}

    def view(self, data: bytes, stride=None, target=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        v = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        if target:
            v["target"] = target
        self.bin += data
        self.j["bufferViews"].append(v)
        return len(self.j["bufferViews"]) - 1

    def accessor(self, view, ctype, count, atype, offset=0, minmax=None):
        a = {"bufferView": view, "componentType": ctype, "count": count, "type": atype}
        if offset:
            a["byteOffset"] = offset
        if minmax is not None:
            a["min"], a["max"] = [float(x) for x in minmax[0]], [float(x) for x in minmax[1]]
        self.j["accessors"].append(a)
        return len(self.j["accessors"]) - 1

    def vec_accessor(self, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        atype = {2: "VEC2", 3: "VEC3"}[arr.shape[1]]
        mm = (arr.min(0), arr.max(0)) if arr.shape[1] == 3 else None
        return self.accessor(self.view(arr.tobytes(), target=34962), FLOAT, arr.shape[0], atype, minmax=mm)

    def index_accessor(self, idx):
        idx = np.ascontiguousarray(idx, np.uint16)
        return self.accessor(self.view(idx.tobytes(), target=34963), USHORT, idx.size, "SCALAR")

    def material(self, base=None, metallic=None, roughness=None, emissive=None, strength=None, name=None):
        m = {}
        if name:
            m["name"] = name
        if base is not None or metallic is not None or roughness is not None:
            p = {}
            if base is not None:
                p["baseColorFactor"] = [float(x) for x in base]
            if metallic is not None:
                p["metallicFactor"] = float(metallic)
            if roughness is not None:
                p["roughnessFactor"] = float(roughness)
            m["pbrMetallicRoughness"] = p
        if emissive is not None:
            m["emissiveFactor"] = [float(x) for x in emissive]
        if strength is not None:
            m["extensions"] = {"KHR_materials_emissive_strength": {"emissiveStrength": float(strength)}}
            self.j.setdefault("extensionsUsed", [])
            if "KHR_materials_emissive_strength" not in self.j["extensionsUsed"]:
                self.j["extensionsUsed"].append("KHR_materials_emissive_strength")
        self.j["materials"].append(m)
        return len(self.j["materials"]) - 1

    def primitive(self, pos, nrm, idx, uv=None, material=None):
        attrs = {"POSITION": self.vec_accessor(pos), "NORMAL": self.vec_accessor(nrm)}
        if uv is not None:
            attrs["TEXCOORD_0"] = self.vec_accessor(uv)
        p = {"attributes": attrs, "indices": self.index_accessor(idx), "mode": 4}
        if material is not None:
            p["material"] = material
        return p

    def mesh(self, primitives):
        self.j["meshes"].append({"primitives": primitives})
        return len(self.j["meshes"]) - 1

    def node(self, mesh=None, translation=None, rotation=None, scale=None, matrix=None, camera=None):
        n = {}
        if mesh is not None:
            n["mesh"] = mesh
        if translation is not None:
            n["translation"] = [float(x) for x in translation]
        if rotation is not None:
            n["rotation"] = [float(x) for x in rotation]
        if scale is not None:
            n["scale"] = [float(x) for x in scale]
        if matrix is not None:
            n["matrix"] = [float(x) for x in matrix]
        if camera is not None:
            self.j.setdefault("cameras", [{"type": "perspective", "perspective": {"yfov": 0.8, "znear": 0.1}}])
            n["camera"] = camera
        self.j["nodes"].append(n)
        self.j["scenes"][0]["nodes"].append(len(self.j["nodes"]) - 1)
        return len(self.j["nodes"]) - 1

    def tobytes(self):
        j = dict(self.j)
        for k in ("materials", "nodes", "meshes"):
            if not j[k]:
                del j[k]
        while len(self.bin) % 4:
            self.bin.append(0)
        j["buffers"] = [{"byteLength": len(self.bin)}]
        js = json.dumps(j, separators=(",", ":")).encode()
        js += b" " * (-len(js) % 4)
        total = 12 + 8 + len(js) + 8 + len(self.bin)
        return (struct.pack("<III", 0x46546C67, 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js +
                struct.pack("<II", len(self.bin), 0x004E4942) + bytes(self.bin))


def box(center, half, seed=0):
    """24-vertex box (flat normals, per-face uvs) -> pos [24,3], nrm [24,3], uv [24,2], idx [36]."""
    cx, cy, cz = center
    hx, hy, hz = half
    faces = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 0, 1), (0, 1, 0)), ((0, 1, 0), (0, 0, 1), (1, 0, 0)),
             ((0, -1, 0), (1, 0, 0), (0, 0, 1)), ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (0, 1, 0), (1, 0, 0))]
    pos, nrm, uv, idx = [], [], [], []
    h = np.array([hx, hy, hz])
    for n, a, b in faces:
        n, a, b = np.array(n, float), np.array(a, float), np.array(b, float)
        base = len(pos)
        for (s, t) in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            pos.append(np.array(center) + (n + s * a + t * b) * h)
            nrm.append(n)
            uv.append(((s + 1) / 2, (t + 1) / 2))
        idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    return np.array(pos, np.float32), np.array(nrm, np.float32), np.array(uv, np.float32), np.array(idx, np.uint16)


def smooth_blob(n_lat, n_lon, radius, center, seed):
    """UV sphere with smooth (non-axis-aligned) normals and a seeded radial wobble."""
    rng = np.random.default_rng(seed)
    pos, nrm, uv, idx = [], [], [], []
    for i in range(n_lat + 1):
        th = np.pi * i / n_lat
        for j in range(n_lon + 1):
            ph = 2 * np.pi * j / n_lon
            d = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
            r = radius * (1 + 0.1 * np.sin(3 * th) * np.cos(2 * ph))
            pos.append(np.array(center) + r * d)
            nn = d + 0.05 * rng.normal(size=3)
            nrm.append(nn / np.linalg.norm(nn))
            uv.append((j / n_lon, i / n_lat))
    for i in range(n_lat):
        for j in range(n_lon):
            a, b = i * (n_lon + 1) + j, (i + 1) * (n_lon + 1) + j
            idx += [a, b, a + 1, a + 1, b, b + 1]
    return np.array(pos, np.float32), np.array(nrm, np.float32), np.array(uv, np.float32), np.array(idx, np.uint16)


def quat(axis, angle):
    axis = np.array(axis, float) / np.linalg.norm(axis)
    s = np.sin(angle / 2)
    return [axis[0] * s, axis[1] * s, axis[2] * s, np.cos(angle / 2)]


def build_cases():
    """name -> GLB bytes.  Every file is texture-less (the reference loader needs a GL context once textures exist)."""
    out = {}
    # 1. one mesh, one primitive, rough dielectric (-> Oren-Nayar, roughness * 0.35), no node transform at all
    g = Glb()
    p, n, uv, i = box((0, 0, 0), (1, 0.5, 0.25))
    m = g.material(base=(0.8, 0.4, 0.2, 1.0), metallic=0.0, roughness=0.5)
    g.node(mesh=g.mesh([g.primitive(p, n, i, uv, m)]))
    out["single"] = g.tobytes()
    # 2. one mesh with three primitives: the per-mesh arrays are appended per primitive and the de-index loop re-walks the
    #    whole index array (loader.cpp:62-65, 304-345) -- later primitives re-emit the earlier ones' triangles
    g = Glb()
    prims = []
    for k, (c, mat) in enumerate([((0, 0, 0), g.material(base=(1, 1, 1, 1), metallic=0.0, roughness=0.0)),
                                  ((3, 0, 0), g.material(base=(0.9, 0.8, 0.1, 1), metallic=1.0, roughness=0.3)),
                                  ((0, 3, 0), g.material(emissive=(1.0, 0.5, 0.25), strength=12.0))]):
        p, n, uv, i = box(c, (0.5 + 0.1 * k, 0.5, 0.5))
        prims.append(g.primitive(p, n, i, uv if k != 1 else None, mat))
    g.node(mesh=g.mesh(prims))
    out["multiprim"] = g.tobytes()
    # 3. two meshes, nodes with scale / rotation / translation (y <-> z swapped by the loader, loader.cpp:375,388), a matrix
    #    node, a camera node (skipped) -- every non-camera node's transform is baked into ALL triangles (loader.cpp:349-393)
    g = Glb()
    p, n, uv, i = smooth_blob(6, 8, 0.7, (0, 0, 0), 3)
    m0 = g.mesh([g.primitive(p, n, i, uv, g.material(base=(0.2, 0.6, 0.9, 1), metallic=0.0, roughness=1.0))])
    p, n, uv, i = box((1, 2, 3), (0.3, 0.6, 0.9))
    m1 = g.mesh([g.primitive(p, n, i, None, None)])   # no material: the loader's magenta default
    g.node(mesh=m0, scale=(1.5, 0.5, 2.0), rotation=quat((1, 2, 3), 0.7), translation=(0.25, -1.5, 4.0))
    g.node(camera=0, translation=(9, 9, 9))
    g.node(mesh=m1, matrix=(0.0, 0.0, -2.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.5, 0.0, 0.0, 0.0, 1.0, -2.0, 0.5, 1.0))
    g.node(mesh=m1, rotation=quat((0, 1, 0), -1.2))
    out["nodes"] = g.tobytes()
    # 4. interleaved vertex buffer (byteStride) and an index accessor with a byteOffset inside its view: the loader adds
    #    accessor->offset for attributes but NOT for indices (loader.cpp:96-97 vs 156-160), so it reads from the view's start
    g = Glb()
    p, n, uv, i = smooth_blob(4, 5, 1.0, (0.5, 0.5, -2), 9)
    inter = np.concatenate([p, n, uv], 1).astype(np.float32)        # stride 32
    v = g.view(inter.tobytes(), stride=32, target=34962)
    ap = g.accessor(v, FLOAT, p.shape[0], "VEC3", 0, (p.min(0), p.max(0)))
    an = g.accessor(v, FLOAT, p.shape[0], "VEC3", 12)
    au = g.accessor(v, FLOAT, p.shape[0], "VEC2", 24)
    lead = np.array([0, 1, 2, 2, 1, 3, 4, 3, 5, 1, 4, 0], np.uint16)  # 12 valid indices in front of the accessor's range
    iv = g.view(np.concatenate([lead, i]).astype(np.uint16).tobytes(), target=34963)
    ai = g.accessor(iv, USHORT, i.size, "SCALAR", offset=lead.size * 2)
    prim = {"attributes": {"POSITION": ap, "NORMAL": an, "TEXCOORD_0": au}, "indices": ai, "mode": 4,
            "material": g.material(base=(0.5, 0.5, 0.5, 1.0), metallic=0.00005, roughness=0.00005)}
    g.node(mesh=g.mesh([prim]), translation=(0, 0, 0))
    out["interleaved"] = g.tobytes()
    # 5. the shape of src/main.cpp's own asset: a lit Cornell box in one mesh -- five wall quads, a ceiling light with emissive
    #    strength, two boxes -- one primitive per material, so that quirk 14 shapes the very scene main() would render
    g = Glb()
    prims = []

    def quad(c, a, b, nrm):
        c, a, b, nrm = (np.array(x, np.float32) for x in (c, a, b, nrm))
        pos = np.array([c - a - b, c + a - b, c + a + b, c - a + b], np.float32)
        return pos, np.tile(nrm, (4, 1)).astype(np.float32), np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32), np.array([0, 1, 2, 0, 2, 3], np.uint16)
    white = g.material(base=(0.73, 0.73, 0.73, 1), metallic=0.0, roughness=0.0)
    red = g.material(base=(0.65, 0.05, 0.05, 1), metallic=0.0, roughness=0.0)
    green = g.material(base=(0.12, 0.45, 0.15, 1), metallic=0.0, roughness=0.6)
    light = g.material(emissive=(1.0, 0.7, 0.25), strength=17.0)
    for args, mat in ((((0, 0, 0), (1, 0, 0), (0, 0, 1), (0, 1, 0)), white), (((0, 2, 0), (1, 0, 0), (0, 0, -1), (0, -1, 0)), white),
                      (((0, 1, -1), (1, 0, 0), (0, 1, 0), (0, 0, 1)), white), (((-1, 1, 0), (0, 0, 1), (0, 1, 0), (1, 0, 0)), red),
                      (((1, 1, 0), (0, 0, -1), (0, 1, 0), (-1, 0, 0)), green), (((0, 1.99, 0), (0.3, 0, 0), (0, 0, -0.3), (0, -1, 0)), light)):
        p, n, uv, i = quad(*args)
        prims.append(g.primitive(p, n, i, uv, mat))
    p, n, uv, i = box((-0.35, 0.6, -0.3), (0.3, 0.6, 0.3))
    prims.append(g.primitive(p, n, i, uv, white))
    g.node(mesh=g.mesh(prims))
    out["cornell_lit"] = g.tobytes()
    return out
