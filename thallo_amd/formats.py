"""Readers / writers for the data formats on either side of the hot path, so that the reference's shipped
data sets can be fed to `libThallo.so` unchanged.  numpy + zlib only.

  .constraints      image_warping markers            examples/image_warping/src/main.cpp:4-27
  .png              8-bit gray / RGB / RGBA           (the harnesses load them with LodePNG; only non-interlaced 8-bit here)
  .imagedump        shape_from_shading buffers        examples/shape_from_shading/src/SimpleBuffer.cpp:12-52
  SFS parameters    160-byte struct dump              examples/shape_from_shading/src/TerraSolverParameters.h:7-45
  BAL text          bundle adjustment in the large    examples/bundle_adjustment/src/bal_problem.cpp:61-150
  .off / .ply       triangle meshes (ARAP)            read through OpenMesh in the reference (arap_mesh_deformation/src/main.cpp:56-62)
  .mrk              mesh landmarks                    examples/arap_mesh_deformation/src/LandMarkSet.h
"""
import struct
import zlib

import numpy as np


# ------------------------------------------------------------------------------------------------ image_warping
def read_constraints(path):
    """-> int array [n, 4] of (x, y, target_x, target_y).  File: count, then 4 ints per marker (main.cpp:13-24)."""
    tok = open(path).read().split()
    n = int(tok[0])
    v = np.array([int(t) for t in tok[1:1 + 4 * n]], dtype=np.int64)
    if v.size != 4 * n:
        raise ValueError(f"{path}: expected {4 * n} integers after the count, found {v.size}")
    return v.reshape(n, 4)


def write_constraints(path, c):
    c = np.asarray(c, dtype=np.int64).reshape(-1, 4)
    with open(path, "w") as f:
        f.write(f"{len(c)}\n")
        for row in c:
            f.write(" ".join(str(int(x)) for x in row) + "\n")


def add_border_constraints(c, W, H):
    """The harness pins every border pixel to itself (main.cpp:119-129), appended after the file's markers in y-major order."""
    extra = [(x, y, x, y) for y in range(H) for x in range(W) if y == 0 or x == 0 or y == H - 1 or x == W - 1]
    return np.concatenate([np.asarray(c, dtype=np.int64).reshape(-1, 4), np.array(extra, dtype=np.int64)])


def constraint_image(c, mask, alpha=1.0):
    """float32 [H, W, 2]: (-1,-1) everywhere except markers on unmasked pixels, whose target is interpolated from the pixel
    itself (alpha = 0) to the marker target (alpha = 1) -- CombinedSolver.h:178-204 (float32 arithmetic as there)."""
    H, W = mask.shape
    out = np.full((H, W, 2), -1.0, dtype=np.float32)
    a = np.float32(alpha)
    for x, y, tx, ty in np.asarray(c, dtype=np.int64):
        if mask[y, x] == 0:
            out[y, x, 0] = (np.float32(1.0) - a) * np.float32(x) + a * np.float32(tx)
            out[y, x, 1] = (np.float32(1.0) - a) * np.float32(y) + a * np.float32(ty)
    return out


# ------------------------------------------------------------------------------------------------ PNG (8-bit, non-interlaced)
_PNG_SIG = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


def read_png(path):
    """-> uint8 array [H, W, C] (C = 1, 2, 3 or 4).  8-bit, non-interlaced, non-palette PNGs only."""
    b = open(path, "rb").read()
    if b[:8] != _PNG_SIG:
        raise ValueError(f"{path}: not a PNG")
    pos, idat, hdr = 8, [], None
    while pos < len(b):
        (ln,), typ = struct.unpack(">I", b[pos:pos + 4]), b[pos + 4:pos + 8]
        data = b[pos + 8:pos + 8 + ln]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif typ == b"IDAT":
            idat.append(data)
        elif typ == b"IEND":
            break
        pos += 12 + ln
    W, H, depth, ctype, _, _, interlace = hdr
    if depth != 8 or ctype not in _CHANNELS or interlace != 0:
        raise ValueError(f"{path}: only 8-bit non-interlaced gray/RGB/RGBA PNGs are supported")
    C = _CHANNELS[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(H, 1 + W * C)
    out = np.zeros((H, W * C), dtype=np.uint8)
    prev = np.zeros(W * C, dtype=np.int32)
    for y in range(H):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:                       # Sub: running sum per channel
            cur = line.reshape(W, C).cumsum(axis=0).reshape(-1) & 255
        else:                               # Average / Paeth: sequential in x
            cur = np.zeros(W * C, dtype=np.int32)
            for i in range(W * C):
                a = cur[i - C] if i >= C else 0
                bb = prev[i]
                if ft == 3:
                    pred = (a + bb) >> 1
                else:
                    c = prev[i - C] if i >= C else 0
                    p = a + bb - c
                    pa, pb, pc = abs(p - a), abs(p - bb), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (bb if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    return out.reshape(H, W, C)


def write_png(path, img):
    img = np.asarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    H, W, C = img.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[C]
    raw = np.concatenate([np.zeros((H, 1), dtype=np.uint8), img.reshape(H, W * C)], axis=1).tobytes()

    def chunk(typ, data):
        return struct.pack(">I", len(data)) + typ + data + struct.pack(">I", zlib.crc32(typ + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


# ------------------------------------------------------------------------------------------------ shape_from_shading
_DUMP_TYPES = {0: np.float32, 1: np.uint8}        # SimpleBuffer.h DataType {FLOAT, UCHAR}


def read_imagedump(path, clamp_infinity=True):
    """-> array [H, W, C].  Header: int32 width, height, channels, datatype (0 float32, 1 uint8), then raw rows.
    clamp_infinity mirrors SimpleBuffer.cpp:29-41 (+inf -> FLT_MAX, -inf -> -10000) for single-channel float data."""
    b = open(path, "rb").read()
    w, h, c, t = struct.unpack("<4i", b[:16])
    if t not in _DUMP_TYPES:
        raise ValueError(f"{path}: unknown datatype {t}")
    a = np.frombuffer(b, dtype=_DUMP_TYPES[t], count=w * h * c, offset=16).reshape(h, w, c).copy()
    if t == 0 and clamp_infinity:
        flat = a.reshape(-1)[:w * h]                  # the reference clamps the first w*h floats only
        flat[np.isposinf(flat)] = np.finfo(np.float32).max
        flat[np.isneginf(flat)] = -10000.0
    return a


def write_imagedump(path, a):
    a = np.asarray(a)
    if a.ndim == 2:
        a = a[:, :, None]
    t = 0 if a.dtype == np.float32 else 1 if a.dtype == np.uint8 else None
    if t is None:
        raise ValueError("imagedump holds float32 or uint8")
    h, w, c = a.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", w, h, c, t) + np.ascontiguousarray(a).tobytes())


_SFS_FIELDS = ("weightFitting", "weightRegularizer", "weightPrior", "weightShading", "weightShadingStart", "weightShadingIncrement",
               "weightBoundary", "fx", "fy", "ux", "uy")


def read_sfs_params(path):
    """160-byte struct: 11 floats, a 4x4 float transform, 9 lighting coefficients, 3 unused uints (TerraSolverParameters.h:7-31)."""
    b = open(path, "rb").read()
    if len(b) < 160:
        raise ValueError(f"{path}: {len(b)} bytes, expected 160")
    f = struct.unpack("<36f", b[:144])
    d = dict(zip(_SFS_FIELDS, f[:11]))
    d["deltaTransform"] = np.array(f[11:27], dtype=np.float32).reshape(4, 4)
    d["lightingCoefficients"] = np.array(f[27:36], dtype=np.float32)
    d["unused"] = struct.unpack("<3I", b[144:156])
    return d


def write_sfs_params(path, d):
    f = [float(d[k]) for k in _SFS_FIELDS] + [float(x) for x in np.asarray(d["deltaTransform"]).reshape(16)] + \
        [float(x) for x in np.asarray(d["lightingCoefficients"]).reshape(9)]
    with open(path, "wb") as fh:
        fh.write(struct.pack("<36f", *f) + struct.pack("<3I", *d.get("unused", (0, 0, 0))) + b"\0\0\0\0")


# ------------------------------------------------------------------------------------------------ bundle adjustment
def read_bal(path, sort_for_coherency=True):
    """BAL text: 'C P O', O lines 'cam point x y', then 9*C camera and 3*P point parameters, one per line.
    -> dict(cameras float64 [C,9], points float64 [P,3], observations float64 [O,2], cam_idx, pt_idx int32 [O]).
    sort_for_coherency orders observations by (camera, point) like bal_problem.cpp:101-131."""
    tok = open(path).read().split()
    C, P, O = int(tok[0]), int(tok[1]), int(tok[2])
    body = np.array(tok[3:3 + 4 * O], dtype=np.float64).reshape(O, 4)
    cam_idx, pt_idx, obs = body[:, 0].astype(np.int32), body[:, 1].astype(np.int32), body[:, 2:4].copy()
    par = np.array(tok[3 + 4 * O:3 + 4 * O + 9 * C + 3 * P], dtype=np.float64)
    if par.size != 9 * C + 3 * P:
        raise ValueError(f"{path}: truncated parameter block")
    if sort_for_coherency:
        order = np.lexsort((pt_idx, cam_idx))
        cam_idx, pt_idx, obs = cam_idx[order], pt_idx[order], obs[order]
    return {"cameras": par[:9 * C].reshape(C, 9), "points": par[9 * C:].reshape(P, 3), "observations": obs, "cam_idx": cam_idx, "pt_idx": pt_idx}


def write_bal(path, cameras, points, observations, cam_idx, pt_idx):
    with open(path, "w") as f:
        f.write(f"{len(cameras)} {len(points)} {len(observations)}\n")
        for c, p, o in zip(cam_idx, pt_idx, observations):
            f.write(f"{int(c)} {int(p)}     {o[0]:.6e} {o[1]:.6e}\n")
        for v in np.asarray(cameras).reshape(-1):
            f.write(f"{v:.16e}\n")
        for v in np.asarray(points).reshape(-1):
            f.write(f"{v:.16e}\n")


# ------------------------------------------------------------------------------------------------ meshes
def read_off(path):
    tok = open(path).read().split()
    if tok[0] != "OFF":
        raise ValueError(f"{path}: not an OFF file")
    nv, nf = int(tok[1]), int(tok[2])
    V = np.array(tok[4:4 + 3 * nv], dtype=np.float32).reshape(nv, 3)
    faces, pos = [], 4 + 3 * nv
    for _ in range(nf):
        k = int(tok[pos])
        faces.append([int(t) for t in tok[pos + 1:pos + 1 + k]])
        pos += 1 + k
    return V, faces


def write_off(path, V, faces):
    with open(path, "w") as f:
        f.write(f"OFF\n{len(V)} {len(faces)} 0\n")
        for v in V:
            f.write(f"{v[0]:.6f} {v[1]:.6f} {v[2]:.6f}\n")
        for fc in faces:
            f.write(f"{len(fc)} " + " ".join(str(int(i)) for i in fc) + "\n")


_PLY_T = {"char": "b", "uchar": "B", "short": "h", "ushort": "H", "int": "i", "uint": "I", "float": "f", "double": "d",
          "int8": "b", "uint8": "B", "int16": "h", "uint16": "H", "int32": "i", "uint32": "I", "float32": "f", "float64": "d"}


def read_ply(path):
    """-> (V float32 [n,3], faces list).  ascii and binary_little_endian; vertex x/y/z plus a face index list."""
    b = open(path, "rb").read()
    end = b.index(b"end_header") + len(b"end_header")
    end = b.index(b"\n", end) + 1
    fmt, elems = None, []
    for ln in b[:end].decode("ascii", "replace").splitlines():
        t = ln.split()
        if not t:
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            elems.append([t[1], int(t[2]), []])
        elif t[0] == "property":
            elems[-1][2].append(t[1:])
    V, faces = None, []
    if fmt == "ascii":
        tok, pos = b[end:].split(), 0
        for name, n, props in elems:
            rows = []
            for _ in range(n):
                row = {}
                for p in props:
                    if p[0] == "list":
                        k = int(tok[pos]); row[p[3]] = [int(x) for x in tok[pos + 1:pos + 1 + k]]; pos += 1 + k
                    else:
                        row[p[1]] = float(tok[pos]); pos += 1
                rows.append(row)
            if name == "vertex":
                V = np.array([[r["x"], r["y"], r["z"]] for r in rows], dtype=np.float32)
            elif name == "face":
                faces = [next(v for v in r.values() if isinstance(v, list)) for r in rows]
    elif fmt == "binary_little_endian":
        pos = end
        for name, n, props in elems:
            if all(p[0] != "list" for p in props):
                dt = np.dtype([(p[1], "<" + _PLY_T[p[0]]) for p in props])
                arr = np.frombuffer(b, dtype=dt, count=n, offset=pos)
                pos += n * dt.itemsize
                if name == "vertex":
                    V = np.stack([arr["x"], arr["y"], arr["z"]], axis=1).astype(np.float32)
            else:
                for _ in range(n):
                    row = None
                    for p in props:
                        if p[0] == "list":
                            (k,) = struct.unpack_from("<" + _PLY_T[p[1]], b, pos); pos += struct.calcsize(_PLY_T[p[1]])
                            row = list(struct.unpack_from(f"<{k}" + _PLY_T[p[2]], b, pos)); pos += k * struct.calcsize(_PLY_T[p[2]])
                        else:
                            pos += struct.calcsize(_PLY_T[p[0]])
                    if name == "face":
                        faces.append(row)
    else:
        raise ValueError(f"{path}: unsupported PLY format {fmt}")
    return V, faces


def write_ply(path, V, faces, binary=True):
    V = np.asarray(V, dtype=np.float32)
    hdr = ["ply", "format binary_little_endian 1.0" if binary else "format ascii 1.0", f"element vertex {len(V)}",
           "property float x", "property float y", "property float z", f"element face {len(faces)}",
           "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode())
        if binary:
            f.write(V.astype("<f4").tobytes())
            for fc in faces:
                f.write(struct.pack(f"<B{len(fc)}i", len(fc), *fc))
        else:
            for v in V:
                f.write(f"{v[0]:.7g} {v[1]:.7g} {v[2]:.7g}\n".encode())
            for fc in faces:
                f.write((f"{len(fc)} " + " ".join(str(i) for i in fc) + "\n").encode())


def read_mrk(path):
    """Landmarks: count, then 'x y z radius vertex_index' per line -> (idx int32 [n], target float32 [n,3])."""
    tok = open(path).read().split()
    n = int(tok[0])
    a = np.array(tok[1:1 + 5 * n], dtype=np.float64).reshape(n, 5)
    return a[:, 4].astype(np.int32), a[:, :3].astype(np.float32)


def write_mrk(path, idx, target, radius=0.0224524):
    with open(path, "w") as f:
        f.write(f"{len(idx)}\n")
        for i, t in zip(idx, target):
            f.write(f"{t[0]:.6g} {t[1]:.6g} {t[2]:.6g} {radius} {int(i)}\n")


def mesh_directed_edges(faces, n_vertices):
    """V0, V1 int32 [E]: both directions of every undirected mesh edge, grouped by V0 in vertex order -- the layout
    the reference builds from the one-ring of every vertex (arap_mesh_deformation/src/CombinedSolver.h:39-66)."""
    und = set()
    for fc in faces:
        k = len(fc)
        for i in range(k):
            a, b = int(fc[i]), int(fc[(i + 1) % k])
            if a != b:
                und.add((min(a, b), max(a, b)))
    nbr = [[] for _ in range(n_vertices)]
    for a, b in sorted(und):
        nbr[a].append(b); nbr[b].append(a)
    v0 = np.array([v for v in range(n_vertices) for _ in nbr[v]], dtype=np.int32)
    v1 = np.array([w for v in range(n_vertices) for w in sorted(nbr[v])], dtype=np.int32)
    return v0, v1
