"""The mesh of the reference's raster pass as its loader hands it to GL (ObjParser::parse, App.cpp:71-72): interleaved vertices
(position, normal, texcoord: GLUtils.hpp:29-34) and 32-bit triangle indices, ready for raytracer_amd.api.OverlayMesh."""
import ctypes as C

import numpy as np


def load_obj(path: str):
    """Vertex and index arrays as the reference's ObjParser builds them (ObjParser.cpp:196-441): one vertex per distinct
    (position, texcoord, normal) index triple in order of first use; a missing texcoord index selects (0, 0); a quad is split
    along 0-2 when the angles at its corners 1 and 3 sum to at most pi, else along 1-3.  Faces with more than four corners and
    faces without normals are not supported here.  (Parity unpinned: the loader cannot be built in this image; only the arrays it
    yields matter to the raster pass, and both sides of every comparison receive the same arrays.)  Returns ([n, 8] f32, [m] u32)."""
    libm = C.CDLL("libm.so.6"); libm.acosf.restype = C.c_float; libm.acosf.argtypes = [C.c_float]
    f32 = np.float32
    pos, tex, nrm = [], [], []
    verts, index, seen = [], [], {}
    for line in open(path):
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "v":
            pos.append(np.array(tok[1:4], f32))
        elif tok[0] == "vt":
            tex.append(np.array(tok[1:3], f32))
        elif tok[0] == "vn":
            nrm.append(np.array(tok[1:4], f32))
        elif tok[0] == "f":
            corners = []
            for c in tok[1:]:
                parts = (c.split("/") + ["", ""])[:3]
                corners.append((int(parts[0]) - 1, int(parts[1]) - 1 if parts[1] else 0, int(parts[2]) - 1))
            if len(corners) == 4:
                p = [pos[c[0]] for c in corners]
                def angle(a, b, c):      # at b
                    u, w = a - b, c - b
                    return libm.acosf(f32(np.dot(u, w)) / f32(np.sqrt(f32(np.dot(u, u)) * f32(np.dot(w, w)))))
                if f32(angle(p[0], p[1], p[2])) + f32(angle(p[2], p[3], p[0])) <= f32(np.pi):
                    corners = [corners[0], corners[1], corners[2], corners[0], corners[2], corners[3]]
                else:
                    corners = [corners[0], corners[1], corners[3], corners[1], corners[2], corners[3]]
            elif len(corners) != 3:
                raise ValueError("faces with more than four corners are not supported")
            if not tex:
                tex.append(np.zeros(2, f32))
            for c in corners:
                if c not in seen:
                    seen[c] = len(verts)
                    verts.append(np.concatenate([pos[c[0]], nrm[c[2]], tex[c[1]]]))
                index.append(seen[c])
    return np.array(verts, f32), np.array(index, np.uint32)
