"""Scene layout the reference's notes plan for the GPU (Documentation/gpu.meshes.txt:16-34):
per triangle three float4 = (v0, packed n0), (e0 = v1-v0, packed n1), (e1 = v2-v0, packed n2),
and the corrected form of its experimental normal packing (UnitTests/NormalPackingTest.cpp:10-23).

pack: byte_k = floor(n_k*127 + 127.5); packed = byte_0/2^8 + byte_1/2^16 + byte_2/2^24 (exact in
fp32: 24 fraction bits).  unpack: byte_k = floor(fract(packed * 256^k) * 256); n_k = byte_k/127 - 1.
(The reference's unpack multiplies by 1, 65536, 16777216 and therefore returns (0, 1, -1) for
(0, 0, 1): its own test cannot pass as written)"""
import numpy as np

F = np.float32


def pack_normal(n):
    n = np.asarray(n, F)
    b = np.floor(n * F(127.0) + F(127.5)).astype(F)
    return F(b[..., 0] / F(256.0) + b[..., 1] / F(65536.0) + b[..., 2] / F(16777216.0))


def unpack_normal(packed):
    p = np.asarray(packed, F)
    out = []
    for m in (F(1.0), F(256.0), F(65536.0)):
        s = (p * m).astype(F)
        frac = (s - np.floor(s)).astype(F)
        out.append((np.floor(frac * F(256.0)) / F(127.0) - F(1.0)).astype(F))
    return np.stack(out, axis=-1)


def to_edge_format(vertices, normals=None):
    """(3N, 4) absolute-vertex float4 rows -> (3N, 4) rows (v0, e0, e1); e0/e1 are the same fp32
    subtractions the trace path performs (Kernels.cuh:37-38).  normals: optional (3N, 3)."""
    v = np.ascontiguousarray(vertices, F).reshape(-1, 3, 4)
    out = np.zeros_like(v)
    out[:, 0, :3] = v[:, 0, :3]
    out[:, 1, :3] = v[:, 1, :3] - v[:, 0, :3]
    out[:, 2, :3] = v[:, 2, :3] - v[:, 0, :3]
    if normals is not None:
        out[:, :, 3] = pack_normal(np.asarray(normals, F).reshape(-1, 3, 3))
    return out.reshape(-1, 4)


def vertex_normals(edge_rows):
    """(3N, 4) edge-format rows -> (3N, 3) unpacked per-vertex normals."""
    return unpack_normal(np.ascontiguousarray(edge_rows, F).reshape(-1, 4)[:, 3])


def uv_sphere(center=(0.0, 0.0, -4.0), radius=1.0, n_lat=8, n_lon=16):
    """A latitude/longitude tessellation of a sphere in edge format with per-vertex normals
    (the outward unit normals, quantised by the packing), wound so that the outside faces
    have det > 0 for rays arriving from outside (Kernels.cuh:39-45).  2*n_lon*(n_lat-1) triangles."""
    c = np.asarray(center, np.float64)

    def vert(i, j):
        th = np.pi * i / n_lat                # polar angle from +y
        ph = 2.0 * np.pi * (j % n_lon) / n_lon
        n = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
        return c + radius * n, n

    tris, nrm = [], []
    for i in range(n_lat):
        for j in range(n_lon):
            quad = [vert(i, j), vert(i + 1, j), vert(i + 1, j + 1), vert(i, j + 1)]
            for (a, b, d) in ((0, 2, 1), (0, 3, 2)):
                if (i == 0 and (a, b, d) == (0, 3, 2)) or (i == n_lat - 1 and (a, b, d) == (0, 2, 1)):
                    continue                  # degenerate at the poles
                tris.append([quad[a][0], quad[b][0], quad[d][0]])
                nrm.append([quad[a][1], quad[b][1], quad[d][1]])
    t = np.asarray(tris, F)
    rows = np.zeros((t.shape[0], 3, 4), F)
    rows[:, :, :3] = t
    return to_edge_format(rows.reshape(-1, 4), normals=np.asarray(nrm, F).reshape(-1, 3))
