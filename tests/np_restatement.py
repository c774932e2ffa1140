"""A second, independent restatement of the hot-path arithmetic in numpy (test infrastructure).

Its purpose is to cross-check oracle/pcs_oracle.c: two restatements written separately, in
different languages, that must agree bit for bit. float32 FMA does not exist in numpy, so it is
emulated exactly: the product of two float32 is exact in float64; the sum with the addend is
rounded to ODD in float64 (via TwoSum), and rounding that to float32 is then correctly rounded.
"""
import numpy as np


def fma32(a, b, c):
    a = np.asarray(a, np.float32).astype(np.float64)
    b = np.asarray(b, np.float32).astype(np.float64)
    c = np.asarray(c, np.float32).astype(np.float64)
    with np.errstate(all="ignore"):
        p = a * b                      # exact: 24 + 24 bits
        s = p + c
        # TwoSum error term
        bb = s - p
        e = (p - (s - bb)) + (c - bb)
        fin = np.isfinite(s) & np.isfinite(e) & (e != 0)
        bits = s.view(np.int64).copy()
        even = (bits & 1) == 0
        # round to odd: if inexact and the float64 result is even, move one ulp toward the true value
        toward_up = (e > 0)
        pos = s > 0
        step = np.where(toward_up == pos, 1, -1)
        adj = fin & even
        bits = np.where(adj, bits + step, bits)
        s_odd = bits.view(np.float64)
        # s == 0 with e != 0 cannot happen for inexact sums of this magnitude; keep s as is otherwise
        return np.where(adj, s_odd, s).astype(np.float32)


def cvtt(f):
    f = np.asarray(f, np.float32)
    with np.errstate(all="ignore"):
        ok = (f >= np.float32(-2147483648.0)) & (f < np.float32(2147483648.0))
        return np.where(ok, np.trunc(np.where(ok, f, 0)).astype(np.int64), -2**31)


def pack_np(sc, V, T, color):
    V = np.asarray(V, np.float32); T = np.asarray(T, np.float32)
    M = np.array(list(sc.cam_to_world), np.float32)
    W, H = sc.color.width, sc.color.height
    xi = np.clip(cvtt(fma32(T[:, 0], np.float32(W), np.float32(0.5))), 0, W - 1)
    yi = np.clip(cvtt(fma32(T[:, 1], np.float32(H), np.float32(0.5))), 0, H - 1)
    idx = xi * sc.color_bpp + yi * sc.color_stride
    out = np.zeros((V.shape[0], 5), np.int16)
    for r in range(3):
        a = fma32(V[:, 0], M[4 * r], M[4 * r + 3])
        a = fma32(V[:, 1], M[4 * r + 1], a)
        a = fma32(V[:, 2], M[4 * r + 2], a)
        with np.errstate(all="ignore"):
            a = (a * np.float32(1000.0)).astype(np.float32)
        out[:, r] = (cvtt(a) & 0xFFFF).astype(np.uint16).view(np.int16)
    color = np.asarray(color, np.uint8).astype(np.int64)
    out[:, 3] = (color[idx] | (color[idx + 1] << 8)).astype(np.uint16).view(np.int16)
    out[:, 4] = color[idx + 2].astype(np.int16)
    return out


def deproject_np(sc, depth, half_pixel=False):
    f32 = np.float32
    di, ci = sc.depth, sc.color
    W, H = di.width, di.height
    d = np.asarray(depth, np.uint16).reshape(H, W)
    with np.errstate(all="ignore"):
        z = (f32(sc.depth_scale) * d.astype(f32)).astype(f32)
        mx = ((np.arange(W, dtype=f32) - f32(di.ppx)) / f32(di.fx)).astype(f32)[None, :]
        my = ((np.arange(H, dtype=f32) - f32(di.ppy)) / f32(di.fy)).astype(f32)[:, None]
        X = (z * mx).astype(f32); Y = (z * my).astype(f32); Z = z
        R = [f32(x) for x in sc.depth_to_color.rotation]; t = [f32(x) for x in sc.depth_to_color.translation]
        def row(i):
            return (((R[i] * X).astype(f32) + (R[i + 3] * Y).astype(f32)).astype(f32)
                    + (R[i + 6] * Z).astype(f32)).astype(f32) + t[i]
        P0, P1, P2 = row(0).astype(f32), row(1).astype(f32), row(2).astype(f32)
        x = (P0 / P2).astype(f32); y = (P1 / P2).astype(f32)
        px = ((x * f32(ci.fx)).astype(f32) + f32(ci.ppx)).astype(f32)
        py = ((y * f32(ci.fy)).astype(f32) + f32(ci.ppy)).astype(f32)
        if half_pixel:                      # PCS_FLAG_TEXCOORD_HALF_PIXEL: older librealsense pixel_to_texcoord
            px = (px + f32(0.5)).astype(f32); py = (py + f32(0.5)).astype(f32)
        u = (px / f32(ci.width)).astype(f32); v = (py / f32(ci.height)).astype(f32)
    valid = Z != 0
    u = np.where(valid, u, f32(0)); v = np.where(valid, v, f32(0))
    vtx = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(f32)
    tex = np.stack([u, v], -1).reshape(-1, 2).astype(f32)
    return vtx, tex


def transform_payload_np(payload, m16, downsample=1):
    """The centre's decode / affine / re-encode (src/pcs-multicamera-optimized.cpp:226-265, 289) in numpy float32: division by
    1000.0f, ((m0*x + m1*y) + m2*z) + m3 with every product and sum rounded to float32, * 1000.0f, truncation, low 16 bits."""
    f32 = np.float32
    d = max(int(downsample), 1)
    p = np.asarray(payload, np.int16).reshape(-1, 5)
    p = p[::d][:p.shape[0] // d]          # the cloud's width is size / downsample, rounded down (:230)
    M = np.asarray(m16, f32).reshape(-1)
    out = np.empty_like(p)
    with np.errstate(all="ignore"):
        x, y, z = [(p[:, k].astype(f32) / f32(1000.0)).astype(f32) for k in range(3)]
        for r in range(3):
            a = (M[4 * r] * x).astype(f32)
            a = (a + (M[4 * r + 1] * y).astype(f32)).astype(f32)
            a = (a + (M[4 * r + 2] * z).astype(f32)).astype(f32)
            a = (a + M[4 * r + 3]).astype(f32)
            a = (a * f32(1000.0)).astype(f32)
            out[:, r] = (cvtt(a) & 0xFFFF).astype(np.uint16).view(np.int16)
    out[:, 3] = p[:, 3]
    out[:, 4] = (p[:, 4].view(np.uint16) & 0xFF).astype(np.int16)
    return out
