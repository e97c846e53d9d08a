"""Independent numpy / pure-Python restatement of the path's contracts, written from the same published
algorithms as oracle/*.c but sharing no code with it.  Used by make_golden.py to mint the committed fixtures and by
tests/test_oracle_golden.py to pin the C oracle.  Small inputs only (pure-Python loops)."""
import math
import numpy as np

PALETTE_BGR = [(128, 128, 128), (0, 0, 128), (128, 192, 192), (0, 69, 255), (128, 64, 128), (222, 40, 60),
               (0, 128, 128), (128, 128, 192), (128, 64, 64), (128, 0, 64), (0, 64, 64), (192, 128, 0)]

f32 = np.float32


def cv_round(x):
    """cvRound: nearest, ties to even (Python's round does exactly that on floats)"""
    return int(round(float(x)))


# ---------------------------------------------------------------- matcher (src/orb.cpp:16-29 + cv::BFMatcher)
_POP8 = np.array([bin(i).count("1") for i in range(256)], np.uint8)


def hamming_matrix(q, t):
    out = np.zeros((len(q), len(t)), np.int32)
    for i in range(0, len(q), 128):
        out[i:i + 128] = _POP8[q[i:i + 128, None, :] ^ t[None, :, :]].sum(axis=2, dtype=np.int32)
    return out


def knn2(q, t):
    d = hamming_matrix(q, t)
    order = np.argsort(d, axis=1, kind="stable")[:, :2]          # stable: equal distances keep the lower train index
    return order.astype(np.int32), np.take_along_axis(d, order, axis=1)


def ratio_keep(d0, d1, ratio=0.8):
    return float(f32(d0)) < ratio * float(f32(d1))               # float < double*float, evaluated in double


def match(q, t, ratio=0.8):
    idx, dist = knn2(q, t)
    return [(i, int(idx[i, 0]), 0, float(dist[i, 0])) for i in range(len(q)) if ratio_keep(dist[i, 0], dist[i, 1], ratio)]


# ---------------------------------------------------------------- mapper front half
def project2dTo3d(d, u, v, cam):
    cx, cy, fx, fy, scale = cam
    if d == 0:
        return f32(0), f32(0), f32(0)
    z = f32(float(d) / scale)
    return f32((u - cx) * float(z) / fx), f32((v - cy) * float(z) / fy), z


def moving_mask(sem):
    from scipy import ndimage
    b, g, r = sem[..., 0], sem[..., 1], sem[..., 2]
    m = (((b == 0) & (g == 64) & (r == 64)) | ((b == 192) & (g == 128) & (r == 0))).astype(np.uint8) * 255
    return ndimage.maximum_filter(m, size=5, mode="constant", cval=0)


def label_of(b, g, r):
    try:
        return PALETTE_BGR.index((int(b), int(g), int(r)))
    except ValueError:
        return 255


def backproject(depth, rgb, sem, mask, cam, T, max_distance):
    """returns list of (x,y,z,b,g,r,label) in row-major order; T row-major 4x4 or None"""
    out = []
    h, w = depth.shape
    lim = max_distance * cam[4]
    for m in range(h):
        for n in range(w):
            d = int(depth[m, n])
            if d == 0 or d > lim or mask[m, n] == 255:
                continue
            pb, pg, pr = (int(v) for v in sem[m, n])
            if (pb, pg, pr) in ((128, 128, 128), (128, 192, 192), (192, 128, 0)):
                continue
            x, y, z = project2dTo3d(d, n, m, cam)
            if T is not None:
                X, Y, Z = float(x), float(y), float(z)
                x = f32(T[0][0] * X + T[0][1] * Y + T[0][2] * Z + T[0][3])
                y = f32(T[1][0] * X + T[1][1] * Y + T[1][2] * Z + T[1][3])
                z = f32(T[2][0] * X + T[2][1] * Y + T[2][2] * Z + T[2][3])
            out.append((x, y, z, int(rgb[m, n, 0]), int(rgb[m, n, 1]), int(rgb[m, n, 2]), label_of(pb, pg, pr)))
    return out


def voxel_filter(points, leaf):
    """points: list of (x,y,z,b,g,r,label).  Exact-sum contract.  Returns list of (x,y,z,b,g,r,label) sorted by (k,j,i)."""
    inv = f32(1.0) / f32(leaf)
    cells = {}
    for (x, y, z, b, g, r, lab) in points:
        fi, fj, fk = (float(f32(x) * inv), float(f32(y) * inv), float(f32(z) * inv))
        if not all(math.isfinite(q) and abs(math.floor(q)) < 1048576 for q in (fi, fj, fk)):
            continue                                    # range contract: not keyable in 21 bits per axis -> skipped like a non-finite point in PCL
        i = int(math.floor(fi)); j = int(math.floor(fj)); k = int(math.floor(fk))
        c = cells.setdefault((k, j, i), [0, 0, 0, 0, 0, 0, 0, [0] * 12])
        c[0] += cv_round(float(x) * 16777216.0); c[1] += cv_round(float(y) * 16777216.0); c[2] += cv_round(float(z) * 16777216.0)
        c[3] += r; c[4] += g; c[5] += b; c[6] += 1
        if lab < 12:
            c[7][lab] += 1
    out = []
    for key in sorted(cells):
        sx, sy, sz, sr, sg, sb, n, hist = cells[key]
        best, lab = 0, 255
        for q in range(12):
            if hist[q] > best:
                best, lab = hist[q], q
        out.append((f32((float(sx) / float(n)) * (1.0 / 16777216.0)), f32((float(sy) / float(n)) * (1.0 / 16777216.0)),
                    f32((float(sz) / float(n)) * (1.0 / 16777216.0)), sb // n, sg // n, sr // n, lab))
    return out


# ---------------------------------------------------------------- ORB image primitives
def bgr2gray(bgr):
    b = bgr[..., 0].astype(np.int64); g = bgr[..., 1].astype(np.int64); r = bgr[..., 2].astype(np.int64)
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def resize_tables(ssize, dsize):
    scale = 1.0 / (float(dsize) / ssize)
    ofs, c0, c1 = [], [], []
    for d in range(dsize):
        f = f32((d + 0.5) * scale - 0.5)
        s = int(math.floor(float(f)))
        f = f32(f - f32(s))
        if s < 0:
            f, s = f32(0), 0
        if s >= ssize - 1:
            f, s = f32(0), ssize - 1
        ofs.append(s); c0.append(cv_round(float(f32(f32(1) - f) * f32(2048)))); c1.append(cv_round(float(f * f32(2048))))
    return np.array(ofs), np.array(c0, np.int64), np.array(c1, np.int64)


def resize_linear(src, dw, dh):
    sh, sw = src.shape
    xo, a0, a1 = resize_tables(sw, dw)
    yo, b0, b1 = resize_tables(sh, dh)
    s = src.astype(np.int64)
    x1 = np.minimum(xo + 1, sw - 1); y1 = np.minimum(yo + 1, sh - 1)
    h = s[:, xo] * a0[None, :] + s[:, x1] * a1[None, :]           # rows of horizontally resized ints
    h0 = h[yo, :]; h1 = h[y1, :]
    out = (((b0[:, None] * (h0 >> 4)) >> 16) + ((b1[:, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def gaussian_taps():
    cf = [f32(math.exp(-0.5 / 4.0 * (i - 3.0) ** 2)) for i in range(7)]
    s = f32(0)
    for c in cf:
        s = f32(s + c)
    s = f32(f32(1) / s)
    return [cv_round(float(f32(f32(float(c) * float(s)) * f32(256)))) for c in cf]


def gaussian7(src):
    k = np.array(gaussian_taps(), np.int64)
    p = np.pad(src.astype(np.int64), 3, mode="reflect")          # numpy 'reflect' == BORDER_REFLECT_101
    h, w = src.shape
    row = sum(k[t] * p[:, t:t + w] for t in range(7))
    col = sum(k[t] * row[t:t + h, :] for t in range(7))
    return np.minimum((col + 32768) >> 16, 255).astype(np.uint8)


RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast_S(img, x, y):
    v = int(img[y, x])
    ring = [int(img[y + dy, x + dx]) for dx, dy in RING]
    best = -256
    for k in range(16):
        arc = [ring[(k + j) % 16] for j in range(9)]
        best = max(best, min(a - v for a in arc), min(v - a for a in arc))
    return best


def fast_atan2(y, x):
    c = f32(180 / math.pi)
    p1, p3, p5, p7 = f32(f32(0.9997878412794807) * c), f32(f32(-0.3258083974640975) * c), f32(f32(0.1555786518463281) * c), f32(f32(-0.04432655554792128) * c)
    y, x = f32(y), f32(x)
    ax, ay = abs(x), abs(y)
    eps = f32(2.220446049250313e-16)
    if ax >= ay:
        cc = f32(ay / f32(ax + eps)); c2 = f32(cc * cc)
        a = f32(f32(f32(f32(f32(f32(p7 * c2) + p5) * c2) + p3) * c2 + p1) * cc)
    else:
        cc = f32(ax / f32(ay + eps)); c2 = f32(cc * cc)
        a = f32(f32(90) - f32(f32(f32(f32(f32(f32(p7 * c2) + p5) * c2) + p3) * c2 + p1) * cc))
    if x < 0:
        a = f32(f32(180) - a)
    if y < 0:
        a = f32(f32(360) - a)
    return a


def contract_sincos(angle_rad):
    HI, LO = 1.57079632673412561417e+00, 6.07710050650619224932e-11
    x = float(f32(angle_rad))
    kd = float(round(x * 0.63661977236758134308))
    k = int(kd)
    r = (x - kd * HI) - kd * LO
    r2 = r * r
    ps = -1.0 / 1307674368000.0
    for c in (1.0 / 6227020800.0, -1.0 / 39916800.0, 1.0 / 362880.0, -1.0 / 5040.0, 1.0 / 120.0, -1.0 / 6.0):
        ps = ps * r2 + c
    sn = r + r * (r2 * ps)
    pc = 1.0 / 20922789888000.0
    for c in (-1.0 / 87178291200.0, 1.0 / 479001600.0, -1.0 / 3628800.0, 1.0 / 40320.0, -1.0 / 720.0, 1.0 / 24.0, -0.5):
        pc = pc * r2 + c
    cs = 1.0 + r2 * pc
    S, C = [(sn, cs), (cs, -sn), (-sn, -cs), (-cs, sn)][k & 3]
    return f32(S), f32(C)


# ---------------------------------------------------------------- ORB extractor (ORB_SLAM2::ORBextractor restated)
EDGE, HALF = 19, 15


def umax_table():
    vmax = int(math.floor(HALF * math.sqrt(2.0) / 2 + 1)); vmin = int(math.ceil(HALF * math.sqrt(2.0) / 2))
    um = [0] * (HALF + 2)
    for v in range(vmax + 1):
        um[v] = cv_round(math.sqrt(HALF * HALF - v * v))
    v0 = 0
    for v in range(HALF, vmin - 1, -1):
        while um[v0] == um[v0 + 1]:
            v0 += 1
        um[v] = v0; v0 += 1
    return um[:HALF + 1]


def level_params(nfeatures, scale, nlevels):
    sfd = float(f32(scale))
    sf = [f32(1)]
    for _ in range(1, nlevels):
        sf.append(f32(float(sf[-1]) * sfd))
    inv = [f32(f32(1) / s) for s in sf]
    factor = f32(1.0 / sfd)
    nd = f32(f32(f32(nfeatures) * f32(f32(1) - factor)) / f32(f32(1) - f32(math.pow(float(factor), float(nlevels)))))
    feat, tot = [], 0
    for _ in range(nlevels - 1):
        feat.append(cv_round(float(nd))); tot += feat[-1]; nd = f32(nd * factor)
    feat.append(max(nfeatures - tot, 0))
    return sf, inv, feat


def cell_candidates(img, ini_th, min_th):
    """list of (x, y, response) relative to (minBorder, minBorder), in cell-major / raster order"""
    h, w = img.shape
    minB = EDGE - 3; maxBX = w - EDGE + 3; maxBY = h - EDGE + 3
    width = f32(maxBX - minB); height = f32(maxBY - minB)
    nCols = int(width / f32(30)); nRows = int(height / f32(30))
    wCell = int(math.ceil(float(f32(width / f32(nCols))))); hCell = int(math.ceil(float(f32(height / f32(nRows)))))
    out = []
    for i in range(nRows):
        iniY = minB + i * hCell; maxY = iniY + hCell + 6
        if iniY >= maxBY - 3:
            continue
        maxY = min(maxY, maxBY)
        for j in range(nCols):
            iniX = minB + j * wCell; maxX = iniX + wCell + 6
            if iniX >= maxBX - 6:
                continue
            maxX = min(maxX, maxBX)
            cw, ch = maxX - iniX, maxY - iniY
            S = np.zeros((ch, cw), np.int32)
            for y in range(3, ch - 3):
                for x in range(3, cw - 3):
                    S[y, x] = fast_S(img, iniX + x, iniY + y)
            for th in (ini_th, min_th):
                score = np.where(S > th, S - 1, 0)                # cv::FAST response, 0 where not a corner
                found = []
                for y in range(3, ch - 3):
                    for x in range(3, cw - 3):
                        s = score[y, x]
                        if s == 0 and S[y, x] <= th:
                            continue
                        nb = score[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
                        if S[y, x] > th and s > nb.max():
                            found.append((x + j * wCell, y + i * hCell, int(s)))
                if found:
                    out.extend(found)
                    break
    return out


class _Node:
    __slots__ = ("UL", "UR", "BL", "BR", "keys", "no_more", "seq")


def distribute_octtree(cands, minX, maxX, minY, maxY, N):
    """literal port of ORBextractor::DistributeOctTree with Python lists standing in for std::list"""
    if not cands:
        return []
    seq = [0]

    def new_node():
        n = _Node(); n.keys = []; n.no_more = False; n.seq = seq[0]; seq[0] += 1
        return n

    nIni = max(int(math.floor(float(f32(maxX - minX) / f32(maxY - minY)) + 0.5)), 1)
    hX = f32(f32(maxX - minX) / f32(nIni))
    L, ini = [], []
    for i in range(nIni):
        n = new_node()
        n.UL = (int(f32(hX * f32(i))), 0); n.UR = (int(f32(hX * f32(i + 1))), 0)
        n.BL = (n.UL[0], maxY - minY); n.BR = (n.UR[0], maxY - minY)
        L.append(n); ini.append(n)
    for k in cands:
        ini[min(int(f32(f32(k[0]) / hX)), nIni - 1)].keys.append(k)
    L = [n for n in L if n.keys]
    for n in L:
        if len(n.keys) == 1:
            n.no_more = True

    def divide(p):
        halfX = int(math.ceil(float(f32(p.UR[0] - p.UL[0]) / f32(2)))); halfY = int(math.ceil(float(f32(p.BR[1] - p.UL[1]) / f32(2))))
        n1, n2, n3, n4 = new_node(), new_node(), new_node(), new_node()
        n1.UL = p.UL; n1.UR = (p.UL[0] + halfX, p.UL[1]); n1.BL = (p.UL[0], p.UL[1] + halfY); n1.BR = (p.UL[0] + halfX, p.UL[1] + halfY)
        n2.UL = n1.UR; n2.UR = p.UR; n2.BL = n1.BR; n2.BR = (p.UR[0], p.UL[1] + halfY)
        n3.UL = n1.BL; n3.UR = n1.BR; n3.BL = p.BL; n3.BR = (n1.BR[0], p.BL[1])
        n4.UL = n3.UR; n4.UR = n2.BR; n4.BL = n3.BR; n4.BR = p.BR
        for k in p.keys:
            if k[0] < n1.UR[0]:
                (n1 if k[1] < n1.BR[1] else n3).keys.append(k)
            else:
                (n2 if k[1] < n1.BR[1] else n4).keys.append(k)
        for n in (n1, n2, n3, n4):
            if len(n.keys) == 1:
                n.no_more = True
        return n1, n2, n3, n4

    finish = False
    while not finish:
        prev = len(L); to_expand = 0; vs = []
        for p in [n for n in L]:                                   # iteration over the list as it was at the start
            if p.no_more:
                continue
            for c in divide(p):
                if c.keys:
                    L.insert(0, c)
                    if len(c.keys) > 1:
                        to_expand += 1; vs.append(c)
            L.remove(p)
        if len(L) >= N or len(L) == prev:
            finish = True
        elif len(L) + to_expand * 3 > N:
            while not finish:
                prev = len(L)
                vp = sorted(vs, key=lambda n: (len(n.keys), n.seq)); vs = []
                for p in reversed(vp):
                    for c in divide(p):
                        if c.keys:
                            L.insert(0, c)
                            if len(c.keys) > 1:
                                vs.append(c)
                    L.remove(p)
                    if len(L) >= N:
                        break
                if len(L) >= N or len(L) == prev:
                    finish = True
    res = []
    for n in L:
        best = n.keys[0]
        for k in n.keys[1:]:
            if k[2] > best[2]:
                best = k
        res.append(best)
    return res


def ic_angle(img, x, y, umax):
    m01 = m10 = 0
    for u in range(-HALF, HALF + 1):
        m10 += u * int(img[y, x + u])
    for v in range(1, HALF + 1):
        vs = 0; d = umax[v]
        for u in range(-d, d + 1):
            p, m = int(img[y + v, x + u]), int(img[y - v, x + u])
            vs += p - m; m10 += u * (p + m)
        m01 += v * vs
    return fast_atan2(f32(m01), f32(m10))


def orb_descriptor(blur, x, y, angle_deg, pattern):
    ang = f32(f32(angle_deg) * f32(math.pi / float(f32(180.0))))
    b, a = contract_sincos(ang)
    desc = []
    for i in range(32):
        val = 0
        for j in range(8):
            q = pattern[(i * 8 + j) * 4:(i * 8 + j) * 4 + 4]
            t = []
            for e in range(2):
                px, py = f32(q[2 * e]), f32(q[2 * e + 1])
                yy = cv_round(float(f32(f32(px * b) + f32(py * a)))); xx = cv_round(float(f32(f32(px * a) - f32(py * b))))
                t.append(int(blur[y + yy, x + xx]))
            val |= (1 if t[0] < t[1] else 0) << j
        desc.append(val)
    return desc


def orb_extract(gray, nfeatures, scale, nlevels, ini_th, min_th, pattern):
    sf, inv, feat = level_params(nfeatures, scale, nlevels)
    umax = umax_table()
    h, w = gray.shape
    imgs = []
    for l in range(nlevels):
        lw, lh = cv_round(float(f32(f32(w) * inv[l]))), cv_round(float(f32(f32(h) * inv[l])))
        imgs.append(gray.copy() if l == 0 else resize_linear(imgs[-1], lw, lh))
    kps, descs = [], []
    for l, img in enumerate(imgs):
        lh, lw = img.shape
        minB = EDGE - 3
        cands = cell_candidates(img, ini_th, min_th)
        sel = distribute_octtree(cands, minB, lw - EDGE + 3, minB, lh - EDGE + 3, feat[l])
        blur = gaussian7(img)
        for (cx, cy, s) in sel:
            x, y = cx + minB, cy + minB
            ang = ic_angle(img, x, y, umax)
            descs.append(orb_descriptor(blur, x, y, ang, pattern))
            fx, fy = f32(x), f32(y)
            if l:
                fx, fy = f32(fx * sf[l]), f32(fy * sf[l])
            kps.append((fx, fy, f32(f32(31) * sf[l]), ang, f32(s), l, -1))
    return kps, np.array(descs, np.uint8).reshape(-1, 32)


# ---------------------------------------------------------------- stereo quad matcher (cv::goodFeaturesToTrack, cv::calcOpticalFlowPyrLK restated)
def _refl(i, n):
    i = np.abs(i); i = np.where(i >= n, 2 * n - 2 - i, i)
    return np.clip(i, 0, n - 1)


def min_eigen_map(img):
    h, w = img.shape
    p = img.astype(np.int64)
    ys, xs = np.arange(h), np.arange(w)
    R = lambda dy, dx: p[_refl(ys + dy, h)][:, _refl(xs + dx, w)]
    dx = (R(-1, 1) - R(-1, -1)) + 2 * (R(0, 1) - R(0, -1)) + (R(1, 1) - R(1, -1))
    dy = (R(1, -1) - R(-1, -1)) + 2 * (R(1, 0) - R(-1, 0)) + (R(1, 1) - R(-1, 1))
    box = lambda a: sum(a[_refl(ys + j, h)][:, _refl(xs + i, w)] for j in (-1, 0, 1) for i in (-1, 0, 1))
    sxx, sxy, syy = box(dx * dx), box(dx * dy), box(dy * dy)
    s = f32(1.0 / (255.0 * 4.0 * 3.0)); s2 = f32(s * s)
    a = (sxx.astype(np.float32) * s2) * f32(0.5); b = sxy.astype(np.float32) * s2; c = (syy.astype(np.float32) * s2) * f32(0.5)
    d = a - c
    return ((a + c) - np.sqrt(d * d + b * b)).astype(np.float32)


def gftt(img, max_corners, quality, min_distance):
    e = min_eigen_map(img); h, w = e.shape
    thr = f32(float(max(e.max(), f32(0))) * quality)
    t = np.where(e > thr, e, f32(0))
    cands = []
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            v = e[y, x]
            if v > thr and v == t[y - 1:y + 2, x - 1:x + 2].max():
                cands.append((-float(v), y * w + x))
    cands.sort()
    out = []
    md2 = f32(min_distance * min_distance)
    for _, idx in cands:
        y, x = divmod(idx, w)
        if all(f32(f32(x - px) * f32(x - px) + f32(y - py) * f32(y - py)) >= md2 for px, py in out):
            out.append((x, y))
            if len(out) == max_corners:
                break
    return np.array(out, np.float32).reshape(-1, 2)


def pyrdown(img):
    h, w = img.shape; dh, dw = (h + 1) // 2, (w + 1) // 2
    k = np.array([1, 4, 6, 4, 1], np.int64); p = img.astype(np.int64)
    ys, xs = 2 * np.arange(dh), 2 * np.arange(dw)
    rows = sum(k[i + 2] * p[:, _refl(xs + i, w)] for i in range(-2, 3))
    out = sum(k[j + 2] * rows[_refl(ys + j, h)] for j in range(-2, 3))
    return ((out + 128) >> 8).astype(np.uint8)


def scharr(img):
    h, w = img.shape; p = img.astype(np.int64); ys, xs = np.arange(h), np.arange(w)
    R = lambda dy, dx: p[_refl(ys + dy, h)][:, _refl(xs + dx, w)]
    dx = 3 * (R(-1, 1) - R(-1, -1)) + 10 * (R(0, 1) - R(0, -1)) + 3 * (R(1, 1) - R(1, -1))
    dy = 3 * (R(1, -1) - R(-1, -1)) + 10 * (R(1, 0) - R(-1, 0)) + 3 * (R(1, 1) - R(-1, 1))
    return np.stack([dx, dy], 2).astype(np.int16)


def lk_track(prev, nxt, pts, max_count=200, epsilon=0.01, min_eig_thr=1e-6, win=11, levels=4):
    P, N = [prev], [nxt]
    for _ in range(1, levels):
        P.append(pyrdown(P[-1])); N.append(pyrdown(N[-1]))
    D = [scharr(p) for p in P]
    n = len(pts); out = np.zeros((n, 2), np.float32); status = np.ones(n, np.uint8); err = np.zeros(n, np.float32)
    half = f32((win - 1) * 0.5); SC = f32(1.0 / (1 << 20)); eps2 = f32(epsilon * epsilon)
    desc = lambda v, b: (v + (1 << (b - 1))) >> b

    def wts(fx, fy):
        ix, iy = int(math.floor(float(fx))), int(math.floor(float(fy)))
        a, b = f32(fx - f32(ix)), f32(fy - f32(iy))
        w00 = cv_round(float(f32(f32(f32(1) - a) * f32(f32(1) - b)) * f32(1 << 14))); w01 = cv_round(float(f32(a * f32(f32(1) - b)) * f32(1 << 14)))
        w10 = cv_round(float(f32(f32(f32(1) - a) * b) * f32(1 << 14)))
        return ix, iy, w00, w01, w10, (1 << 14) - w00 - w01 - w10

    for lv in range(levels - 1, -1, -1):
        H, W = P[lv].shape
        pix = lambda im, x, y: int(im[int(_refl(np.array(y), H)), int(_refl(np.array(x), W))])
        der = lambda x, y, c: 0 if (x < 0 or y < 0 or x >= W or y >= H) else int(D[lv][y, x, c])
        for i in range(n):
            ppx = f32(f32(pts[i][0]) * f32(1.0 / (1 << lv))); ppy = f32(f32(pts[i][1]) * f32(1.0 / (1 << lv)))
            if lv == levels - 1:
                nx, ny = ppx, ppy
            else:
                nx, ny = f32(out[i][0] * f32(2)), f32(out[i][1] * f32(2))
            out[i] = (nx, ny)
            ppx = f32(ppx - half); ppy = f32(ppy - half)
            ipx, ipy, w00, w01, w10, w11 = wts(ppx, ppy)
            if ipx < -win or ipx >= W or ipy < -win or ipy >= H:
                if lv == 0:
                    status[i] = 0; err[i] = 0
                continue
            I, Ix, Iy = [], [], []
            for y in range(win):
                for x in range(win):
                    gx, gy = ipx + x, ipy + y
                    I.append(desc(pix(P[lv], gx, gy) * w00 + pix(P[lv], gx + 1, gy) * w01 + pix(P[lv], gx, gy + 1) * w10 + pix(P[lv], gx + 1, gy + 1) * w11, 9))
                    Ix.append(desc(der(gx, gy, 0) * w00 + der(gx + 1, gy, 0) * w01 + der(gx, gy + 1, 0) * w10 + der(gx + 1, gy + 1, 0) * w11, 14))
                    Iy.append(desc(der(gx, gy, 1) * w00 + der(gx + 1, gy, 1) * w01 + der(gx, gy + 1, 1) * w10 + der(gx + 1, gy + 1, 1) * w11, 14))
            A11 = f32(f32(sum(a * a for a in Ix)) * SC); A12 = f32(f32(sum(a * b for a, b in zip(Ix, Iy))) * SC); A22 = f32(f32(sum(b * b for b in Iy)) * SC)
            Dt = f32(f32(A11 * A22) - f32(A12 * A12))
            dd = f32(A11 - A22)
            minEig = f32(f32(f32(A22 + A11) - np.sqrt(f32(f32(dd * dd) + f32(f32(f32(4) * A12) * A12)))) / f32(2 * win * win))
            err[i] = minEig
            if minEig < f32(min_eig_thr) or Dt < f32(np.finfo(np.float32).eps):
                if lv == 0:
                    status[i] = 0
                continue
            Dt = f32(f32(1) / Dt)
            npx, npy = f32(nx - half), f32(ny - half)
            pdx = pdy = f32(0)
            for j in range(max_count):
                inx, iny, w00, w01, w10, w11 = wts(npx, npy)
                if inx < -win or inx >= W or iny < -win or iny >= H:
                    if lv == 0:
                        status[i] = 0
                    break
                sb1 = sb2 = 0; k = 0
                for y in range(win):
                    for x in range(win):
                        gx, gy = inx + x, iny + y
                        diff = desc(pix(N[lv], gx, gy) * w00 + pix(N[lv], gx + 1, gy) * w01 + pix(N[lv], gx, gy + 1) * w10 + pix(N[lv], gx + 1, gy + 1) * w11, 9) - I[k]
                        sb1 += diff * Ix[k]; sb2 += diff * Iy[k]; k += 1
                b1, b2 = f32(f32(sb1) * SC), f32(f32(sb2) * SC)
                ddx = f32(f32(f32(A12 * b2) - f32(A22 * b1)) * Dt); ddy = f32(f32(f32(A12 * b1) - f32(A11 * b2)) * Dt)
                npx, npy = f32(npx + ddx), f32(npy + ddy)
                out[i] = (f32(npx + half), f32(npy + half))
                if f32(f32(ddx * ddx) + f32(ddy * ddy)) <= eps2:
                    break
                if j > 0 and abs(f32(ddx + pdx)) < f32(0.01) and abs(f32(ddy + pdy)) < f32(0.01):
                    out[i] = (f32(out[i][0] - f32(ddx * f32(0.5))), f32(out[i][1] - f32(ddy * f32(0.5))))
                    break
                pdx, pdy = ddx, ddy
    return out, status, err


# ---------------------------------------------------------------- stereo visual odometry (src/vo_stereo.cpp, src/vo.cpp)
# An independent restatement of oracle/vo.c in Python floats (IEEE double, never fused): same contracts (glibc rand,
# the sin/cos polynomial, OpenCV 2.4 LU, the 64-way summation order of the refinement).
class GlibcRand:
    def __init__(self, seed):
        seed = seed or 1
        r = [seed]
        for _ in range(30):
            hi, lo = divmod(r[-1], 127773)
            w = 16807 * lo - 2836 * hi
            if w < 0:
                w += 2147483647
            r.append(w)
        self.r = [x & 0xFFFFFFFF for x in r]; self.f = 3; self.b = 0
        for _ in range(310):
            self.next()

    def next(self):
        self.r[self.f] = (self.r[self.f] + self.r[self.b]) & 0xFFFFFFFF
        out = self.r[self.f] >> 1
        self.f = (self.f + 1) % 31; self.b = (self.b + 1) % 31
        return out

    def sample(self, n, num):
        pool = list(range(n)); out = []
        for _ in range(num):
            out.append(pool.pop(self.next() % len(pool)))
        return out


def vo_sincos(x):
    fn = float(np.rint(x * 6.36619772367581382433e-01))
    r = x - fn * 1.57079632673412561417e+00
    r = r - fn * 6.07710050650619224932e-11
    r = r - fn * 2.02226624879595063154e-21
    z = r * r
    S = (-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04, 2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10)
    Cc = (4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05, -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11)
    ps = S[1] + z * (S[2] + z * (S[3] + z * (S[4] + z * S[5])))
    sr = r + (r * z) * (S[0] + z * ps)
    pc = Cc[0] + z * (Cc[1] + z * (Cc[2] + z * (Cc[3] + z * (Cc[4] + z * Cc[5]))))
    cr = (1.0 - 0.5 * z) + (z * z) * pc
    return ((sr, cr), (cr, -sr), (-sr, -cr), (-cr, sr))[int(fn) & 3]


def vo_solve6(A, b):
    A = [list(map(float, row)) for row in A]; b = list(map(float, b)); eps = 2.220446049250313e-16 * 100
    for i in range(6):
        k = i
        for j in range(i + 1, 6):
            if abs(A[j][i]) > abs(A[k][i]):
                k = j
        if abs(A[k][i]) < eps:
            return None
        if k != i:
            for j in range(i, 6):
                A[i][j], A[k][j] = A[k][j], A[i][j]
            b[i], b[k] = b[k], b[i]
        d = -1 / A[i][i]
        for j in range(i + 1, 6):
            alpha = A[j][i] * d
            for kk in range(i + 1, 6):
                A[j][kk] += alpha * A[i][kk]
            b[j] += alpha * b[i]
        A[i][i] = -d
    for i in range(5, -1, -1):
        s = b[i]
        for k in range(i + 1, 6):
            s -= A[i][k] * b[k]
        b[i] = s * A[i][i]
    return b


def _vo_rot(tr):
    (sx, cx), (sy, cy), (sz, cz) = vo_sincos(tr[0]), vo_sincos(tr[1]), vo_sincos(tr[2])
    R = dict(r00=+cy*cz, r01=-cy*sz, r02=+sy, r10=+sx*sy*cz+cx*sz, r11=-sx*sy*sz+cx*cz, r12=-sx*cy, r20=-cx*sy*cz+sx*sz, r21=+cx*sy*sz+sx*cz, r22=+cx*cy,
             x10=+cx*sy*cz-sx*sz, x11=-cx*sy*sz-sx*sz, x12=-cx*cy, x20=+sx*sy*cz+cx*sz, x21=-sx*sy*sz+cx*cz, x22=-sx*cy,
             y00=-sy*cz, y01=+sy*sz, y02=+cy, y10=+sx*cy*cz, y11=-sx*cy*sz, y12=+sx*sy, y20=-cx*cy*cz, y21=+cx*cy*sz, y22=-cx*sy,
             z00=-cy*sz, z01=-cy*cz, z10=-sx*sy*sz+cx*cz, z11=-sx*sy*cz-cx*sz, z20=+cx*sy*sz+sx*cz, z21=+cx*sy*cz-sx*sz, tx=tr[3], ty=tr[4], tz=tr[5])
    return R


def _vo_point(m, P, R, want_j):
    f, cu, cv, base, thr, rw = P
    dd = max(float(np.float32(m["u1p"]) - np.float32(m["u2p"])), 1.0)
    X = (float(m["u1p"]) - cu) * base / dd; Y = (float(m["v1p"]) - cv) * base / dd; Z = f * base / dd
    X1c = R["r00"]*X + R["r01"]*Y + R["r02"]*Z + R["tx"]; Y1c = R["r10"]*X + R["r11"]*Y + R["r12"]*Z + R["ty"]; Z1c = R["r20"]*X + R["r21"]*Y + R["r22"]*Z + R["tz"]
    obs = [float(m["u1c"]), float(m["v1c"]), float(m["u2c"]), float(m["v2c"])]
    w = 1.0 / (abs(obs[0] - cu) / abs(cu) + 0.05) if rw else 1.0
    X2c = X1c - base
    pred = [f * X1c / Z1c + cu, f * Y1c / Z1c + cv, f * X2c / Z1c + cu, f * Y1c / Z1c + cv]
    if not want_j:
        return pred, None, None
    J = [[0.0] * 6 for _ in range(4)]
    for j in range(6):
        if j == 0: d = (0.0, R["x10"]*X + R["x11"]*Y + R["x12"]*Z, R["x20"]*X + R["x21"]*Y + R["x22"]*Z)
        elif j == 1: d = (R["y00"]*X + R["y01"]*Y + R["y02"]*Z, R["y10"]*X + R["y11"]*Y + R["y12"]*Z, R["y20"]*X + R["y21"]*Y + R["y22"]*Z)
        elif j == 2: d = (R["z00"]*X + R["z01"]*Y, R["z10"]*X + R["z11"]*Y, R["z20"]*X + R["z21"]*Y)
        else: d = (1.0 if j == 3 else 0.0, 1.0 if j == 4 else 0.0, 1.0 if j == 5 else 0.0)
        J[0][j] = w * f * (d[0]*Z1c - X1c*d[2]) / (Z1c*Z1c); J[1][j] = w * f * (d[1]*Z1c - Y1c*d[2]) / (Z1c*Z1c)
        J[2][j] = w * f * (d[0]*Z1c - X2c*d[2]) / (Z1c*Z1c); J[3][j] = w * f * (d[1]*Z1c - Y1c*d[2]) / (Z1c*Z1c)
    return pred, J, [w * (obs[k] - pred[k]) for k in range(4)]


def _vo_update(ms, active, P, tr, eps, lanes):
    if len(active) < 3:
        return 1
    R = _vo_rot(tr)
    acc = [[0.0] * 42 for _ in range(lanes)]
    for q, idx in enumerate(active):
        _, J, res = _vo_point(ms[idx], P, R, True)
        a = acc[q % lanes]
        for r in range(4):
            for mm in range(6):
                for nn in range(6):
                    a[mm * 6 + nn] += J[r][mm] * J[r][nn]
                a[36 + mm] += J[r][mm] * res[r]
    s = 1
    while s < lanes:
        acc = [[acc[l][k] + acc[l ^ s][k] for k in range(42)] for l in range(lanes)]
        s <<= 1
    x = vo_solve6([acc[0][6 * i:6 * i + 6] for i in range(6)], acc[0][36:])
    if x is None:
        return 1
    conv = True
    for k in range(6):
        tr[k] += 1.0 * x[k]
        if abs(x[k]) > eps:
            conv = False
    return 2 if conv else 0


def vo_estimate(ms, P, samples):
    """P = (f, cu, cv, base, inlier_threshold, reweighting).  returns (success, tr, inliers)"""
    n = len(ms)
    if n < 6:
        return False, [0.0] * 6, []
    best, tr_best = [], [0.0] * 6
    for smp in samples:
        tr = [0.0] * 6; res = 0; it = 0
        while res == 0:
            res = _vo_update(ms, list(smp), P, tr, 1e-6, 1)
            it += 1
            if it - 1 > 20 or res == 2:
                break
        if res != 1:
            R = _vo_rot(tr); cur = []
            for i in range(n):
                pred, _, _ = _vo_point(ms[i], P, R, False)
                o = [float(ms[i]["u1c"]), float(ms[i]["v1c"]), float(ms[i]["u2c"]), float(ms[i]["v2c"])]
                if sum((o[k] - pred[k]) * (o[k] - pred[k]) for k in range(4)) < P[4] * P[4]:
                    cur.append(i)
            if len(cur) > len(best):
                best, tr_best = cur, list(tr)
    ok = True
    if len(best) >= 6:
        res = 0; it = 0
        while res == 0:
            res = _vo_update(ms, best, P, tr_best, 1e-8, 64)
            it += 1
            if it - 1 > 100 or res == 2:
                break
        ok = res == 2
    else:
        ok = False
    return ok, tr_best, best


# ---------------------------------------------------------------- cv::StereoSGBM (OpenCV 2.4, single-pass mode), second restatement
# Written from the algorithm's definition in VOLUME form (whole cost volumes, numpy over the disparity axis), sharing no code and no structure
# with oracle/sgbm.c, which follows OpenCV's row-by-row ring buffers: prefilter -> Birchfield-Tomasi pixel cost -> SADWindow box sums -> the
# five scan directions -> winner / uniqueness / sub-pixel / left-right check -> medianBlur(3) -> filterSpeckles.  Contract: oracle/sgbm.c header.
def _sgbm_prefiltered(img, ftzero):
    """two channels per image: the x-Sobel (rows above / below replicated) clipped to +-ftzero and shifted to [0, 2 ftzero], and the raw intensity;
    the first and last column of BOTH channels hold ftzero (the clip-table entry of 0, as OpenCV initialises them)"""
    a = img.astype(np.int64); h, w = a.shape
    up = np.vstack([a[:1], a[:-1]]); dn = np.vstack([a[1:], a[-1:]])
    sob = np.zeros_like(a)
    sob[:, 1:-1] = (a[:, 2:] - a[:, :-2]) * 2 + (up[:, 2:] - up[:, :-2]) + (dn[:, 2:] - dn[:, :-2])
    c0 = np.clip(sob, -ftzero, ftzero) + ftzero
    c1 = a.copy()
    for c in (c0, c1):
        c[:, 0] = ftzero; c[:, -1] = ftzero
    return c0, c1


def _sgbm_interp_range(ch):
    """per pixel the min / max of the value and its two half-way interpolations to the horizontal neighbours (Birchfield-Tomasi)"""
    lo = ch.copy(); hi = ch.copy()
    l = (ch[:, 1:] + ch[:, :-1]) // 2
    lo[:, 1:] = np.minimum(lo[:, 1:], l); hi[:, 1:] = np.maximum(hi[:, 1:], l)       # neighbour on the left
    lo[:, :-1] = np.minimum(lo[:, :-1], l); hi[:, :-1] = np.maximum(hi[:, :-1], l)   # neighbour on the right
    return lo, hi


def _trunc_div(a, b):
    """C integer division (towards zero), b > 0"""
    return np.where(a >= 0, a // b, -((-a) // b))


def sgbm_raw(left, right, minD=0, ndisp=80, SAD=11, P1=None, P2=None, disp12MaxDiff=1, preFilterCap=63, uniquenessRatio=10):
    left = np.asarray(left, np.uint8); right = np.asarray(right, np.uint8)
    h, w = left.shape
    P1 = 4 * SAD * SAD if P1 is None else P1; P2 = 32 * SAD * SAD if P2 is None else P2
    P2 = max(P2, P1 + 1)
    D = ndisp; maxD = minD + D
    ftzero = max(preFilterCap, 15) | 1
    INV = (minD - 1) * 16
    minX1 = max(maxD, 0); maxX1 = w + min(minD, 0); W1 = maxX1 - minX1
    out = np.full((h, w), INV, np.int64)
    if W1 <= 0:
        return out.astype(np.int16)
    half = SAD // 2
    # ---- pixel cost volume pix[y, x', d], x' = x - minX1, right pixel x - (d + minD)
    pix = np.zeros((h, W1, D), np.int64)
    for ch, (lc, rc) in enumerate(zip(_sgbm_prefiltered(left, ftzero), _sgbm_prefiltered(right, ftzero))):
        u0, u1 = _sgbm_interp_range(lc); v0, v1 = _sgbm_interp_range(rc)
        xs = np.arange(minX1, maxX1)
        for d in range(D):
            xr = xs - (d + minD)
            ok = (xr >= 0) & (xr < w)
            xr = np.clip(xr, 0, w - 1)
            u = lc[:, xs]; v = rc[:, xr]
            ca = np.maximum(0, np.maximum(u - v1[:, xr], v0[:, xr] - u))
            cb = np.maximum(0, np.maximum(v - u1[:, xs], u0[:, xs] - v))
            pix[:, :, d] += np.where(ok, np.minimum(ca, cb) >> (2 if ch else 0), 0)
    # ---- SAD window: horizontal box with replicated columns, then vertical box with the top row replicated; OpenCV 2.4 freezes column x' = 0 after
    # row 0 and stops updating once the window would need rows >= h (see oracle/sgbm.c header)
    xi = np.clip(np.arange(W1)[:, None] + np.arange(-half, half + 1)[None, :], 0, W1 - 1)
    hs = pix[:, xi, :].sum(axis=2)                                  # [h, W1, D]
    C = np.zeros((h, W1, D), np.int64)
    C[0] = P2 + sum(hs[min(k, h - 1)] * (half + 1 if k == 0 else 1) for k in range(half + 1))
    for y in range(1, h):
        C[y] = C[y - 1]
        if y + half < h:
            C[y, 1:] = C[y - 1, 1:] + hs[y + half, 1:] - hs[max(y - half - 1, 0), 1:]
    BIG = 32767

    def step(prev, prev_min, cost):
        """L(d) = cost(d) + min(prev(d), prev(d-1) + P1, prev(d+1) + P1, prev_min + P2) - (prev_min + P2); cost carries the P2 (it is C)"""
        lo = np.concatenate([[BIG], prev[:-1]]) + P1; hi = np.concatenate([prev[1:], [BIG]]) + P1
        delta = prev_min + P2
        return cost + np.minimum(np.minimum(prev, lo), np.minimum(hi, delta)) - delta

    zeros = np.zeros(D, np.int64)
    prevL = [np.zeros((W1, D), np.int64) for _ in range(4)]; prevMin = [np.zeros(W1, np.int64) for _ in range(4)]   # previous row, directions 0..3
    for y in range(h):
        L = [np.zeros((W1, D), np.int64) for _ in range(4)]
        # directions from the previous row: up-left, up, up-right (a neighbour outside the image is an all-zero path with minimum 0)
        for k, dx in ((1, -1), (2, 0), (3, 1)):
            for x in range(W1):
                xp = x + dx
                if 0 <= xp < W1:
                    L[k][x] = step(prevL[k][xp], prevMin[k][xp], C[y, x])
                else:
                    L[k][x] = step(zeros, 0, C[y, x])
        for x in range(W1):                                           # from the left neighbour, sequential in x
            L[0][x] = step(L[0][x - 1], L[0][x - 1].min(), C[y, x]) if x > 0 else step(zeros, 0, C[y, x])
        S = np.minimum(L[0] + L[1] + L[2] + L[3], BIG)                # saturating int16 sum, term by term it never goes negative
        mins = [l.min(axis=1) for l in L]
        prevL, prevMin = L, mins
        # from the right neighbour, sequential from the right; winner selection on the way
        disp2cost = np.full(w, BIG, np.int64); disp2 = np.full(w, INV, np.int64)
        row = np.full(w, INV, np.int64)
        R = zeros; Rmin = 0
        for x in range(W1 - 1, -1, -1):
            R = step(R, Rmin, C[y, x]) if x < W1 - 1 else step(zeros, 0, C[y, x])
            Rmin = R.min()
            Sx = np.minimum(S[x] + R, BIG)
            best = int(np.argmin(Sx)); minS = int(Sx[best])          # first minimum
            far = np.abs(np.arange(D) - best) > 1
            if (far & (Sx * (100 - uniquenessRatio) < minS * 100)).any():
                continue
            x2 = x + minX1 - best - minD
            if disp2cost[x2] > minS:
                disp2cost[x2] = minS; disp2[x2] = best + minD
            if 0 < best < D - 1:
                den = max(int(Sx[best - 1] + Sx[best + 1] - 2 * Sx[best]), 1)
                dsub = best * 16 + int(_trunc_div(np.int64((Sx[best - 1] - Sx[best + 1]) * 16 + den), den * 2))
            else:
                dsub = best * 16
            row[x + minX1] = dsub + minD * 16
        for x in range(minX1, maxX1):                                 # left-right consistency on both roundings of the sub-pixel disparity
            d1 = int(row[x])
            if d1 == INV:
                continue
            dlo, dhi = d1 >> 4, (d1 + 15) >> 4
            xa, xb = x - dlo, x - dhi
            if 0 <= xa < w and disp2[xa] >= minD and abs(int(disp2[xa]) - dlo) > disp12MaxDiff and \
               0 <= xb < w and disp2[xb] >= minD and abs(int(disp2[xb]) - dhi) > disp12MaxDiff:
                row[x] = INV
        out[y] = row
    return out.astype(np.int16)


def median3_s16(img):
    p = np.pad(np.asarray(img, np.int16), 1, mode="edge"); h, w = img.shape
    return np.sort(np.stack([p[i:i + h, j:j + w] for i in range(3) for j in range(3)]), axis=0)[4].astype(np.int16)


def filter_speckles(img, new_val, max_size, max_diff):
    """components of the graph whose nodes are the pixels != new_val and whose edges join 4-neighbours differing by <= max_diff; small ones become new_val"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    a = np.asarray(img, np.int64); h, w = a.shape; idx = np.arange(h * w).reshape(h, w); valid = a != new_val
    eh = valid[:, 1:] & valid[:, :-1] & (np.abs(a[:, 1:] - a[:, :-1]) <= max_diff)
    ev = valid[1:, :] & valid[:-1, :] & (np.abs(a[1:, :] - a[:-1, :]) <= max_diff)
    r = np.concatenate([idx[:, 1:][eh], idx[1:, :][ev]]); c = np.concatenate([idx[:, :-1][eh], idx[:-1, :][ev]])
    n, lab = connected_components(coo_matrix((np.ones(len(r)), (r, c)), shape=(h * w, h * w)), directed=False)
    size = np.bincount(lab, minlength=n)
    out = a.copy()
    out[valid & (size[lab].reshape(h, w) <= max_size)] = new_val
    return out.astype(np.int16)


def sgbm(left, right, speckle_window=100, speckle_range=32, **kw):
    raw = sgbm_raw(left, right, **kw)
    d = median3_s16(raw)
    return filter_speckles(d, (kw.get("minD", 0) - 1) * 16, speckle_window, 16 * speckle_range) if speckle_window > 0 else d


# ---------------------------------------------------------------- PnPSolver::solvePnP (/root/reference/src/pnp.cpp:5-118), second restatement
# g2o's pose-only bundle adjustment as the reference sets it up (EdgeSE3ProjectXYZOnlyPose, Huber kernel with delta = float(sqrt(5.991)), Levenberg
# with the dense 6 x 6 LDL^T solve, four rounds of optimize(10)) and pnp.cpp's inlier bookkeeping as written.  Written from the algorithm's published
# description in plain Python floats (IEEE double, no fused operations); shares no code with oracle/pnp.c or include/ssm/pnp_core.h.  The numeric contract
# of the product is followed where it defines the bits: sums over edges run over 1024 "lanes" (edge i -> lane i % 1024, in list order), a lane group of
# 64 is added pairwise (neighbours first), the 16 groups in order; sin / cos = vo_sincos; the cube in Levenberg's lambda rule is t * t * t.
def _pnp_tree(vals):
    """vals: 1024 floats (one per lane) -> the contract's sum"""
    groups = []
    for g in range(16):
        a = list(vals[64 * g:64 * g + 64])
        while len(a) > 1:
            a = [a[i] + a[i + 1] for i in range(0, len(a), 2)]
        groups.append(a[0])
    s = groups[0]
    for g in groups[1:]:
        s = s + g
    return s


class _PnpEdge:
    __slots__ = ("id", "level", "robust", "X", "u", "v", "e")


def _pnp_map(R, t, X):
    return [R[3 * r] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2] + t[r] for r in range(3)]


def _pnp_error(e, R, t, cam):
    cx, cy, fx, fy = cam[0], cam[1], cam[2], cam[3]
    p = _pnp_map(R, t, e.X)
    e.e = (e.u - (p[0] / p[2] * fx + cx), e.v - (p[1] / p[2] * fy + cy))


def _pnp_huber(e2, delta):
    d2 = delta * delta
    if e2 <= d2:
        return e2, 1.0
    s = math.sqrt(e2)
    return 2 * s * delta - d2, delta / s


def _pnp_chi(edges, R, t, cam, delta):
    lanes = [0.0] * 1024
    for i, e in enumerate(edges):
        if e.level != 0:
            continue
        _pnp_error(e, R, t, cam)
        e2 = e.e[0] * e.e[0] + e.e[1] * e.e[1]
        lanes[i % 1024] += _pnp_huber(e2, delta)[0] if e.robust else e2
    return _pnp_tree(lanes)


def _pnp_system(edges, R, t, cam, delta):
    fx, fy = cam[2], cam[3]
    lanes = [[0.0] * 27 for _ in range(1024)]
    for i, e in enumerate(edges):
        if e.level != 0:
            continue
        acc = lanes[i % 1024]
        x, y, z = _pnp_map(R, t, e.X)
        iz = 1.0 / z; iz2 = iz * iz
        J = ((x * y * iz2 * fx, -(1 + (x * x * iz2)) * fx, y * iz * fx, -iz * fx, 0.0, x * iz2 * fx),
             ((1 + y * y * iz2) * fy, -x * y * iz2 * fy, -x * iz * fy, 0.0, -iz * fy, y * iz2 * fy))
        w = _pnp_huber(e.e[0] * e.e[0] + e.e[1] * e.e[1], delta)[1] if e.robust else 1.0
        for r in range(2):
            wr = -e.e[r] * w
            q = 0
            for a in range(6):
                acc[21 + a] += J[r][a] * wr
                for c in range(a + 1):
                    acc[q] += J[r][a] * w * J[r][c]; q += 1
    tot = [_pnp_tree([lanes[l][q] for l in range(1024)]) for q in range(27)]
    H = [[0.0] * 6 for _ in range(6)]
    q = 0
    for a in range(6):
        for c in range(a + 1):
            H[a][c] = tot[q]; q += 1
    return H, tot[21:]


def _pnp_ldlt(H, lam, b):
    A = [[H[i][j] + (lam if i == j else 0.0) for j in range(6)] for i in range(6)]
    L = [[0.0] * 6 for _ in range(6)]; D = [0.0] * 6
    for j in range(6):
        d = A[j][j]
        for k in range(j):
            d -= L[j][k] * L[j][k] * D[k]
        if not d > 0:
            return None
        D[j] = d; L[j][j] = 1.0
        for i in range(j + 1, 6):
            s = A[i][j]
            for k in range(j):
                s -= L[i][k] * L[j][k] * D[k]
            L[i][j] = s / d
    y = [0.0] * 6
    for i in range(6):
        s = b[i]
        for k in range(i):
            s -= L[i][k] * y[k]
        y[i] = s
    y = [y[i] / D[i] for i in range(6)]
    x = [0.0] * 6
    for i in range(5, -1, -1):
        s = y[i]
        for k in range(i + 1, 6):
            s -= L[k][i] * x[k]
        x[i] = s
    return x


def _pnp_mm(A, B):
    return [A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c] for r in range(3) for c in range(3)]


def _pnp_oplus(R, t, d):
    th = math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    W = [0.0, -d[2], d[1], d[2], 0.0, -d[0], -d[1], d[0], 0.0]
    W2 = _pnp_mm(W, W)
    I = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]
    if th < 0.00001:
        dR = [I[k] + W[k] + W2[k] for k in range(9)]; V = list(dR)
    else:
        sn, cs = vo_sincos(th)
        a = sn / th; b = (1 - cs) / (th * th); c = (th - sn) / (th * th * th)
        dR = [I[k] + a * W[k] + b * W2[k] for k in range(9)]; V = [I[k] + b * W[k] + c * W2[k] for k in range(9)]
    nR = _pnp_mm(dR, R)
    nt = []
    for r in range(3):
        vt = V[3 * r] * d[3] + V[3 * r + 1] * d[4] + V[3 * r + 2] * d[5]
        nt.append(dR[3 * r] * t[0] + dR[3 * r + 1] * t[1] + dR[3 * r + 2] * t[2] + vt)
    return nR, nt


def _pnp_optimize(edges, R, t, cam, delta, iterations):
    if not any(e.level == 0 for e in edges):
        return R, t
    lam, nu = 0.0, 2.0
    for it in range(iterations):
        chi = _pnp_chi(edges, R, t, cam, delta)
        H, b = _pnp_system(edges, R, t, cam, delta)
        if it == 0:
            lam = 1e-5 * max(abs(H[j][j]) for j in range(6)); nu = 2.0
        gain, trials = 0.0, 0
        while True:
            sR, st = R, t
            x = _pnp_ldlt(H, lam, b)
            ok = x is not None
            if not ok:
                x = [0.0] * 6
            R, t = _pnp_oplus(R, t, x)
            chi_new = _pnp_chi(edges, R, t, cam, delta)
            if not ok:
                chi_new = 1.7976931348623157e308
            gain = chi - chi_new
            scale = 0.0
            for j in range(6):
                scale += x[j] * (lam * x[j] + b[j])
            scale += 1e-3
            gain /= scale
            if gain > 0 and math.isfinite(chi_new):
                tt = 2 * gain - 1
                alpha = 1.0 - tt * tt * tt
                alpha = alpha if alpha < 2.0 / 3.0 else 2.0 / 3.0
                lam *= alpha if alpha > 1.0 / 3.0 else 1.0 / 3.0
                nu = 2.0; chi = chi_new
            else:
                lam *= nu; nu *= 2; R, t = sR, st
                if not math.isfinite(lam):
                    break
            trials += 1
            if not (gain < 0 and trials < 10):
                break
        if trials == 10 or gain == 0:
            break
    _pnp_chi(edges, R, t, cam, delta)
    return R, t


def pnp_solve(img, obj, cam, T, min_inliers=10):
    """img n x 2 float32, obj n x 3 float32 (all-zero row = no depth), cam (cx, cy, fx, fy, ...), T 4 x 4 initial transform.
    Returns (success, T 4 x 4, inlier indices)."""
    n = len(img)
    delta = float(np.float32(math.sqrt(5.991)))
    inl = [True] * n
    edges = []; good = 0
    for i in range(n):
        if float(obj[i][0]) == 0.0 and float(obj[i][1]) == 0.0 and float(obj[i][2]) == 0.0:
            inl[i] = False
            continue
        good += 1
        e = _PnpEdge(); e.id = i; e.level = 0; e.robust = True
        e.X = [float(obj[i][0]), float(obj[i][1]), float(obj[i][2])]; e.u = float(img[i][0]); e.v = float(img[i][1]); e.e = (0.0, 0.0)
        edges.append(e)
    T = np.asarray(T, np.float64)
    R0 = [float(T[r, c]) for r in range(3) for c in range(3)]; t0 = [float(T[r, 3]) for r in range(3)]
    R, t = R0, t0
    for it in range(4):
        R, t = _pnp_optimize(edges, R0, t0, cam, delta, 10)
        for i, e in enumerate(edges):
            if inl[e.id]:
                _pnp_error(e, R, t, cam)
            if e.e[0] * e.e[0] + e.e[1] * e.e[1] > 5.991:
                inl[e.id] = False; e.level = 1; good -= 1
            else:
                inl[i] = True; e.level = 0
            if it == 2:
                e.robust = False
        if good < 5:
            break
    out = np.eye(4)
    for r in range(3):
        for c in range(3):
            out[r, c] = R[3 * r + c]
        out[r, 3] = t[r]
    return n > min_inliers, out, np.array([i for i in range(n) if inl[i]], np.int32)
