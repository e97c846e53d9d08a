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
        i = int(math.floor(float(f32(x) * inv))); j = int(math.floor(float(f32(y) * inv))); k = int(math.floor(float(f32(z) * inv)))
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
