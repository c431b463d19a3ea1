"""TEST INFRASTRUCTURE ONLY.  Line-by-line Python restatement (float32 arithmetic, plain loops: small images only) of Telea's
fast-marching inpainting as the product implements it (elevation_mapping_cupy_amd/csrc/emap_inpaint_host.hip), i.e. of the step the
reference delegates to OpenCV: ``cv2.inpaint(h, mask, 1, cv2.INPAINT_TELEA)`` (reference plugins/inpainting.py:59).

Third-party dependency: opencv-python, NOT pinned by the reference (requirements.txt) and absent from this image and from
/root/reference, so no golden vector of OpenCV's output exists here: **parity unpinned**.  The algorithm is the published one
(A. Telea, "An Image Inpainting Technique Based on the Fast Marching Method", Journal of Graphics Tools 9(1), 2004): narrow band
around the region, arrival time T from the eikonal update over the four quadrant pairs, pixel value = normalised sum over the known
pixels within the radius of w (I + grad I . r) with w = direction x distance x level factors; the frame / flag / queue conventions
follow the form OpenCV's implementation is documented to have (FIFO among equal T); the value is stored with one rounding to nearest."""
import heapq
import itertools

import numpy as np

KNOWN, BAND, INSIDE, CHANGE = 0, 1, 2, 3
F32 = np.float32


class _Heap:
    def __init__(self):
        self.h, self.n = [], itertools.count()

    def push(self, i, j, T):
        heapq.heappush(self.h, (float(T), next(self.n), i, j))       # FIFO among equal keys

    def pop(self):
        if not self.h:
            return None
        _, _, i, j = heapq.heappop(self.h)
        return i, j


def _solve(t, f, i1, j1, i2, j2):
    a11, a22 = t[i1, j1], t[i2, j2]
    m12 = min(a11, a22)
    if f[i1, j1] != INSIDE:
        if f[i2, j2] != INSIDE:
            if abs(F32(a11 - a22)) >= F32(1.0):
                return F32(F32(1.0) + m12)
            return F32(F32(F32(a11 + a22) + np.sqrt(F32(F32(2.0) - F32(F32(a11 - a22) * F32(a11 - a22))))) * F32(0.5))
        return F32(F32(1.0) + a11)
    if f[i2, j2] != INSIDE:
        return F32(F32(1.0) + a22)
    return F32(F32(1.0) + m12)


def _min4(a, b, c, d):
    return min(min(a, b), min(c, d))


def inpaint_telea(image, mask, radius=1):
    """image, mask: (rows, cols) uint8; returns the inpainted uint8 image"""
    image = np.asarray(image, np.uint8); mask = np.asarray(mask) != 0
    rows, cols = image.shape
    R, C = rows + 2, cols + 2
    rng = max(1, min(100, int(radius)))
    f = np.zeros((R, C), np.uint8); t = np.full((R, C), 1.0e6, F32)
    out = image.copy()
    m = np.zeros((R, C), bool); m[1:-1, 1:-1] = mask

    def dilate(src, r):
        d = np.zeros_like(src)
        for k in range(-r, r + 1):
            a = np.zeros_like(src)
            if k >= 0:
                a[k:, :] = src[:R - k, :]
            else:
                a[:R + k, :] = src[-k:, :]
            d |= a
            a = np.zeros_like(src)
            if k >= 0:
                a[:, k:] = src[:, :C - k]
            else:
                a[:, :C + k] = src[:, -k:]
            d |= a
        return d
    band = dilate(m, 1) & ~m
    band[0, :] = band[-1, :] = False; band[:, 0] = band[:, -1] = False
    H = _Heap()
    for i, j in zip(*np.nonzero(band)):
        H.push(int(i), int(j), 0.0)
    f[band] = BAND; t[band] = 0; f[m] = INSIDE
    ring = dilate(m, rng) & ~m & ~band
    ring[0, :] = ring[-1, :] = False; ring[:, 0] = ring[:, -1] = False
    fr = np.where(ring, INSIDE, 0).astype(np.uint8)
    Out = _Heap()
    for i, j in zip(*np.nonzero(band)):
        Out.push(int(i), int(j), 0.0)
    while True:                                              # distances outside the region (negated at the end)
        p = Out.pop()
        if p is None:
            break
        ii, jj = p
        fr[ii, jj] = CHANGE
        for di, dj in ((-1, 0), (0, -1), (1, 0), (0, 1)):
            i, j = ii + di, jj + dj
            if i <= 0 or j <= 0 or i >= R - 1 or j >= C - 1:
                continue
            if fr[i, j] == INSIDE:
                dist = _min4(_solve(t, fr, i - 1, j, i, j - 1), _solve(t, fr, i + 1, j, i, j - 1), _solve(t, fr, i - 1, j, i, j + 1), _solve(t, fr, i + 1, j, i, j + 1))
                t[i, j] = dist; fr[i, j] = BAND
                Out.push(i, j, dist)
    t[fr == CHANGE] = -t[fr == CHANGE]

    def O(i, j):
        return F32(out[i, j])
    while True:
        p = H.pop()
        if p is None:
            break
        ii, jj = p
        f[ii, jj] = KNOWN
        for di, dj in ((-1, 0), (0, -1), (1, 0), (0, 1)):
            i, j = ii + di, jj + dj
            if i <= 0 or j <= 0 or i >= R - 1 or j >= C - 1 or f[i, j] != INSIDE:
                continue
            dist = _min4(_solve(t, f, i - 1, j, i, j - 1), _solve(t, f, i + 1, j, i, j - 1), _solve(t, f, i - 1, j, i, j + 1), _solve(t, f, i + 1, j, i, j + 1))
            t[i, j] = dist
            if f[i, j + 1] != INSIDE:
                gx = F32(F32(t[i, j + 1] - t[i, j - 1]) * F32(0.5)) if f[i, j - 1] != INSIDE else F32(t[i, j + 1] - t[i, j])
            else:
                gx = F32(t[i, j] - t[i, j - 1]) if f[i, j - 1] != INSIDE else F32(0)
            if f[i + 1, j] != INSIDE:
                gy = F32(F32(t[i + 1, j] - t[i - 1, j]) * F32(0.5)) if f[i - 1, j] != INSIDE else F32(t[i + 1, j] - t[i, j])
            else:
                gy = F32(t[i, j] - t[i - 1, j]) if f[i - 1, j] != INSIDE else F32(0)
            Ia = Jx = Jy = F32(0); s = F32(1.0e-20)
            for k in range(i - rng, i + rng + 1):
                km, kp = k - 1 + (k == 1), k - 1 - (k == R - 2)
                for l in range(j - rng, j + rng + 1):
                    lm, lp = l - 1 + (l == 1), l - 1 - (l == C - 2)
                    if k <= 0 or l <= 0 or k >= R - 1 or l >= C - 1:
                        continue
                    if f[k, l] == INSIDE or (l - j) ** 2 + (k - i) ** 2 > rng * rng:
                        continue
                    ry, rx = F32(i - k), F32(j - l)
                    len2 = F32(F32(rx * rx) + F32(ry * ry))
                    dst = F32(F32(1.0) / F32(len2 * np.sqrt(len2)))
                    lev = F32(F32(1.0) / F32(F32(1.0) + abs(F32(t[k, l] - t[i, j]))))
                    dr = F32(F32(rx * gx) + F32(ry * gy))
                    if abs(dr) <= F32(0.01):
                        dr = F32(0.000001)
                    w = abs(F32(F32(dst * lev) * dr))
                    if f[k, l + 1] != INSIDE:
                        gIx = F32(F32(O(km, lp + 1) - O(km, lm - 1)) * F32(2.0)) if f[k, l - 1] != INSIDE else F32(O(km, lp + 1) - O(km, lm))
                    else:
                        gIx = F32(O(km, lp) - O(km, lm - 1)) if f[k, l - 1] != INSIDE else F32(0)
                    if f[k + 1, l] != INSIDE:
                        gIy = F32(F32(O(kp + 1, lm) - O(km - 1, lm)) * F32(2.0)) if f[k - 1, l] != INSIDE else F32(O(kp + 1, lm) - O(km, lm))
                    else:
                        gIy = F32(O(kp, lm) - O(km - 1, lm)) if f[k - 1, l] != INSIDE else F32(0)
                    Ia = F32(Ia + F32(w * O(km, lm)))
                    Jx = F32(Jx - F32(F32(w * gIx) * rx))
                    Jy = F32(Jy - F32(F32(w * gIy) * ry))
                    s = F32(s + w)
            sat = F32(F32(Ia / s) + F32(F32(Jx + Jy) / F32(np.sqrt(F32(F32(Jx * Jx) + F32(Jy * Jy))) + F32(1.0e-20))))
            out[i - 1, j - 1] = np.uint8(min(255, max(0, int(np.rint(sat)))))
            f[i, j] = BAND
            H.push(i, j, dist)
    return out
