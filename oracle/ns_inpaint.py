"""TEST INFRASTRUCTURE ONLY.  Line-by-line Python restatement (float32 / float64 arithmetic as in the C++, plain loops: small images only)
of the Navier-Stokes based fast-marching inpainting the product implements (elevation_mapping_cupy_amd/csrc/emap_inpaint_ns.cpp), i.e. of
the step the reference delegates to OpenCV: ``cv2.inpaint(h, mask, 1, cv2.INPAINT_NS)`` (reference plugins/inpainting.py:33-38,59).

Third-party dependency: opencv-python, NOT pinned by the reference (requirements.txt) and absent from this image and from
/root/reference, so no golden vector of OpenCV's output exists here: **parity unpinned**.  The method is the published one (Bertalmio,
Bertozzi, Sapiro, CVPR 2001) in the fast-marching form OpenCV gives it: the fill order is the arrival time of a front started at the
region's boundary (eikonal update over the four quadrant pairs, first in first out among equal times); a pixel is the mean of the
known pixels within the radius weighted by 1 / (|r|^2 + 1) x |r . iso| / sqrt(|r| |iso|), iso = the isophote direction at the known
pixel from absolute differences of its known neighbours; the value is stored with one rounding to nearest."""
import heapq
import itertools

import numpy as np

KNOWN, BAND, INSIDE = 0, 1, 2
F32, F64 = np.float32, np.float64


def _solve(t, f, i1, j1, i2, j2):
    a, b = t[i1, j1], t[i2, j2]
    m = min(a, b)
    ka, kb = f[i1, j1] != INSIDE, f[i2, j2] != INSIDE
    if ka and kb:
        if abs(F32(a - b)) >= F32(1.0):
            return F32(F32(1.0) + m)
        return F32(F32(F32(a + b) + np.sqrt(F32(F32(2.0) - F32(F32(a - b) * F32(a - b))))) * F32(0.5))
    if ka:
        return F32(F32(1.0) + a)
    if kb:
        return F32(F32(1.0) + b)
    return F32(F32(1.0) + m)


def inpaint_ns(image, mask, radius=1):
    """image, mask: (rows, cols) uint8; returns the inpainted uint8 image"""
    img = np.asarray(image, np.uint8).copy(); mask = np.asarray(mask) != 0
    rows, cols = img.shape
    if rows < 2 or cols < 2:                   # (the clamped neighbour indices need two rows / columns; emap_inpaint_ns_u8 rejects such shapes too)
        raise ValueError("inpaint_ns needs at least 2 x 2 pixels")
    R, C = rows + 2, cols + 2
    rng = max(1, min(100, int(radius)))
    f = np.zeros((R, C), np.uint8); t = np.full((R, C), 1.0e6, F32)
    f[1:-1, 1:-1][mask] = INSIDE
    heap, order = [], itertools.count()

    def push(i, j, T):
        heapq.heappush(heap, (float(T), next(order), i, j))
    band = []
    for i in range(1, R - 1):
        for j in range(1, C - 1):
            if f[i, j] != INSIDE and INSIDE in (f[i - 1, j], f[i + 1, j], f[i, j - 1], f[i, j + 1]):
                t[i, j] = 0; push(i, j, 0.0); band.append((i, j))
    for i, j in band:
        f[i, j] = BAND

    def O(i, j):
        return int(img[i, j])
    while heap:
        _, _, ii, jj = heapq.heappop(heap)
        f[ii, jj] = KNOWN
        for di, dj in ((-1, 0), (0, -1), (1, 0), (0, 1)):
            i, j = ii + di, jj + dj
            if i <= 0 or j <= 0 or i >= R - 1 or j >= C - 1 or f[i, j] != INSIDE:
                continue
            dist = min(min(_solve(t, f, i - 1, j, i, j - 1), _solve(t, f, i + 1, j, i, j - 1)),
                       min(_solve(t, f, i - 1, j, i, j + 1), _solve(t, f, i + 1, j, i, j + 1)))
            t[i, j] = dist
            Ia, s = F32(0), F32(1.0e-20)
            for k in range(i - rng, i + rng + 1):
                km, kp = k - 1 + (k == 1), k - 1 - (k == R - 2)
                for l in range(j - rng, j + rng + 1):
                    lm, lp = l - 1 + (l == 1), l - 1 - (l == C - 2)
                    if k <= 0 or l <= 0 or k >= R - 1 or l >= C - 1:
                        continue
                    if f[k, l] == INSIDE or (l - j) ** 2 + (k - i) ** 2 > rng * rng:
                        continue
                    ry, rx = F32(i - k), F32(j - l)
                    rlen = np.sqrt(F32(F32(rx * rx) + F32(ry * ry)))
                    dst = F32(F32(1.0) / F32(F32(rlen * rlen) + F32(1.0)))
                    if f[k + 1, l] != INSIDE:
                        if f[k - 1, l] != INSIDE:
                            gr = F32(abs(O(kp + 1, lm) - O(kp, lm)) + abs(O(kp, lm) - O(km - 1, lm)))
                        else:
                            gr = F32(F32(abs(O(kp + 1, lm) - O(kp, lm))) * F32(2.0))
                    else:
                        gr = F32(F32(abs(O(kp, lm) - O(km - 1, lm))) * F32(2.0)) if f[k - 1, l] != INSIDE else F32(0)
                    if f[k, l + 1] != INSIDE:
                        if f[k, l - 1] != INSIDE:
                            gc = F32(abs(O(km, lp + 1) - O(km, lm)) + abs(O(km, lm) - O(km, lm - 1)))
                        else:
                            gc = F32(F32(abs(O(km, lp + 1) - O(km, lm))) * F32(2.0))
                    else:
                        gc = F32(F32(abs(O(km, lm) - O(km, lm - 1))) * F32(2.0)) if f[k, l - 1] != INSIDE else F32(0)
                    ix, iy = F32(-gr), gc
                    dr = F32(F32(rx * ix) + F32(ry * iy))
                    if abs(dr) <= F32(0.01):
                        dr = F32(0.000001)
                    else:
                        glen = np.sqrt(F32(F32(ix * ix) + F32(iy * iy)))
                        dr = F32(abs(F64(dr) / np.sqrt(F64(F32(rlen * glen)))))
                    w = F32(dst * dr)
                    Ia = F32(Ia + F32(w * F32(O(km, lm))))
                    s = F32(s + w)
            img[i - 1, j - 1] = np.uint8(min(255, max(0, int(np.rint(F64(Ia) / F64(s))))))
            f[i, j] = BAND
            push(i, j, dist)
    return img
