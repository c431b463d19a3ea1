"""Host side of the safety-polygon service (reference EM/traversability_polygon.py:10-63): masked untraversability statistics and
the convex hull of the untraversable cells.  The reference builds the hull with shapely (``MultiPoint(...).convex_hull``, an
unpinned third-party dependency that is absent here); ``scipy.spatial.ConvexHull`` returns the same vertex set -- the ring is
emitted counter-clockwise and closed (first vertex repeated) like a shapely exterior, but its starting vertex is not pinned by
any reference test."""
import numpy as np


def get_masked_traversability(map_array, mask, traversability):
    traversability = traversability[1:-1, 1:-1]
    is_valid = map_array[2][1:-1, 1:-1]
    mask = mask[1:-1, 1:-1]
    untraversability = np.where(is_valid > 0.5, 1 - traversability, 0)      # invalid place is 0 traversability value
    masked = untraversability * mask
    masked_isvalid = is_valid * mask
    return masked, masked_isvalid


def is_traversable(masked_untraversability, thresh, min_thresh, max_over_n):
    untraversable_thresh = 1 - thresh
    max_thresh = 1 - min_thresh
    over_thresh = np.where(masked_untraversability > untraversable_thresh, 1, 0)
    polygon = calculate_untraversable_polygon(over_thresh)
    max_untraversability = masked_untraversability.max()
    if over_thresh.sum() > max_over_n:
        is_safe = False
    elif max_untraversability > max_thresh:
        is_safe = False
    else:
        is_safe = True
    return is_safe, polygon


def calculate_area(polygon):
    area = 0
    for i in range(len(polygon)):
        p1 = polygon[i - 1]
        p2 = polygon[i]
        area += (p1[0] * p2[1] - p1[1] * p2[0]) / 2.0
    return abs(area)


def calculate_untraversable_polygon(over_thresh):
    x, y = np.where(over_thresh > 0.5)
    points = np.stack([x, y]).T.astype(np.float64)
    if points.shape[0] < 3:
        return None                                  # shapely: empty / Point / LineString
    from scipy.spatial import ConvexHull, QhullError
    try:
        hull = ConvexHull(points)
    except QhullError:                               # collinear cells: a LineString in shapely terms
        return None
    ring = points[hull.vertices]                     # counter-clockwise in 2-D
    return np.vstack([ring, ring[:1]])


def transform_to_map_position(polygon, center, cell_n, resolution):
    return center.reshape(1, 2) + (polygon - cell_n / 2.0) * resolution


def transform_to_map_index(points, center, cell_n, resolution):
    return ((points - center.reshape(1, 2)) / resolution + cell_n / 2).astype(np.int32)
