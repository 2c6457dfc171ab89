"""DAVIS region (J) and boundary (F) measures (names of the reference's lib/davis.py:19-189, itself a port of the public
DAVIS toolkit; that file cannot run here: it needs skimage and uses np.bool, removed in numpy >= 1.24).

Restated from the published definitions (Perazzi et al., CVPR 2016):
  J  = |M & G| / |M | G|  (1 when both are empty)
  F  = 2PR/(P+R) between the boundary maps of M and G, a boundary pixel counting as matched when the other boundary has
       a pixel within bound_th * image diagonal (>= 1 px), implemented as a disk dilation.
Pixel-exact agreement with skimage's disk()/binary_dilation is UNPINNED (skimage is not installed); the disk here is
{(dy,dx): dy^2 + dx^2 <= r^2}.  CPU only; not on the hot path.
"""
import numpy as np
from scipy import ndimage


def db_eval_iou(annotation, segmentation):
    a = np.asarray(annotation).astype(bool)
    s = np.asarray(segmentation).astype(bool)
    union = np.logical_or(a, s).sum()
    if union == 0:
        return 1.0
    return float(np.logical_and(a, s).sum()) / float(union)


def seg2bmap(seg):
    """Boundary map: a pixel is a boundary pixel if it differs from its east, south or south-east neighbour."""
    seg = np.asarray(seg).astype(bool)
    e = np.zeros_like(seg)
    s = np.zeros_like(seg)
    se = np.zeros_like(seg)
    e[:, :-1] = seg[:, 1:]
    e[:, -1] = seg[:, -1]
    s[:-1, :] = seg[1:, :]
    s[-1, :] = seg[-1, :]
    se[:-1, :-1] = seg[1:, 1:]
    se[:-1, -1] = seg[1:, -1]
    se[-1, :-1] = seg[-1, 1:]
    se[-1, -1] = seg[-1, -1]
    b = (seg ^ e) | (seg ^ s) | (seg ^ se)
    b[-1, :] = seg[-1, :] ^ e[-1, :]
    b[:, -1] = seg[:, -1] ^ s[:, -1]
    b[-1, -1] = False
    return b


def _disk(r):
    y, x = np.ogrid[-r:r + 1, -r:r + 1]
    return (x * x + y * y) <= r * r


def _within(points, of, r):
    """Which `points` pixels have an `of` pixel within the disk {dy^2 + dx^2 <= r^2}: exactly `points & binary_dilation(of, _disk(r))`
    (zero border), through one exact Euclidean distance transform instead of a (2r+1)^2-element dilation -- 17x17 at 480p, where the
    dilation took ~0.1 s per map and made the boundary measure the cost of every dataset evaluation (of needs at least one pixel)."""
    ys, xs = np.flatnonzero(of.any(1)), np.flatnonzero(of.any(0))
    y0, y1 = max(int(ys[0]) - r, 0), min(int(ys[-1]) + r + 1, of.shape[0])         # nothing outside the bounding box of `of` grown by r can match
    x0, x1 = max(int(xs[0]) - r, 0), min(int(xs[-1]) + r + 1, of.shape[1])
    d = ndimage.distance_transform_edt(~of[y0:y1, x0:x1])   # sqrt of the exact integer squared distance: d <= r  <=>  dy^2 + dx^2 <= r^2
    out = np.zeros_like(points)
    out[y0:y1, x0:x1] = points[y0:y1, x0:x1] & (d <= r)
    return out


def db_eval_boundary(foreground_mask, gt_mask, bound_th=0.008):
    fg = np.asarray(foreground_mask).astype(bool)
    gt = np.asarray(gt_mask).astype(bool)
    bound_pix = bound_th if bound_th >= 1 else int(np.ceil(bound_th * np.linalg.norm(fg.shape)))
    fg_b, gt_b = seg2bmap(fg), seg2bmap(gt)
    r = max(int(bound_pix), 1)
    n_fg, n_gt = fg_b.sum(), gt_b.sum()
    if n_fg == 0 and n_gt > 0:
        precision, recall = 1.0, 0.0
    elif n_fg > 0 and n_gt == 0:
        precision, recall = 0.0, 1.0
    elif n_fg == 0 and n_gt == 0:
        precision, recall = 1.0, 1.0
    else:
        gt_match, fg_match = _within(gt_b, fg_b, r), _within(fg_b, gt_b, r)
        precision, recall = fg_match.sum() / float(n_fg), gt_match.sum() / float(n_gt)
    if precision + recall == 0:
        return 0.0
    return float(2 * precision * recall / (precision + recall))


def db_statistics(per_frame_values):
    """Mean, recall (fraction > 0.5) and decay (first-quarter mean minus last-quarter mean) over the frames."""
    v = np.asarray(per_frame_values, dtype=np.float64)
    v = v[~np.isnan(v)]
    if v.size == 0:
        return float('nan'), float('nan'), float('nan')
    ids = np.round(np.linspace(1, len(v), 5) + 1e-10).astype(np.int64) - 1
    bins = [v[ids[i]:ids[i + 1] + 1] for i in range(4)]
    return float(v.mean()), float((v > 0.5).mean()), float(np.mean(bins[0]) - np.mean(bins[3]))


# ---- the reference's names (lib/davis.py:19-236), pinned by tests/golden/g10_davis.npz --------------------------------------

def davis_jaccard_measure(fg_mask, gt_mask):
    """Reference lib/davis.py:54-71 (argument order: segmentation first)."""
    return db_eval_iou(gt_mask, fg_mask)


def davis_f_measure(foreground_mask, gt_mask, bound_th=0.008):
    """Reference lib/davis.py:75-131."""
    return db_eval_boundary(foreground_mask, gt_mask, bound_th)


def nanmean(*args, **kwargs):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        return np.nanmean(*args, **kwargs)


def mean(X):
    return nanmean(np.asarray(X, dtype=np.float64))


def std(X):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        return np.nanstd(np.asarray(X, dtype=np.float64))


def recall(X, threshold=0.5):
    x = np.asarray(X, dtype=np.float64)
    x = x[~np.isnan(x)]
    return nanmean(x > threshold) if x.size else float('nan')


def decay(X, n_bins=4):
    """First-quarter mean minus last-quarter mean over the non-NaN values (reference lib/davis.py:214-227; the bin edges go
    through uint8 there, so sequences beyond 256 evaluated frames wrap -- kept: DAVIS / YouTube-VOS sequences are shorter)."""
    x = np.asarray(X, dtype=np.float64)
    x = x[~np.isnan(x)]
    ids = (np.round(np.linspace(1, len(x), n_bins + 1) + 1e-10) - 1).astype(np.uint8)
    bins = [x[ids[i]:ids[i + 1] + 1] for i in range(4)]
    return nanmean(bins[0]) - nanmean(bins[3])


def evaluate_sequence(segmentations, annotations, object_info, measure='J'):
    """Reference lib/davis.py:19-50.  segmentations / annotations: ordered dicts frame name -> label image ((1,H,W) tensor or
    (H,W) array); object_info: {object id: name of its first frame}.  A frame counts for an object strictly after the
    object's first frame and strictly before the last frame of the sequence; the others stay NaN."""
    fn = {'J': davis_jaccard_measure, 'F': davis_f_measure}[measure]
    names = list(annotations.keys())
    out = dict(raw={})

    def arr(v):
        v = v.numpy() if hasattr(v, 'numpy') else np.asarray(v)
        return v.reshape(v.shape[-2:])
    for obj_id, first in object_info.items():
        r = np.full(len(names), np.nan)
        i0 = names.index(first)
        for i, (an, sg) in enumerate(zip(annotations, segmentations)):
            if i0 < i < len(names) - 1:
                r[i] = fn(arr(segmentations[sg]) == obj_id, arr(annotations[an]) == obj_id)
        out['raw'][obj_id] = r
    for name, f in (('decay', decay), ('mean', mean), ('recall', recall), ('std', std)):
        out[name] = [float(f(r)) for r in out['raw'].values()]
    return out
