"""J / F evaluation of label sequences held in memory (counterpart of the reference's lib/evaluation.py:9-85, which reads
PNGs from disk).  Frames 0 and the last one are skipped like in the DAVIS protocol (lib/evaluation.py:35-41)."""
import numpy as np

from .davis import db_eval_boundary, db_eval_iou, db_statistics


def evaluate_sequence(pred_labels, gt_labels, obj_ids, measure='J', skip_first_last=True):
    """pred_labels / gt_labels: lists of (H,W) integer label images; returns {obj_id: per-frame values}."""
    fn = db_eval_iou if measure == 'J' else db_eval_boundary
    frames = range(1, len(pred_labels) - 1) if skip_first_last and len(pred_labels) > 2 else range(len(pred_labels))
    out = {}
    for oid in obj_ids:
        out[oid] = [fn(np.asarray(gt_labels[t]) == oid, np.asarray(pred_labels[t]) == oid) for t in frames]
    return out


def evaluate_dataset(results, measure='J'):
    """results: iterable of (name, pred_labels, gt_labels, obj_ids) -> dict(mean, recall, decay, per_sequence)."""
    per_seq, all_means = {}, []
    for name, pred, gt, ids in results:
        vals = evaluate_sequence(pred, gt, ids, measure)
        stats = {oid: db_statistics(v) for oid, v in vals.items()}
        per_seq[name] = stats
        all_means.extend(s[0] for s in stats.values())
    m = float(np.nanmean(all_means)) if all_means else float('nan')
    return dict(measure=measure, mean=m, per_sequence=per_seq)


def j_and_f(pred_labels, gt_labels, obj_ids):
    """Mean of J-mean and F-mean over objects, in percent (the DAVIS-2017 'J&F' number) for one sequence."""
    j = evaluate_sequence(pred_labels, gt_labels, obj_ids, 'J')
    f = evaluate_sequence(pred_labels, gt_labels, obj_ids, 'F')
    jm = np.mean([np.mean(v) for v in j.values()])
    fm = np.mean([np.mean(v) for v in f.values()])
    return 100.0 * (jm + fm) / 2.0, 100.0 * jm, 100.0 * fm
