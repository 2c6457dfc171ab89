"""J / F evaluation of label sequences held in memory (counterpart of the reference's lib/evaluation.py:9-85, which reads
PNGs from disk).  Frames 0 and the last one are skipped like in the DAVIS protocol (lib/evaluation.py:35-41)."""
import numpy as np

from . import davis as _davis
from .davis import db_eval_boundary, db_eval_iou, db_statistics


def evaluate_sequence(pred_labels, gt_labels, obj_ids, measure='J', skip_first_last=True):
    """pred_labels / gt_labels: lists of (H,W) integer label images; returns {obj_id: per-frame values}."""
    fn = db_eval_iou if measure == 'J' else db_eval_boundary
    frames = range(1, len(pred_labels) - 1) if skip_first_last and len(pred_labels) > 2 else range(len(pred_labels))
    out = {}
    for oid in obj_ids:
        out[oid] = [fn(np.asarray(gt_labels[t]) == oid, np.asarray(pred_labels[t]) == oid) for t in frames]
    return out


def evaluate_results(results, measure='J'):
    """In-memory form: ``results`` = iterable of (name, pred_labels, gt_labels, obj_ids); returns dict(measure, mean, per_sequence)."""
    per_seq, all_means = {}, []
    for name, pred, gt, ids in results:
        vals = evaluate_sequence(pred, gt, ids, measure)
        stats = {oid: db_statistics(v) for oid, v in vals.items()}
        per_seq[name] = stats
        all_means.extend(s[0] for s in stats.values())
    m = float(np.nanmean(all_means)) if all_means else float('nan')
    return dict(measure=measure, mean=m, per_sequence=per_seq)


def evaluate_dataset(dset, results_path=None, measure='J', to_file=True):
    """The reference's call (lib/evaluation.py:9-85; driver: evaluate.py:159-165): ``dset`` yields sequences with .name / .annos /
    .obj_ids / .start_frames / .merge_objects, the tracker's PNGs are read from ``results_path``/<sequence>/<frame>.png (a Path or a
    str), per-sequence lines and the final "<measure>: mean, recall, decay" line go to stdout and results_path/evaluation-<measure>.txt.
    Returns the summary dict.  Results held in memory go through ``evaluate_results``; for backward compatibility
    ``evaluate_dataset(results)`` / ``evaluate_dataset(results, 'J')`` with a list of tuples still dispatches there -- decided by the
    TYPE of the first argument, never by what results_path looks like (round-2 ADVICE)."""
    in_memory = isinstance(dset, (list, tuple)) and (len(dset) == 0 or isinstance(dset[0], (list, tuple)))
    if in_memory:
        if isinstance(results_path, str) and results_path in ('J', 'F'):
            measure = results_path
        elif results_path is not None:
            raise TypeError('evaluate_dataset(results, ...): in-memory results take no results_path (got %r)' % (results_path,))
        return evaluate_results(dset, measure)
    if results_path is None:
        raise TypeError('evaluate_dataset(dset, results_path, measure): results_path (directory of the PNGs) is required for a dataset object')
    return _evaluate_dataset_files(dset, results_path, measure, to_file)


def j_and_f(pred_labels, gt_labels, obj_ids):
    """Mean of J-mean and F-mean over objects, in percent (the DAVIS-2017 'J&F' number) for one sequence."""
    j = evaluate_sequence(pred_labels, gt_labels, obj_ids, 'J')
    f = evaluate_sequence(pred_labels, gt_labels, obj_ids, 'F')
    jm = np.mean([np.mean(v) for v in j.values()])
    fm = np.mean([np.mean(v) for v in f.values()])
    return 100.0 * (jm + fm) / 2.0, 100.0 * jm, 100.0 * fm


def _evaluate_dataset_files(dset, results_path, measure='J', to_file=True):
    from collections import OrderedDict as odict
    from pathlib import Path
    from .image import imread
    from .utils import text_bargraph
    results_path = Path(results_path)
    scores, decays, recalls, results = [], [], [], odict()
    f = open(results_path / ('evaluation-%s.txt' % measure), 'w') if to_file else None

    def out(msg):
        print(msg)
        if f is not None:
            print(msg, file=f)
            f.flush()
    n_seqs = len(dset)
    for j, sequence in enumerate(dset):
        annotations, segmentations = odict(), odict()
        for file in sequence.annos:
            lb = imread(file)
            annotations[file.stem] = (lb != 0).to(lb.dtype) if sequence.merge_objects else lb
            segmentations[file.stem] = imread(results_path / sequence.name / file.name)
        object_info = {}
        for obj_id in sequence.obj_ids:                       # one start frame per object, no background object
            for frame, ids in sequence.start_frames.items():
                if obj_id in ids:
                    assert obj_id not in object_info
                    object_info[obj_id] = frame
        assert 0 not in object_info
        n_objs = len(object_info)
        out('%d/%d: %s: %d object%s' % (j + 1, n_seqs, sequence.name, n_objs, 's' if n_objs > 1 else ''))
        r = _davis.evaluate_sequence(segmentations, annotations, object_info, measure=measure)
        results[sequence.name] = r
        per_obj, per_frame = [], []
        for obj_id, score in r['raw'].items():
            per_frame.append(score)
            per_obj.append(_davis.mean(score))
            if n_objs > 1:
                out('joint {obj}: acc {score:.3f} \u250a{apf}\u250a'.format(obj=obj_id, score=per_obj[-1], apf=text_bargraph(score)))
        decays.extend(r['decay'])
        recalls.extend(r['recall'])
        scores.extend(per_obj)
        out('final  : acc {seq:.3f} ({dset:.3f}) \u250a{apf}\u250a'.format(
            seq=_davis.mean(per_obj), dset=float(np.mean(scores)), apf=text_bargraph(_davis.nanmean(np.array(per_frame), axis=0))))
    out('%s: %.3f, recall: %.3f, decay: %.3f' % (measure, _davis.mean(scores), _davis.mean(recalls), _davis.mean(decays)))
    if f is not None:
        f.close()
    return dict(measure=measure, mean=float(_davis.mean(scores)), recall=float(_davis.mean(recalls)), decay=float(_davis.mean(decays)),
                per_sequence=results)
