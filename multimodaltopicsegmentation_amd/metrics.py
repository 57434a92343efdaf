"""Segmentation metrics used by ``TextSegmenter.test_step`` (reference: models/lightning_model.py:16-152).

``segeval`` (third party, pinned 2.0.11 in requirements.txt:4) is what the reference calls for Pk / WindowDiff / B; it is
not installed here and not part of the reference tree.  If it is importable it is used, so results are the reference's own
on a machine that has it.  Otherwise Pk and WindowDiff follow segeval 2.0.11's published behaviour -- PARITY UNPINNED for
exactly these conventions (nothing here can execute segeval; tests/test_product_host_cpu.py holds hand-computed cases):

  * masses -> one segment index per unit; hypothesis and reference must cover the same number of units;
  * default window k = int(round(mean reference mass / 2)) with python's round (half to even; segeval rounds a Decimal
    the same way), and k = 2 whenever that is below 2;
  * N - k windows, window i spans units i .. i + k;
  * Pk (Beeferman et al. 1999): fraction of windows whose two end units lie in the same segment in exactly one of the
    two segmentations;  WindowDiff (Pevzner & Hearst 2002): fraction of windows in which the two segmentations place a
    different NUMBER of boundaries (segeval's default lamprier_et_al_2007_fix=False: no phantom padding);
  * the error itself is returned (one_minus=False), as a float (segeval returns Decimal; the reference casts to float);
  * a document not longer than the window has no windows: 0.0 here (segeval's Pk also returns 0 there).

WinPR is pure Python upstream and pinned by fixture g16.  B (boundary edit distance, Fournier 2013) stays a documented
raise without segeval: its transposition weighting lives in segeval's source, which cannot be consulted or executed here.
"""
import numpy as np

try:  # pragma: no cover
    import segeval as _segeval
except Exception:  # noqa: BLE001
    _segeval = None


def get_boundaries(boundaries):
    """bool list -> list of segment masses (lightning_model.py:16-24)."""
    tot_sents, masses = 0, []
    for boundary in boundaries:
        tot_sents += 1
        if boundary:
            masses.append(tot_sents)
            tot_sents = 0
    return masses


def _positions(masses):
    pos = []
    for seg, m in enumerate(masses):
        pos.extend([seg] * int(m))
    return pos


def _default_k(ref_masses):
    return max(int(round(sum(ref_masses) / float(len(ref_masses)) / 2.0)), 2)


def pk(h, t, window_size=None):
    if _segeval is not None:
        return _segeval.pk(h, t) if window_size is None else _segeval.pk(h, t, window_size=window_size)
    hp, tp = _positions(h), _positions(t)
    assert len(hp) == len(tp)
    k = window_size or _default_k(t)
    n = len(tp) - k
    if n <= 0:
        return 0.0
    return sum(1 for i in range(n) if (hp[i] == hp[i + k]) != (tp[i] == tp[i + k])) / float(n)


def window_diff(h, t, window_size=None):
    if _segeval is not None:
        return _segeval.window_diff(h, t) if window_size is None else _segeval.window_diff(h, t, window_size=window_size)
    hp, tp = _positions(h), _positions(t)
    assert len(hp) == len(tp)
    k = window_size or _default_k(t)
    n = len(tp) - k
    if n <= 0:
        return 0.0
    return sum(1 for i in range(n) if (hp[i + k] - hp[i]) != (tp[i + k] - tp[i])) / float(n)


def compute_Pk(boundaries, ground_truth, window_size=None, boundary_symb='1'):
    """lightning_model.py:26-39: the last position is a boundary on both sides while scoring, then restored."""
    boundaries[-1] = 1
    ground_truth[-1] = 1
    try:
        return pk(get_boundaries(boundaries), get_boundaries(ground_truth), window_size)
    finally:
        boundaries[-1] = 0
        ground_truth[-1] = 0


def compute_window_diff(boundaries, ground_truth, window_size=None, segval=True, boundary_symb='1'):
    """lightning_model.py:41-55."""
    boundaries[-1] = 1
    ground_truth[-1] = 1
    try:
        return window_diff(get_boundaries(boundaries), get_boundaries(ground_truth), window_size)
    finally:
        boundaries[-1] = 0
        ground_truth[-1] = 0


def f1_boundary(y_true, y_pred):
    """sklearn.metrics.f1_score(y_true, y_pred, labels=[1], average=None)[0] (lightning_model.py:631-632)."""
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    tp = float(np.sum((y_true == 1) & (y_pred == 1)))
    fp = float(np.sum((y_true != 1) & (y_pred == 1)))
    fn = float(np.sum((y_true == 1) & (y_pred != 1)))
    return 0.0 if tp == 0 else 2 * tp / (2 * tp + fp + fn)


def WinPR(reference, hypothesis, k=10):
    """Scaiano & Inkpen 2012 window-based precision / recall (lightning_model.py:57-124)."""
    reference, hypothesis = list(reference), list(hypothesis)
    assert len(reference) == len(hypothesis), 'Hypothesis and reference should be the same length!'
    N = len(reference)
    RC, last_r, last_c = [], None, None
    for i in range(1 - k, N + 1):
        prev_br = 1 if (last_r is not None and len(last_r) > 0 and last_r[0] == 1) else 0
        prev_bc = 1 if (last_c is not None and len(last_c) > 0 and last_c[0] == 1) else 0
        # python slicing with a negative start wraps around, exactly as the reference's reference[i:i+k]
        last_r = reference[i:i + k]
        last_c = hypothesis[i:i + k]
        RC.append((sum(reference[max(i, 0):i + k]) + prev_br, sum(hypothesis[max(i, 0):i + k]) + prev_bc))
    TP = sum(min(R, C) for R, C in RC)
    FP = sum(max(0, C - R) for R, C in RC)
    FN = sum(max(0, R - C) for R, C in RC)
    if TP + FP == 0:
        return 0, 0, 0
    precision = TP / (TP + FP)
    recall = TP / (TP + FN)
    return precision, recall, 2 * (precision * recall / (precision + recall))


def B_measure(boundaries, ground_truth):
    """Boundary-edit-distance measures (Fournier 2013) come from segeval only (lightning_model.py:126-152)."""
    if _segeval is None:
        raise NotImplementedError("metric 'b' needs the third-party package segeval (boundary edit distance); not installed")
    boundaries[-1] = 1
    ground_truth[-1] = 1
    h, t = get_boundaries(boundaries), get_boundaries(ground_truth)
    cm = _segeval.boundary_confusion_matrix(h, t, n_t=4)
    p = float(_segeval.precision(cm, classification=1))
    r = float(_segeval.recall(cm, classification=1))
    f1 = 2 * p * r / (p + r) if (p + r) else 0.0
    return p, r, f1, float(_segeval.boundary_similarity(h, t, n_t=10))
