"""Confusion counts, mirror of reference catfish/metrics.py:3-36 (vectorised)."""
import numpy as np


def confusion_matrix(true_labels, predicted_labels):
    """Returns (true_pos, false_pos, true_neg, false_neg) -- catfish/metrics.py:3-36.

    Like the reference, labels other than 0/1 in ``predicted_labels`` are ignored
    and a length mismatch raises ValueError.
    """
    if len(true_labels) != len(predicted_labels):
        raise ValueError("Length of labels to compare is not equal.")
    t = np.asarray(true_labels)
    p = np.asarray(predicted_labels)
    true_pos = int(np.count_nonzero((p == 1) & (t == 1)))
    false_pos = int(np.count_nonzero((p == 1) & (t != 1)))
    true_neg = int(np.count_nonzero((p == 0) & (t == 0)))
    false_neg = int(np.count_nonzero((p == 0) & (t != 0)))
    return true_pos, false_pos, true_neg, false_neg
