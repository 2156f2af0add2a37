"""Confusion counts and the scores derived from them -- mirror of reference catfish/metrics.py:3-36 and
networks/trainingDB/metrics.py:40-63,117-135 (vectorised counting, same results)."""
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


def precision_recall(true_pos, false_pos, false_neg):
    """networks/trainingDB/metrics.py:40-55: an empty denominator gives 0 (with the reference's message)."""
    try:
        precision = true_pos / (true_pos + false_pos)
    except ZeroDivisionError:
        precision = 0
        print("Precision could not be calculated.")
    try:
        recall = true_pos / (true_pos + false_neg)
    except ZeroDivisionError:
        recall = 0
        print("Recall could not be calculated.")
    return precision, recall


def calculate_accuracy(true_pos, false_pos, true_neg, false_neg):
    """networks/trainingDB/metrics.py:58-63."""
    try:
        return (true_pos + true_neg) / (true_pos + false_pos + true_neg + false_neg)
    except ZeroDivisionError:
        return 0


def f1(precision, recall):
    """networks/trainingDB/metrics.py:117-135: balanced F1 of one class, 0 when precision + recall = 0."""
    try:
        return 2 * (precision * recall) / (precision + recall)
    except ZeroDivisionError:
        print("Precision, recall or both are zero. Unable of calculating weighted F1.")
        return 0


def weighted_f1(precision, recall, n, N):
    """networks/trainingDB/metrics.py:95-113: F1 of one class weighted by its share n / N of the samples."""
    try:
        return 2 * n / N * (precision * recall) / (precision + recall)
    except ZeroDivisionError:
        print("Precision, recall or both are zero. Unable of calculating weighted F1.")
        return 0


def class_from_threshold(predicted_scores, threshold):
    """networks/trainingDB/metrics.py:66-76 -> list of ints."""
    import numpy as np
    return (np.asarray(predicted_scores) >= threshold).astype(np.int64).tolist()
