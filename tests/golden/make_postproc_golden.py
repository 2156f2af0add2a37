"""Generate post-processing golden vectors by EXECUTING the reference's own pure-Python functions.

The reference modules cannot be imported whole (catfish/infer.py imports h5py and the
TensorFlow model classes at the top; the ``catfish`` click script imports them too), so the
function definitions that depend only on numpy are extracted with ``ast`` and executed:

  catfish/infer.py : normalize_raw_signal (:96-105), reshape_input (:108-124),
                     class_from_threshold (:128-138), hp_in_pred (:141-162), correct_short (:174-198)
  catfish/catfish  : center_hp (:121-135)
  catfish/metrics.py : confusion_matrix (imports cleanly)

plus the padding rule of infer_class_from_signal (:31-38) and the span-merging loop of the
CLI (:58-81), which are inline statements and therefore restated here verbatim-in-behaviour
around the extracted functions.  Outputs: tests/golden/postproc_golden.json (data only).
Run in the build container (needs /root/reference); the fixture travels, the reference does not.
"""
import ast
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/catfish"


def extract(path, names):
    with open(path) as fh:
        tree = ast.parse(fh.read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    mod = ast.Module(body=body, type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, path, "exec"), ns)
    return ns


def main():
    inf = extract(os.path.join(REF, "infer.py"),
                  {"normalize_raw_signal", "reshape_input", "class_from_threshold", "hp_in_pred", "correct_short"})
    cli = extract(os.path.join(REF, "catfish"), {"center_hp"})
    rng = np.random.default_rng(20261004)
    cases = {"postproc": [], "normalize": [], "center_hp": [], "merge": [], "padding": []}

    # scores -> labels -> corrected -> spans
    def scores_case(scores):
        labels = inf["class_from_threshold"](scores)
        corrected = inf["correct_short"](labels)
        spans = inf["hp_in_pred"](corrected)
        return {"scores": [float(s) for s in scores], "labels": [int(v) for v in labels],
                "corrected": [int(v) for v in corrected], "spans": [[int(a), int(b)] for a, b in spans]}

    for n in (1, 2, 14, 15, 16, 35, 70, 200, 1000):
        for _ in range(3):
            # smooth-ish random walk so that runs of many lengths (incl. 14/15/16) appear
            base = np.cumsum(rng.normal(0, 0.12, size=n)) + rng.normal(0, 0.5)
            scores = 1.0 / (1.0 + np.exp(-base))
            cases["postproc"].append(scores_case(scores))
    for runs in ([(1, 14)], [(1, 15)], [(0, 3), (1, 15), (0, 1), (1, 14), (0, 2), (1, 40)], [(1, 20), (0, 20)],
                 [(0, 50)], [(1, 50)], [(0, 1), (1, 15)], [(1, 15), (0, 1)]):
        lab = np.concatenate([np.full(c, 0.9 if v else 0.1) for v, c in runs])
        cases["postproc"].append(scores_case(lab))
    cases["postproc"].append(scores_case(np.array([0.5, 0.5, 0.4999999, 0.5000001] * 5)))

    # normalisation (int16 DAC in, float64 out)
    for n in (5, 64, 1001, 4096):
        raw = np.clip(np.rint(rng.normal(500, 60, size=n)), 0, 2047).astype(np.int16)
        out = inf["normalize_raw_signal"](raw, "median")
        cases["normalize"].append({"raw": raw.tolist(), "out": [float(v) for v in out]})

    # padding rule of infer_class_from_signal (:31-38) around reshape_input
    for length in (1, 34, 35, 36, 69, 70, 105, 4096, 4130):
        window_size = 35
        raw = np.arange(length, dtype=np.float64)
        if not (len(raw) / window_size).is_integer():
            padding_size = window_size - (len(raw) - (len(raw) // window_size * window_size))
        else:
            padding_size = 35
        padded = np.hstack((raw, np.array(padding_size * [0])))
        shaped = inf["reshape_input"](padded, window_size, 1)
        cases["padding"].append({"length": length, "padding_size": int(padding_size),
                                 "shape": [int(s) for s in shaped.shape]})

    # center_hp (catfish:121-135)
    for _ in range(40):
        len_read = int(rng.integers(200, 6000))
        start = int(rng.integers(-11, len_read))
        end = start + int(rng.integers(1, 1500))
        chunk = int(rng.choice([100, 500, 1000]))
        merged = [[0, 5], [start, end]]
        out = cli["center_hp"](merged, len_read, chunk)
        cases["center_hp"].append({"in": [start, end], "len_read": len_read, "chunk_size": chunk,
                                   "out": [int(out[-1][0]), int(out[-1][1])]})

    # span merging + non-HP complement of the CLI (catfish:58-81), restated around center_hp
    def merge(hp_positions, len_read, chunk_size):
        hp_positions = [list(p) for p in hp_positions]
        merged_positions = [hp_positions[0]]
        for i in range(len(hp_positions)):
            if hp_positions[i][1] >= chunk_size + merged_positions[-1][0]:
                merged_positions[-1][-1] = hp_positions[i - 1][1]
                cli["center_hp"](merged_positions, len_read, chunk_size)
                merged_positions.append(hp_positions[i])
        cli["center_hp"](merged_positions, len_read, chunk_size)
        nonhp = []
        m_start = 0
        for m in range(len(merged_positions)):
            if merged_positions[m][0] > m_start:
                m_end = merged_positions[m][0] - 1
                nonhp.append([m_start, m_end])
            m_start = merged_positions[m][1]
        if merged_positions[-1][1] != len_read:
            nonhp.append([merged_positions[-1][1], len_read])
        return merged_positions, nonhp

    for _ in range(30):
        len_read = int(rng.integers(1500, 20000))
        n_hp = int(rng.integers(1, 12))
        starts = np.sort(rng.choice(np.arange(0, len_read - 60), size=n_hp, replace=False))
        spans = []
        pos = 0
        for s in starts:
            s = max(int(s), pos)
            ln = int(rng.integers(15, 60))
            if s + ln >= len_read:
                break
            spans.append([s - 11, s + ln + 16])
            pos = s + ln + 1
        if not spans:
            continue
        chunk = int(rng.choice([300, 1000]))
        merged, nonhp = merge(spans, len_read, chunk)
        cases["merge"].append({"spans": spans, "len_read": len_read, "chunk_size": chunk,
                               "merged": [[int(a), int(b)] for a, b in merged],
                               "nonhp": [[int(a), int(b)] for a, b in nonhp]})

    with open(os.path.join(HERE, "postproc_golden.json"), "w") as fh:
        json.dump(cases, fh)
    print({k: len(v) for k, v in cases.items()})


if __name__ == "__main__":
    main()
