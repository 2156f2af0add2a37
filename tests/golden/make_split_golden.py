"""Generate golden vectors for the split step by EXECUTING the reference's own ``split_signal`` (catfish/split_f5.py:8-81).

The module imports h5py at the top (absent here), so the one function is lifted with ``ast`` -- like make_postproc_golden.py
does for infer.py -- and run against a stand-in for the four things it touches: ``h5py.File`` (nested groups as dicts with
``create_dataset`` / ``close``, the Signal dataset a numpy int16 vector sliced ``[a:b]``), ``copyfile`` (a deep copy of the
stand-in tree under the new name) and ``os.path``.  What is recorded per case is DATA: the chunk lists that went in, the
read's samples, and for every file the reference created -- in creation order -- its directory, its name and the samples of
its new Signal dataset.  That pins the naming (stem up to the first dot, one running index over HP then non-HP chunks), the
order and the slicing; HDF5 encoding (gzip-9 datasets, the deleted basecall groups) is outside this path and not recorded.

The chunk lists are the reference-executed ``merged`` / ``nonhp`` rows of postproc_golden.json plus hand-made edge cases (a
start below zero, an end beyond the read, an empty slice, no non-HP rows, names with several dots or none).
Outputs: tests/golden/split_golden.json + split_golden.npz.  Run in the build container (needs /root/reference).
"""
import ast
import copy
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/catfish/split_f5.py"


class Group(dict):
    """An HDF5 group as far as split_signal goes: members by name, ``del``, item assignment, ``create_dataset``."""

    def create_dataset(self, name, data=None, dtype=None, compression=None, compression_opts=None):
        assert compression == "gzip" and compression_opts == 9
        self[name] = np.asarray(data).astype(dtype)
        return self[name]


def run_reference(cases):
    with open(REF) as fh:
        tree = ast.parse(fh.read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "split_signal"]
    registry, created = {}, []

    class File(Group):
        def __init__(self, path, mode="r"):
            if path not in registry:
                raise IOError("no such file")
            Group.__init__(self, registry[path])
            self.path = path

        def close(self):
            registry[self.path] = Group(self)

    def copyfile(src, dst):
        registry[dst] = copy.deepcopy(registry[src])
        created.append(dst)

    h5py = type("h5py", (), {"File": File})
    ns = {"h5py": h5py, "copyfile": copyfile, "os": os}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    out = []
    for c in cases:
        path = "/in/" + c["name"]
        registry[path] = Group(Raw=Group(Reads=Group(Read_7=Group(Signal=c["signal"]))),
                               Analyses=Group(Basecall_1D_000=Group(x=1), RawGenomeCorrected_000=Group(x=1)))
        del created[:]
        ns["split_signal"](path, c["hp"], c["nonhp"], "/out/HP", "/out/nonHP")
        out.append([(os.path.basename(os.path.dirname(d)), os.path.basename(d), registry[d]["Raw"]["Reads"]["Read_7"]["Signal"])
                    for d in created])
    return out


def main():
    rng = np.random.default_rng(20261005)
    with open(os.path.join(HERE, "postproc_golden.json")) as fh:
        merges = [c for c in json.load(fh)["merge"] if c["len_read"] < 9000]
    cases = [{"name": "read_%03d.fast5" % i, "len_read": c["len_read"], "hp": c["merged"], "nonhp": c["nonhp"]}
             for i, c in enumerate(merges)]
    cases += [
        {"name": "ch12.read7.strand.fast5", "len_read": 400, "hp": [[-11, 120]], "nonhp": [[120, 400]]},       # start below zero
        {"name": "noextension", "len_read": 300, "hp": [[-5, 40], [100, 340]], "nonhp": [[40, 99], [340, 300]]},   # end beyond, empty slice
        {"name": "whole.fast5", "len_read": 250, "hp": [[0, 250]], "nonhp": []},                                    # no non-HP rows
        {"name": ".hidden.fast5", "len_read": 64, "hp": [[3, 20]], "nonhp": [[0, 2], [20, 64]]},                    # empty stem
        {"name": "tiny.fast5", "len_read": 1, "hp": [[0, 1]], "nonhp": [[1, 1]]},
    ]
    for c in cases:
        c["signal"] = np.clip(np.rint(rng.normal(500, 60, size=c["len_read"])), 0, 2047).astype(np.int16)
    results = run_reference(cases)
    arrays, doc = {}, []
    for i, (c, files) in enumerate(zip(cases, results)):
        arrays["signal_%d" % i] = c["signal"]
        listed = []
        for j, (folder, name, data) in enumerate(files):
            assert data.dtype == np.int16
            arrays["out_%d_%d" % (i, j)] = data
            listed.append([folder, name])
        doc.append({"name": c["name"], "len_read": c["len_read"], "hp": c["hp"], "nonhp": c["nonhp"], "files": listed})
    with open(os.path.join(HERE, "split_golden.json"), "w") as fh:
        json.dump(doc, fh)
    np.savez_compressed(os.path.join(HERE, "split_golden.npz"), **arrays)
    print(len(doc), "cases,", sum(len(d["files"]) for d in doc), "files")


if __name__ == "__main__":
    main()
