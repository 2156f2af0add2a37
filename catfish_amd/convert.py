"""FAST5 reads -> the int16 ``.npy`` reads the MI355X path ingests natively.

The reference opens every read with h5py (catfish/infer.py:27-29) and keeps the DAC samples behind the leader
(``process_signal``, infer.py:77-93: ``Signal[first_sample_template:]``).  Neither image of this build holds h5py / libhdf5, so
the directory the CLI classifies is made of one-dimensional little-endian int16 ``.npy`` files holding exactly those samples (what
``infer.load_dac`` returns for the FAST5) -- the format the library's loader pool reads straight into pinned memory
(``cf_listing_load_npy_int16``) and the split step writes.  This module is the bridge, to be run once WHERE the FAST5 files and an
h5py live:

    python -m catfish_amd.convert -i fast5_dir -o npy_dir

``read_x.fast5`` becomes ``read_x.npy`` (the name up to its first dot -- what the split step names its pieces after -- is kept).
With h5py installed the CLI also takes FAST5 directly (``infer.load_dac``, one file at a time through the general loader).
"""
from __future__ import annotations

import os

import numpy as np

from . import infer
from .split import npy_header


def convert_read(src, dst):
    """One read: ``infer.load_dac(src)`` (leader trimmed, NOT normalised) written as an int16 ``.npy``; -> its length in samples.
    ValueError for a wrong path or samples that are not int16 codes, ImportError without h5py (``load_dac``'s own errors)."""
    dac = np.asarray(infer.load_dac(src)).reshape(-1)
    if not infer.is_dac(dac):
        raise ValueError("%s: the signal does not hold int16 DAC codes (dtype %s)" % (src, dac.dtype))
    dac = np.ascontiguousarray(dac, dtype="<i2")
    tmp = dst + ".part"
    with open(tmp, "wb") as fh:
        fh.write(npy_header(dac.shape[0]) + dac.tobytes())
    os.replace(tmp, dst)                                    # a reader of the directory never sees half a file
    return int(dac.shape[0])


def convert_directory(input_dir, output_dir, keep_going=False):
    """Every entry of ``input_dir`` (sorted) -> ``output_dir/<name without its last extension>.npy``.
    -> dict(converted, samples, failed = [(name, error text)]); the first failure raises unless ``keep_going``."""
    os.makedirs(output_dir, exist_ok=True)
    done = {"converted": 0, "samples": 0, "failed": []}
    for name in sorted(os.listdir(input_dir)):
        src = os.path.join(input_dir, name)
        if os.path.isdir(src):
            continue
        try:
            done["samples"] += convert_read(src, os.path.join(output_dir, os.path.splitext(name)[0] + ".npy"))
            done["converted"] += 1
        except Exception as exc:                            # noqa: BLE001 -- reported per file; the reference aborts on the first (infer.py:25-29)
            if not keep_going:
                raise
            done["failed"].append((name, "%s: %s" % (type(exc).__name__, exc)))
    return done


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="FAST5 (or .npz / .bin) reads -> int16 .npy reads for catfish_amd")
    ap.add_argument("-i", "--input-dir", required=True)
    ap.add_argument("-o", "--output-dir", required=True)
    ap.add_argument("--keep-going", action="store_true", help="report unreadable files at the end instead of stopping at the first")
    args = ap.parse_args(argv)
    done = convert_directory(args.input_dir, args.output_dir, keep_going=args.keep_going)
    print("converted %d reads (%d samples) into %s" % (done["converted"], done["samples"], args.output_dir))
    for name, err in done["failed"]:
        print("FAILED %s: %s" % (name, err))
    return 1 if done["failed"] else 0


if __name__ == "__main__":
    raise SystemExit(main())
