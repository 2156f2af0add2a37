"""``catfish`` command line -- mirror of the reference's click script (catfish/catfish:18-94).

Same options (``-i/--input-dir``, ``-s/--split-dir``, ``-c/--chunk-size`` default 1000) and the
same pipeline: load the bundled ResNetRNN, predict homopolymer stretches per read, merge them
into chunks of at least ``chunk_size`` samples, derive the non-HP complement, split the reads.
All reads of the directory go through packed multi-read launches instead of one call per file.

The FAST5 splitter (catfish/split_f5.py) is disk I/O around h5py and out of scope: when h5py
is not installed the chunk coordinates are written as JSON next to the would-be split files.
"""
from __future__ import annotations

import datetime
import json
import os

from . import batching
from . import infer
from . import neural_network


def center_hp(merged_positions, len_read, chunk_size=1000):
    """catfish/catfish:121-135: widen the LAST merged span to chunk_size around its centre, then
    shift it back inside [0, len_read] (in-place, quirks of the reference kept)."""
    last = merged_positions[-1]
    len_hp = last[-1] - last[0]
    if len_hp < chunk_size:
        left_padding = (chunk_size - len_hp) // 2
        right_padding = (chunk_size - len_hp) - left_padding
        last[0] = last[0] - left_padding
        last[1] = last[1] + right_padding
        if last[0] < 0:
            last[1] -= last[0]
            last[0] = 0
        if last[1] > len_read:
            last[0] -= len_read - last[1]
            last[1] = len_read
    return merged_positions


def merge_positions(hp_positions, len_read, chunk_size=1000):
    """catfish/catfish:58-65: greedy merge of consecutive HP spans into chunks >= chunk_size."""
    merged_positions = [hp_positions[0]]
    for i in range(len(hp_positions)):
        if hp_positions[i][1] >= chunk_size + merged_positions[-1][0]:
            merged_positions[-1][-1] = hp_positions[i - 1][1]
            center_hp(merged_positions, len_read, chunk_size)
            merged_positions.append(hp_positions[i])
    center_hp(merged_positions, len_read, chunk_size)
    return merged_positions


def nonhp_complement(merged_positions, len_read):
    """catfish/catfish:70-81: stretches between merged HP chunks."""
    out = []
    m_start = 0
    for m in range(len(merged_positions)):
        if merged_positions[m][0] > m_start:
            out.append([m_start, merged_positions[m][0] - 1])
        m_start = merged_positions[m][1]
    if merged_positions[-1][1] != len_read:
        out.append([merged_positions[-1][1], len_read])
    return out


def run_pipeline(input_dir, split_dir, chunk_size=1000, network_path="ResNetRNN", network_type="ResNetRNN",
                 checkpoint=30000, device=0):
    """Body of the reference's ``main`` (catfish/catfish:23-94) up to the split step."""
    hp_dict = {}
    nonhp_dict = {}
    temp_dir = "{}/TEMP".format(os.path.abspath(split_dir))
    temp_dir_hp = "{}/HP".format(os.path.abspath(temp_dir))
    temp_dir_nonhp = "{}/nonHP".format(os.path.abspath(temp_dir))
    os.makedirs(temp_dir_hp)          # raises if they exist, like the reference (:37-38)
    os.mkdir(temp_dir_nonhp)

    input_dir = os.path.abspath(input_dir)
    input_files = os.listdir(input_dir)

    t1 = datetime.datetime.now()
    network_path = os.path.abspath(network_path)
    # Big jobs run 131 072 windows per launch (~1100 reads of 4096 samples): the biGRU launches then end in a 1-2 %
    # tail instead of 8 % and the three layers go out as one dynamically scheduled launch.
    max_windows = 131072 if len(input_files) > 400 else 32768
    model = neural_network.load_network(network_type, network_path, checkpoint=checkpoint, device=device,
                                        max_windows_per_pass=max_windows)
    print("Loaded model in {}".format(datetime.datetime.now() - t1))

    print("Checking for homopolymers in raw signal..")
    t2 = datetime.datetime.now()
    signals = [infer.load_raw("{}/{}".format(input_dir, f)) for f in input_files]
    results = batching.infer_reads(model, signals, max_windows=max_windows)
    for fast5_file, (hp_positions, len_read) in zip(input_files, results):
        if hp_positions != []:
            merged_positions = merge_positions(hp_positions, len_read, chunk_size)
            hp_dict[fast5_file] = merged_positions
            nonhp_dict[fast5_file] = nonhp_complement(merged_positions, len_read)
        else:
            nonhp_dict[fast5_file] = [([(0, len_read), len_read])]     # catfish:82 (kept verbatim)
    print("Finished determining possible HP stretches in {}".format(datetime.datetime.now() - t2))

    print("Splitting reads...")
    t3 = datetime.datetime.now()
    with open(os.path.join(temp_dir, "hp_positions.json"), "w") as fh:
        json.dump(hp_dict, fh)
    with open(os.path.join(temp_dir, "nonhp_positions.json"), "w") as fh:
        json.dump(nonhp_dict, fh)
    print("Chunk coordinates written to {} (FAST5 splitting needs h5py and is outside this path) in {}".format(
        temp_dir, datetime.datetime.now() - t3))
    return hp_dict, nonhp_dict


def _build_click_main():
    import click

    @click.command()
    @click.option("--input-dir", "-i", help="Path to input directory of reads in FAST5 format")
    @click.option("--split-dir", "-s", help="Path to directory to save split reads to")
    @click.option("--chunk-size", "-c", help="Chunk size for homopolymer containing stretches", default=1000,
                  show_default=True)
    def main(input_dir, split_dir, chunk_size):
        """
        A tool with a neural network as basis to predict the presence of
        homopolymers in the raw signal from a MinION sequencer.
        """
        run_pipeline(input_dir, split_dir, chunk_size)

    return main


def main():
    _build_click_main()()


if __name__ == "__main__":
    main()
