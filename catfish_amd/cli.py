"""``catfish`` command line -- mirror of the reference's click script (catfish/catfish:18-94).

Same options (``-i/--input-dir``, ``-s/--split-dir``, ``-c/--chunk-size`` default 1000) and the
same pipeline: load the bundled ResNetRNN, predict homopolymer stretches per read, merge them
into chunks of at least ``chunk_size`` samples, derive the non-HP complement, split the reads.
All reads of the directory go through packed multi-read launches instead of one call per file, and
the per-file loop with its merge tail (catfish/catfish:50-82) shards over the GPUs of the node: ``--gpus N`` (an MI355X
addition) or ``python -m torch.distributed.run --nproc-per-node N -m catfish_amd.cli ...``.

The chunk coordinates are written as two JSON documents under ``<split-dir>/TEMP`` and the reads are cut there into
``HP/<stem>_<k>.npy`` / ``nonHP/<stem>_<k>.npy`` (``catfish_amd/split.py``: the reference's split_f5.split_signal for int16
``.npy`` / ``.npz`` / ``.bin`` reads; its HDF5 container is outside this path).
"""
from __future__ import annotations

import datetime
import os

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


def chunks_of_read(hp_positions, len_read, chunk_size=1000):
    """catfish/catfish:57-82 for one read: (merged HP chunks or None, non-HP stretches).  The per-read Python form of the
    rules (pinned by tests/golden/postproc_golden.json, made from the reference's own functions); the pipeline runs them
    for all reads of a rank at once in ``chunks.ChunkTable.from_spans``."""
    if hp_positions != []:
        merged_positions = merge_positions(hp_positions, len_read, chunk_size)
        return merged_positions, nonhp_complement(merged_positions, len_read)
    return None, [([(0, len_read), len_read])]                 # catfish:82 (kept verbatim)


def run_pipeline(input_dir, split_dir, chunk_size=1000, network_path="ResNetRNN", network_type="ResNetRNN",
                 checkpoint=30000, device=None, precision="fp32", timings=None, gather_table=False, bind=False):
    """Body of the reference's ``main`` (catfish/catfish:23-94), split step included.

    Under ``torch.distributed.run`` (RANK / WORLD_SIZE / LOCAL_RANK in the environment) the per-file loop of
    catfish/catfish:50-82 is sharded WITH its tail, and so is the writing: every rank reads and classifies its own block of
    the (sorted) file list on its own MI355X, merges / centres / complements the spans (``sharding.chunk_files_local``),
    formats its part of the two JSON documents and writes it at its offset (``chunks.write_json_documents``).  The ranks
    exchange three small objects over a gloo group -- set-up status, shard status, byte counts -- and no results at all;
    rank 0 does nothing that grows with the number of ranks.  Files are taken in sorted (bytewise) order (the reference iterates
    in ``os.listdir`` order, which is arbitrary), so N ranks write the same bytes as one.

    Returns on every rank the totals ``dict(reads, samples, reads_with_hp, hp_chunks, bytes)`` plus ``files`` / ``file_range``: the
    names of the files THIS rank classified and their index range in the agreed order of the directory (the whole list is never
    built as Python strings: ``sharding.DirListing``) and ``split``, this rank's counts of the split step (``split.split_reads``).
    With ``gather_table=True`` rank 0's dict also holds ``table``, the
    ``chunks.ChunkTable`` over all files gathered from the ranks, and its ``files`` are all names
    (``table.to_dicts(files)`` = the reference's ``hp_dict`` / ``nonhp_dict``).  ``timings`` (optional dict) receives
    this rank's ``listing_s`` (output directories, the directory listing and the ranks' agreement on it), ``model_s`` (network
    load up to the ranks' agreement on it), ``setup_s`` (their sum), ``infer_s``, ``chunks_s``, ``write_s``, ``split_s`` and ``placement``
    (the CPUs the rank is bound to, ``placement.summary``).  ``bind=True``: bind this PROCESS to the CPUs next to the rank's GPU before
    the first GPU call (``catfish_amd/placement.py``; process-wide and permanent, like a ``taskset`` around the job).  The command line
    (``main``) does that; a library caller keeps its affinity unless it asks (the default), and a binding the process took earlier
    (``placement.bind`` at start-up, as bench.py does) is reported as it is.
    """
    import time
    from . import chunks, placement, sharding, split
    rank, world, local_rank = sharding.dist_env()
    # before the first GPU call and the first pinned allocation: this rank, its loader thread and the library's file pool run
    # on the CPUs next to its MI355X (a no-op when the caller -- bench.py -- bound the process already; CATFISH_BIND=0 turns it off)
    place = placement.bind(local_rank, device_of_rank=(lambda r: _pick_device(r) if (device is None or r != local_rank) else device)) \
        if bind else (placement.current() or {"bound": False, "source": "off (bind=False)",
                                              "cpus": sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []})
    own_group = sharding.init_host_group()
    timings = {} if timings is None else timings
    timings["placement"] = placement.summary(place)
    finished = False
    try:
        # every host-side exchange of this job goes over gloo, also when the caller's default group is RCCL: created
        # first, while no rank can have failed yet
        host_group = sharding.host_gather_group()
        temp_dir = "{}/TEMP".format(os.path.abspath(split_dir))
        input_dir = os.path.abspath(input_dir)
        network_path = os.path.abspath(network_path)
        t1 = datetime.datetime.now()
        # 1. output directories + the listing (catfish/catfish:35-38, 49-50).  Part of the job: it grows with the directory, so it is
        # timed apart from the network load (``listing_s``) and counted by the end-to-end benchmark.  Rank 0 reads the names once and
        # broadcasts them, every rank stats only its n/world block, one all-gather completes the sizes (``sharding.shared_listing``).
        t_list = time.perf_counter()
        listing_error = None
        if rank == 0:
            try:
                os.makedirs("{}/HP".format(temp_dir))      # raises if they exist, like the reference (:37-38)
                os.mkdir("{}/nonHP".format(temp_dir))
            except Exception as exc:                      # noqa: BLE001 -- every rank must learn of it before the data path
                listing_error = exc
        listing, file_sizes = sharding.shared_listing(input_dir, rank, world, group=host_group, error=listing_error)
        timings["listing_s"] = time.perf_counter() - t_list
        # 2. the network (catfish/catfish:40-47), outside what the benchmark counts (``model_s``, up to the ranks' agreement on it)
        t_model = time.perf_counter()
        model = setup_error = None
        # Big jobs run 131 072 windows per launch (~1100 reads of 4096 samples): the biGRU launches then end in a 1-2 %
        # tail instead of 8 % and the three layers go out as one dynamically scheduled launch.
        max_windows = 131072 if len(listing) > 400 * world else 32768
        try:
            model = neural_network.load_network(network_type, network_path, checkpoint=checkpoint,
                                                device=_pick_device(local_rank) if device is None else device,
                                                max_windows_per_pass=max_windows, precision=precision)
            if getattr(model, "engine", None) is not None:   # the card the runtime really gave this rank: re-bind if the sysfs guess was another
                placement.verify(place, model.engine.device_identity()[0], local_rank)
                timings["placement"] = placement.summary(place)
        except Exception as exc:                          # noqa: BLE001 -- every rank must learn of it before the data path
            setup_error = exc
        sharding.agree_or_raise(setup_error, "set-up (loading the network, opening the device)", group=host_group)
        timings["model_s"] = time.perf_counter() - t_model
        timings["setup_s"] = timings["listing_s"] + timings["model_s"]
        if rank == 0:
            print("Loaded model in {}".format(datetime.datetime.now() - t1))
            print("Checking for homopolymers in raw signal..")
        t2 = datetime.datetime.now()
        mine = table = shard_error = None
        try:
            # path strings are built for this rank's block only (sharding.ListingPaths): 100 000 of them on each of 8 ranks was 15 ms
            mine, table = sharding.chunk_files_local(model, sharding.ListingPaths(listing), chunk_size,
                                                     max_samples_per_batch=max_windows * infer.WINDOW_SIZE, rank=rank,
                                                     world_size=world, timings=timings, file_sizes=file_sizes)
        except Exception as exc:                          # noqa: BLE001 -- a bad file on one rank fails the whole job, at once
            shard_error = exc
        sharding.agree_or_raise(shard_error, "classification", group=host_group)
        if rank == 0:
            print("Finished determining possible HP stretches in {}".format(datetime.datetime.now() - t2))
            print("Splitting reads...")
        t3 = time.perf_counter()
        my_files = listing.names(mine[0], mine[-1] + 1) if mine else []
        result = chunks.write_json_documents(temp_dir, table, my_files, group=host_group)
        timings["write_s"] = time.perf_counter() - t3
        result["files"], result["file_range"] = my_files, ((mine[0], mine[-1] + 1) if mine else (0, 0))
        # 3. the split (catfish/catfish:85-92 -> split_f5.split_signal): every rank cuts the reads IT classified, after the documents
        # are agreed on; no gather.  Timed apart (``split_s``): it is file writing, not part of the classification rate.
        t4 = time.perf_counter()
        split_error, split_counts = None, None
        try:
            my_paths = ["{}/{}".format(input_dir, name) for name in my_files]
            split_counts = split.split_reads(table, my_paths, "{}/HP".format(temp_dir), "{}/nonHP".format(temp_dir),
                                             listing=listing, lo=mine[0] if mine else 0)
        except Exception as exc:                          # noqa: BLE001 -- one rank's full disk fails the job on every rank
            split_error = exc
        sharding.agree_or_raise(split_error, "splitting the reads", group=host_group)
        timings["split_s"] = time.perf_counter() - t4
        result["split"] = split_counts                    # THIS rank's counts: reads cut, files_hp, files_nonhp, samples
        if rank == 0:
            print("Finished splitting the raw signals in {}".format(datetime.timedelta(seconds=timings["split_s"])))
        if gather_table:
            import torch.distributed as dist
            if world > 1:
                parts = [None] * world if rank == 0 else None
                dist.gather_object(table, parts, dst=0, group=host_group)
                table = chunks.ChunkTable.concat(parts) if rank == 0 else None
            if rank == 0:
                result["table"] = table
                result["files"], result["file_range"] = listing.names(), (0, len(listing))
        finished = True
        return result
    finally:
        if own_group:
            import torch.distributed as dist
            if finished:                      # never a collective on the way out of a failure: the peers may not get there
                dist.barrier()
            dist.destroy_process_group()


def _pick_device(local_rank):
    """One process per GPU: rank r of the node drives device r.  ``CATFISH_DEVICE`` pins every rank to one device
    instead (rehearsing the multi-rank path on a box with fewer GPUs than ranks)."""
    return int(os.environ["CATFISH_DEVICE"]) if "CATFISH_DEVICE" in os.environ else local_rank


def free_port():
    """A TCP port nobody listens on right now (for the launcher's rendezvous, so that two jobs on one node do not collide)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(n_gpus, argv):
    """Start ``n_gpus`` ranks of this CLI under torch.distributed.run as a CHILD process (never an exec: nothing in
    this process may have touched the GPU, and the child's exit code becomes ours).  Rendezvous on 127.0.0.1 at
    ``MASTER_PORT`` when the caller set one, else at a port that is free now."""
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_gpus)),
           "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT") or str(free_port()),
           "-m", "catfish_amd.cli"] + list(argv)
    return subprocess.call(cmd)


def _build_click_main():
    import click

    @click.command()
    @click.option("--input-dir", "-i", help="Path to input directory of reads in FAST5 format")
    @click.option("--split-dir", "-s", help="Path to directory to save split reads to")
    @click.option("--chunk-size", "-c", help="Chunk size for homopolymer containing stretches", default=1000,
                  show_default=True)
    @click.option("--gpus", "-g", default=1, show_default=True,
                  help="MI355X devices of this node to shard the reads over (one process per GPU)")
    @click.option("--network-path", default="ResNetRNN", show_default=True, help="Directory of the trained network")
    @click.option("--precision", default="fp32", show_default=True, type=click.Choice(["fp32", "bf16x3", "bf16"]),
                  help="Arithmetic of the biGRU matmuls")
    def main(input_dir, split_dir, chunk_size, gpus, network_path, precision):
        """
        A tool with a neural network as basis to predict the presence of
        homopolymers in the raw signal from a MinION sequencer.
        """
        if gpus > 1 and "WORLD_SIZE" not in os.environ:
            import sys
            argv = ["-i", input_dir, "-s", split_dir, "-c", str(chunk_size), "--network-path", network_path,
                    "--precision", precision]
            sys.exit(launch_ranks(gpus, argv))
        run_pipeline(input_dir, split_dir, chunk_size, network_path=network_path, precision=precision, bind=True)

    return main


def main():
    _build_click_main()()


if __name__ == "__main__":
    main()
