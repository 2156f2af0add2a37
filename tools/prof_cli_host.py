"""cProfile of `cli.run_pipeline` on one rank (12 500 x 4096-sample .npy files, page cache warm): where the HOST spends its time when the
GPU is not the limit (bf16).  usage: python tools/prof_cli_host.py [precision=bf16]"""
import contextlib
import cProfile
import io
import os
import pstats
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from catfish_amd import checkpoint, cli  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
root = tempfile.mkdtemp(prefix="catfish_prof_")
try:
    reads = os.path.join(root, "reads")
    os.makedirs(reads)
    rng = np.random.default_rng(0)
    for i in range(12500):
        np.save(os.path.join(reads, "read_%06d.npy" % i), bench.squiggle_dac(rng, 4096))
    os.makedirs(os.path.join(root, "ResNetRNN", "checkpoints"))
    with open(os.path.join(root, "ResNetRNN", "ResNetRNN.txt"), "w") as fh:
        fh.write("MODEL TYPE: ResNet-RNN\n\nbatch_size: 256\noptimizer_choice: RMSProp\nlearning_rate: 0.001\nlayer_size: 64\nn_layers: 3\n"
                 "keep_prob: 0.8\nlayer_size_res: 32\nn_layers_res: 2\n")
    checkpoint.write_checkpoint(os.path.join(root, "ResNetRNN", "checkpoints", "ckpnt-30000"), bench.load_weights())
    with contextlib.redirect_stdout(io.StringIO()):
        cli.run_pipeline(reads, os.path.join(root, "warm"), network_path=os.path.join(root, "ResNetRNN"), device=0, precision=prec)
    pr = cProfile.Profile()
    with contextlib.redirect_stdout(io.StringIO()):
        pr.enable()
        cli.run_pipeline(reads, os.path.join(root, "out"), network_path=os.path.join(root, "ResNetRNN"), device=0, precision=prec)
        pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
finally:
    shutil.rmtree(root, ignore_errors=True)
