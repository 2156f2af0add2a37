"""Generate the network fixtures under tests/golden/ (run in the build container only).

* ckpnt-30000-inference.npz -- the 74 inference tensors of the reference's bundled
  checkpoint (catfish/ResNetRNN/checkpoints/ckpnt-30000.*), read with
  catfish_amd.checkpoint (every tensor CRC-32C verified against the .index).
  This is DATA (weights), re-exported because /root/reference does not exist on
  the GPU box.
* ckpt_table.json -- name / dtype / shape / offset / size / masked crc32c of all 190
  bundle entries (reader KAT).
* golden_read_4096_seed0.npz -- one synthetic 4096-sample read (SURVEY.md 8d
  generator): int16 DAC, normalised windows [118,35] f32, fp64 and fp32 oracle
  probabilities, and per-stage oracle activations for the first 16 windows.
  NOTE: the expected outputs come from oracle/catfish_oracle.py (a restatement;
  network parity is "unpinned", see its header), not from TensorFlow.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from catfish_amd import checkpoint  # noqa: E402
from oracle import catfish_oracle as oracle  # noqa: E402

CKPT_DIR = "/root/reference/catfish/ResNetRNN/checkpoints"


def main():
    entries = checkpoint.read_index(CKPT_DIR + "/ckpnt-30000.index")
    with open(os.path.join(HERE, "ckpt_table.json"), "w") as fh:
        json.dump([e.as_dict() for e in entries.values()], fh, indent=0)
    w = checkpoint.read_inference_weights(CKPT_DIR, "ckpnt-30000")
    np.savez(os.path.join(HERE, "ckpnt-30000-inference.npz"), **w)

    dac = oracle.synthetic_dac(1, 4096, seed=0)[0]
    sig = oracle.normalize_raw_signal(dac)
    x, pad = oracle.pad_and_window(sig)
    x32 = x[:, :, 0].astype(np.float32)
    p64, st = oracle.forward(x32, w, np.float64, return_stages=True)
    p32 = oracle.forward(x32, w, np.float32)
    np.savez_compressed(
        os.path.join(HERE, "golden_read_4096_seed0.npz"),
        dac=dac, x=x32, pad=np.int64(pad), probs_fp64=p64, probs_fp32=p32.astype(np.float32),
        res0_w16=st["res0"][:16].astype(np.float32), res1_w16=st["res1"][:16].astype(np.float32),
        gru0_w16=st["gru0"][:16].astype(np.float32), gru1_w16=st["gru1"][:16].astype(np.float32),
        gru2_w16=st["gru2"][:16].astype(np.float32), logits=st["logits"])
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
