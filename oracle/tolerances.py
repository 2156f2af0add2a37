"""ACCEPTANCE CRITERIA -- TEST INFRASTRUCTURE ONLY, stated once.

The numbers ``tests/`` and ``bench.py`` judge the HIP path by, so that a test, the bench's parity gate and a bench leg
cannot drift apart (three different bounds were in use for BASELINE configs[3]).

* fp32 / bf16x3 (BASELINE configs[1], ``north_star``): per-sample sigmoid outputs within 1e-4 of the fp64 oracle.
* bf16 (BASELINE configs[3]): SURVEY.md section 8(d) judges this configuration by the LABEL MATCH RATE against the fp32
  oracle (bf16 probe in the build container: about 99.8 %); the probability bound is secondary and loose: bf16 operands
  carry 8 significant bits through 3 x 35 recurrent steps.  Measured on the MI355X: per-read max |dp| 6.9e-3 over the 27 reads
  of ``test_config4_bf16_packed_varlen``, 1.19e-2 over the 32 reads of the bench leg (the worst read is one of the longest).
"""
GATE_MAX_ABS_DP = 1e-4                    # fp32 and bf16x3 against the fp64 oracle

CONFIG4_MIN_LABEL_MATCH = 0.998           # over all samples of the checked reads, against the fp32 oracle
CONFIG4_MIN_LABEL_MATCH_PER_READ = 0.99   # no single read may fall below this
# max |dp| against the fp32 oracle.  PARITY UNPINNED BY THE REFERENCE: it holds no bf16 fixture and states no bf16 tolerance, so
# both bounds below are the builder's, set from measurements on the MI355X.  A maximum over samples grows with the number of
# samples looked at, hence two scales (one loose constant for both let the per-sample check of the test drift):
CONFIG4_MAX_ABS_DP_TEST = 1e-2            # tests/test_gpu_pipeline.py::test_config4_bf16_packed_varlen: 27 reads <= 16 384 samples,
                                          # 93 k samples in all, measured 6.9e-3 -- a k-block whose operands went missing lands far above
CONFIG4_MAX_ABS_DP = 2e-2                 # bench.py's config4 leg: 32 reads spread up to the longest of 10 000 (measured 1.19e-2)
