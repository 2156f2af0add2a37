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
CONFIG4_MAX_ABS_DP = 2e-2                 # max |dp| against the fp32 oracle over all samples of the checked reads
