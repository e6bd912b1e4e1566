#!/usr/bin/env python3
"""
tests/golden/measured_deviations.json <- the per-process files a GPU run of the suite left in a directory:

    PPOAF_RECORD_DEVIATIONS=gpurun_out/dev python -m pytest tests -m gpu -q        (on the MI355X box)
    python tests/golden/merge_deviations.py gpurun_out/dev                          (here)

Per (case, key) the largest value any process recorded.  What the numbers are: tests/test_gpu_reference_golden.py
(`_bound`): deviations of per-epoch statistics and of the weights after all optimiser steps from the reference's
fixtures; the asserts allow MARGIN x these.
"""
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    with open(f) as fh:
        for case, kv in json.load(fh).items():
            slot = out.setdefault(case, {})
            for k, v in kv.items():
                slot[k] = max(float(v), slot.get(k, 0.0))
with open(os.path.join(HERE, "measured_deviations.json"), "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
print(f"{len(out)} cases -> tests/golden/measured_deviations.json")
