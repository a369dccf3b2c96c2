#!/usr/bin/env python3
"""The general dense QP kernel alone (wbc_qp_dense_batch) for rocprofv3: `rocprofv3 --kernel-trace --stats -- python3 tools/qp_general_profile.py`
prints the bench's qp_dense_general object (GPU launch time, iterations, CPU oracle time on the same problems) as one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import wbc_quadruped_dob_amd as W  # noqa: E402

print(json.dumps(bench.qp_dense_general(W, torch, "f64", with_cpu=True)))
