"""Runs the counter-calibration gather (rto_probe_gather): 32 Mi lines = 4 GiB buffer, 3 launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rt_octree_amd as R  # noqa: E402

n_lines = 32 << 20
rc = R.lib().rto_probe_gather(n_lines, 3)
print("probe rc", rc, "lines", n_lines)
